// pbf_oracle.cpp — CPU restatement of the reference's per-step PBF-SPH algorithm.
//
// TEST INFRASTRUCTURE ONLY (see pbf_oracle.h): the checker for the HIP path, never the product.
// PARITY UNPINNED for the floating-point stages (reference ships no goldens and its OpenMP
// backend is unbuildable here without glm); integer stages are pinned against oracle/_ref.
//
// Written from the algorithm in /root/reference/src/omp/ompsph.hpp, src/sph.hpp, src/curves.h,
// src/sph_constants.h (cited per function, paths relative to the reference root).  glm 0.9.9.8
// (CMakeLists.txt:27-29, absent from the image) supplies only elementary vector arithmetic on
// this path; the published semantics restated here are:
//   distance(a,b) = length(b-a) = sqrt(dot(d,d)),  dot = (x*x + y*y) + z*z,
//   length2(v) = dot(v,v), pow2(x) = x*x, pow3(x) = x*x*x, pow = std::pow,
//   mix(a,b,t) = a*(1-t) + b*t, clamp(x,lo,hi) = min(max(x,lo),hi),
//   normalize(v) = v * (1/sqrt(dot(v,v))), vec/scalar = per-component true division.
//
// Build: g++ -O2 -ffp-contract=off -fopenmp -shared -fPIC (no fast-math: SURVEY §8c).

#include "pbf_oracle.h"

#include "mc_tables.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <thread>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;

// ---- sph_constants.h:5-16 — all `float`, promoted to N at the use site -------------------
constexpr float VD = 0.49f;
constexpr float RHO = 6378.0f;
constexpr float RHO_RECIP = 1.f / RHO;
constexpr float EPSILON = 0.00000001f;
constexpr float CFM_EPSILON = 600.0f;
constexpr float CorrDeltaQ = 0.3f;
constexpr float C_XSPH = 0.00001f;             // sph_constants.h:13 (unused by the reference)
constexpr float VORTICITY_EPSILON = 0.0005f;   // sph_constants.h:14 (unused by the reference)
constexpr float CorrK = 0.0001f;
constexpr float CorrN = 4.f;

// ---- curves.h:72-88 — 10-bit-per-axis Morton encode -------------------------------------
inline uint64_t spread10(uint64_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
inline uint64_t mortonEncode(uint64_t x, uint64_t y, uint64_t z) {
  return spread10(x) | spread10(y) << 1 | spread10(z) << 2;
}
// ---- curves.h:46-65 — decode -----------------------------------------------------------
inline uint64_t uninterleave(uint64_t value) {
  uint64_t ret = 0;
  for (int b = 0; b < 10; ++b) ret |= (value & (uint64_t(1) << (3 * b))) >> (2 * b);
  return ret;
}
inline uint64_t mortonDecode(uint64_t index, int axis) { return uninterleave((index >> axis) & 0x9249249); }

// ---- sph.hpp:217-234 — the 27 neighbour codes, x fastest, then y, then z -----------------
inline void neighbourCodes(uint64_t zIndex, uint64_t out[27]) {
  const uint64_t x = mortonDecode(zIndex, 0), y = mortonDecode(zIndex, 1), z = mortonDecode(zIndex, 2);
  int k = 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx)  // size_t wrap-around (x-1 at x=0) is kept: spread10 keeps the low 10 bits
        out[k++] = mortonEncode(x + uint64_t(int64_t(dx)), y + uint64_t(int64_t(dy)), z + uint64_t(int64_t(dz)));
}

template <typename N> struct V3 {
  N x, y, z;
};
template <typename N> struct V4 {
  N x, y, z, w;
};
template <typename N> inline V3<N> operator+(V3<N> a, V3<N> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename N> inline V3<N> operator-(V3<N> a, V3<N> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename N> inline V3<N> operator*(V3<N> a, N s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename N> inline V3<N> operator*(N s, V3<N> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename N> inline V3<N> operator/(V3<N> a, N s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename N> inline N dot(V3<N> a, V3<N> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename N> inline N distance(V3<N> a, V3<N> b) {
  const V3<N> d = b - a;
  return std::sqrt(dot(d, d));
}

// ---- sph.hpp:251-253 — pi<N>, poly6Factor, spikyKernelFactor -----------------------------
// std::pow(N, int) yields double for N=float, so the float case is evaluated partly in double
// exactly as the C++ usual arithmetic conversions dictate, then rounded to N on return.
template <typename N> N piN() { return std::acos(-N(1)); }
template <typename N> N poly6Factor(N h) { return N(N(315.0) / (N(64.0) * piN<N>() * std::pow(h, 9))); }
template <typename N> N spikyKernelFactor(N h) { return N(-(N(45.0) / (piN<N>() * std::pow(h, 6)))); }

// ---- ompsph.hpp:67-75 ------------------------------------------------------------------
template <typename N> inline N poly6Kernel(N r, N factor, N h) {
  if (!(r <= h)) return N(0);
  const N d = (h * h) - r * r;
  return factor * (d * d * d);
}
template <typename N> inline V3<N> spikyKernelGradient(V3<N> x, V3<N> y, N r, N h, N factor) {
  if (r >= EPSILON && r <= h) {
    const N hr = h - r;
    return (x - y) * (factor * ((hr * hr) / r));
  }
  return {N(0), N(0), N(0)};
}

struct Params {
  const pbf_oracle_params *p;
};

}  // namespace

struct pbf_oracle {
  int fp64;
  virtual ~pbf_oracle() = default;
};

namespace {

template <typename N> struct Oracle final : pbf_oracle {
  // particle (sph.hpp:36-54), SoA
  std::vector<uint64_t> id;
  std::vector<uint8_t> type;
  std::vector<N> mass;
  std::vector<V3<N>> pos, vel;
  std::vector<V4<N>> colour;
  // PartiallyAdvected scratch (sph.hpp:255-261)
  std::vector<uint64_t> zIndex;
  std::vector<V3<N>> pStar, deltaP;
  std::vector<N> lambda;
  // grid
  std::vector<uint64_t> table;
  std::array<uint64_t, 3> extent{};
  V3<N> minExtent{};
  N lastH = N(0.1);  // h of the last predict(), for candidate_stats
  // marching cubes (ompsph.hpp:277-477)
  std::array<uint64_t, 3> sample{};
  std::vector<V4<N>> latticePN, latticeC;
  std::vector<V3<N>> meshV, meshN;
  std::vector<V4<N>> meshC;
  bool pow4 = false;

  size_t n() const { return id.size(); }

  template <typename F> void foreach_1d(int threads, size_t size, const F &f) const {
    // ompsph.hpp:39-44
#if defined(PBF_ORACLE_STD_THREADS)
    // ThreadSanitizer build (oracle/Makefile `san`): the same static partition on std::thread — libgomp's barriers are
    // invisible to TSan (every parallel-for would be reported), pthread create / join are not.
    const size_t nt = std::max<size_t>(1, std::min<size_t>(threads > 0 ? size_t(threads) : 4, size));
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nt; ++t)
      pool.emplace_back([&, t] {
        for (size_t i = size * t / nt; i < size * (t + 1) / nt; ++i) f(i);
      });
    for (auto &th : pool) th.join();
#else
#ifdef _OPENMP
    const int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (long i = 0; i < static_cast<long>(size); ++i) f(static_cast<size_t>(i));
#endif
  }

  // sph.hpp:203-213 — walk the 27 ranges; last table entry yields an empty range
  template <typename F> void foreach_grid(uint64_t z, const F &f) const {
    uint64_t offs[27];
    neighbourCodes(z, offs);
    const uint64_t tn = table.size();
    for (uint64_t off : offs) {
      if (off >= tn) continue;
      const uint64_t start = table[off];
      const uint64_t end = (off + 1) < tn ? table[off + 1] : start;
      for (uint64_t b = start; b < end; ++b) f(static_cast<size_t>(b));
    }
  }

  void grid_extent(const pbf_oracle_params &c) {
    // ompsph.hpp:132-135
    const N h = N(c.h), scale = N(c.scale);
    const N padding = h * 2;
    const V3<N> minB{N(c.min_bound[0]), N(c.min_bound[1]), N(c.min_bound[2])};
    const V3<N> maxB{N(c.max_bound[0]), N(c.max_bound[1]), N(c.max_bound[2])};
    const V3<N> lo = minB / scale, hi = maxB / scale;
    minExtent = {lo.x - padding, lo.y - padding, lo.z - padding};
    const V3<N> maxExtent = {hi.x + padding, hi.y + padding, hi.z + padding};
    const V3<N> e = (maxExtent - minExtent) / h;
    extent = {static_cast<uint64_t>(e.x), static_cast<uint64_t>(e.y), static_cast<uint64_t>(e.z)};
  }

  static uint64_t cellCoord(N v) {
    // static_cast<size_t>(x / h) (sph.hpp:199).  Negative input is UB in the reference; this
    // restatement pins what x86-64 gcc emits (cvttss2si to int64, reinterpret) so that both
    // oracle and HIP path agree; spread10 then keeps the low 10 bits.
    return static_cast<uint64_t>(static_cast<int64_t>(v));
  }

  int predict(const pbf_oracle_params &c) {
    // ompsph.hpp:128-154.  Obstacles follow the OpenCL backend (ocl/oclsph.cpp:66-69): pStar =
    // position/scale, velocity untouched — the OpenMP early-return (ompsph.hpp:139) leaves the
    // slot default-constructed, i.e. garbage.
    grid_extent(c);
    lastH = N(c.h);
    const size_t cnt = n();
    zIndex.assign(cnt, 0);
    pStar.assign(cnt, {});
    deltaP.assign(cnt, {});
    lambda.assign(cnt, N(0));
    const N h = N(c.h), dt = N(c.dt), scale = N(c.scale);
    const V3<N> force{N(c.constant_force[0]), N(c.constant_force[1]), N(c.constant_force[2])};
    foreach_1d(c.threads, cnt, [&](size_t i) {
      if (type[i] & 1) {
        pStar[i] = pos[i] / scale;
      } else {
        V3<N> combinedForce = mass[i] * force;
        for (int w = 0; w < c.n_wells; ++w) {
          const V3<N> centre{N(c.wells[4 * w]), N(c.wells[4 * w + 1]), N(c.wells[4 * w + 2])};
          const N wforce = N(c.wells[4 * w + 3]);
          const N dist = distance(pos[i], centre);
          if (dist < N(75)) {
            const V3<N> d = centre - pos[i];
            const V3<N> rHat = d * (N(1) / std::sqrt(dot(d, d)));
            const V3<N> t = ((rHat * wforce) * mass[i]) / (dist * dist);
            auto cl = [](N v) { return std::min(std::max(v, N(-10)), N(10)); };
            combinedForce = combinedForce + V3<N>{cl(t.x), cl(t.y), cl(t.z)};
          }
        }
        vel[i] = combinedForce * dt + vel[i];
        pStar[i] = (vel[i] * dt) + (pos[i] / scale);
      }
      zIndex[i] = mortonEncode(cellCoord((pStar[i].x - minExtent.x) / h), cellCoord((pStar[i].y - minExtent.y) / h),
                               cellCoord((pStar[i].z - minExtent.z) / h));
    });
    return 0;
  }

  template <typename T> static void permute(std::vector<T> &v, const std::vector<uint32_t> &perm) {
    std::vector<T> out(v.size());
    for (size_t i = 0; i < perm.size(); ++i) out[i] = v[perm[i]];
    v.swap(out);
  }

  int sort(const pbf_oracle_params &c) {
    // ompsph.hpp:157-159: std::sort of the AoS array by zIndex only.  std::sort's permutation is
    // driven purely by comparison outcomes, so sorting (key, index) pairs with the same key-only
    // comparator under the same libstdc++ reproduces the reference's tie order.
    const size_t cnt = n();
    struct KI {
      uint64_t key;
      uint32_t idx;
    };
    std::vector<KI> ki(cnt);
    for (size_t i = 0; i < cnt; ++i) ki[i] = {zIndex[i], static_cast<uint32_t>(i)};
    auto cmp = [](const KI &l, const KI &r) { return l.key < r.key; };
    if (c.sort == PBF_ORACLE_SORT_STABLE)
      std::stable_sort(ki.begin(), ki.end(), cmp);
    else
      std::sort(ki.begin(), ki.end(), cmp);
    std::vector<uint32_t> perm(cnt);
    for (size_t i = 0; i < cnt; ++i) perm[i] = ki[i].idx;
    permute(id, perm);
    permute(type, perm);
    permute(mass, perm);
    permute(pos, perm);
    permute(vel, perm);
    permute(colour, perm);
    permute(zIndex, perm);
    permute(pStar, perm);
    permute(deltaP, perm);
    permute(lambda, perm);
    return 0;
  }

  int grid_table(const pbf_oracle_params &) {
    // sph.hpp:238-250
    const uint64_t maxZIndex = mortonEncode(extent[0], extent[1], extent[2]);
    table.assign(maxZIndex, 0);
    uint64_t gridIndex = 0;
    const uint64_t size = n();
    for (uint64_t z = 0; z < maxZIndex; ++z) {
      table[z] = gridIndex;
      while (gridIndex != size && zIndex[gridIndex] == z) gridIndex++;
    }
    return 0;
  }

  int diffuse(const pbf_oracle_params &c) {
    // ompsph.hpp:188-207.  GS: in place, ascending index (what one thread does).
    // Jacobi: read the pre-stage colours, like the OpenCL kernel (ocl/oclsph_kernel.h:67-93).
    const size_t cnt = n();
    const N t = N(c.dt) / N(750.0);
    auto body = [&](size_t a, const std::vector<V4<N>> &in, std::vector<V4<N>> &out) {
      if (type[a] != 0) return;  // obstacle, or a ghost copy (bit 1) owned by another slab (tests of the slab driver)
      int nNeighbours = 0;
      V4<N> mixture{N(0), N(0), N(0), N(0)};
      foreach_grid(zIndex[a], [&](size_t b) {
        if (!(type[b] & 1)) {
          mixture = {mixture.x + in[b].x, mixture.y + in[b].y, mixture.z + in[b].z, mixture.w + in[b].w};
          nNeighbours++;
        }
      });
      if (nNeighbours != 0) {
        const N nn = N(nNeighbours);
        auto one = [&](N x, N m) {
          const N y = (m / nn) * N(1.33);
          const N o = x * (N(1) - t) + y * t;
          return std::min(std::max(o, N(0.03)), N(1.0));
        };
        out[a] = {one(in[a].x, mixture.x), one(in[a].y, mixture.y), one(in[a].z, mixture.z), one(in[a].w, mixture.w)};
      }
    };
    if (c.mode == PBF_ORACLE_JACOBI) {
      std::vector<V4<N>> out = colour;
      foreach_1d(c.threads, cnt, [&](size_t a) { body(a, colour, out); });
      colour.swap(out);
    } else {
      for (size_t a = 0; a < cnt; ++a) body(a, colour, colour);
    }
    return 0;
  }

  int lambda_stage(const pbf_oracle_params &c) {
    // ompsph.hpp:211-212,217-232 (race-free in the reference)
    const size_t cnt = n();
    const N h = N(c.h);
    const N Poly6Factor = poly6Factor(h);
    const N SpikyKernelFactor = spikyKernelFactor(h);
    foreach_1d(c.threads, cnt, [&](size_t a) {
      if (type[a] != 0) {
        if (type[a] & 1) lambda[a] = 0;  // a ghost copy keeps the lambda its owner sent
        return;
      }
      V3<N> norm2V{N(0), N(0), N(0)};
      N rho = 0;
      foreach_grid(zIndex[a], [&](size_t b) {
        const N r = distance(pStar[a], pStar[b]);
        norm2V = norm2V + spikyKernelGradient(pStar[a], pStar[b], r, h, SpikyKernelFactor) * N(RHO_RECIP);
        rho += mass[a] * poly6Kernel(r, Poly6Factor, h);
      });
      const N norm2 = dot(norm2V, norm2V);
      const N Ci = (rho / RHO - N(1));
      lambda[a] = -Ci / (norm2 + CFM_EPSILON);
    });
    return 0;
  }

  int delta_stage(const pbf_oracle_params &c) {
    // ompsph.hpp:213,235-248.  GS = in place, ascending index (the reference at 1 thread);
    // Jacobi = all reads see the pre-stage pStar.
    const size_t cnt = n();
    const N h = N(c.h), scale = N(c.scale);
    const N Poly6Factor = poly6Factor(h);
    const N SpikyKernelFactor = spikyKernelFactor(h);
    const N P6DeltaQ = poly6Kernel(N(CorrDeltaQ * h), Poly6Factor, h);
    const V3<N> minB{N(c.min_bound[0]), N(c.min_bound[1]), N(c.min_bound[2])};
    const V3<N> maxB{N(c.max_bound[0]), N(c.max_bound[1]), N(c.max_bound[2])};
    auto body = [&](size_t a, const std::vector<V3<N>> &in, std::vector<V3<N>> &out) {
      if (type[a] != 0) return;
      V3<N> deltaPAcc{N(0), N(0), N(0)};
      foreach_grid(zIndex[a], [&](size_t b) {
        const N r = distance(in[a], in[b]);
        const N q = poly6Kernel(r, Poly6Factor, h) / P6DeltaQ;
        // xsph/vorticity flags aside, `pow4` is a sensitivity probe only (tests): (q*q)*(q*q) as the
        // device evaluates it, instead of the reference's std::pow(q, 4)
        const N corr = N(-CorrK) * (pow4 ? (q * q) * (q * q) : std::pow(q, N(CorrN)));
        const N factor = (lambda[a] + lambda[b] + corr) / N(RHO);
        deltaPAcc = deltaPAcc + spikyKernelGradient(in[a], in[b], r, h, SpikyKernelFactor) * factor;
      });
      deltaP[a] = deltaPAcc;
      V3<N> p = (in[a] + deltaPAcc) * scale;
      p = {std::min(maxB.x, std::max(minB.x, p.x)), std::min(maxB.y, std::max(minB.y, p.y)),
           std::min(maxB.z, std::max(minB.z, p.z))};
      out[a] = p / scale;
    };
    if (c.mode == PBF_ORACLE_JACOBI) {
      std::vector<V3<N>> out = pStar;
      foreach_1d(c.threads, cnt, [&](size_t a) { body(a, pStar, out); });
      pStar.swap(out);
    } else {
      for (size_t a = 0; a < cnt; ++a) body(a, pStar, pStar);
    }
    return 0;
  }

  // Opt-in extras, ABSENT from the reference (only the constants survive, sph_constants.h:13-14;
  // SURVEY finding 3).  Macklin & Mueller 2013 eq. 15-17 on the post-solve velocity; Jacobi only.
  // parity unpinned: there is no reference implementation to compare with.
  // The three sub-stages are also entry points of their own (pbf_oracle_vorticity / _vorticity_force / _xsph): the slab
  // twin (tests/slab_engines.py) refreshes the ghost copies' velocity / vorticity between them.
  std::vector<V3<N>> omega;  // vorticity of the last vorticity_omega()
  void vorticity_omega(const pbf_oracle_params &c) {
    const size_t cnt = n();
    const N h = N(c.h);
    const N SpikyKernelFactor = spikyKernelFactor(h);
    const std::vector<V3<N>> &v = vel;
    omega.assign(cnt, V3<N>{N(0), N(0), N(0)});
    foreach_1d(c.threads, cnt, [&](size_t a) {
      if (type[a] != 0) return;
      V3<N> w{N(0), N(0), N(0)};
      foreach_grid(zIndex[a], [&](size_t b) {
        const N r = distance(pStar[a], pStar[b]);
        const V3<N> g = spikyKernelGradient(pStar[a], pStar[b], r, h, SpikyKernelFactor);
        const V3<N> vij = v[b] - v[a];
        // Macklin & Mueller 2013 eq. 15: omega_i = sum_j v_ij x grad_{p_j} W(p_i - p_j), v_ij = v_j - v_i.  The gradient is
        // with respect to the NEIGHBOUR: grad_{p_j} W = -grad_{p_i} W = -g, hence v_ij x (-g) = g x v_ij.
        w = w + V3<N>{g.y * vij.z - g.z * vij.y, g.z * vij.x - g.x * vij.z, g.x * vij.y - g.y * vij.x};
      });
      omega[a] = w;
    });
  }
  void vorticity_force(const pbf_oracle_params &c) {
    const size_t cnt = n();
    const N h = N(c.h), dt = N(c.dt);
    const N SpikyKernelFactor = spikyKernelFactor(h);
    if (omega.size() != cnt) omega.assign(cnt, V3<N>{N(0), N(0), N(0)});
    std::vector<V3<N>> vnew = vel;
    foreach_1d(c.threads, cnt, [&](size_t a) {
      if (type[a] != 0) return;
      V3<N> eta{N(0), N(0), N(0)};
      foreach_grid(zIndex[a], [&](size_t b) {
        const N r = distance(pStar[a], pStar[b]);
        const V3<N> g = spikyKernelGradient(pStar[a], pStar[b], r, h, SpikyKernelFactor);
        eta = eta + g * std::sqrt(dot(omega[b], omega[b]));
      });
      const N len = std::sqrt(dot(eta, eta));
      if (len > N(EPSILON)) {
        const V3<N> nn = eta * (N(1) / len);
        const V3<N> w = omega[a];
        const V3<N> f{nn.y * w.z - nn.z * w.y, nn.z * w.x - nn.x * w.z, nn.x * w.y - nn.y * w.x};
        vnew[a] = vnew[a] + f * (N(VORTICITY_EPSILON) * dt);
      }
    });
    vel.swap(vnew);
  }
  void xsph(const pbf_oracle_params &c) {
    const size_t cnt = n();
    const N h = N(c.h);
    const N Poly6Factor = poly6Factor(h);
    const std::vector<V3<N>> base = vel;
    foreach_1d(c.threads, cnt, [&](size_t a) {
      if (type[a] != 0) return;
      V3<N> acc{N(0), N(0), N(0)};
      foreach_grid(zIndex[a], [&](size_t b) {
        const N r = distance(pStar[a], pStar[b]);
        acc = acc + (base[b] - base[a]) * poly6Kernel(r, Poly6Factor, h);
      });
      vel[a] = base[a] + acc * N(C_XSPH);
    });
  }
  void extras(const pbf_oracle_params &c, std::vector<V3<N>> &v) {
    (void)v;  // (always `vel`)
    if (c.vorticity) {
      vorticity_omega(c);
      vorticity_force(c);
    }
    if (c.xsph) xsph(c);
  }


  // ---- marching-cubes surface (ompsph.hpp:277-477).  glm's fastDistance / fastLength / fastNormalize
  // (gtx/fast_square_root, absent offline) are approximations without a bit contract; restated with the
  // exact sqrt.  Triangles are emitted in cube order (the reference appends through an atomic counter,
  // i.e. in no particular order); the case tables are oracle/mc_tables.h (generated, see tools/gen_mc_tables.py).
  int mc_field(const pbf_oracle_params &c, const pbf_oracle_mc &m) {
    const N h = N(c.h), scale = N(c.scale), res = N(m.resolution);
    const N particleSize = N(m.particle_size), particleInfluence = N(m.particle_influence);
    for (int k = 0; k < 3; ++k) sample[k] = static_cast<uint64_t>(std::floor(N(extent[k]) * res)) + 1;  // :283-284
    const size_t latticeN = sample[0] * sample[1] * sample[2];
    latticePN.assign(latticeN, V4<N>{N(0), N(0), N(0), N(0)});
    latticeC.assign(latticeN, V4<N>{N(0), N(0), N(0), N(0)});
    const N step = h / res, threshold = h * scale * 1;
    const uint64_t tn = table.size();
    foreach_1d(c.threads, latticeN, [&](size_t idx) {
      const size_t x = idx / (sample[1] * sample[2]), y = (idx / sample[2]) % sample[1], z = idx % sample[2];
      const V3<N> pos{N(x), N(y), N(z)};
      const V3<N> a = (minExtent + (pos * step)) * scale;
      const uint64_t zIndex = mortonEncode(uint64_t(pos.x / res), uint64_t(pos.y / res), uint64_t(pos.z / res));
      const uint64_t zX = mortonDecode(zIndex, 0), zY = mortonDecode(zIndex, 1), zZ = mortonDecode(zIndex, 2);
      if (zX == extent[0] && zY == extent[1] && zZ == extent[2]) return;  // :300-303
      auto cl = [](int v, int hi) { return uint64_t(std::min(std::max(v, 0), hi)); };
      const uint64_t xs[3] = {cl(int(zX) - 1, int(extent[0]) - 1), zX, cl(int(zX) + 1, int(extent[0]) - 1)};
      const uint64_t ys[3] = {cl(int(zY) - 1, int(extent[1]) - 1), zY, cl(int(zY) + 1, int(extent[1]) - 1)};
      const uint64_t zs[3] = {cl(int(zZ) - 1, int(extent[2]) - 1), zZ, cl(int(zZ) + 1, int(extent[2]) - 1)};
      N v = 0;
      V3<N> normal{N(0), N(0), N(0)};
      V4<N> colourAcc{N(0), N(0), N(0), N(0)};
      size_t nNeighbours = 0;
      for (int dz = 0; dz < 3; ++dz)
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx) {  // :312-325, clamped cells may repeat at the domain faces
            const uint64_t off = mortonEncode(xs[dx], ys[dy], zs[dz]);
            if (off >= tn) continue;
            const uint64_t s0 = table[off], e0 = (off + 1) < tn ? table[off + 1] : s0;
            for (uint64_t b = s0; b < e0; ++b) {
              if ((type[b] & 1) || (type[b] & 2)) continue;
              const V3<N> l = this->pos[b] - a;
              const N len = std::sqrt(dot(l, l));
              if (!(len < threshold)) continue;
              const N denominator = std::pow(len, particleInfluence);
              v += (particleSize / denominator);
              normal = normal + (l / denominator) * ((-particleInfluence) * particleSize);
              colourAcc = {colourAcc.x + colour[b].x, colourAcc.y + colour[b].y, colourAcc.z + colour[b].z,
                           colourAcc.w + colour[b].w};
              nNeighbours++;
            }
          }
      normal = normal * (N(1) / std::sqrt(dot(normal, normal)));
      latticePN[idx] = {v, normal.x, normal.y, normal.z};
      const N nn = N(nNeighbours);
      latticeC[idx] = {colourAcc.x / nn, colourAcc.y / nn, colourAcc.z / nn, colourAcc.w / nn};
    });
    return 0;
  }

  int mc_emit(const pbf_oracle_params &c, const pbf_oracle_mc &m, uint64_t *nTriangles) {
    const N h = N(c.h), scale = N(c.scale), res = N(m.resolution), isolevel = N(m.isolevel);
    const N step = h / res;
    static const uint64_t CUBE[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    static const int EDGE[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    meshV.clear(), meshN.clear(), meshC.clear();
    if (sample[0] < 2 || sample[1] < 2 || sample[2] < 2) {
      *nTriangles = 0;
      return 0;
    }
    const uint64_t rx = sample[0] - 1, ry = sample[1] - 1, rz = sample[2] - 1;
    for (uint64_t i = 0; i < rx * ry * rz; ++i) {
      const uint64_t px = i / (ry * rz), py = (i / rz) % ry, pz = i % rz;  // utils::to3d
      N values[8];
      V3<N> offs[8], nrm[8];
      V4<N> cols[8];
      unsigned ci = 0;
      for (int k = 0; k < 8; ++k) {
        const uint64_t ox = px + CUBE[k][0], oy = py + CUBE[k][1], oz = pz + CUBE[k][2];
        const size_t idx = ox * sample[1] * sample[2] + oy * sample[2] + oz;
        const V4<N> point = latticePN[idx];
        values[k] = point.x;
        offs[k] = (minExtent + (V3<N>{N(ox), N(oy), N(oz)} * step)) * scale;
        nrm[k] = {point.y, point.z, point.w};
        cols[k] = latticeC[idx];
        if (values[k] < isolevel) ci |= 1u << k;
      }
      if (kMcEdgeTable[ci] == 0) continue;
      V3<N> ts[12], ns[12];
      V4<N> cs[12];
      for (int e = 0; e < 12; ++e)
        if (kMcEdgeTable[ci] & (1u << e)) {
          const int f = EDGE[e][0], t = EDGE[e][1];
          const N w = (isolevel - values[f]) / (values[t] - values[f]);  // utils::scale
          auto mix = [&](N a, N b) { return a * (N(1) - w) + b * w; };
          ts[e] = {mix(offs[f].x, offs[t].x), mix(offs[f].y, offs[t].y), mix(offs[f].z, offs[t].z)};
          ns[e] = {mix(nrm[f].x, nrm[t].x), mix(nrm[f].y, nrm[t].y), mix(nrm[f].z, nrm[t].z)};
          cs[e] = {mix(cols[f].x, cols[t].x), mix(cols[f].y, cols[t].y), mix(cols[f].z, cols[t].z),
                   mix(cols[f].w, cols[t].w)};
        }
      for (int k = 0; kMcTriTable[ci][k] != 255; ++k) {
        const int e = kMcTriTable[ci][k];
        meshV.push_back(ts[e]), meshN.push_back(ns[e]), meshC.push_back(cs[e]);
      }
    }
    *nTriangles = meshV.size() / 3;
    return 0;
  }

  int finalise(const pbf_oracle_params &c) {
    // ompsph.hpp:256-264
    const size_t cnt = n();
    const N dt = N(c.dt), scale = N(c.scale);
    foreach_1d(c.threads, cnt, [&](size_t a) {
      if (type[a] != 0) return;
      const V3<N> deltaX = pStar[a] - pos[a] / scale;
      pos[a] = pStar[a] * scale;
      vel[a] = (deltaX * (N(1) / dt) + vel[a]) * N(VD);
    });
    extras(c, vel);
    return 0;
  }

  int step(const pbf_oracle_params &c) {
    if (n() == 0) return 0;  // "Particles depleted" (ompsph.hpp:122-126)
    predict(c);
    sort(c);
    grid_table(c);
    diffuse(c);
    for (uint64_t itr = 0; itr < c.iteration; ++itr) {
      lambda_stage(c);
      delta_stage(c);
    }
    finalise(c);
    return 0;
  }
};

template <typename F> auto dispatch(pbf_oracle *o, F &&f) {
  if (o->fp64) return f(*static_cast<Oracle<double> *>(o));
  return f(*static_cast<Oracle<float> *>(o));
}
template <typename F> auto dispatch(const pbf_oracle *o, F &&f) {
  if (o->fp64) return f(*static_cast<const Oracle<double> *>(o));
  return f(*static_cast<const Oracle<float> *>(o));
}

inline size_t icbrt(size_t v) {
  size_t r = static_cast<size_t>(std::cbrt(static_cast<double>(v)));
  while ((r + 1) * (r + 1) * (r + 1) <= v) ++r;
  while (r * r * r > v) --r;
  return r;
}

// sph.hpp:127-145 — len^3 lattice, x outer, z inner, ids sequential, mass 1, v 0
template <typename N>
uint64_t makeCube(uint64_t offset, N spacing, size_t count, V3<N> origin, V4<N> col, uint64_t *id, N *mass, N *pos,
                  N *vel, N *colour, size_t &w) {
  const auto len = static_cast<size_t>(std::cbrt(count));  // same truncation hazard as sph.hpp:134
  for (size_t x = 0; x < len; ++x)
    for (size_t y = 0; y < len; ++y)
      for (size_t z = 0; z < len; ++z) {
        if (id) {
          const V3<N> p = (V3<N>{N(x), N(y), N(z)} * spacing) + origin;
          id[w] = offset;
          mass[w] = N(1.0);
          pos[3 * w] = p.x, pos[3 * w + 1] = p.y, pos[3 * w + 2] = p.z;
          vel[3 * w] = vel[3 * w + 1] = vel[3 * w + 2] = N(0);
          colour[4 * w] = col.x, colour[4 * w + 1] = col.y, colour[4 * w + 2] = col.z, colour[4 * w + 3] = col.w;
        }
        ++offset;
        ++w;
      }
  return offset;
}

template <typename N>
size_t sceneCubes(size_t count, uint64_t *id, void *mass, void *pos, void *vel, void *colour) {
  // sph.hpp:160-166
  size_t w = 0;
  uint64_t tag = 0;
  tag = makeCube<N>(tag, N(22.f), count / 2, V3<N>{N(100), N(0), N(100)}, V4<N>{N(0), N(0.1), N(0.8), N(1)}, id,
                    (N *)mass, (N *)pos, (N *)vel, (N *)colour, w);
  tag = makeCube<N>(tag, N(22.f), count / 2, V3<N>{N(600), N(0), N(600)}, V4<N>{N(0.1), N(0.8), N(0.1), N(1)}, id,
                    (N *)mass, (N *)pos, (N *)vel, (N *)colour, w);
  return w;
}

template <typename N>
size_t sceneDambreak(size_t nominal, uint64_t *id, void *mass_, void *pos_, void *vel_, void *colour_,
                     double *box_side) {
  // SURVEY.md §8d "scene dam-break" (ours; the reference has no such scene): nx = nz = icbrt(n/2),
  // ny = 2 nx, spacing 22 (sph.hpp:165), L = 50*ceil(2.5*nx*22/50 + 4); the column sits 100 world
  // units from the x=0 and z=0 walls and from the +Y floor (gravity is +Y, sph.hpp:171).
  const size_t nx = icbrt(nominal / 2), ny = 2 * nx, nz = nx;
  const double L = 50.0 * std::ceil(2.5 * double(nx) * 22.0 / 50.0 + 4.0);
  if (box_side) *box_side = L;
  N *mass = (N *)mass_, *pos = (N *)pos_, *vel = (N *)vel_, *colour = (N *)colour_;
  const N spacing = N(22.f);
  const V3<N> origin{N(100), N(L - 100.0 - double(ny - 1) * 22.0), N(100)};
  size_t w = 0;
  for (size_t x = 0; x < nx; ++x)
    for (size_t y = 0; y < ny; ++y)
      for (size_t z = 0; z < nz; ++z) {
        if (id) {
          const V3<N> p = (V3<N>{N(x), N(y), N(z)} * spacing) + origin;
          id[w] = w;
          mass[w] = N(1.0);
          pos[3 * w] = p.x, pos[3 * w + 1] = p.y, pos[3 * w + 2] = p.z;
          vel[3 * w] = vel[3 * w + 1] = vel[3 * w + 2] = N(0);
          colour[4 * w] = N(0), colour[4 * w + 1] = N(0.1), colour[4 * w + 2] = N(0.8), colour[4 * w + 3] = N(1);
        }
        ++w;
      }
  return w;
}

}  // namespace

extern "C" {

pbf_oracle *pbf_oracle_create(int fp64) {
  pbf_oracle *o = fp64 ? static_cast<pbf_oracle *>(new Oracle<double>()) : static_cast<pbf_oracle *>(new Oracle<float>());
  o->fp64 = fp64 ? 1 : 0;
  return o;
}
void pbf_oracle_destroy(pbf_oracle *o) { delete o; }

int pbf_oracle_set_particles(pbf_oracle *o, size_t n, const uint64_t *id, const uint8_t *type, const void *mass,
                             const void *pos, const void *vel, const void *colour) {
  return dispatch(o, [&](auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    s.id.assign(id, id + n);
    s.type.assign(type, type + n);
    s.mass.assign((const N *)mass, (const N *)mass + n);
    s.pos.resize(n), s.vel.resize(n), s.colour.resize(n);
    std::memcpy(s.pos.data(), pos, n * 3 * sizeof(N));
    std::memcpy(s.vel.data(), vel, n * 3 * sizeof(N));
    std::memcpy(s.colour.data(), colour, n * 4 * sizeof(N));
    s.zIndex.assign(n, 0);
    s.pStar.assign(n, {});
    s.deltaP.assign(n, {});
    s.lambda.assign(n, N(0));
    return 0;
  });
}
size_t pbf_oracle_count(const pbf_oracle *o) {
  return dispatch(o, [&](const auto &s) { return s.n(); });
}
int pbf_oracle_get_particles(const pbf_oracle *o, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel,
                             void *colour) {
  return dispatch(o, [&](const auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    const size_t n = s.n();
    if (id) std::memcpy(id, s.id.data(), n * 8);
    if (type) std::memcpy(type, s.type.data(), n);
    if (mass) std::memcpy(mass, s.mass.data(), n * sizeof(N));
    if (pos) std::memcpy(pos, s.pos.data(), n * 3 * sizeof(N));
    if (vel) std::memcpy(vel, s.vel.data(), n * 3 * sizeof(N));
    if (colour) std::memcpy(colour, s.colour.data(), n * 4 * sizeof(N));
    return 0;
  });
}

#define STAGE(name, call)                                           \
  int name(pbf_oracle *o, const pbf_oracle_params *p) {             \
    return dispatch(o, [&](auto &s) { return s.call(*p); });        \
  }
STAGE(pbf_oracle_step, step)
STAGE(pbf_oracle_predict, predict)
STAGE(pbf_oracle_sort, sort)
STAGE(pbf_oracle_grid_table, grid_table)
STAGE(pbf_oracle_diffuse, diffuse)
STAGE(pbf_oracle_lambda, lambda_stage)
STAGE(pbf_oracle_delta, delta_stage)
STAGE(pbf_oracle_finalise, finalise)
#undef STAGE

int pbf_oracle_get_keys(const pbf_oracle *o, uint64_t *keys) {
  return dispatch(o, [&](const auto &s) {
    std::memcpy(keys, s.zIndex.data(), s.n() * 8);
    return 0;
  });
}
int pbf_oracle_get_pstar(const pbf_oracle *o, void *pstar) {
  return dispatch(o, [&](const auto &s) {
    std::memcpy(pstar, s.pStar.data(), s.n() * sizeof(s.pStar[0]));
    return 0;
  });
}
int pbf_oracle_get_lambda(const pbf_oracle *o, void *lambda) {
  return dispatch(o, [&](const auto &s) {
    std::memcpy(lambda, s.lambda.data(), s.n() * sizeof(s.lambda[0]));
    return 0;
  });
}
size_t pbf_oracle_table_size(const pbf_oracle *o) {
  return dispatch(o, [&](const auto &s) { return s.table.size(); });
}
int pbf_oracle_get_table(const pbf_oracle *o, uint64_t *table) {
  return dispatch(o, [&](const auto &s) {
    std::memcpy(table, s.table.data(), s.table.size() * 8);
    return 0;
  });
}
int pbf_oracle_get_extent(const pbf_oracle *o, uint64_t extent[3], void *min_extent) {
  return dispatch(o, [&](const auto &s) {
    for (int i = 0; i < 3; ++i) extent[i] = s.extent[i];
    if (min_extent) std::memcpy(min_extent, &s.minExtent, sizeof(s.minExtent));
    return 0;
  });
}
int pbf_oracle_candidate_stats(const pbf_oracle *o, double *mean, uint64_t *max, double *mean_within_h) {
  return dispatch(o, [&](const auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    uint64_t total = 0, mx = 0, within = 0;
    const N h = s.lastH;
    for (size_t a = 0; a < s.n(); ++a) {
      uint64_t c = 0;
      s.foreach_grid(s.zIndex[a], [&](size_t b) {
        ++c;
        if (distance(s.pStar[a], s.pStar[b]) <= h) ++within;
      });
      total += c;
      mx = std::max(mx, c);
    }
    if (mean) *mean = s.n() ? double(total) / double(s.n()) : 0.0;
    if (max) *max = mx;
    if (mean_within_h) *mean_within_h = s.n() ? double(within) / double(s.n()) : 0.0;
    return 0;
  });
}

uint64_t pbf_oracle_morton_encode(uint64_t x, uint64_t y, uint64_t z) { return mortonEncode(x, y, z); }
uint64_t pbf_oracle_morton_decode(uint64_t code, int axis) { return mortonDecode(code, axis); }
void pbf_oracle_neighbour_codes(uint64_t zindex, uint64_t out[27]) { neighbourCodes(zindex, out); }
double pbf_oracle_poly6_factor(int fp64, double h) {
  return fp64 ? double(poly6Factor<double>(h)) : double(poly6Factor<float>(float(h)));
}
double pbf_oracle_spiky_factor(int fp64, double h) {
  return fp64 ? double(spikyKernelFactor<double>(h)) : double(spikyKernelFactor<float>(float(h)));
}

size_t pbf_oracle_scene_cubes(int fp64, size_t count, uint64_t *id, void *mass, void *pos, void *vel, void *colour) {
  return fp64 ? sceneCubes<double>(count, id, mass, pos, vel, colour)
              : sceneCubes<float>(count, id, mass, pos, vel, colour);
}
size_t pbf_oracle_scene_dambreak(int fp64, size_t nominal, uint64_t *id, void *mass, void *pos, void *vel,
                                 void *colour, double *box_side) {
  return fp64 ? sceneDambreak<double>(nominal, id, mass, pos, vel, colour, box_side)
              : sceneDambreak<float>(nominal, id, mass, pos, vel, colour, box_side);
}
void pbf_oracle_motion_offset(int fp64, uint64_t frame, double out[3]) {
  // sph.hpp:147-158: computed in float, `* 0.3` promotes the z term to double before N()
  const float offsetScale = 300.f, offsetRate = 20.f;
  const double ox = double(std::sin(float(frame) / offsetRate) * offsetScale);
  const double oz = double(std::cos(float(frame) / offsetRate) * offsetScale) * 0.3;
  out[0] = fp64 ? ox : double(float(ox));
  out[1] = 0.0;
  out[2] = fp64 ? oz : double(float(oz));
}

int pbf_oracle_surface(pbf_oracle *o, const pbf_oracle_params *p, const pbf_oracle_mc *m, uint64_t *n_triangles) {
  return dispatch(o, [&](auto &s) {
    s.mc_field(*p, *m);
    return s.mc_emit(*p, *m, n_triangles);
  });
}
int pbf_oracle_surface_from_lattice(pbf_oracle *o, const pbf_oracle_params *p, const pbf_oracle_mc *m,
                                    const uint64_t sample[3], const void *pn, const void *c, uint64_t *n_triangles) {
  return dispatch(o, [&](auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    for (int k = 0; k < 3; ++k) s.sample[k] = sample[k];
    const size_t n = sample[0] * sample[1] * sample[2];
    s.latticePN.resize(n), s.latticeC.resize(n);
    std::memcpy(s.latticePN.data(), pn, n * 4 * sizeof(N));
    std::memcpy(s.latticeC.data(), c, n * 4 * sizeof(N));
    return s.mc_emit(*p, *m, n_triangles);
  });
}
int pbf_oracle_get_lattice(const pbf_oracle *o, uint64_t sample[3], void *pn, void *c) {
  return dispatch(o, [&](const auto &s) {
    for (int k = 0; k < 3; ++k) sample[k] = s.sample[k];
    if (pn) std::memcpy(pn, s.latticePN.data(), s.latticePN.size() * sizeof(s.latticePN[0]));
    if (c) std::memcpy(c, s.latticeC.data(), s.latticeC.size() * sizeof(s.latticeC[0]));
    return 0;
  });
}
int pbf_oracle_get_mesh(const pbf_oracle *o, void *vs, void *ns, void *cs) {
  return dispatch(o, [&](const auto &s) {
    if (vs) std::memcpy(vs, s.meshV.data(), s.meshV.size() * sizeof(s.meshV[0]));
    if (ns) std::memcpy(ns, s.meshN.data(), s.meshN.size() * sizeof(s.meshN[0]));
    if (cs) std::memcpy(cs, s.meshC.data(), s.meshC.size() * sizeof(s.meshC[0]));
    return 0;
  });
}

int pbf_oracle_set_scratch(pbf_oracle *o, const uint64_t *keys, const void *pstar, const void *lambda) {
  return dispatch(o, [&](auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    const size_t n = s.n();
    if (keys) s.zIndex.assign(keys, keys + n);
    if (pstar) std::memcpy(s.pStar.data(), pstar, n * 3 * sizeof(N));
    if (lambda) std::memcpy(s.lambda.data(), lambda, n * sizeof(N));
    return 0;
  });
}

void pbf_oracle_set_pow4(pbf_oracle *o, int on) {
  dispatch(o, [&](auto &s) { s.pow4 = on != 0; return 0; });
}

// ---- scene dynamics on the host side of advance() (ompsph.hpp:91-126, 167-186) ------------------------------
// sources[k] = {centre.xyz, velocity.xyz, colour.rgba, rate} (11 doubles), tags[k] = Source::tag
int pbf_oracle_scene_emit(pbf_oracle *o, double h, double scale, size_t n_sources, const uint64_t *tags,
                          const double *sources) {
  return dispatch(o, [&](auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    const N spacing = (N(h) * N(scale) / 2);  // ompsph.hpp:93
    for (size_t k = 0; k < n_sources; ++k) {
      const double *q = sources + 11 * k;
      const V3<N> centre{N(q[0]), N(q[1]), N(q[2])}, velocity{N(q[3]), N(q[4]), N(q[5])};
      const V4<N> colour{N(q[6]), N(q[7]), N(q[8]), N(q[9])};
      const N size = std::sqrt(static_cast<N>(q[10]));  // ompsph.hpp:95
      const size_t width = size_t(std::floor(size)), depth = size_t(std::ceil(size));
      // offset = centre - (V3(width, 0, depth) * 0.5 * spacing)                                   ompsph.hpp:98
      const V3<N> half{N(width) * N(0.5) * spacing, N(0) * N(0.5) * spacing, N(depth) * N(0.5) * spacing};
      const V3<N> offset = centre - half;
      for (size_t x = 0; x < width; ++x)
        for (size_t z = 0; z < depth; ++z) {
          const V3<N> step{N(x) * spacing, N(0) * spacing, N(z) * spacing};
          s.id.push_back(tags[k]);
          s.type.push_back(0);
          s.mass.push_back(N(1));
          s.pos.push_back(offset + step);
          s.vel.push_back(velocity);
          s.colour.push_back(colour);
        }
    }
    const size_t n = s.id.size();
    s.zIndex.assign(n, 0), s.pStar.assign(n, {}), s.deltaP.assign(n, {}), s.lambda.assign(n, N(0));
    return 0;
  });
}

// drains[k] = {centre.xyz, width}: fluid strictly closer than `width` to a drain centre is erased, obstacles stay
int pbf_oracle_scene_drain(pbf_oracle *o, size_t n_drains, const double *drains) {
  return dispatch(o, [&](auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    size_t w = 0;
    for (size_t a = 0; a < s.id.size(); ++a) {
      bool gone = false;
      if (s.type[a] != 1)  // ompsph.hpp:109
        for (size_t k = 0; k < n_drains && !gone; ++k) {
          const double *q = drains + 4 * k;
          const V3<N> d = s.pos[a] - V3<N>{N(q[0]), N(q[1]), N(q[2])};  // glm::distance = length(b - a)
          gone = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z) < N(q[3]);
        }
      if (!gone) {
        s.id[w] = s.id[a], s.type[w] = s.type[a], s.mass[w] = s.mass[a], s.pos[w] = s.pos[a], s.vel[w] = s.vel[a];
        s.colour[w] = s.colour[a];
        ++w;
      }
    }
    s.id.resize(w), s.type.resize(w), s.mass.resize(w), s.pos.resize(w), s.vel.resize(w), s.colour.resize(w);
    s.zIndex.assign(w, 0), s.pStar.assign(w, {}), s.deltaP.assign(w, {}), s.lambda.assign(w, N(0));
    return 0;
  });
}

// ids of the FLUID particles in the cell of a query point, in sorted order (ompsph.hpp:167-186); needs sort +
// grid_table of the current step.  Returns the count, writes at most `cap` ids.
size_t pbf_oracle_query(const pbf_oracle *o, const pbf_oracle_params *p, const double point[3], uint64_t *out, size_t cap) {
  return dispatch(o, [&](const auto &s) -> size_t {
    using N = std::decay_t<decltype(s.mass[0])>;
    const N h = N(p->h), scale = N(p->scale);
    const V3<N> scaled = V3<N>{N(point[0]), N(point[1]), N(point[2])} / scale - s.minExtent;
    using S = std::decay_t<decltype(s)>;  // zCurveGridIndexAtCoordAt (sph.hpp:198-201)
    const uint64_t z = mortonEncode(S::cellCoord(scaled.x / h), S::cellCoord(scaled.y / h), S::cellCoord(scaled.z / h));
    const uint64_t tn = s.table.size();
    size_t k = 0;
    if (z < tn && z + 1 < tn)
      for (uint64_t a = s.table[z]; a < s.table[z + 1]; ++a) {
        if (s.type[a] != 0) continue;
        if (k < cap) out[k] = s.id[a];
        ++k;
      }
    return k;
  });
}

// ---- the opt-in extras as stages of their own (the slab twin exchanges ghost data between them) ------------
int pbf_oracle_vorticity(pbf_oracle *o, const pbf_oracle_params *p) {
  return dispatch(o, [&](auto &s) { s.vorticity_omega(*p); return 0; });
}
int pbf_oracle_vorticity_force(pbf_oracle *o, const pbf_oracle_params *p) {
  return dispatch(o, [&](auto &s) { s.vorticity_force(*p); return 0; });
}
int pbf_oracle_xsph(pbf_oracle *o, const pbf_oracle_params *p) {
  return dispatch(o, [&](auto &s) { s.xsph(*p); return 0; });
}
// which: 0 = velocity, 1 = vorticity; n x 3 values of N
int pbf_oracle_get_vec(const pbf_oracle *o, int which, void *out) {
  return dispatch(o, [&](const auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    const auto &v = which ? s.omega : s.vel;
    if (v.size() != s.n()) return 1;
    std::memcpy(out, v.data(), s.n() * 3 * sizeof(N));
    return 0;
  });
}
int pbf_oracle_set_vec(pbf_oracle *o, int which, const void *in) {
  return dispatch(o, [&](auto &s) {
    using N = std::decay_t<decltype(s.mass[0])>;
    auto &v = which ? s.omega : s.vel;
    v.resize(s.n());
    std::memcpy(v.data(), in, s.n() * 3 * sizeof(N));
    return 0;
  });
}

const char *pbf_oracle_last_error(void) { return g_err.c_str(); }

}  // extern "C"
