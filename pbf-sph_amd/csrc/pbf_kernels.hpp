// pbf_kernels.hpp — gfx950 device kernels of the PBF-SPH step (wave64, no MFMA: nothing on this
// path is a dense contraction).  Each kernel cites the reference lines whose result it
// reproduces; the decomposition (device counting sort, packed pStar+lambda, Jacobi buffers) is ours.
#pragma once

#include "pbf_common.hpp"

namespace pbf {

constexpr int BLOCK = 256;  // 4 waves of 64

// ------------------------------------------------------------------------------------------------
// predict + Morton key + cell histogram                       (reference: ompsph.hpp:137-154)
// One thread per particle, all streams coalesced (16 B or 32 B per lane).
// pos4 = {position.xyz (world), mass}; vel4 = {velocity.xyz, 0}; pstar = {pStar.xyz, lambda}.
// Obstacles follow the OpenCL backend (ocl/oclsph.cpp:66-69): pStar = position/scale, v untouched.
// ------------------------------------------------------------------------------------------------
template <typename N> __device__ inline int64_t cell_coord(N v) {
  // static_cast<size_t>(v) of sph.hpp:199; negative v is UB there — pinned to what x86-64 emits
  // (truncate to int64, reinterpret); the Morton spread then keeps the low 10 bits.
  return static_cast<int64_t>(v);
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_predict(StepConsts<N> c, const vec4<N> *__restrict__ pos4,
                                                   vec4<N> *__restrict__ vel4, const uint8_t *__restrict__ type,
                                                   const N *__restrict__ wells, vec4<N> *__restrict__ pstar,
                                                   uint32_t *__restrict__ key, uint32_t *__restrict__ count) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= c.n) return;
  const vec4<N> p = pos4[i];
  vec4<N> v = vel4[i];
  N px, py, pz;
  if (c.hasObstacles && type[i] == 1) {
    px = p.x / c.scale, py = p.y / c.scale, pz = p.z / c.scale;
  } else {
    const N mass = p.w;
    N fx = mass * c.force[0], fy = mass * c.force[1], fz = mass * c.force[2];
    for (uint32_t w = 0; w < c.nWells; ++w) {  // ompsph.hpp:141-148
      const N cx = wells[4 * w], cy = wells[4 * w + 1], cz = wells[4 * w + 2], wf = wells[4 * w + 3];
      const N dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
      const N d2 = dx * dx + dy * dy + dz * dz;
      const N dist = sqrt(d2);
      if (dist < N(75)) {
        const N inv = N(1) / sqrt(d2);
        const N dd = dist * dist;
        const N tx = ((dx * inv) * wf * mass) / dd, ty = ((dy * inv) * wf * mass) / dd,
                tz = ((dz * inv) * wf * mass) / dd;
        fx += min(max(tx, N(-10)), N(10));
        fy += min(max(ty, N(-10)), N(10));
        fz += min(max(tz, N(-10)), N(10));
      }
    }
    v.x = fx * c.dt + v.x, v.y = fy * c.dt + v.y, v.z = fz * c.dt + v.z;
    vel4[i] = v;
    px = (v.x * c.dt) + (p.x / c.scale);
    py = (v.y * c.dt) + (p.y / c.scale);
    pz = (v.z * c.dt) + (p.z / c.scale);
  }
  pstar[i] = make_vec4<N>(px, py, pz, N(0));
  const uint32_t k = morton_encode(static_cast<uint32_t>(cell_coord((px - c.minExtent[0]) / c.h)),
                                   static_cast<uint32_t>(cell_coord((py - c.minExtent[1]) / c.h)),
                                   static_cast<uint32_t>(cell_coord((pz - c.minExtent[2]) / c.h)));
  key[i] = k;
  // bucket tableN collects particles outside the table: they are "in no cell" (sph.hpp:206)
  atomicAdd(&count[min(k, c.tableN)], 1u);
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan of the cell histogram = the reference's gridTable (sph.hpp:238-250):
// table[c] = number of particles with key < c = first sorted index with key >= c.
// Three passes (block sums -> scan of sums -> scan + offset), 2048 cells per block.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

__device__ inline uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, total in *total
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *total) {
  __shared__ uint32_t waveSum[BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan(v, lane);
  __syncthreads();  // protect waveSum across back-to-back calls
  if (lane == 63) waveSum[wave] = incl;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w) {
    const uint32_t s = waveSum[w];
    if (w < wave) off += s;
    tot += s;
  }
  *total = tot;
  return off + incl - v;
}

__global__ __launch_bounds__(BLOCK) void k_scan_block_sums(const uint32_t *__restrict__ count, uint32_t len,
                                                           uint32_t *__restrict__ blockSums) {
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= len) {
    const uint4 a = *reinterpret_cast<const uint4 *>(count + base);
    const uint4 b = *reinterpret_cast<const uint4 *>(count + base + 4);
    s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  } else {
    for (int j = 0; j < SCAN_ITEMS; ++j)
      if (base + j < len) s += count[base + j];
  }
  uint32_t total;
  block_excl_scan(s, &total);
  if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

__global__ __launch_bounds__(BLOCK) void k_scan_sums(uint32_t *__restrict__ blockSums, uint32_t nb) {
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? blockSums[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan(v, &total);
    if (i < nb) blockSums[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(BLOCK) void k_scan_apply(const uint32_t *__restrict__ count, uint32_t len,
                                                      const uint32_t *__restrict__ blockSums,
                                                      uint32_t *__restrict__ table) {
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = (base + j < len) ? count[base + j] : 0u;
    s += v[j];
  }
  uint32_t total;
  uint32_t run = block_excl_scan(s, &total) + blockSums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < len) table[base + j] = run;
    run += v[j];
  }
}

// ------------------------------------------------------------------------------------------------
// Counting-sort scatter, made deterministic                      (reference: ompsph.hpp:157-159)
//   pass A: slot inside the cell from an atomic (arbitrary order), records the source index;
//           atomicSub returns the histogram to zero for the next step (no memset);
//   pass B: each particle's final rank inside its cell = number of cell-mates with a smaller
//           source index, i.e. a STABLE sort by key — run-to-run reproducible — then the whole
//           record moves to its sorted slot (writes stay inside one cell's short range).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_scatter_slots(uint32_t n, uint32_t tableN, const uint32_t *__restrict__ key,
                                                         const uint32_t *__restrict__ table,
                                                         uint32_t *__restrict__ count,
                                                         uint32_t *__restrict__ permTmp) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t b = min(key[i], tableN);
  const uint32_t r = atomicSub(&count[b], 1u) - 1u;
  permTmp[table[b] + r] = i;
}

template <typename N> struct ParticleArrays {
  vec4<N> *pos4, *vel4, *col4, *pstar;
  uint64_t *id;
  uint8_t *type;
  uint32_t *key;
};

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_rank_move(uint32_t n, uint32_t tableN,
                                                     const uint32_t *__restrict__ permTmp,
                                                     const uint32_t *__restrict__ table, ParticleArrays<N> src,
                                                     ParticleArrays<N> dst) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = permTmp[i];
  const uint32_t k = src.key[s];
  const uint32_t b = min(k, tableN);
  const uint32_t lo = table[b], hi = table[b + 1];
  uint32_t rank = 0;
  if (b < tableN) {
    for (uint32_t j = lo; j < hi; ++j) rank += (permTmp[j] < s) ? 1u : 0u;
  } else {
    // overflow bucket (particles in no cell): keys differ, order by (key, source index) so that the
    // whole array is exactly the stable sort by key the reference's write-back order implies
    for (uint32_t j = lo; j < hi; ++j) {
      const uint32_t sj = permTmp[j];
      const uint32_t kj = src.key[sj];
      rank += (kj < k || (kj == k && sj < s)) ? 1u : 0u;
    }
  }
  const uint32_t d = lo + rank;
  dst.pos4[d] = src.pos4[s];
  dst.vel4[d] = src.vel4[s];
  dst.col4[d] = src.col4[s];
  dst.pstar[d] = src.pstar[s];
  dst.id[d] = src.id[s];
  dst.type[d] = src.type[s];
  dst.key[d] = k;
}

// ------------------------------------------------------------------------------------------------
// 27-cell walk in the reference's order (sph.hpp:203-236): x fastest, then y, then z; a code
// >= tableN is skipped; the last table entry yields an empty range.  Neighbour codes come from
// dilated-integer +-1 on the key (equal to decode / +-1 in size_t / re-encode of curves.h:
// both keep the low 10 bits per axis, so x-1 at x = 0 becomes 1023 in both).
// ------------------------------------------------------------------------------------------------
struct Neigh {
  uint32_t xs[3], ys[3], zs[3];
};
__device__ inline Neigh neigh_codes(uint32_t key) {
  Neigh nb;
  const uint32_t xm = key & MORTON_X, ym = key & MORTON_Y, zm = key & MORTON_Z;
  nb.xs[0] = (xm - 1u) & MORTON_X, nb.xs[1] = xm, nb.xs[2] = ((xm | ~MORTON_X) + 1u) & MORTON_X;
  nb.ys[0] = (ym - 2u) & MORTON_Y, nb.ys[1] = ym, nb.ys[2] = ((ym | ~MORTON_Y) + 2u) & MORTON_Y;
  nb.zs[0] = (zm - 4u) & MORTON_Z, nb.zs[1] = zm, nb.zs[2] = ((zm | ~MORTON_Z) + 4u) & MORTON_Z;
  return nb;
}

template <typename F>
__device__ inline void for_each_candidate(uint32_t key, const uint32_t *__restrict__ table, uint32_t tableN, F &&f) {
  const Neigh nb = neigh_codes(key);
#pragma unroll 1
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll 1
    for (int dy = 0; dy < 3; ++dy) {
      const uint32_t yz = nb.ys[dy] | nb.zs[dz];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const uint32_t code = nb.xs[dx] | yz;
        if (code >= tableN) continue;
        const uint32_t start = table[code];
        const uint32_t end = (code + 1u) < tableN ? table[code + 1u] : start;
        for (uint32_t b = start; b < end; ++b) f(b);
      }
    }
}

// Pair terms shared by lambda / delta.  PRECISE follows the oracle's operation order with IEEE
// sqrt and divide; FAST uses one v_rsq (the reference's own builds are -Ofast / native_divide).
template <typename N, bool FAST> struct PairGeom {
  N dx, dy, dz;  // a - b
  N r;
  N hr2_over_r;  // (h - r)^2 / r, valid when inSpiky
  bool inH, inSpiky;
};
template <typename N, bool FAST>
__device__ inline PairGeom<N, FAST> pair_geom(const vec4<N> &a, const vec4<N> &b, N h) {
  PairGeom<N, FAST> g;
  const N bx = b.x - a.x, by = b.y - a.y, bz = b.z - a.z;  // distance(a,b) = length(b - a)
  const N d2 = bx * bx + by * by + bz * bz;
  g.dx = a.x - b.x, g.dy = a.y - b.y, g.dz = a.z - b.z;
  if constexpr (FAST) {
    const N rinv = rsqrt(d2);
    g.r = d2 > N(0) ? d2 * rinv : N(0);
    g.inH = g.r <= h;
    g.inSpiky = g.inH && g.r >= N(EPSILON);
    const N hr = h - g.r;
    g.hr2_over_r = (hr * hr) * rinv;
  } else {
    g.r = sqrt(d2);
    g.inH = g.r <= h;
    g.inSpiky = g.inH && g.r >= N(EPSILON);
    const N hr = h - g.r;
    g.hr2_over_r = (hr * hr) / g.r;
  }
  return g;
}

// ------------------------------------------------------------------------------------------------
// diffuse (ompsph.hpp:188-207), Jacobi like the OpenCL kernel (ocl/oclsph_kernel.h:67-93)
// ------------------------------------------------------------------------------------------------
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_diffuse(StepConsts<N> c, const uint32_t *__restrict__ key,
                                                   const uint32_t *__restrict__ table,
                                                   const uint8_t *__restrict__ type,
                                                   const vec4<N> *__restrict__ colIn, vec4<N> *__restrict__ colOut) {
  const uint32_t a = blockIdx.x * BLOCK + threadIdx.x;
  if (a >= c.n) return;
  const vec4<N> ca = colIn[a];
  if (c.hasObstacles && type[a] == 1) {
    colOut[a] = ca;
    return;
  }
  N mx = 0, my = 0, mz = 0, mw = 0;
  int nn = 0;
  for_each_candidate(key[a], table, c.tableN, [&](uint32_t b) {
    if (c.hasObstacles && type[b] == 1) return;
    const vec4<N> cb = colIn[b];
    mx += cb.x, my += cb.y, mz += cb.z, mw += cb.w;
    ++nn;
  });
  vec4<N> out = ca;
  if (nn != 0) {
    const N fn = N(nn), t = c.diffuseT;
    auto one = [&](N x, N m) {
      const N y = (m / fn) * N(1.33);
      const N o = x * (N(1) - t) + y * t;
      return min(max(o, N(0.03)), N(1.0));
    };
    out = make_vec4<N>(one(ca.x, mx), one(ca.y, my), one(ca.z, mz), one(ca.w, mw));
  }
  colOut[a] = out;
}

// ------------------------------------------------------------------------------------------------
// lambda (ompsph.hpp:217-232): rho = sum m_a poly6; g = sum grad spiky / rho0;
// lambda = -(rho/rho0 - 1) / (|g|^2 + 600).  Written into pstar[a].w (only xyz is read here).
// ------------------------------------------------------------------------------------------------
template <typename N, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_lambda(StepConsts<N> c, const uint32_t *__restrict__ key,
                                                  const uint32_t *__restrict__ table,
                                                  const uint8_t *__restrict__ type, const vec4<N> *__restrict__ pos4,
                                                  vec4<N> *__restrict__ pstar) {
  const uint32_t a = blockIdx.x * BLOCK + threadIdx.x;
  if (a >= c.n) return;
  if (c.hasObstacles && type[a] == 1) {
    pstar[a].w = N(0);
    return;
  }
  const vec4<N> pa = pstar[a];
  const N mass = pos4[a].w;
  N gx = 0, gy = 0, gz = 0, rho = 0;
  for_each_candidate(key[a], table, c.tableN, [&](uint32_t b) {
    const vec4<N> pb = pstar[b];
    const auto g = pair_geom<N, FAST>(pa, pb, c.h);
    if (g.inSpiky) {
      const N s = c.spikyFactor * g.hr2_over_r;
      gx += (g.dx * s) * N(RHO_RECIP), gy += (g.dy * s) * N(RHO_RECIP), gz += (g.dz * s) * N(RHO_RECIP);
    }
    if (g.inH) {
      const N d = (c.h * c.h) - g.r * g.r;
      rho += mass * (c.poly6Factor * (d * d * d));
    }
  });
  const N norm2 = gx * gx + gy * gy + gz * gz;
  const N Ci = rho / N(RHO) - N(1);
  pstar[a].w = -Ci / (norm2 + N(CFM_EPSILON));
}

// ------------------------------------------------------------------------------------------------
// delta-p + clamp (ompsph.hpp:235-248), Jacobi: reads pstarIn (xyz + lambda), writes pstarOut.
// ------------------------------------------------------------------------------------------------
template <typename N, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_delta(StepConsts<N> c, const uint32_t *__restrict__ key,
                                                 const uint32_t *__restrict__ table,
                                                 const uint8_t *__restrict__ type,
                                                 const vec4<N> *__restrict__ pstarIn,
                                                 vec4<N> *__restrict__ pstarOut) {
  const uint32_t a = blockIdx.x * BLOCK + threadIdx.x;
  if (a >= c.n) return;
  const vec4<N> pa = pstarIn[a];
  if (c.hasObstacles && type[a] == 1) {
    pstarOut[a] = pa;
    return;
  }
  N ax = 0, ay = 0, az = 0;
  for_each_candidate(key[a], table, c.tableN, [&](uint32_t b) {
    const vec4<N> pb = pstarIn[b];
    const auto g = pair_geom<N, FAST>(pa, pb, c.h);
    if (g.inSpiky) {  // outside it the gradient is zero, so corr / factor are irrelevant
      const N d = (c.h * c.h) - g.r * g.r;
      const N q = (c.poly6Factor * (d * d * d)) / c.p6DeltaQ;
      const N q2 = q * q;
      const N corr = N(-CorrK) * (q2 * q2);
      const N factor = (pa.w + pb.w + corr) / N(RHO);
      const N s = c.spikyFactor * g.hr2_over_r;
      ax += (g.dx * s) * factor, ay += (g.dy * s) * factor, az += (g.dz * s) * factor;
    }
  });
  N x = (pa.x + ax) * c.scale, y = (pa.y + ay) * c.scale, z = (pa.z + az) * c.scale;
  x = min(c.maxB[0], max(c.minB[0], x));
  y = min(c.maxB[1], max(c.minB[1], y));
  z = min(c.maxB[2], max(c.minB[2], z));
  pstarOut[a] = make_vec4<N>(x / c.scale, y / c.scale, z / c.scale, pa.w);
}

// ------------------------------------------------------------------------------------------------
// finalise (ompsph.hpp:256-264): pure stream, 48 B in / 32 B out per particle (fp32)
// ------------------------------------------------------------------------------------------------
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_finalise(StepConsts<N> c, const uint8_t *__restrict__ type,
                                                    const vec4<N> *__restrict__ pstar, vec4<N> *__restrict__ pos4,
                                                    vec4<N> *__restrict__ vel4) {
  const uint32_t a = blockIdx.x * BLOCK + threadIdx.x;
  if (a >= c.n) return;
  if (c.hasObstacles && type[a] == 1) return;
  const vec4<N> ps = pstar[a];
  vec4<N> p = pos4[a];
  vec4<N> v = vel4[a];
  const N dxx = ps.x - p.x / c.scale, dyy = ps.y - p.y / c.scale, dzz = ps.z - p.z / c.scale;
  const N invdt = N(1) / c.dt;
  p.x = ps.x * c.scale, p.y = ps.y * c.scale, p.z = ps.z * c.scale;
  v.x = (dxx * invdt + v.x) * N(VD), v.y = (dyy * invdt + v.y) * N(VD), v.z = (dzz * invdt + v.z) * N(VD);
  pos4[a] = p;
  vel4[a] = v;
}

// ------------------------------------------------------------------------------------------------
// AoS <-> SoA on the device for the advance() shim (std::vector<Particle>, sph.hpp:36-54)
// ------------------------------------------------------------------------------------------------
struct AosLayout {
  uint32_t stride, off_id, off_type, off_mass, off_pos, off_vel, off_colour;
};

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_unpack_aos(uint32_t n, const uint8_t *__restrict__ aos, AosLayout l,
                                                      ParticleArrays<N> dst) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint8_t *p = aos + size_t(i) * l.stride;
  const N *pos = reinterpret_cast<const N *>(p + l.off_pos);
  const N *vel = reinterpret_cast<const N *>(p + l.off_vel);
  const N *col = reinterpret_cast<const N *>(p + l.off_colour);
  dst.id[i] = *reinterpret_cast<const uint64_t *>(p + l.off_id);
  dst.type[i] = p[l.off_type];
  dst.pos4[i] = make_vec4<N>(pos[0], pos[1], pos[2], *reinterpret_cast<const N *>(p + l.off_mass));
  dst.vel4[i] = make_vec4<N>(vel[0], vel[1], vel[2], N(0));
  dst.col4[i] = make_vec4<N>(col[0], col[1], col[2], col[3]);
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_pack_aos(uint32_t n, uint8_t *__restrict__ aos, AosLayout l,
                                                    ParticleArrays<N> src) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint8_t *p = aos + size_t(i) * l.stride;
  N *pos = reinterpret_cast<N *>(p + l.off_pos);
  N *vel = reinterpret_cast<N *>(p + l.off_vel);
  N *col = reinterpret_cast<N *>(p + l.off_colour);
  const vec4<N> P = src.pos4[i], V = src.vel4[i], C = src.col4[i];
  *reinterpret_cast<uint64_t *>(p + l.off_id) = src.id[i];
  p[l.off_type] = src.type[i];
  *reinterpret_cast<N *>(p + l.off_mass) = P.w;
  pos[0] = P.x, pos[1] = P.y, pos[2] = P.z;
  vel[0] = V.x, vel[1] = V.y, vel[2] = V.z;
  col[0] = C.x, col[1] = C.y, col[2] = C.z, col[3] = C.w;
}

}  // namespace pbf
