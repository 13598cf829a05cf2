// ref_grid.cpp — builds oracle/_ref/libref_grid.so from the REFERENCE'S OWN headers, where they lie.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit contains no reference source: it #includes
// /root/reference/src/sph.hpp (which includes src/curves.h) by path and exports thin C wrappers so
// tests can compare the oracle restatement (pbf_oracle.cpp) and the HIP path with the real thing.
// Those two headers are glm-free (sph.hpp is generic over the vector template V), so no stand-in
// for a missing library is involved; `Vec` below is the template argument a caller of the
// reference's public API must supply anyway.  src/omp/ompsph.hpp is NOT built (it needs glm).
//
// Covered: Morton encode/decode (curves.h:46-88), zCurveGridIndexAtCoordAt (sph.hpp:198-201),
// makeGridTable (sph.hpp:238-250), foreach_grid (sph.hpp:203-236), kernel factors (sph.hpp:251-253),
// scene factory + box motion (sph.hpp:127-186).
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <tuple>

#include "sph.hpp"  // -I/root/reference/src

namespace {
template <size_t L, typename C> struct Vec;
template <typename C> struct Vec<3, C> {
  C x{}, y{}, z{};
  Vec() = default;
  template <typename A, typename B, typename D> Vec(A a, B b, D d) : x(C(a)), y(C(b)), z(C(d)) {}
  Vec &operator+=(const Vec &o) {
    x += o.x, y += o.y, z += o.z;
    return *this;
  }
  bool operator==(const Vec &o) const { return x == o.x && y == o.y && z == o.z; }
};
template <typename C> struct Vec<4, C> {
  C x{}, y{}, z{}, w{};
  Vec() = default;
  template <typename A, typename B, typename D, typename E> Vec(A a, B b, D d, E e) : x(C(a)), y(C(b)), z(C(d)), w(C(e)) {}
  bool operator==(const Vec &o) const { return x == o.x && y == o.y && z == o.z && w == o.w; }
};
template <typename C> Vec<3, C> operator*(const Vec<3, C> &a, C s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename C> Vec<3, C> operator+(const Vec<3, C> &a, const Vec<3, C> &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

template <typename N>
size_t sceneCubes(size_t count, uint64_t solverIter, double scaling, uint64_t *id, uint8_t *type, N *mass, N *pos,
                  N *vel, N *colour, double *cfg /* dt, scale, iteration, force3, min3, max3 */) {
  auto [mc, config, prepared] = sph::simpleConfigWith2Cubes<size_t, N, Vec>(count, solverIter, N(scaling));
  (void)mc;
  if (cfg) {
    cfg[0] = double(config.dt), cfg[1] = double(config.scale), cfg[2] = double(config.iteration);
    cfg[3] = config.constantForce.x, cfg[4] = config.constantForce.y, cfg[5] = config.constantForce.z;
    cfg[6] = config.minBound.x, cfg[7] = config.minBound.y, cfg[8] = config.minBound.z;
    cfg[9] = config.maxBound.x, cfg[10] = config.maxBound.y, cfg[11] = config.maxBound.z;
  }
  if (id)
    for (size_t i = 0; i < prepared.size(); ++i) {
      const auto &p = prepared[i];
      id[i] = p.id, type[i] = uint8_t(p.type), mass[i] = p.mass;
      pos[3 * i] = p.position.x, pos[3 * i + 1] = p.position.y, pos[3 * i + 2] = p.position.z;
      vel[3 * i] = p.velocity.x, vel[3 * i + 1] = p.velocity.y, vel[3 * i + 2] = p.velocity.z;
      colour[4 * i] = p.colour.x, colour[4 * i + 1] = p.colour.y, colour[4 * i + 2] = p.colour.z,
                 colour[4 * i + 3] = p.colour.w;
    }
  return prepared.size();
}

template <typename N> void motion(uint64_t frame, double out[6]) {
  sph::SphParams<size_t, N, Vec> c{};
  c.minBound = Vec<3, N>(0, 0, 0);
  c.maxBound = Vec<3, N>(1000, 1000, 1000);
  auto w = sph::applyMotionSinXCosZ(c, size_t(frame));
  out[0] = w.minBound.x, out[1] = w.minBound.y, out[2] = w.minBound.z;
  out[3] = w.maxBound.x, out[4] = w.maxBound.y, out[5] = w.maxBound.z;
}
}  // namespace

extern "C" {
uint64_t ref_morton_encode(uint64_t x, uint64_t y, uint64_t z) { return zCurveGridIndexAtCoord(x, y, z); }
uint64_t ref_morton_decode(uint64_t code, int axis) {
  return axis == 0 ? coordAtZCurveGridIndex0(code) : axis == 1 ? coordAtZCurveGridIndex1(code) : coordAtZCurveGridIndex2(code);
}
uint64_t ref_grid_index_at_f32(float x, float y, float z, float h) { return sph::zCurveGridIndexAtCoordAt<float>(x, y, z, h); }
uint64_t ref_grid_index_at_f64(double x, double y, double z, double h) { return sph::zCurveGridIndexAtCoordAt<double>(x, y, z, h); }

// returns table length; out may be NULL to query the length
uint64_t ref_make_grid_table(uint64_t ex, uint64_t ey, uint64_t ez, uint64_t n, const uint64_t *sortedKeys, uint64_t *out) {
  if (!out) return zCurveGridIndexAtCoord(ex, ey, ez);
  auto t = sph::makeGridTable(ex, ey, ez, n, [&](size_t i) -> size_t { return sortedKeys[i]; });
  std::memcpy(out, t.data(), t.size() * sizeof(uint64_t));
  return t.size();
}
// visits in the reference's order; returns the number of candidates (writes at most cap)
uint64_t ref_foreach_grid(uint64_t zIndex, const uint64_t *table, uint64_t tableN, uint64_t *out, uint64_t cap) {
  uint64_t k = 0;
  sph::foreach_grid(size_t(zIndex), table, size_t(tableN), [&](size_t b) {
    if (k < cap) out[k] = b;
    ++k;
  });
  return k;
}
double ref_poly6_factor_f32(float h) { return double(sph::poly6Factor<float>(h)); }
double ref_poly6_factor_f64(double h) { return sph::poly6Factor<double>(h); }
double ref_spiky_factor_f32(float h) { return double(sph::spikyKernelFactor<float>(h)); }
double ref_spiky_factor_f64(double h) { return sph::spikyKernelFactor<double>(h); }

uint64_t ref_scene_cubes_f32(uint64_t count, uint64_t iter, double scaling, uint64_t *id, uint8_t *type, float *mass,
                             float *pos, float *vel, float *colour, double *cfg) {
  return sceneCubes<float>(count, iter, scaling, id, type, mass, pos, vel, colour, cfg);
}
uint64_t ref_scene_cubes_f64(uint64_t count, uint64_t iter, double scaling, uint64_t *id, uint8_t *type, double *mass,
                             double *pos, double *vel, double *colour, double *cfg) {
  return sceneCubes<double>(count, iter, scaling, id, type, mass, pos, vel, colour, cfg);
}
void ref_motion_f32(uint64_t frame, double out[6]) { motion<float>(frame, out); }
void ref_motion_f64(uint64_t frame, double out[6]) { motion<double>(frame, out); }
uint64_t ref_sizeof_partially_advected_f32() { return sizeof(sph::PartiallyAdvected<size_t, float, Vec>); }
}
