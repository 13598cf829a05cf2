// pbf_common.hpp — shared host/device definitions for the gfx950 PBF-SPH hot path.
//
// Constants restate src/sph_constants.h:5-16 of the reference: all `float`, promoted to the
// working type N at the use site (so the fp64 path uses float-rounded constants).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pbf {

constexpr float VD = 0.49f;          // sph_constants.h:5  velocity dampening
constexpr float RHO = 6378.0f;       // sph_constants.h:6  reference density
constexpr float RHO_RECIP = 1.f / RHO;
constexpr float EPSILON = 0.00000001f;
constexpr float CFM_EPSILON = 600.0f;
constexpr float CorrDeltaQ = 0.3f;
constexpr float C_XSPH = 0.00001f;            // sph_constants.h:13 — unused by the reference (opt-in here)
constexpr float VORTICITY_EPSILON = 0.0005f;  // sph_constants.h:14 — unused by the reference (opt-in here)
constexpr float CorrK = 0.0001f;
// CorrN = 4 (sph_constants.h:16): pow(x, 4) is evaluated as (x*x)*(x*x) on the device.

// Dilated-integer masks of the 10-bit-per-axis Morton code (src/curves.h:72-88).
constexpr uint32_t MORTON_X = 0x09249249u;
constexpr uint32_t MORTON_Y = MORTON_X << 1;
constexpr uint32_t MORTON_Z = MORTON_X << 2;

__host__ __device__ inline uint32_t spread10(uint32_t x) {
  // curves.h:73-76 — keeps the low 10 bits of x
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}
__host__ __device__ inline uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
  return spread10(x) | (spread10(y) << 1) | (spread10(z) << 2);
}
__host__ __device__ inline uint32_t compact10(uint32_t v) {
  // inverse of spread10 (curves.h:46-59 does it bit by bit)
  v &= 0x09249249u;
  v = (v | (v >> 2)) & 0x030C30C3u;
  v = (v | (v >> 4)) & 0x0300F00Fu;
  v = (v | (v >> 8)) & 0x030000FFu;
  v = (v | (v >> 16)) & 0x000003FFu;
  return v;
}

template <typename N> struct vec4_of;
template <> struct vec4_of<float> {
  using type = float4;
};
template <> struct vec4_of<double> {
  using type = double4;
};
template <typename N> using vec4 = typename vec4_of<N>::type;

template <typename N> __host__ __device__ inline vec4<N> make_vec4(N x, N y, N z, N w) {
  vec4<N> v;
  v.x = x, v.y = y, v.z = z, v.w = w;
  return v;
}

// Per-step constants, computed on the host in N exactly as the reference computes them
// (ompsph.hpp:132-135, 211-213; sph.hpp:251-253) and passed by value to every kernel.
template <typename N> struct StepConsts {
  N h, dt, scale;
  N force[3];
  N minB[3], maxB[3];
  N minExtent[3];
  N poly6Factor, spikyFactor, p6DeltaQ;
  N diffuseT;  // dt / 750 (ompsph.hpp:203)
  N h2filter;  // h^2 (1 + 1e-5): conservative candidate filter, see maybe_within_h
  uint32_t n;
  uint32_t tableN;  // Morton(extent) (sph.hpp:240)
  uint32_t nWells;
  uint32_t hasObstacles;
  uint32_t xoff;  // slab mode: keys use the x cell coordinate minus xoff (rank-local, compact table); 0 otherwise
  // pbf_slab_step (round 3): predict itself sorts the particles into stayers / leavers / last step's copies
  uint32_t slabOn;           // 0: everything below is ignored
  uint32_t sxlo, sxhi;       // owned cell columns [sxlo, sxhi) in the keys' x frame
  uint32_t sHasL, sHasR;     // is there a slab on that side
};

// particle type bits: sph::Type (src/sph.hpp:15) + the slab decomposition's copies
constexpr uint8_t TYPE_OBSTACLE = 1;
constexpr uint8_t TYPE_GHOST = 2;  // a copy of a neighbouring slab's boundary particle: a candidate, never updated locally
// pbf_slab_step: a slot whose particle has left (a migrant that was packed for the neighbour, or last step's copy).  It
// takes no part in the histogram and the scatter, so the step's sort drops it — no compaction pass of its own.
constexpr uint32_t DEAD_KEY = 0xFFFFFFFFu;

}  // namespace pbf
