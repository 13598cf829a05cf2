// pbf_slab.hpp — device side of the multi-GPU slab decomposition (no reference counterpart: the
// reference is single-device).  The domain is cut into slabs of whole cell columns
// along x; the cell geometry is the global one (every rank passes the same bounds), keys may live in
// a rank-local x frame (pbf_slab_configure: compact per-rank table), and all the single-GPU kernels
// run unchanged.  What is added here:
//   * a deterministic 3-pass "select" (count / scan / emit) that, in array order,
//       mode MIGRATE: compacts the particles that stay, and packs those whose cell column left
//                     the slab into wire records for the left / right neighbour (ghosts of the
//                     previous step are dropped on the way);
//       mode GHOST:   packs a copy of the particles in the slab's first / last column for the
//                     neighbour and remembers which particle each copy came from;
//   * append kernels for what arrives, a histogram kernel for the re-assembled set,
//   * field pack / unpack: after every lambda and delta launch the owners refresh their copies'
//     {pStar, lambda} on the neighbour, addressed through the sort's source->slot map.
// Ghost copies carry type bit 1 (TYPE_GHOST): they are candidates but never updated locally.
#pragma once

#include "pbf_kernels.hpp"

namespace pbf {

constexpr int SEL_ITEMS = 4;
constexpr int SEL_TILE = BLOCK * SEL_ITEMS;

template <typename N> struct MigrantRec {
  vec4<N> pos4, vel4, col4, pstar;
  uint64_t id;
  uint32_t key, type;
};
template <typename N> struct GhostRec {
  vec4<N> pstar, col4;
  uint32_t key, type, pad0, pad1;
};

struct SlabCut {
  uint32_t xlo, xhi;  // owned cell columns [xlo, xhi)
  uint32_t hasLeft, hasRight;
};

enum { SEL_MIGRATE = 0, SEL_GHOST = 1 };
// class ids: MIGRATE: 0 keep, 1 to-left, 2 to-right.  GHOST: 0 copy-for-left, 1 copy-for-right.
template <int MODE> __device__ inline uint32_t slab_classes(uint32_t key, uint8_t type, const SlabCut &s) {
  const uint32_t cx = compact10(key);
  if (MODE == SEL_MIGRATE) {
    if (type & TYPE_GHOST) return 0u;  // last step's copies are dropped
    if (s.hasLeft && cx < s.xlo) return 2u;
    if (s.hasRight && cx >= s.xhi) return 4u;
    return 1u;
  } else {
    uint32_t m = 0;
    if (s.hasLeft && cx == s.xlo) m |= 1u;
    if (s.hasRight && cx + 1u == s.xhi) m |= 2u;
    return m;
  }
}

// pass 1: per-block class counts -> counts[cls * nb + block]
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_sel_count(uint32_t n, SlabCut s, const uint32_t *__restrict__ key,
                                                     const uint8_t *__restrict__ type, uint32_t nb,
                                                     uint32_t *__restrict__ counts) {
  uint32_t c0 = 0, c1 = 0, c2 = 0;
  const uint32_t base = blockIdx.x * SEL_TILE + threadIdx.x * SEL_ITEMS;
#pragma unroll
  for (int j = 0; j < SEL_ITEMS; ++j) {
    const uint32_t i = base + j;
    if (i < n) {
      const uint32_t m = slab_classes<MODE>(key[i], type[i], s);
      c0 += m & 1u, c1 += (m >> 1) & 1u, c2 += (m >> 2) & 1u;
    }
  }
  uint32_t t0, t1, t2;
  block_excl_scan(c0, &t0);
  block_excl_scan(c1, &t1);
  block_excl_scan(c2, &t2);
  if (threadIdx.x == 0) counts[blockIdx.x] = t0, counts[nb + blockIdx.x] = t1, counts[2 * nb + blockIdx.x] = t2;
}

// pass 2: exclusive scan per class over the blocks; totals[cls] = grand total
__global__ __launch_bounds__(BLOCK) void k_sel_scan(uint32_t nb, uint32_t *__restrict__ counts,
                                                    uint32_t *__restrict__ totals) {
  for (int cls = 0; cls < 3; ++cls) {
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += BLOCK) {
      const uint32_t i = base + threadIdx.x;
      const uint32_t v = i < nb ? counts[cls * nb + i] : 0u;
      uint32_t total;
      const uint32_t ex = block_excl_scan(v, &total);
      if (i < nb) counts[cls * nb + i] = carry + ex;
      carry += total;
    }
    if (threadIdx.x == 0) totals[cls] = carry;
  }
}

// pass 3: emit in array order.  A wave takes 64 x SEL_ITEMS consecutive particles of the workgroup's tile in SEL_ITEMS
// rounds of 64 (lane = particle: every load and, but for the holes, every store of a round is one contiguous run — a
// thread per SEL_ITEMS consecutive particles, as the count pass has it, reads 16-byte records at a 64-byte stride); the
// position inside a round is the ballot prefix, the wave's start inside the tile comes from a first sweep over keys and
// types, the tile's start from pass 2.  Same order as the count pass: array order.
template <typename N, int MODE>
__global__ __launch_bounds__(BLOCK) void k_sel_emit(uint32_t n, SlabCut s, ParticleArrays<N> src, ParticleArrays<N> dst,
                                                    uint32_t nb, const uint32_t *__restrict__ bases, void *sendL,
                                                    void *sendR, uint32_t capRecords, uint32_t *__restrict__ srcL,
                                                    uint32_t *__restrict__ srcR) {
  __shared__ uint32_t waveTot[3][BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t waveBase = blockIdx.x * SEL_TILE + wave * 64u * SEL_ITEMS;
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t m[SEL_ITEMS];
  uint32_t t0 = 0, t1 = 0, t2 = 0;  // the wave's class totals (uniform)
#pragma unroll
  for (int j = 0; j < SEL_ITEMS; ++j) {
    const uint32_t i = waveBase + j * 64u + lane;
    m[j] = i < n ? slab_classes<MODE>(src.key[i], src.type[i], s) : 0u;
    t0 += uint32_t(__builtin_popcountll(__ballot((m[j] & 1u) != 0u)));
    t1 += uint32_t(__builtin_popcountll(__ballot((m[j] & 2u) != 0u)));
    t2 += uint32_t(__builtin_popcountll(__ballot((m[j] & 4u) != 0u)));
  }
  if (lane == 0) waveTot[0][wave] = t0, waveTot[1][wave] = t1, waveTot[2][wave] = t2;
  __syncthreads();
  uint32_t p0 = bases[blockIdx.x], p1 = bases[nb + blockIdx.x], p2 = bases[2 * nb + blockIdx.x];
  for (uint32_t w = 0; w < wave; ++w) p0 += waveTot[0][w], p1 += waveTot[1][w], p2 += waveTot[2][w];
#pragma unroll
  for (int j = 0; j < SEL_ITEMS; ++j) {
    const uint32_t i = waveBase + j * 64u + lane;
    const uint64_t b0 = __ballot((m[j] & 1u) != 0u), b1 = __ballot((m[j] & 2u) != 0u), b2 = __ballot((m[j] & 4u) != 0u);
    const uint32_t r0 = p0 + uint32_t(__builtin_popcountll(b0 & below)), r1 = p1 + uint32_t(__builtin_popcountll(b1 & below)),
                   r2 = p2 + uint32_t(__builtin_popcountll(b2 & below));
    p0 += uint32_t(__builtin_popcountll(b0)), p1 += uint32_t(__builtin_popcountll(b1)), p2 += uint32_t(__builtin_popcountll(b2));
    if (MODE == SEL_MIGRATE) {
      if (m[j] & 1u) {  // stays: stable compaction into the other array set
        const uint32_t d = r0;
        dst.pos4[d] = src.pos4[i], dst.vel4[d] = src.vel4[i], dst.col4[d] = src.col4[i], dst.pstar[d] = src.pstar[i];
        dst.id[d] = src.id[i], dst.type[d] = src.type[i], dst.key[d] = src.key[i];
      } else if (m[j] & 6u) {
        const bool left = (m[j] & 2u) != 0;
        const uint32_t d = left ? r1 : r2;
        if (d < capRecords) {
          MigrantRec<N> r;
          r.pos4 = src.pos4[i], r.vel4 = src.vel4[i], r.col4 = src.col4[i], r.pstar = src.pstar[i];
          r.id = src.id[i], r.key = src.key[i], r.type = src.type[i];
          static_cast<MigrantRec<N> *>(left ? sendL : sendR)[d] = r;
        }
      }
    } else {
      if (m[j] & 1u) {
        const uint32_t d = r0;
        if (d < capRecords) {
          GhostRec<N> r;
          r.pstar = src.pstar[i], r.col4 = src.col4[i], r.key = src.key[i], r.type = src.type[i] | TYPE_GHOST;
          r.pad0 = r.pad1 = 0;
          static_cast<GhostRec<N> *>(sendL)[d] = r;
          srcL[d] = i;
        }
      }
      if (m[j] & 2u) {
        const uint32_t d = r1;
        if (d < capRecords) {
          GhostRec<N> r;
          r.pstar = src.pstar[i], r.col4 = src.col4[i], r.key = src.key[i], r.type = src.type[i] | TYPE_GHOST;
          r.pad0 = r.pad1 = 0;
          static_cast<GhostRec<N> *>(sendR)[d] = r;
          srcR[d] = i;
        }
      }
    }
  }
}

// Assembly rounds of pbf_slab_step: the wire message is {header (WIRE_HDR bytes: record count) | capacity-sized record
// array}; the counts never travel on their own.
constexpr size_t WIRE_HDR = 32;  // keeps the records 32-byte aligned (double4)
__global__ void k_wire_headers(const uint32_t *__restrict__ totals, int idxL, int idxR, uint32_t *__restrict__ hdrL,
                               uint32_t *__restrict__ hdrR, uint32_t *__restrict__ recvL, uint32_t *__restrict__ recvR) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    hdrL[0] = totals[idxL];
    hdrR[0] = totals[idxR];
  }
  // a rank without that neighbour neither sends nor receives on that side: its incoming header reads 0
  if (blockIdx.x == 0 && threadIdx.x < WIRE_HDR / 4) recvL[threadIdx.x] = 0u, recvR[threadIdx.x] = 0u;
}

// After the exchange: {own totals[3], -, count from the left, count from the right} straight into the caller's pinned
// host words (one launch instead of three small copies)
__global__ void k_wire_counts(const uint32_t *__restrict__ totals, const uint32_t *__restrict__ recvL,
                              const uint32_t *__restrict__ recvR, uint32_t *__restrict__ host) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    host[0] = totals[0], host[1] = totals[1], host[2] = totals[2];
    host[4] = recvL[0], host[5] = recvR[0];
    __threadfence_system();
  }
}

// A record keyed in the sender's rank-local x frame is re-keyed into ours: x' = x + shift (cells).
__device__ inline uint32_t shift_key_x(uint32_t key, int32_t shift) {
  const uint32_t x = (compact10(key) + uint32_t(shift)) & 1023u;
  return (key & ~MORTON_X) | spread10(x);
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_append_migrants(uint32_t at, const MigrantRec<N> *__restrict__ recvL,
                                                           uint32_t nL, const MigrantRec<N> *__restrict__ recvR,
                                                           uint32_t nR, int32_t shiftL, int32_t shiftR,
                                                           ParticleArrays<N> dst) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= nL + nR) return;
  const MigrantRec<N> r = j < nL ? recvL[j] : recvR[j - nL];
  const uint32_t d = at + j;
  dst.pos4[d] = r.pos4, dst.vel4[d] = r.vel4, dst.col4[d] = r.col4, dst.pstar[d] = r.pstar;
  dst.id[d] = r.id, dst.type[d] = uint8_t(r.type), dst.key[d] = shift_key_x(r.key, j < nL ? shiftL : shiftR);
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_append_ghosts(uint32_t at, const GhostRec<N> *__restrict__ recvL,
                                                         uint32_t nL, const GhostRec<N> *__restrict__ recvR,
                                                         uint32_t nR, int32_t shiftL, int32_t shiftR,
                                                         ParticleArrays<N> dst) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= nL + nR) return;
  const GhostRec<N> r = j < nL ? recvL[j] : recvR[j - nL];
  const uint32_t d = at + j;
  const vec4<N> zero = make_vec4<N>(N(0), N(0), N(0), N(0));
  dst.pos4[d] = zero, dst.vel4[d] = zero, dst.col4[d] = r.col4, dst.pstar[d] = r.pstar;
  dst.id[d] = ~uint64_t(0), dst.type[d] = uint8_t(r.type), dst.key[d] = shift_key_x(r.key, j < nL ? shiftL : shiftR);
}

__global__ __launch_bounds__(BLOCK) void k_count_keys(uint32_t n, uint32_t tableN, const uint32_t *__restrict__ key,
                                                      uint32_t *__restrict__ count) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t before, rank;  // one atomic per distinct cell per wave (the keeps are still in Z order)
  wave_bucket_atomic(min(key[i], tableN), [&](uint32_t b, uint32_t cnt) { return atomicAdd(&count[b], cnt); }, before, rank);
}

// owned particles per GLOBAL grid column (1024 bins, LDS-privatised): the load-balance input of the slab driver
__global__ __launch_bounds__(BLOCK) void k_column_histogram(uint32_t n, uint32_t xoff, const uint32_t *__restrict__ key,
                                                            const uint8_t *__restrict__ type,
                                                            uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[1024];
  for (uint32_t j = threadIdx.x; j < 1024; j += BLOCK) h[j] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
    if (!(type[i] & TYPE_GHOST)) atomicAdd(&h[(compact10(key[i]) + xoff) & 1023u], 1u);
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < 1024; j += BLOCK)
    if (h[j]) atomicAdd(&hist[j], h[j]);
}

// owners -> copies: {pStar, lambda} of the particles listed in srcL / srcR, read at their sorted slot
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_pack_field(uint32_t nL, uint32_t nR, const uint32_t *__restrict__ srcL,
                                                      const uint32_t *__restrict__ srcR,
                                                      const uint32_t *__restrict__ slotOf,
                                                      const vec4<N> *__restrict__ pstar, vec4<N> *__restrict__ outL,
                                                      vec4<N> *__restrict__ outR, const uint32_t *__restrict__ rowSlotOf) {
  // (rowSlotOf != NULL: `pstar` is the iterations' row-major copy, sorted particle d sits at rowSlotOf[d])
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= nL + nR) return;
  uint32_t d = slotOf[j < nL ? srcL[j] : srcR[j - nL]];
  if (rowSlotOf) d = rowSlotOf[d];
  (j < nL ? outL[j] : outR[j - nL]) = pstar[d];
}

// copies <- owners: the copies sit at pre-sort indices ghostAt + j in arrival order (left, then right)
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_unpack_field(StepConsts<N> c, uint32_t ghostAt, uint32_t nL, uint32_t nR,
                                                        const vec4<N> *__restrict__ inL,
                                                        const vec4<N> *__restrict__ inR,
                                                        const uint32_t *__restrict__ slotOf,
                                                        vec4<N> *__restrict__ pstar, uint2 *__restrict__ qpos,
                                                        const uint32_t *__restrict__ rowSlotOf) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= nL + nR) return;
  const vec4<N> v = j < nL ? inL[j] : inR[j - nL];
  uint32_t d = slotOf[ghostAt + j];
  if (rowSlotOf) d = rowSlotOf[d];  // (pstar / qpos are then the row-major copies)
  pstar[d] = v;
  if (qpos) {  // (NULL when the field is not pStar: the extras' velocity / vorticity refresh)
    bool usable;
    qpos[d] = quantise_position<N>(c, v, &usable);  // keeps the list build's quantised copy in step with pStar
  }
}

// ================================================================================================
// pbf_slab_step's assembly (round 3): ONE select pass for both rounds and no compaction.
//   k_predict (slab mode) has already dropped last step's copies (key = DEAD_KEY) and left the leavers out of the
//   histogram.  k_slab_count / k_slab_scan / k_slab_emit then, in array order,
//       pack the leavers into the migrant wire and mark their slots DEAD_KEY            (classes 0 = to left, 1 = to right)
//       pack copies of the STAYERS in the first / last owned column into the ghost wire (classes 2 = for left, 3 = for right)
//   the stayers themselves are not moved: the sort — which moves every record once anyway — skips dead slots.  After the
//   migrants' exchange k_append_migrants appends the arrivals (and adds them to the histogram), k_slab_arrival_ghosts
//   appends the copies of those arrivals that landed in a boundary column to the ghost wire — after the stayers', in
//   arrival order: the very sequence a select over [stayers, arrivals from the left, arrivals from the right] gives.
// ================================================================================================
__device__ inline uint32_t slab_classes4(uint32_t key, uint8_t type, const SlabCut &s) {
  if (key == DEAD_KEY || (type & TYPE_GHOST)) return 0u;
  const uint32_t cx = compact10(key);
  if (s.hasLeft && cx < s.xlo) return 1u;
  if (s.hasRight && cx >= s.xhi) return 2u;
  uint32_t m = 0;
  if (s.hasLeft && cx == s.xlo) m |= 4u;
  if (s.hasRight && cx + 1u == s.xhi) m |= 8u;
  return m;
}

__global__ __launch_bounds__(BLOCK) void k_slab_count(uint32_t n, SlabCut s, const uint32_t *__restrict__ key,
                                                      const uint8_t *__restrict__ type, uint32_t nb,
                                                      uint32_t *__restrict__ counts) {
  __shared__ uint32_t waveTot[4][BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t waveBase = blockIdx.x * SEL_TILE + wave * 64u * SEL_ITEMS;
  uint32_t t[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < SEL_ITEMS; ++j) {
    const uint32_t i = waveBase + j * 64u + lane;
    const uint32_t m = i < n ? slab_classes4(key[i], type[i], s) : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] += uint32_t(__builtin_popcountll(__ballot(((m >> k) & 1u) != 0u)));
  }
  if (lane == 0)
    for (int k = 0; k < 4; ++k) waveTot[k][wave] = t[k];
  __syncthreads();
  if (threadIdx.x < 4) {
    uint32_t sum = 0;
    for (uint32_t w = 0; w < BLOCK / 64; ++w) sum += waveTot[threadIdx.x][w];
    counts[threadIdx.x * nb + blockIdx.x] = sum;
  }
}

// exclusive scan per class over the tiles (one WAVE per class, 64 tiles a trip); totals[0..3]; the wire headers of BOTH
// rounds (the ghost headers hold the stayers' counts until k_slab_arrival_ghosts adds the arrivals'); incoming headers read
// 0 until a message lands
__global__ __launch_bounds__(BLOCK) void k_slab_scan(uint32_t nb, uint32_t *__restrict__ counts, uint32_t *__restrict__ totals,
                                                     uint32_t *__restrict__ migL, uint32_t *__restrict__ migR,
                                                     uint32_t *__restrict__ ghoL, uint32_t *__restrict__ ghoR,
                                                     uint32_t *__restrict__ recvL, uint32_t *__restrict__ recvR) {
  static_assert(BLOCK / 64 == 4, "one wave per class");
  const int lane = threadIdx.x & 63, cls = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += 64) {
    const uint32_t i = base + uint32_t(lane);
    const uint32_t v = i < nb ? counts[cls * nb + i] : 0u;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (i < nb) counts[cls * nb + i] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) {
    totals[cls] = carry;
    (cls == 0 ? migL : cls == 1 ? migR : cls == 2 ? ghoL : ghoR)[0] = carry;
  }
  if (threadIdx.x < WIRE_HDR / 4) recvL[threadIdx.x] = 0u, recvR[threadIdx.x] = 0u;
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_slab_emit(uint32_t n, SlabCut s, ParticleArrays<N> src, uint32_t nb,
                                                     const uint32_t *__restrict__ bases, MigrantRec<N> *__restrict__ migL,
                                                     MigrantRec<N> *__restrict__ migR, GhostRec<N> *__restrict__ ghoL,
                                                     GhostRec<N> *__restrict__ ghoR, uint32_t capRecords,
                                                     uint32_t *__restrict__ srcL, uint32_t *__restrict__ srcR) {
  __shared__ uint32_t waveTot[4][BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t waveBase = blockIdx.x * SEL_TILE + wave * 64u * SEL_ITEMS;
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t m[SEL_ITEMS];
  uint32_t t[4] = {0, 0, 0, 0};
  bool any = false;
#pragma unroll
  for (int j = 0; j < SEL_ITEMS; ++j) {
    const uint32_t i = waveBase + j * 64u + lane;
    m[j] = i < n ? slab_classes4(src.key[i], src.type[i], s) : 0u;
    any |= m[j] != 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] += uint32_t(__builtin_popcountll(__ballot(((m[j] >> k) & 1u) != 0u)));
  }
  if (lane == 0)
    for (int k = 0; k < 4; ++k) waveTot[k][wave] = t[k];
  __syncthreads();
  if (!__any(any)) return;  // (most waves: nothing of theirs leaves or borders a neighbour)
  uint32_t p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    p[k] = bases[k * nb + blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) p[k] += waveTot[k][w];
  }
#pragma unroll
  for (int j = 0; j < SEL_ITEMS; ++j) {
    const uint32_t i = waveBase + j * 64u + lane;
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t b = __ballot(((m[j] >> k) & 1u) != 0u);
      r[k] = p[k] + uint32_t(__builtin_popcountll(b & below));
      p[k] += uint32_t(__builtin_popcountll(b));
    }
    if (m[j] & 3u) {  // leaves: the whole record travels, the slot dies
      const bool left = (m[j] & 1u) != 0u;
      const uint32_t d = left ? r[0] : r[1];
      if (d < capRecords) {
        MigrantRec<N> rec;
        rec.pos4 = src.pos4[i], rec.vel4 = src.vel4[i], rec.col4 = src.col4[i], rec.pstar = src.pstar[i];
        rec.id = src.id[i], rec.key = src.key[i], rec.type = src.type[i];
        (left ? migL : migR)[d] = rec;
      }
      src.key[i] = DEAD_KEY;
    } else if (m[j] & 12u) {
      GhostRec<N> rec;
      rec.pstar = src.pstar[i], rec.col4 = src.col4[i], rec.key = src.key[i], rec.type = src.type[i] | TYPE_GHOST;
      rec.pad0 = rec.pad1 = 0;
      if ((m[j] & 4u) && r[2] < capRecords) ghoL[r[2]] = rec, srcL[r[2]] = i;
      if ((m[j] & 8u) && r[3] < capRecords) ghoR[r[3]] = rec, srcR[r[3]] = i;
    }
  }
}

// the six words the host needs after an exchange, written straight into its pinned memory:
// {a, b, c, d = totals[0..3], count from the left, count from the right}
// host[7] = seq is written LAST, behind a system-scope fence: the host polls that word instead of synchronising the stream
__global__ void k_slab_counts(const uint32_t *__restrict__ totals, uint32_t *__restrict__ recvL,
                              uint32_t *__restrict__ recvR, volatile uint32_t *__restrict__ host, uint32_t seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    host[0] = totals[0], host[1] = totals[1], host[2] = totals[2], host[3] = totals[3];
    host[4] = recvL[0], host[5] = recvR[0];
    __threadfence_system();
    host[7] = seq;
    __threadfence_system();
    recvL[0] = 0u, recvR[0] = 0u;  // the next round's incoming headers read 0 until its message lands
  }
}

// arrivals: appended behind everything the arrays hold, re-keyed into this rank's frame, added to the histogram
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_append_migrants_h(uint32_t at, const MigrantRec<N> *__restrict__ recvL,
                                                             uint32_t nL, const MigrantRec<N> *__restrict__ recvR,
                                                             uint32_t nR, int32_t shiftL, int32_t shiftR,
                                                             ParticleArrays<N> dst, uint32_t tableN,
                                                             uint32_t *__restrict__ count) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= nL + nR) return;
  const MigrantRec<N> r = j < nL ? recvL[j] : recvR[j - nL];
  const uint32_t d = at + j, k = shift_key_x(r.key, j < nL ? shiftL : shiftR);
  dst.pos4[d] = r.pos4, dst.vel4[d] = r.vel4, dst.col4[d] = r.col4, dst.pstar[d] = r.pstar;
  dst.id[d] = r.id, dst.type[d] = uint8_t(r.type), dst.key[d] = k;
  atomicAdd(&count[min(k, tableN)], 1u);
}

// copies of the arrivals that landed in a boundary column, appended to the ghost wire behind the stayers' copies in
// arrival order (ONE workgroup: the arrivals of a step are few, and only their keys are read to find the handful that
// qualifies); updates the totals and the outgoing headers
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_slab_arrival_ghosts(uint32_t at, uint32_t nArrived, SlabCut s, ParticleArrays<N> arr,
                                                               uint32_t *__restrict__ totals, GhostRec<N> *__restrict__ ghoL,
                                                               GhostRec<N> *__restrict__ ghoR, uint32_t *__restrict__ hdrL,
                                                               uint32_t *__restrict__ hdrR, uint32_t capRecords,
                                                               uint32_t *__restrict__ srcL, uint32_t *__restrict__ srcR) {
  uint32_t baseL = totals[2], baseR = totals[3];
  for (uint32_t b0 = 0; b0 < nArrived; b0 += BLOCK) {
    const uint32_t j = b0 + threadIdx.x, i = at + j;
    uint32_t m = 0;
    if (j < nArrived) m = slab_classes4(arr.key[i], arr.type[i], s) >> 2;  // (an arrival never leaves again in the same step)
    uint32_t totL, totR;
    const uint32_t eL = block_excl_scan(m & 1u, &totL);
    const uint32_t eR = block_excl_scan((m >> 1) & 1u, &totR);
    if (m) {
      GhostRec<N> rec;
      rec.pstar = arr.pstar[i], rec.col4 = arr.col4[i], rec.key = arr.key[i], rec.type = arr.type[i] | TYPE_GHOST;
      rec.pad0 = rec.pad1 = 0;
      if ((m & 1u) && baseL + eL < capRecords) ghoL[baseL + eL] = rec, srcL[baseL + eL] = i;
      if ((m & 2u) && baseR + eR < capRecords) ghoR[baseR + eR] = rec, srcR[baseR + eR] = i;
    }
    baseL += totL, baseR += totR;
  }
  __syncthreads();
  if (threadIdx.x == 0) totals[2] = baseL, totals[3] = baseR, hdrL[0] = baseL, hdrR[0] = baseR;
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_append_ghosts_h(uint32_t at, const GhostRec<N> *__restrict__ recvL,
                                                           uint32_t nL, const GhostRec<N> *__restrict__ recvR,
                                                           uint32_t nR, int32_t shiftL, int32_t shiftR,
                                                           ParticleArrays<N> dst, uint32_t tableN,
                                                           uint32_t *__restrict__ count) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= nL + nR) return;
  const GhostRec<N> r = j < nL ? recvL[j] : recvR[j - nL];
  const uint32_t d = at + j, k = shift_key_x(r.key, j < nL ? shiftL : shiftR);
  const vec4<N> zero = make_vec4<N>(N(0), N(0), N(0), N(0));
  dst.pos4[d] = zero, dst.vel4[d] = zero, dst.col4[d] = r.col4, dst.pstar[d] = r.pstar;
  dst.id[d] = ~uint64_t(0), dst.type[d] = uint8_t(r.type), dst.key[d] = k;
  atomicAdd(&count[min(k, tableN)], 1u);
}

}  // namespace pbf
