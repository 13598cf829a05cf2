// pbf_tiles.hpp — the solver iteration out of LDS tiles (option gather = 3).
//
// Why: counters and two microbenchmarks (tools/valu_rate.hip, tools/gather_rate.hip; profiles/r02_gather_rates.md) show
// the list-driven readers bound by the GATHER path, not by arithmetic: one wave-wide gather whose 64 addresses fall
// into a window of a few thousand particles costs ~38 ns per CU out of L1 / L2 (exactly what lambda spends per list
// entry), the same gather out of an LDS tile 6-7 ns.  The list build likewise spends most of its time waiting for its
// ~100 candidate loads per lane.  So the three kernels of a solver iteration run per BRICK (reference walk:
// sph.hpp:220-234, ompsph.hpp:217-248):
//   * a workgroup owns a Morton-aligned brick of 4 x 4 x 4 cells — 64 consecutive codes, ONE contiguous run of the
//     sorted arrays — and stages the brick's 6 x 6 x 6 halo of cells into LDS once, x-fastest, so that the three x
//     cells of a (dy, dz) row are ONE contiguous LDS run (cell start / end offsets from the grid table, scanned in LDS);
//   * k_tile_build: the tile holds the 8-byte quantised positions; every home particle walks its 9 runs out of LDS
//     in the reference's order and records the TILE SLOTS of the candidates inside h (superset test, as
//     k_build_lists_q) into its neighbour-list row in HBM;
//   * k_tile_from_lists<Op>: the tile holds the candidates' pStar (+ lambda); every home particle reads its row and
//     takes the candidates from LDS by slot — same candidates, same order, same op code as every other gather
//     kernel, hence the same bits.
// Persistent workgroups pull bricks from the sort stage's list of non-empty bricks through an atomic ticket; the loop
// ends when the ticket passes the list's end, an exit every wave reaches.  A brick whose halo holds more than `cap`
// records (a pile-up) is handled with global loads by the same workgroup: its rows then hold global indices — every
// kernel of the iteration takes the same decision from the same table.  Particles in no cell (key >= tableN) are
// swept the same way.  Rows are the ones of k_build_lists_q / k_gather_from_lists: [i / 256][slot][i % 256].
#pragma once

#include "pbf_kernels.hpp"

namespace pbf {

constexpr int TILE_BZ = 4;          // the sort stage's brick list (k_brick_list) is the 4 x 4 x 4 one
constexpr int TILE_THREADS = 512;   // ~450 home particles per brick at rest density
using TileBrick = Brick2<TILE_BZ>;  // HOME 64 codes, HALO 216 cells, HDR2 header bytes (offsets, global starts, ticket)

// Phase 1: the halo's cell ranges (grid table) and their exclusive scan; returns the number of records in the halo.
template <typename N, int THREADS>
__device__ inline uint32_t tile_header(const StepConsts<N> &c, const uint32_t *__restrict__ table, uint32_t code0,
                                       uint32_t *off, uint32_t *gstart) {
  using B = TileBrick;
  static_assert(B::HALO <= THREADS, "one halo cell per thread");
  const uint32_t tid = threadIdx.x;
  const uint32_t bx = compact10(code0), by = compact10(code0 >> 1), bz = compact10(code0 >> 2);
  uint32_t cnt = 0;
  if (tid < B::HALO) {
    const uint32_t lx = tid % 6, ly = (tid / 6) % 6, lz = tid / 36;
    const uint32_t code = morton_encode((bx + lx - 1u) & 1023u, (by + ly - 1u) & 1023u, (bz + lz - 1u) & 1023u);
    uint32_t s = 0, e = 0;
    if (code < c.tableN) {  // sph.hpp:206-208: a cell outside the table, and the table's last cell, are empty
      s = table[code];
      e = (code + 1u) < c.tableN ? table[code + 1u] : s;
    }
    gstart[tid] = s;
    cnt = e - s;
  }
  uint32_t total;
  const uint32_t ex = block_excl_scan<THREADS>(cnt, &total);
  if (tid < B::HALO) off[tid] = ex;
  if (tid == 0) off[B::HALO] = total;
  __syncthreads();
  return total;
}

// Phase 2: record r of the tile <- its halo cell by bisection of off[] (all threads, consecutive records: coalesced)
template <int THREADS, typename T, typename F>
__device__ inline void tile_stage(uint32_t total, const uint32_t *off, const uint32_t *gstart, T *tile, F &&load) {
  using B = TileBrick;
  for (uint32_t r = threadIdx.x; r < total; r += THREADS) {
    uint32_t lo = 0, hi = B::HALO;  // the last cell h with off[h] <= r (it is not empty: off[h + 1] > r)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const uint32_t mid = (lo + hi) >> 1;
      const bool right = off[mid] <= r;
      lo = right ? mid : lo;
      hi = right ? hi : mid;
    }
    tile[r] = load(gstart[lo] + (r - off[lo]));
  }
}

// first halo cell of row (dy, dz) of the home cell with key k: the row's three x cells are off[l0] .. off[l0 + 3]
__device__ inline uint32_t tile_home_cell(uint32_t k) {
  const uint32_t hx = (k & 1u) | ((k >> 2) & 2u);          // key bits 0, 3
  const uint32_t hy = ((k >> 1) & 1u) | ((k >> 3) & 2u);   // key bits 1, 4
  const uint32_t hz = ((k >> 2) & 1u) | ((k >> 4) & 2u);   // key bits 2, 5
  return (hz * 6u + hy) * 6u + hx;
}

__device__ inline uint32_t *nbr_row(uint32_t *nbrList, uint32_t i) {
  return nbrList + size_t(i / BLOCK) * NBR_ROWS * BLOCK + (i % BLOCK);
}
__device__ inline const uint32_t *nbr_row(const uint32_t *nbrList, uint32_t i) {
  return nbrList + size_t(i / BLOCK) * NBR_ROWS * BLOCK + (i % BLOCK);
}

// One particle's list with global loads and a store per hit: the same list k_build_lists_q writes (global indices, walk
// order).  Only for bricks beyond the tile and particles in no cell.
template <typename N>
__device__ inline void build_one_plain(const StepConsts<N> &c, const vec4<N> *__restrict__ pstar,
                                       const uint2 *__restrict__ qpos, const uint8_t *__restrict__ type,
                                       const uint32_t *__restrict__ key, const uint32_t *__restrict__ table,
                                       uint32_t *__restrict__ nbrList, uint32_t *__restrict__ nbrCount, uint32_t i) {
  if (c.hasObstacles && type[i] != 0) {
    nbrCount[i] = 0;
    return;
  }
  bool usable;
  const uint2 qa = quantise_position<N>(c, pstar[i], &usable);
  const qpair axy = __builtin_bit_cast(qpair, qa.x), azw = __builtin_bit_cast(qpair, qa.y);
  const uint32_t t2 = usable ? QPOS_T * QPOS_T : 0xFFFFFFFFu;
  uint32_t *row = nbr_row(nbrList, i);
  uint32_t written = 0;
  for_each_candidate(key[i], table, c.tableN, [&](uint32_t b) {
    const uint2 q = qpos[b];
    const qpair dxy = __builtin_bit_cast(qpair, q.x) - axy, dzw = __builtin_bit_cast(qpair, q.y) - azw;
    if (uint32_t(qdot2(__builtin_bit_cast(uint32_t, dzw), qdot2(__builtin_bit_cast(uint32_t, dxy)))) <= t2) {
      if (written < NBR_ROWS) row[written * BLOCK] = b;
      ++written;
    }
  });
  nbrCount[i] = written <= NBR_ROWS ? written : NBR_OVERFLOW;
}

// One particle's op from a row of GLOBAL indices (the serial form of k_gather_from_lists)
template <typename N, typename Op>
__device__ inline void from_lists_one_global(const StepConsts<N> &c, const typename Op::Args &args,
                                             const uint32_t *__restrict__ key, const uint32_t *__restrict__ table,
                                             const uint32_t *__restrict__ nbrList, const uint32_t *__restrict__ nbrCount,
                                             uint32_t i) {
  Op op;
  if (!op.begin(c, args, i)) return;
  const uint32_t cnt = nbrCount[i];
  const uint32_t *mine = nbr_row(nbrList, i);
  if (cnt == NBR_OVERFLOW) {
    for_each_candidate(key[i], table, c.tableN, [&](uint32_t b) { op.add(c, Op::load(args, b)); });
  } else {
    for (uint32_t q = 0; q < cnt; q += 4) {
      uint32_t b[4];
      typename Op::Src cnd[4];
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) b[w] = q + w < cnt ? mine[(q + w) * BLOCK] : i;
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) cnd[w] = Op::load(args, b[w]);
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) op.add_bf(c, cnd[w], q + w < cnt);
    }
  }
  op.end(c, args, i);
}

// ------------------------------------------------------------------------------------------------
// List build per brick
// ------------------------------------------------------------------------------------------------
template <typename N, int WAYS, int LMAX>
__global__ __launch_bounds__(TILE_THREADS) void k_tile_build(StepConsts<N> c, const vec4<N> *__restrict__ pstar,
                                                             const uint2 *__restrict__ qpos,
                                                             const uint8_t *__restrict__ type,
                                                             const uint32_t *__restrict__ key,
                                                             const uint32_t *__restrict__ table,
                                                             const uint32_t *__restrict__ active,
                                                             const uint32_t *__restrict__ nActivePtr,
                                                             uint32_t *__restrict__ ticket, uint32_t cap,
                                                             uint32_t *__restrict__ nbrList,
                                                             uint32_t *__restrict__ nbrCount) {
  using B = TileBrick;
  constexpr int THREADS = TILE_THREADS;
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t *off = reinterpret_cast<uint32_t *>(smem);  // [HALO + 1]
  uint32_t *gstart = off + B::HALO + 1;                // [HALO]
  uint32_t *shTicket = gstart + B::HALO;               // [1]
  uint2 *qtile = reinterpret_cast<uint2 *>(smem + B::HDR2);
  uint16_t *list = reinterpret_cast<uint16_t *>(smem + B::HDR2 + size_t(cap + WAYS) * sizeof(uint2));  // [LMAX + WAYS][THREADS]
  const uint32_t tid = threadIdx.x;
  const uint32_t nActive = *nActivePtr;
  {  // particles that lie in no cell (key >= tableN, sph.hpp:206)
    const uint32_t stride = gridDim.x * THREADS;
    for (uint32_t i = table[c.tableN] + blockIdx.x * THREADS + tid; i < c.n; i += stride)
      build_one_plain<N>(c, pstar, qpos, type, key, table, nbrList, nbrCount, i);
  }
  for (;;) {
    __syncthreads();  // the previous brick's LDS reads are done before the header / tile are rewritten
    if (tid == 0) *shTicket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t t = *shTicket;
    if (t >= nActive) break;  // uniform: every wave of the workgroup leaves here
    const uint32_t code0 = active[t] * uint32_t(B::HOME);
    const uint32_t hs = table[code0], he = table[min(code0 + uint32_t(B::HOME), c.tableN)];
    const uint32_t total = tile_header<N, THREADS>(c, table, code0, off, gstart);
    if (total > cap) {  // uniform: a pile-up beyond the tile
      for (uint32_t i = hs + tid; i < he; i += THREADS) build_one_plain<N>(c, pstar, qpos, type, key, table, nbrList, nbrCount, i);
      continue;
    }
    tile_stage<THREADS>(total, off, gstart, qtile, [&](uint32_t g) { return qpos[g]; });
    __syncthreads();
    for (uint32_t i = hs + tid; i < he; i += THREADS) {
      if (c.hasObstacles && type[i] != 0) {
        nbrCount[i] = 0;
        continue;
      }
      bool usable;
      const uint2 qa = quantise_position<N>(c, pstar[i], &usable);
      const qpair axy = __builtin_bit_cast(qpair, qa.x), azw = __builtin_bit_cast(qpair, qa.y);
      const uint32_t t2 = usable ? QPOS_T * QPOS_T : 0xFFFFFFFFu;
      auto within = [&](uint32_t qx, uint32_t qy) {
        const qpair dxy = __builtin_bit_cast(qpair, qx) - axy, dzw = __builtin_bit_cast(qpair, qy) - azw;
        return uint32_t(qdot2(__builtin_bit_cast(uint32_t, dzw), qdot2(__builtin_bit_cast(uint32_t, dxy)))) <= t2;
      };
      uint32_t *row = nbr_row(nbrList, i);
      // the staging list's write cursor as an LDS byte address: one select + one add per candidate
      constexpr uint32_t SLOT = THREADS * 2u;  // bytes per staging slot row
      unsigned char *const lbase = reinterpret_cast<unsigned char *>(list) + tid * 2u;
      uint32_t written = 0, cur = 0;  // cur = staged entries x SLOT
      auto flush = [&]() {
        for (uint32_t q = 0; __any(q < cur); q += SLOT) {
          const uint32_t k = written + q / SLOT;
          if (q < cur && k < NBR_ROWS) row[k * BLOCK] = *reinterpret_cast<const uint16_t *>(lbase + q);
        }
        written += cur / SLOT;
        cur = 0;
      };
      const uint32_t home = tile_home_cell(key[i]);
#pragma unroll 1
      for (uint32_t r = 0; r < 9; ++r) {
        const uint32_t l0 = home + (r / 3u) * 36u + (r % 3u) * 6u;
        const uint32_t s = off[l0], e = off[l0 + 3];
        for (uint32_t j = s; __any(j < e); j += WAYS) {
          if (j < e) {
            uint2 cnd[WAYS];
            const uint2 *src = qtile + j;  // slots past e belong to the next cells (the tile is padded by WAYS): masked
#pragma unroll
            for (uint32_t w = 0; w < WAYS; ++w) cnd[w] = src[w];
#pragma unroll
            for (uint32_t w = 0; w < WAYS; ++w) {
              const bool hit = (j + w < e) & within(cnd[w].x, cnd[w].y);
              *reinterpret_cast<uint16_t *>(lbase + cur) = uint16_t(j + w);  // branch-free append: kept only on a hit
              cur += hit ? SLOT : 0u;
            }
          }
          if (__any(cur >= uint32_t(LMAX) * SLOT)) flush();
        }
      }
      flush();
      nbrCount[i] = written <= NBR_ROWS ? written : NBR_OVERFLOW;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// List-driven op per brick: candidates by tile slot out of LDS
// ------------------------------------------------------------------------------------------------
// (a 16-byte record is fetched whole even where the op reads three of its components: ds_read_b128 is the faster
// LDS gather — 7.5 against 11 ns per wave for ds_read_b96, profiles/r02_gather_rates.md)
__device__ inline void keep_whole(float4 &v) { asm volatile("" : "+v"(v.w)); }
template <typename T> __device__ inline void keep_whole(T &) {}

template <typename N, typename Op>
__global__ __launch_bounds__(TILE_THREADS, sizeof(N) == 4 ? 8 : 4) void k_tile_from_lists(StepConsts<N> c, typename Op::Args args,
                                                                  const uint32_t *__restrict__ key,
                                                                  const uint32_t *__restrict__ table,
                                                                  const uint32_t *__restrict__ active,
                                                                  const uint32_t *__restrict__ nActivePtr,
                                                                  uint32_t *__restrict__ ticket, uint32_t cap,
                                                                  const uint32_t *__restrict__ nbrList,
                                                                  const uint32_t *__restrict__ nbrCount) {
  using B = TileBrick;
  using Src = typename Op::Src;
  constexpr int THREADS = TILE_THREADS;
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t *off = reinterpret_cast<uint32_t *>(smem);  // [HALO + 1]
  uint32_t *gstart = off + B::HALO + 1;                // [HALO]
  uint32_t *shTicket = gstart + B::HALO;               // [1]
  Src *tile = reinterpret_cast<Src *>(smem + B::HDR2);
  const uint32_t tid = threadIdx.x;
  const uint32_t nActive = *nActivePtr;
  {
    const uint32_t stride = gridDim.x * THREADS;
    for (uint32_t i = table[c.tableN] + blockIdx.x * THREADS + tid; i < c.n; i += stride)
      from_lists_one_global<N, Op>(c, args, key, table, nbrList, nbrCount, i);
  }
  for (;;) {
    __syncthreads();
    if (tid == 0) *shTicket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t t = *shTicket;
    if (t >= nActive) break;
    const uint32_t code0 = active[t] * uint32_t(B::HOME);
    const uint32_t hs = table[code0], he = table[min(code0 + uint32_t(B::HOME), c.tableN)];
    const uint32_t total = tile_header<N, THREADS>(c, table, code0, off, gstart);
    if (total > cap) {  // the build took the same branch: this brick's rows hold global indices
      for (uint32_t i = hs + tid; i < he; i += THREADS) from_lists_one_global<N, Op>(c, args, key, table, nbrList, nbrCount, i);
      continue;
    }
    tile_stage<THREADS>(total, off, gstart, tile, [&](uint32_t g) { return Op::load(args, g); });
    __syncthreads();
    for (uint32_t i = hs + tid; i < he; i += THREADS) {
      Op op;
      if (!op.begin(c, args, i)) continue;
      const uint32_t cnt = nbrCount[i];
      const uint32_t *mine = nbr_row(nbrList, i);
      if (cnt == NBR_OVERFLOW) {  // a row longer than NBR_ROWS: the whole walk, out of the tile
        const uint32_t home = tile_home_cell(key[i]);
#pragma unroll 1
        for (uint32_t r = 0; r < 9; ++r) {
          const uint32_t l0 = home + (r / 3u) * 36u + (r % 3u) * 6u;
          for (uint32_t j = off[l0], e = off[l0 + 3]; j < e; ++j) op.add(c, tile[j]);
        }
      } else {
        for (uint32_t q = 0; q < cnt; q += 4) {  // four entries and their candidates in flight per trip
          uint32_t b[4];
          Src cnd[4];
#pragma unroll
          for (uint32_t w = 0; w < 4; ++w) b[w] = mine[(q + w) * BLOCK];  // (NBR_ROWS is a multiple of 4: in the row)
#pragma unroll
          for (uint32_t w = 0; w < 4; ++w) b[w] = q + w < cnt ? b[w] : 0u;  // a tail slot holds anything: record 0, masked
#pragma unroll
          for (uint32_t w = 0; w < 4; ++w) {
            cnd[w] = tile[b[w]];
            keep_whole(cnd[w]);
          }
#pragma unroll
          for (uint32_t w = 0; w < 4; ++w) op.add_bf(c, cnd[w], q + w < cnt);
        }
      }
      op.end(c, args, i);
    }
  }
}

}  // namespace pbf
