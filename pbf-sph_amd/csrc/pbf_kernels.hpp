// pbf_kernels.hpp — gfx950 device kernels of the PBF-SPH step (wave64, no MFMA: nothing on this
// path is a dense contraction).  Each kernel cites the reference lines whose result it
// reproduces; the decomposition (device counting sort, packed pStar+lambda, Jacobi buffers) is ours.
#pragma once

#include <type_traits>

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

// One atomic per distinct bucket per WAVE instead of one per lane: the particles arrive in the previous step's Z order,
// so the 64 lanes of a wave fall into ~10 cells.  Every lane gets the value the bucket held before the wave's add and its
// rank among the wave's lanes of the same bucket (the histogram only needs the side effect; the scatter turns the two
// into its slot).  The loop runs once per distinct bucket of the wave and every lane leaves it.
template <typename F> __device__ inline void wave_bucket_atomic(uint32_t bucket, F &&leader_op, uint32_t &before, uint32_t &rank) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t below = (1ull << lane) - 1ull;
  bool pending = true;
  uint32_t cnt = 0, leader = lane;
  rank = 0;
  while (__any(pending)) {  // phase 1: who shares my bucket (no memory traffic)
    if (pending) {
      const uint32_t b = __builtin_amdgcn_readfirstlane(bucket);  // the first pending lane's bucket
      if (bucket == b) {
        const uint64_t same = __ballot(1);  // the pending lanes of that bucket
        cnt = uint32_t(__builtin_popcountll(same));
        rank = uint32_t(__builtin_popcountll(same & below));
        leader = uint32_t(__builtin_ctzll(same));
        pending = false;
      }
    }
  }
  uint32_t v = 0;
  if (rank == 0) v = leader_op(bucket, cnt);  // phase 2: the leaders' atomics are all in flight together
  before = __shfl(v, int(leader), 64);
}

// predict + key of one particle (reference: ompsph.hpp:137-154): updates v, returns pStar, writes the key
template <typename N>
__device__ inline vec4<N> predict_one(const StepConsts<N> &c, const vec4<N> &p, vec4<N> &v, uint8_t type,
                                      const N *__restrict__ wells, uint32_t &k) {
  N px, py, pz;
  if (c.hasObstacles && (type & 1)) {
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
    px = (v.x * c.dt) + (p.x / c.scale);
    py = (v.y * c.dt) + (p.y / c.scale);
    pz = (v.z * c.dt) + (p.z / c.scale);
  }
  k = morton_encode(static_cast<uint32_t>(cell_coord((px - c.minExtent[0]) / c.h)) - c.xoff,
                    static_cast<uint32_t>(cell_coord((py - c.minExtent[1]) / c.h)),
                    static_cast<uint32_t>(cell_coord((pz - c.minExtent[2]) / c.h)));
  return make_vec4<N>(px, py, pz, N(0));
}

// pbf_slab_step: does a particle with this key still belong to this slab (its cell column inside [sxlo, sxhi), or no slab
// on the side it left to)?  Always true outside slab mode.
template <typename N> __device__ inline bool slab_stays(const StepConsts<N> &c, uint32_t key) {
  if (!c.slabOn) return true;
  const uint32_t cx = compact10(key);
  return !((c.sHasL && cx < c.sxlo) || (c.sHasR && cx >= c.sxhi));
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_predict(StepConsts<N> c, const vec4<N> *__restrict__ pos4,
                                                   vec4<N> *__restrict__ vel4, const uint8_t *__restrict__ type,
                                                   const N *__restrict__ wells, vec4<N> *__restrict__ pstar,
                                                   uint32_t *__restrict__ key, uint32_t *__restrict__ count) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= c.n) return;
  const uint8_t ty = c.hasObstacles ? type[i] : uint8_t(0);
  if (c.slabOn && (ty & TYPE_GHOST)) {  // last step's copy of a neighbour's particle: gone (the sort drops the slot)
    key[i] = DEAD_KEY;
    return;
  }
  const vec4<N> p = pos4[i];
  vec4<N> v = vel4[i];
  uint32_t k;
  pstar[i] = predict_one<N>(c, p, v, ty, wells, k);
  if (!(c.hasObstacles && (ty & 1))) vel4[i] = v;
  key[i] = k;
  if (!slab_stays(c, k)) return;  // its column now belongs to a neighbour: packed and dropped by the slab select
  // bucket tableN collects particles outside the table: they are "in no cell" (sph.hpp:206)
  uint32_t before, rank;
  wave_bucket_atomic(min(k, c.tableN), [&](uint32_t b, uint32_t cnt) { return atomicAdd(&count[b], cnt); }, before, rank);
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan of the cell histogram = the reference's gridTable (sph.hpp:238-250):
// table[c] = number of particles with key < c = first sorted index with key >= c.
// Three passes (block sums -> scan of sums -> scan + offset), 2048 cells per block.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

// Workgroup -> chunk of the sorted particle array, XCD-aware.  The dispatcher deals consecutive workgroup ids round
// robin over the 8 XCDs, each with an L2 of its own, so with chunk = blockIdx two neighbouring chunks — which share most
// of their candidates — never share an L2.  Here XCD k works on the k-th contiguous eighth of the chunks: neighbours in
// the Morton order meet in one L2 (a bijection of [0, gridDim) for any grid size).
constexpr uint32_t NUM_XCD = 8;
__device__ inline uint32_t xcd_chunk() {
#ifdef PBF_NO_XCD_MAP
  return blockIdx.x;
#elif defined(PBF_XCD_STRIPE)
  // experiment: stripes of PBF_XCD_STRIPE consecutive chunks dealt round robin to the XCDs (locality inside a stripe,
  // load balance across stripes: list-driven kernels do work in proportion to the local density)
  constexpr uint32_t S = PBF_XCD_STRIPE;
  const uint32_t g = gridDim.x, full = g / (NUM_XCD * S) * (NUM_XCD * S), b = blockIdx.x;
  if (b >= full) return b;
  const uint32_t k = b % NUM_XCD, j = b / NUM_XCD;
  return ((j / S) * NUM_XCD + k) * S + j % S;
#else
  const uint32_t g = gridDim.x, k = blockIdx.x % NUM_XCD, q = g / NUM_XCD, r = g % NUM_XCD;
  return k * q + min(k, r) + blockIdx.x / NUM_XCD;
#endif
}

__device__ inline uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, total in *total
template <int THREADS = BLOCK> __device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *total) {
  __shared__ uint32_t waveSum[THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan(v, lane);
  __syncthreads();  // protect waveSum across back-to-back calls
  if (lane == 63) waveSum[wave] = incl;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) {
    const uint32_t s = waveSum[w];
    if (w < wave) off += s;
    tot += s;
  }
  *total = tot;
  return off + incl - v;
}

// Up to two independent scans per launch (the Morton grid table and, with the row-major working set, the row table): job 0
// takes the first nb[0] workgroups, job 1 the rest.
struct ScanJobs {
  const uint32_t *count[2];
  uint32_t *sums[2];
  uint32_t *table[2];
  uint32_t len[2], nb[2];
};
__global__ __launch_bounds__(BLOCK) void k_scan_block_sums(ScanJobs jobs) {
  const uint32_t j = blockIdx.x >= jobs.nb[0] ? 1u : 0u, blk = blockIdx.x - (j ? jobs.nb[0] : 0u);
  const uint32_t *__restrict__ count = jobs.count[j];
  const uint32_t len = jobs.len[j];
  const uint32_t base = blk * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= len) {
    const uint4 a = *reinterpret_cast<const uint4 *>(count + base);
    const uint4 b = *reinterpret_cast<const uint4 *>(count + base + 4);
    s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  } else {
    for (int k = 0; k < SCAN_ITEMS; ++k)
      if (base + k < len) s += count[base + k];
  }
  uint32_t total;
  block_excl_scan(s, &total);
  if (threadIdx.x == 0) jobs.sums[j][blk] = total;
}

// one workgroup per job; on its way it zeroes `nZero` control words (the per-step tickets and counters, which would
// otherwise cost a fill launch of their own)
__global__ __launch_bounds__(BLOCK) void k_scan_sums(ScanJobs jobs, uint32_t *__restrict__ zero, uint32_t nZero) {
  if (blockIdx.x == 0)
    for (uint32_t k = threadIdx.x; k < nZero; k += BLOCK) zero[k] = 0u;
  uint32_t *__restrict__ blockSums = jobs.sums[blockIdx.x];
  const uint32_t nb = jobs.nb[blockIdx.x];
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

__global__ __launch_bounds__(BLOCK) void k_scan_apply(ScanJobs jobs) {
  const uint32_t j = blockIdx.x >= jobs.nb[0] ? 1u : 0u, blk = blockIdx.x - (j ? jobs.nb[0] : 0u);
  const uint32_t *__restrict__ count = jobs.count[j];
  uint32_t *__restrict__ table = jobs.table[j];
  const uint32_t len = jobs.len[j];
  const uint32_t base = blk * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < len) ? count[base + k] : 0u;
    s += v[k];
  }
  uint32_t total;
  uint32_t run = block_excl_scan(s, &total) + jobs.sums[j][blk];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < len) table[base + k] = run;
    run += v[k];
  }
}

// ------------------------------------------------------------------------------------------------
// Quantised positions (the list build's 8-byte candidates, see k_build_lists_q): written by every producer of a
// sorted pStar — the sort's move, delta-p's epilogue, the slab refresh — so no pass of its own is needed.
// ------------------------------------------------------------------------------------------------
constexpr int QPOS_BITS = 11;                       // sub-cell resolution h / 2048
// Threshold: a pair the exact test accepts has |d| <= 2048 (1 + 1e-5) units; per axis the quantised difference
// is off by < 1 (two floors) + 1.5 (fp32 rounding of (p - min) * k for |coordinate| < 2^22 units: 3 roundings
// of <= 0.25 each, two particles), so |dq| < 2048.03 + 2.5 sqrt(3) = 2052.4 < T.
constexpr uint32_t QPOS_T = (1u << QPOS_BITS) + 5;
typedef short qpair __attribute__((ext_vector_type(2)));

template <typename N> __device__ inline uint2 quantise_position(const StepConsts<N> &c, const vec4<N> &p, bool *usable) {
  const N k = N(1u << QPOS_BITS) / c.h;
  const N fx = floor((p.x - c.minExtent[0]) * k), fy = floor((p.y - c.minExtent[1]) * k),
          fz = floor((p.z - c.minExtent[2]) * k);
  const N lim = N(1 << 22);
  *usable = fabs(fx) < lim && fabs(fy) < lim && fabs(fz) < lim;  // (false for NaN too)
  const uint32_t x = uint32_t(int32_t(fx)), y = uint32_t(int32_t(fy)), z = uint32_t(int32_t(fz));
  return make_uint2((x & 0xFFFFu) | (y << 16), z & 0xFFFFu);
}


// ------------------------------------------------------------------------------------------------
// Counting-sort scatter, made deterministic                      (reference: ompsph.hpp:157-159)
//   pass A: slot inside the cell from an atomic (arbitrary order), records the source index;
//           atomicSub returns the histogram to zero for the next step (no memset);
//   pass B: each particle's final rank inside its cell = number of cell-mates with a smaller
//           source index, i.e. a STABLE sort by key — run-to-run reproducible — then the whole
//           record moves to its sorted slot (writes stay inside one cell's short range).
// ------------------------------------------------------------------------------------------------
//   pile-ups: pass B's rank is a count over the cell's m members — O(m) per particle.  A cell with more than
//           BIG_CELL members (fluid piled into one cell, or the overflow bucket of particles outside the grid) is
//           listed by pass A and its segment of permTmp is SORTED first (k_sort_big_cells, O(m log^2 m) per cell);
//           pass B then reads the rank off the position.  Same permutation either way.
constexpr uint32_t BIG_CELL = 2048;

__global__ __launch_bounds__(BLOCK) void k_scatter_slots(uint32_t n, uint32_t tableN, const uint32_t *__restrict__ key,
                                                         const uint32_t *__restrict__ table,
                                                         uint32_t *__restrict__ count,
                                                         uint32_t *__restrict__ permTmp,
                                                         uint32_t *__restrict__ bigCells,
                                                         uint32_t *__restrict__ nBig) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = key[i];
  if (k == DEAD_KEY) return;  // pbf_slab_step: a slot whose particle has left takes no place in the sorted array
  const uint32_t b = min(k, tableN);
  uint32_t before, rank;
  wave_bucket_atomic(b, [&](uint32_t bb, uint32_t cnt) { return atomicSub(&count[bb], cnt); }, before, rank);
  const uint32_t r = before - 1u - rank;  // r runs m - 1 .. 0 over the cell's m members (which member gets which: arbitrary)
  permTmp[table[b] + r] = i;
  if (r == BIG_CELL) bigCells[atomicAdd(nBig, 1u)] = b;  // exactly one member of a cell with m > BIG_CELL sees this
}

// In-place bitonic sort (all compare-exchanges ascending: the "mirror" form, so the virtual +inf padding beyond the
// segment never moves) of the permTmp segment of every listed cell; one workgroup per cell, grid-strided.  Ordinary
// cells order by source index; the overflow bucket (keys differ) by (key, source index).
__global__ __launch_bounds__(BLOCK) void k_sort_big_cells(const uint32_t *__restrict__ table, uint32_t tableN,
                                                          const uint32_t *__restrict__ bigCells,
                                                          const uint32_t *__restrict__ nBig,
                                                          uint32_t *__restrict__ permTmp,
                                                          const uint32_t *__restrict__ key) {
  const uint32_t cells = *nBig;
  for (uint32_t c = blockIdx.x; c < cells; c += gridDim.x) {
    const uint32_t b = bigCells[c];
    const uint32_t lo = table[b], m = table[b + 1] - lo;
    uint32_t *a = permTmp + lo;
    const bool byKey = b == tableN;
    auto exchange = [&](uint32_t i, uint32_t l) {  // i < l < m
      const uint32_t x = a[i], y = a[l];
      bool swap;
      if (byKey) {
        const uint32_t kx = key[x], ky = key[y];
        swap = ky < kx || (ky == kx && y < x);
      } else {
        swap = y < x;
      }
      if (swap) a[i] = y, a[l] = x;
    };
    uint32_t P = 1;
    while (P < m) P <<= 1;
    for (uint32_t k = 2; k <= P; k <<= 1) {
      for (uint32_t i = threadIdx.x; i < m; i += BLOCK) {
        const uint32_t l = i ^ (k - 1u);
        if (l > i && l < m) exchange(i, l);
      }
      __syncthreads();
      for (uint32_t j = k >> 2; j > 0; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < m; i += BLOCK) {
          const uint32_t l = i ^ j;
          if (l > i && l < m) exchange(i, l);
        }
        __syncthreads();
      }
    }
  }
}

template <typename N> struct ParticleArrays {
  vec4<N> *pos4, *vel4, *col4, *pstar;
  uint64_t *id;
  uint8_t *type;
  uint32_t *key;
};

// ------------------------------------------------------------------------------------------------
// Row-major working set of the solver iterations (round 3, option "row_major").  The Morton order keeps a cell's 27
// neighbours within a few cache lines of each other, but it scatters the three x cells of a (dy, dz) row: the walk has to
// treat a row as two runs (pair + single, parity-dependent), pad the first to an even length and select, per load, which run
// a slot belongs to — 15.6 VALU per candidate slot against the 5 of the distance test itself.  So the data the K iterations
// work on ({pStar, lambda}, the quantised copy, mass, type) is ALSO laid out cell-row-major — cell (x, y, z) of the box at
// linear index (z EY + y) EX + x, x fastest; inside a cell the Morton-sorted order, i.e. the reference's — where the three
// x cells of a row are ONE contiguous run: one 16-byte table load per row, a walk with no selects and no padding, lists of
// row slots, candidates gathered from the row copy.  Same candidates in the same order as the Morton walk => the same bits.
// The row grid is the cube [0, P)^3, P = the power of two above the largest extent: every cell the reference's table knows
// (Morton code < tableN) has its coordinates below P — including the ones beyond the extent proper, which the walk does
// visit (a particle predicted through the floor sits in such a cell, and the floor's walkers find it there).  A cell of the
// cube whose code is tableN - 1 or more is empty for every walker (sph.hpp:206-208) and stays empty here; the particles with
// such keys — in no cell — sit behind the last cell and walk the Morton table themselves, translating what they find through
// rowSlotOf (`ROW_FALLBACK`).  At the cube's faces x - 1 / x + 1 wrap to codes >= tableN in the reference: the run is clamped.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t ROW_FALLBACK = 1u << 30;  // rowXYZ flag: this walker takes the Morton table
template <typename N> struct RowArrays {
  vec4<N> *pstar;      // [slot] {pStar.xyz, lambda}: the live Jacobi buffer at the time of the sort
  N *mass;             // [slot]
  uint2 *qpos;         // [slot] quantised pStar
  uint32_t *xyz;       // [slot] cell coordinates x | y << 10 | z << 20 (| ROW_FALLBACK)
  uint8_t *type;       // [slot]
  uint32_t *slotOf;    // [Morton-sorted index] -> slot
  vec4<N> *col;        // [slot] colour before the step's diffusion (k_diffuse_rows reads its runs from here); may be NULL
  uint32_t *mortonOf;  // [slot] -> Morton-sorted index (with col)
  const uint32_t *lintab;  // [cell (z P + y) P + x] first slot of the cell; [P^3] = first slot behind the cells
  uint32_t *tail;      // allocator of the slots behind the cells
  uint32_t pshift;     // P = 1 << pshift
};
__host__ __device__ inline bool row_cell_of(uint32_t key, uint32_t tableN, uint32_t pshift, uint32_t *lin) {
  const uint32_t x = compact10(key), y = compact10(key >> 1), z = compact10(key >> 2);
  *lin = (((z << pshift) | y) << pshift) | x;
  return key + 1u < tableN;  // (a key >= tableN - 1 is "in no cell" for its neighbours, sph.hpp:206-208; below it, x, y, z < P)
}

// per cell of the cube: its population, read off the cell histogram k_predict built (one lane per Morton code) — BEFORE the
// scans, so that the Morton table's and the row table's run in the same launches
// (over every code of the cube — a bijection onto its cells, so no memset: a cell the table does not know is written 0)
__global__ __launch_bounds__(BLOCK) void k_lin_count(uint32_t tableN, uint32_t pshift, const uint32_t *__restrict__ count,
                                                     uint32_t *__restrict__ linCount) {
  const uint32_t code = blockIdx.x * BLOCK + threadIdx.x;
  if (code >= (1u << (3u * pshift))) {
    if (code == (1u << (3u * pshift))) linCount[code] = 0u, linCount[code + 1u] = 0u, linCount[code + 2u] = 0u;  // scan sentinel, tail allocator, segment counter
    return;
  }
  uint32_t lin;
  const bool known = row_cell_of(code, tableN, pshift, &lin);
  linCount[lin] = known ? count[code] : 0u;  // (known: code < tableN - 1 — its histogram bin holds exactly the particles with that key)
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_rank_move(StepConsts<N> c, uint32_t n, uint32_t tableN,
                                                     const uint32_t *__restrict__ permTmp,
                                                     const uint32_t *__restrict__ table, ParticleArrays<N> src,
                                                     ParticleArrays<N> dst, uint32_t *__restrict__ slotOf,
                                                     uint2 *__restrict__ qpos, RowArrays<N> row) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = permTmp[i];
  const uint32_t k = src.key[s];
  const uint32_t b = min(k, tableN);
  const uint32_t lo = table[b], hi = table[b + 1];
  uint32_t rank = 0;
  if (hi - lo > BIG_CELL) {
    rank = i - lo;  // k_sort_big_cells has sorted this segment
  } else if (b < tableN) {
    for (uint32_t j = lo; j < hi; j += 4u) {  // four cell mates per trip: their loads are in flight together
      uint32_t v[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) v[k] = permTmp[min(j + k, hi - 1u)];
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) rank += (j + k < hi && v[k] < s) ? 1u : 0u;
    }
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
  const vec4<N> p4 = src.pos4[s];
  dst.pos4[d] = p4;
  dst.vel4[d] = src.vel4[s];
  const vec4<N> cl = src.col4[s];
  dst.col4[d] = cl;
  const vec4<N> ps = src.pstar[s];
  bool usable;
  const uint2 q = quantise_position<N>(c, ps, &usable);
  if (!row.lintab) {  // (row-major iterations: {pStar, lambda} and its quantised copy live in the row arrays from here on)
    dst.pstar[d] = ps;
    qpos[d] = q;  // what the first list build of the step tests candidates on
  }
  dst.id[d] = src.id[s];
  const uint8_t ty = src.type[s];
  dst.type[d] = ty;
  dst.key[d] = k;
  if (slotOf) slotOf[s] = d;  // slab mode: where did pre-sort particle s go
  if (row.lintab) {  // the iterations' row-major copy: cell-row-major, inside a cell the Morton-sorted (= reference) order
    uint32_t lin, slot;
    const uint32_t x = compact10(k), y = compact10(k >> 1), z = compact10(k >> 2);
    uint32_t flags = 0;
    if (row_cell_of(k, tableN, row.pshift, &lin)) {
      slot = row.lintab[lin] + rank;
      // (P = 1024 only: the reference's 10-bit wrap makes cell 1023 a neighbour of cell 0 — those walkers take the Morton table)
      if (row.pshift == 10u && (x == 0u || y == 0u || z == 0u || x == 1023u || y == 1023u || z == 1023u)) flags = ROW_FALLBACK;
    } else {
      slot = row.lintab[1u << (3u * row.pshift)] + atomicAdd(row.tail, 1u);  // (nobody finds these through a row run: any order)
      flags = ROW_FALLBACK;
    }
    row.pstar[slot] = ps, row.mass[slot] = p4.w, row.qpos[slot] = q, row.type[slot] = ty;
    row.xyz[slot] = x | (y << 10) | (z << 20) | flags;
    row.slotOf[d] = slot;
    if (row.col) row.col[slot] = cl, row.mortonOf[slot] = d;
  }
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

// Same walk, candidates handed over two at a time (f2(b, b + 1)) so that their pair terms share a
// basic block; visiting order is unchanged.
template <typename F1, typename F2>
__device__ inline void for_each_candidate2(uint32_t key, const uint32_t *__restrict__ table, uint32_t tableN, F1 &&f1,
                                           F2 &&f2) {
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
        uint32_t b = start;
        for (; b + 1u < end; b += 2u) f2(b, b + 1u);
        if (b < end) f1(b);
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
__device__ inline float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline double fast_rsq(double x) { return 1.0 / sqrt(x); }  // fp64 keeps the IEEE forms

// Correctly rounded sqrt and divide for the PRECISE pair terms, trimmed to the operand range they see and to what the
// VALU issues fastest (tools/valu_rate.hip, profiles/r02_valu_rates.md: fma / add / mul issue at full rate; compares,
// selects, min / max, shifts and the packed forms at 4/7 of it; v_sqrt / v_rcp / v_rsq at 2/7).  hipcc's IEEE sqrtf is
// v_sqrt_f32 + a +-1 ulp fix-up (two compares, two selects) inside a 2^32 rescale for tiny inputs: 16 VALU; its divide
// v_div_scale x 2 + v_rcp + Newton + v_div_fmas + v_div_fixup: 11.
//   * sqrt_rsq: ONE v_rsq_f32 gives y ~ 1/sqrt(x); g = x y, then one Heron step in fma form, g + (x - g^2) (y / 2):
//     no compare, no select, and y comes out as a by-product ~ 1 / r.  Equal to IEEE sqrtf bit for bit for every fp32
//     x >= 2^-75 (k_selftest_math sweeps them all).  The addend 2^-100 keeps x = 0 (a particle meeting itself) away
//     from v_rsq's infinity and is absorbed exactly by every x >= 2^-75; below that — r < 5.5e-12, far under EPSILON —
//     every small r gives the same pair terms: the spiky branch is off and h^2 - r^2 rounds to h^2.
//   * div_seeded: (h - r)^2 / r by Newton from that y: one refinement of the reciprocal, the quotient, one residual
//     correction — 5 fma / mul, no v_rcp (a second correction, as a general-purpose divide would carry, changes no
//     result on the swept operands).  With 1e-8 <= r <= h nothing needs scaling or fixing and the result is the
//     IEEE quotient bit for bit (swept: every fp32 d2 whose root lies in [1e-8, h], four h); outside that range the
//     quotient is never used (selected away).
//   * div_ranged (reciprocal of a per-launch CONSTANT divisor, hoisted out of the loops by the compiler) serves
//     delta-p's poly6(r) / poly6(0.3 h) — numerator 0 or 1e-18 ... 1.6e3 in magnitude; k_selftest_math sweeps EVERY
//     fp32 numerator between 1e-30 and 1e30: identical but for -0 -> +0, and the quotient is only ever squared — and
//     (lambda_a + lambda_b + corr) / RHO, whose numerator can be a denormal (two lambdas cancelling exactly next to
//     r = h): identical for every |numerator| >= 2^-100 (swept; again but for the sign of a zero, which only ever
//     multiplies into a term that is added to a sum), and DeltaOp falls back to the compiler's divide for a whole wave
//     when one lane's numerator is smaller than that.
__device__ inline float sqrt_rsq(float x, float &y) {
  x += 0x1p-100f;
  y = __builtin_amdgcn_rsqf(x);
  const float g = x * y, hy = 0.5f * y;
  return fmaf(fmaf(-g, g, x), hy, g);
}
// fp64 (round 3): hipcc's IEEE sqrt(double) is v_rsq_f64 + one coupled Goldschmidt step + two residual corrections
// (7 fma, 2 mul) inside a 2^256 rescale for x < 2^-767 (compare, two selects, two v_ldexp) and a class test with two more
// selects: 18 VALU.  The pair terms only ever feed it d2 >= 0, so the trimmed form below executes the VERY SAME arithmetic
// instructions without the wrappers — bit-identical BY CONSTRUCTION wherever the rescale is the identity (x >= 2^-767) —
// in 10 VALU, and hands out y ~ 1 / sqrt(x) (2 h, relative error ~2^-51) as a by-product.  The addend 2^-600 keeps x = 0 (a
// particle meeting itself) away from v_rsq's infinity and is absorbed exactly by every x >= 2^-547; below that
// (r < 2^-273, far under EPSILON) every small r yields the same pair terms.  An exhaustive sweep is impossible in fp64:
// k_selftest_math64 compares > 10^10 pseudo-random operands (whole exponent range + the pair terms' own range, densely)
// with the compiler's forms — pbf_selftest_math on an fp64 context.
__device__ inline double sqrt_rsq(double x, double &y) {
  x += 0x1p-600;
  const double r = __builtin_amdgcn_rsq(x);
  double g = x * r, hh = r * 0.5;
  const double e = fma(-hh, g, 0.5);
  g = fma(g, e, g), hh = fma(hh, e, hh);
  double d = fma(-g, g, x);
  g = fma(d, hh, g);
  d = fma(-g, g, x);
  g = fma(d, hh, g);
  y = hh + hh;
  return g;
}
__device__ inline float div_seeded(float a, float b, float y) {
  y = fmaf(fmaf(-b, y, 1.0f), y, y);
  const float q = a * y;
  return fmaf(fmaf(-b, q, a), y, q);  // ONE residual correction: relative error ~2^-47 before the final rounding — enough
                                      // for every operand the sweeps visit (k_selftest_math), which is every operand used
}
// a / b in fp64 from a seed y ~ 1 / b: hipcc's IEEE divide is v_div_scale x 2, v_rcp_f64, two Newton steps on the
// reciprocal, the quotient, its exact residual (fma) and one correction (v_div_fmas), then v_div_fixup: 11 VALU, two of
// them quarter-rate.  With operands in the pair terms' range nothing needs scaling or fixing.  Error-bound argument for the
// form below (Markstein 1990; Muller et al., Handbook of FP Arithmetic, thm. "correcting a quotient with an fma"): let y1 be
// the refined reciprocal, |y1 - 1/b| <= (1/2 + 2^-40) ulp(1/b) (y0 carries <= 2^-26 — v_rcp_f64 — or ~2^-51 — the by-product
// of sqrt_rsq — so NEWTON steps leave (2^-26)^(2^NEWTON) resp. 2^-102 of it plus one rounding), q0 = RN(a y1) is then within 1 ulp of
// a / b, res = a - b q0 is EXACT in one fma (Sterbenz-type cancellation), and q0 + res y1 differs from a / b by less than
// 2^-52 ulp(q0) — below half the smallest distance between a quotient of two doubles and a rounding boundary whenever y1
// is the correctly rounded reciprocal, which the sweep checks rather than assumes: mismatches against the compiler's divide
// over > 10^10 operands must be ZERO (k_selftest_math64).
template <int NEWTON> __device__ inline double div_newton(double a, double b, double y) {
#pragma unroll
  for (int k = 0; k < NEWTON; ++k) y = fma(y, fma(-b, y, 1.0), y);
  const double q = a * y;
  return fma(fma(-b, q, a), y, q);
}
__device__ inline double div_seeded(double a, double b, double y) { return div_newton<1>(a, b, y); }  // seed: sqrt_rsq's y
__device__ inline float div_ranged(float a, float b) { return div_seeded(a, b, __builtin_amdgcn_rcpf(b)); }
// divisor = a per-launch constant: the reciprocal and its two Newton steps are loop-invariant (hoisted), 3 fma / mul remain
__device__ inline double div_ranged(double a, double b) { return div_newton<2>(a, b, __builtin_amdgcn_rcp(b)); }
// The numerators div_ranged is the IEEE divide for: finite and at least 2^-100 in magnitude (x * 0 + x turns an
// infinity into a NaN, which fails the compare like a NaN, a zero or a tiny numerator does)
__device__ inline bool div_ranged_ok(float x) { return fabsf(fmaf(x, 0.0f, x)) >= 0x1p-100f; }
__device__ inline bool div_ranged_ok(double x) { return fabs(fma(x, 0.0, x)) >= 0x1p-700; }  // (no scaling by v_div_scale)

template <typename N, bool FAST>
__device__ inline PairGeom<N, FAST> pair_geom(const vec4<N> &a, const vec4<N> &b, N h) {
  PairGeom<N, FAST> g;
  const N bx = b.x - a.x, by = b.y - a.y, bz = b.z - a.z;  // distance(a,b) = length(b - a)
  // a - b = -(b - a) exactly (round to nearest is sign-symmetric), but for the sign of a zero difference — and a zero
  // difference only ever multiplies into a term that is ADDED to a running sum, which +0 and -0 leave alike (the sums
  // start at +0 and can never become -0).  As a negation it is a free source modifier instead of three subtractions.
  g.dx = -bx, g.dy = -by, g.dz = -bz;
  if constexpr (FAST) {
    const N d2 = fma(bz, bz, fma(by, by, bx * bx));
    const N rinv = fast_rsq(d2);
    g.inSpiky = d2 <= h * h && d2 >= N(EPSILON) * N(EPSILON);
    g.inH = d2 <= h * h;
    g.r = g.inSpiky ? d2 * rinv : N(0);
    const N hr = h - g.r;
    g.hr2_over_r = (hr * hr) * rinv;
  } else {
    const N d2 = bx * bx + by * by + bz * bz;
    N rinv;
    g.r = sqrt_rsq(d2, rinv);
    g.inH = g.r <= h;
    g.inSpiky = g.inH && g.r >= N(EPSILON);
    const N hr = h - g.r;
    g.hr2_over_r = div_seeded(hr * hr, g.r, rinv);  // (only used where inSpiky)
  }
  return g;
}

// Conservative "may be within h" test used to FILTER candidates before the exact pair terms:
// d2 (any rounding order) <= h^2 (1 + 1e-5) holds for every pair the exact test r <= h admits, and a
// candidate it rejects contributes exactly +0 to every sum — so filtering never changes a bit.
template <typename N> __device__ inline bool maybe_within_h(const vec4<N> &a, const vec4<N> &b, N h2filter) {
  const N bx = b.x - a.x, by = b.y - a.y, bz = b.z - a.z;
  return fma(bz, bz, fma(by, by, bx * bx)) <= h2filter;
}

// ------------------------------------------------------------------------------------------------
// The three 27-cell gather stages as "ops": begin(i) loads particle a (false = nothing to do),
// add(b) folds one candidate in the reference's visiting order, end(i) stores the result.
// All gather kernels below (global walk, filtered lists + list-driven, LDS bricks) run the SAME op
// code in the SAME candidate order, so their results are bit-identical to each other and to the oracle.
// ------------------------------------------------------------------------------------------------

// diffuse (ompsph.hpp:188-207), Jacobi like the OpenCL kernel (ocl/oclsph_kernel.h:67-93)
template <typename N> struct DiffuseOp {
  using Src = vec4<N>;
  struct Args {
    const vec4<N> *colIn;
    vec4<N> *colOut;
    const uint8_t *type;
  };
  static constexpr bool kNeedsCandidateType = true;  // obstacles are skipped as candidates (ompsph.hpp:194)
  static constexpr bool kFilter = false;             // no distance test in diffuse: every candidate counts
  static constexpr bool kTileable = true;            // has a single source array the brick kernel can stage
  __device__ bool near(const StepConsts<N> &, const Src &) const { return true; }
  __host__ __device__ static const Src *src(const Args &a) { return a.colIn; }
  __device__ static Src load(const Args &a, uint32_t b) { return a.colIn[b]; }
  vec4<N> ca;
  N mx, my, mz, mw;
  int nn;
  __device__ bool begin(const StepConsts<N> &c, const Args &a, uint32_t i) {
    ca = a.colIn[i];
    mx = my = mz = mw = N(0);
    nn = 0;
    if (c.hasObstacles && a.type[i] != 0) {  // obstacle, or a ghost copy owned by the neighbouring slab
      a.colOut[i] = ca;
      return false;
    }
    return true;
  }
  __device__ void add(const StepConsts<N> &, const Src &cb) {
    mx += cb.x, my += cb.y, mz += cb.z, mw += cb.w;
    ++nn;
  }
  __device__ void add_bf(const StepConsts<N> &c, const Src &cb, bool valid = true) {
    if (valid) add(c, cb);
  }
  __device__ void end(const StepConsts<N> &c, const Args &a, uint32_t i) {
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
    a.colOut[i] = out;
  }
};

// lambda (ompsph.hpp:217-232): rho = sum m_a poly6; g = sum grad spiky / rho0;
// lambda = -(rho/rho0 - 1) / (|g|^2 + 600).  Written into pstar[a].w (only xyz is read here).
template <typename N, bool FAST> struct LambdaOp {
  using Src = vec4<N>;
  struct Args {
    vec4<N> *pstar;
    const vec4<N> *pos4;  // (mass = pos4[i].w) — or, pos4 == NULL, the row-major copy's plain mass array:
    const uint8_t *type;
    const N *mass = nullptr;
  };
  static constexpr bool kNeedsCandidateType = false;
  static constexpr bool kFilter = true;
  static constexpr bool kTileable = true;
  __device__ bool near(const StepConsts<N> &c, const Src &pb) const { return maybe_within_h<N>(pa, pb, c.h2filter); }
  __host__ __device__ static const Src *src(const Args &a) { return a.pstar; }
  __device__ static Src load(const Args &a, uint32_t b) {
#ifdef PBF_LAMBDA_X4
    Src v = a.pstar[b];
    asm volatile("" : "+v"(v.w));  // keep the fourth component: one 16-byte gather instead of a 12-byte one
    return v;
#else
    return a.pstar[b];
#endif
  }
  vec4<N> pa;
  N mass, gx, gy, gz, rho;
  __device__ bool begin(const StepConsts<N> &c, const Args &a, uint32_t i) {
    if (c.hasObstacles && a.type[i] != 0) {
      if (a.type[i] & 1) a.pstar[i].w = N(0);  // obstacle: lambda = 0 (ompsph.hpp:218-221); a ghost keeps its owner's
      return false;
    }
    pa = a.pstar[i];
    mass = a.pos4 ? a.pos4[i].w : a.mass[i];
    gx = gy = gz = rho = N(0);
    return true;
  }
  __device__ void add(const StepConsts<N> &c, const Src &pb) {
    const auto g = pair_geom<N, FAST>(pa, pb, c.h);
    if (g.inSpiky) {  // wave-level skip: lanes of one cell test the same candidate, so whole waves often miss
      const N s = c.spikyFactor * g.hr2_over_r;
      gx += (g.dx * s) * N(RHO_RECIP), gy += (g.dy * s) * N(RHO_RECIP), gz += (g.dz * s) * N(RHO_RECIP);
    }
    if (g.inH) {
      const N d = (c.h * c.h) - g.r * g.r;
      rho += mass * (c.poly6Factor * (d * d * d));
    }
  }
  __device__ void add_bf(const StepConsts<N> &c, const Src &pb, bool valid = true) {
    // branch-free form for the list drain (`valid` = false turns a padding entry into +0): two consecutive calls form one basic block and their long
    // sqrt / divide chains interleave.  An excluded pair contributes exactly +0 (a select, never a
    // multiply: r = 0 makes hr2_over_r infinite), which leaves every partial sum bit-identical.
    const auto g = pair_geom<N, FAST>(pa, pb, c.h);
    const bool sp = g.inSpiky && valid, ih = g.inH && valid;
    // ONE select, on the common factor (r = 0 makes hr2_over_r infinite or NaN): the three products of an excluded
    // pair are then (finite) * 0 * RHO_RECIP = +-0
    const N s = sp ? c.spikyFactor * g.hr2_over_r : N(0);
    gx += (g.dx * s) * N(RHO_RECIP), gy += (g.dy * s) * N(RHO_RECIP), gz += (g.dz * s) * N(RHO_RECIP);
    const N d = (c.h * c.h) - g.r * g.r;
    const N w = mass * (c.poly6Factor * (d * d * d));
    rho += ih ? w : N(0);
  }
  // the partial sums of several lanes that shared this particle's list (k_gather_from_lists_coop)
  template <typename R> __device__ void combine(R &&r) { gx = r(gx), gy = r(gy), gz = r(gz), rho = r(rho); }
  __device__ void end(const StepConsts<N> &, const Args &a, uint32_t i) {
    const N norm2 = gx * gx + gy * gy + gz * gz;
    const N Ci = rho / N(RHO) - N(1);
    a.pstar[i].w = -Ci / (norm2 + N(CFM_EPSILON));
  }
};

// delta-p + clamp (ompsph.hpp:235-248), Jacobi: reads pstarIn (xyz + lambda), writes pstarOut.
template <typename N, bool FAST> struct DeltaOp {
  using Src = vec4<N>;
  struct Args {
    const vec4<N> *pstarIn;
    vec4<N> *pstarOut;
    const uint8_t *type;
    uint2 *qpos;  // may be NULL: the quantised copy of pstarOut for the next iteration's list build
  };
  static constexpr bool kNeedsCandidateType = false;
  static constexpr bool kFilter = true;
  static constexpr bool kTileable = true;
  __device__ bool near(const StepConsts<N> &c, const Src &pb) const { return maybe_within_h<N>(pa, pb, c.h2filter); }
  __host__ __device__ static const Src *src(const Args &a) { return a.pstarIn; }
  __device__ static Src load(const Args &a, uint32_t b) { return a.pstarIn[b]; }
  vec4<N> pa;
  N ax, ay, az;
  __device__ bool begin(const StepConsts<N> &c, const Args &a, uint32_t i) {
    pa = a.pstarIn[i];
    ax = ay = az = N(0);
    if (c.hasObstacles && a.type[i] != 0) {
      a.pstarOut[i] = pa;
      return false;
    }
    return true;
  }
  // (lambda_a + lambda_b + corr) / RHO: the trimmed divide where it IS the IEEE one (div_ranged_ok), the compiler's
  // otherwise — decided per wave, so the common path carries one fma and one compare and no divergence
  __device__ static N over_rho(N num) {
    if constexpr (FAST) {
      return num / N(RHO);
    } else {
      if (__builtin_expect(__any(!div_ranged_ok(num)), 0)) return num / N(RHO);
      return div_ranged(num, N(RHO));
    }
  }
  __device__ void add(const StepConsts<N> &c, const Src &pb) {
    const auto g = pair_geom<N, FAST>(pa, pb, c.h);
    if (g.inSpiky) {  // outside it the gradient is zero, so corr / factor are irrelevant
      const N d = (c.h * c.h) - g.r * g.r;
      const N q = FAST ? (c.poly6Factor * (d * d * d)) / c.p6DeltaQ : div_ranged(c.poly6Factor * (d * d * d), c.p6DeltaQ);
      const N q2 = q * q;
      const N corr = N(-CorrK) * (q2 * q2);  // pow(q, CorrN = 4) of ompsph.hpp:240
      const N factor = (pa.w + pb.w + corr) / N(RHO);
      const N s = c.spikyFactor * g.hr2_over_r;
      ax += (g.dx * s) * factor, ay += (g.dy * s) * factor, az += (g.dz * s) * factor;
    }
  }
  __device__ void add_bf(const StepConsts<N> &c, const Src &pb, bool valid = true) {
    // branch-free like LambdaOp::add_bf; outside the spiky support the gradient is exactly zero, so
    // corr / factor are irrelevant there: both factors of an excluded pair are selected to 0 (a lambda of a particle
    // outside the support may be anything, an infinity included) and the products add +-0
    const auto g = pair_geom<N, FAST>(pa, pb, c.h);
    const bool sp = g.inSpiky && valid;
    const N d = (c.h * c.h) - g.r * g.r;
    const N q = FAST ? (c.poly6Factor * (d * d * d)) / c.p6DeltaQ : div_ranged(c.poly6Factor * (d * d * d), c.p6DeltaQ);
    const N q2 = q * q;
    const N corr = N(-CorrK) * (q2 * q2);  // pow(q, CorrN = 4) of ompsph.hpp:240
    const N f = over_rho(pa.w + pb.w + corr);
    const N factor = sp ? f : N(0), s = sp ? c.spikyFactor * g.hr2_over_r : N(0);
    ax += (g.dx * s) * factor, ay += (g.dy * s) * factor, az += (g.dz * s) * factor;
  }
  template <typename R> __device__ void combine(R &&r) { ax = r(ax), ay = r(ay), az = r(az); }
  __device__ void end(const StepConsts<N> &c, const Args &a, uint32_t i) {
    N x = (pa.x + ax) * c.scale, y = (pa.y + ay) * c.scale, z = (pa.z + az) * c.scale;
    x = min(c.maxB[0], max(c.minB[0], x));
    y = min(c.maxB[1], max(c.minB[1], y));
    z = min(c.maxB[2], max(c.minB[2], z));
    const vec4<N> out = make_vec4<N>(x / c.scale, y / c.scale, z / c.scale, pa.w);
    a.pstarOut[i] = out;
    if (a.qpos) {
      bool usable;
      a.qpos[i] = quantise_position<N>(c, out, &usable);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Opt-in extras named by the north star but ABSENT from the reference (only their constants survive,
// sph_constants.h:13-14; SURVEY finding 3): vorticity confinement and XSPH viscosity after Macklin &
// Mueller 2013 (eq. 15-17), applied to the post-solve velocity, Jacobi.  Parity unpinned: the only
// checker is our own CPU restatement (oracle extras()).  Candidates need position AND velocity.
// ------------------------------------------------------------------------------------------------
template <typename N> struct PosVel {
  vec4<N> p, v;
};

// omega_a = sum_b (v_b - v_a) x grad_{p_b} W(p_a - p_b)        (Macklin & Mueller 2013 eq. 15: the gradient is taken
// with respect to the NEIGHBOUR's position, = -grad_{p_a} W = -spikyKernelGradient(a, b)), accumulated as
// grad_{p_a} W x (v_b - v_a) — the same number, each component the exact negation of (v_b - v_a) x grad_{p_a} W.
// A rigid rotation about +z then gives omega along +z (curl v = 2 Omega), tests/test_physics_gpu.py.
// (Rounds 1-2 accumulated (v_b - v_a) x grad_{p_a} W: the sign was flipped, the confinement force decelerated vortices.)
template <typename N, bool FAST> struct VorticityOp {
  using Src = PosVel<N>;
  struct Args {
    const vec4<N> *pstar, *vel;
    vec4<N> *omega;
    const uint8_t *type;
  };
  static constexpr bool kNeedsCandidateType = false;
  static constexpr bool kFilter = true;
  static constexpr bool kTileable = false;
  __device__ static Src load(const Args &a, uint32_t b) { return {a.pstar[b], a.vel[b]}; }
  vec4<N> pa, va;
  N wx, wy, wz;
  __device__ bool near(const StepConsts<N> &c, const Src &b) const { return maybe_within_h<N>(pa, b.p, c.h2filter); }
  __device__ bool begin(const StepConsts<N> &c, const Args &a, uint32_t i) {
    wx = wy = wz = N(0);
    if (c.hasObstacles && a.type[i] != 0) {
      a.omega[i] = make_vec4<N>(N(0), N(0), N(0), N(0));
      return false;
    }
    pa = a.pstar[i], va = a.vel[i];
    return true;
  }
  __device__ void add(const StepConsts<N> &c, const Src &b) { add_bf(c, b); }
  __device__ void add_bf(const StepConsts<N> &c, const Src &b, bool valid = true) {
    if (!valid) return;
    const auto g = pair_geom<N, FAST>(pa, b.p, c.h);
    const N s = c.spikyFactor * g.hr2_over_r;
    const N gx = g.inSpiky ? g.dx * s : N(0), gy = g.inSpiky ? g.dy * s : N(0), gz = g.inSpiky ? g.dz * s : N(0);
    const N ux = b.v.x - va.x, uy = b.v.y - va.y, uz = b.v.z - va.z;
    wx = wx + (gy * uz - gz * uy), wy = wy + (gz * ux - gx * uz), wz = wz + (gx * uy - gy * ux);
  }
  __device__ void end(const StepConsts<N> &, const Args &a, uint32_t i) { a.omega[i] = make_vec4<N>(wx, wy, wz, N(0)); }
};

// eta = sum_b grad spiky(a, b) |omega_b| ; v_a += (eta/|eta| x omega_a) * (VORTICITY_EPSILON * dt)
template <typename N, bool FAST> struct VorticityForceOp {
  using Src = PosVel<N>;  // .v carries omega
  struct Args {
    const vec4<N> *pstar, *omega, *velIn;
    vec4<N> *velOut;
    const uint8_t *type;
  };
  static constexpr bool kNeedsCandidateType = false;
  static constexpr bool kFilter = true;
  static constexpr bool kTileable = false;
  __device__ static Src load(const Args &a, uint32_t b) { return {a.pstar[b], a.omega[b]}; }
  vec4<N> pa, wa, va;
  N ex, ey, ez;
  __device__ bool near(const StepConsts<N> &c, const Src &b) const { return maybe_within_h<N>(pa, b.p, c.h2filter); }
  __device__ bool begin(const StepConsts<N> &c, const Args &a, uint32_t i) {
    ex = ey = ez = N(0);
    va = a.velIn[i];
    if (c.hasObstacles && a.type[i] != 0) {
      a.velOut[i] = va;
      return false;
    }
    pa = a.pstar[i], wa = a.omega[i];
    return true;
  }
  __device__ void add(const StepConsts<N> &c, const Src &b) { add_bf(c, b); }
  __device__ void add_bf(const StepConsts<N> &c, const Src &b, bool valid = true) {
    if (!valid) return;
    const auto g = pair_geom<N, FAST>(pa, b.p, c.h);
    const N s = c.spikyFactor * g.hr2_over_r;
    const N len = sqrt(b.v.x * b.v.x + b.v.y * b.v.y + b.v.z * b.v.z);
    const N tx = (g.dx * s) * len, ty = (g.dy * s) * len, tz = (g.dz * s) * len;
    ex += g.inSpiky ? tx : N(0), ey += g.inSpiky ? ty : N(0), ez += g.inSpiky ? tz : N(0);
  }
  __device__ void end(const StepConsts<N> &c, const Args &a, uint32_t i) {
    const N len = sqrt(ex * ex + ey * ey + ez * ez);
    vec4<N> v = va;
    if (len > N(EPSILON)) {
      const N inv = N(1) / len;
      const N nx = ex * inv, ny = ey * inv, nz = ez * inv;
      const N k = N(VORTICITY_EPSILON) * c.dt;
      v.x = va.x + (ny * wa.z - nz * wa.y) * k, v.y = va.y + (nz * wa.x - nx * wa.z) * k,
      v.z = va.z + (nx * wa.y - ny * wa.x) * k;
    }
    a.velOut[i] = v;
  }
};

// v_a = v_a + C * sum_b (v_b - v_a) poly6(r)
template <typename N, bool FAST> struct XsphOp {
  using Src = PosVel<N>;
  struct Args {
    const vec4<N> *pstar, *velIn;
    vec4<N> *velOut;
    const uint8_t *type;
  };
  static constexpr bool kNeedsCandidateType = false;
  static constexpr bool kFilter = true;
  static constexpr bool kTileable = false;
  __device__ static Src load(const Args &a, uint32_t b) { return {a.pstar[b], a.velIn[b]}; }
  vec4<N> pa, va;
  N ax, ay, az;
  __device__ bool near(const StepConsts<N> &c, const Src &b) const { return maybe_within_h<N>(pa, b.p, c.h2filter); }
  __device__ bool begin(const StepConsts<N> &c, const Args &a, uint32_t i) {
    ax = ay = az = N(0);
    va = a.velIn[i];
    if (c.hasObstacles && a.type[i] != 0) {
      a.velOut[i] = va;
      return false;
    }
    pa = a.pstar[i];
    return true;
  }
  __device__ void add(const StepConsts<N> &c, const Src &b) { add_bf(c, b); }
  __device__ void add_bf(const StepConsts<N> &c, const Src &b, bool valid = true) {
    if (!valid) return;
    const auto g = pair_geom<N, FAST>(pa, b.p, c.h);
    const N d = (c.h * c.h) - g.r * g.r;
    const N w = g.inH ? c.poly6Factor * (d * d * d) : N(0);
    ax = ax + (b.v.x - va.x) * w, ay = ay + (b.v.y - va.y) * w, az = az + (b.v.z - va.z) * w;
  }
  __device__ void end(const StepConsts<N> &, const Args &a, uint32_t i) {
    vec4<N> v = va;
    v.x = va.x + ax * N(C_XSPH), v.y = va.y + ay * N(C_XSPH), v.z = va.z + az * N(C_XSPH);
    a.velOut[i] = v;
  }
};

// One particle through the global-memory 27-cell walk.
template <typename N, typename Op>
__device__ inline void gather_one_global(const StepConsts<N> &c, const typename Op::Args &args,
                                         const uint32_t *__restrict__ key, const uint32_t *__restrict__ table,
                                         uint32_t i) {
  Op op;
  if (!op.begin(c, args, i)) return;
  if (Op::kNeedsCandidateType && c.hasObstacles) {
    for_each_candidate(key[i], table, c.tableN, [&](uint32_t b) {
      if (!(args.type[b] & 1)) op.add(c, Op::load(args, b));
    });
  } else if (Op::kFilter) {  // lambda / delta: the branchy pair terms skip whole waves, keep one per trip
    for_each_candidate(key[i], table, c.tableN, [&](uint32_t b) { op.add(c, Op::load(args, b)); });
  } else {
    for_each_candidate2(
        key[i], table, c.tableN, [&](uint32_t b) { op.add(c, Op::load(args, b)); },
        [&](uint32_t b0, uint32_t b1) {
          const typename Op::Src p0 = Op::load(args, b0), p1 = Op::load(args, b1);
          op.add(c, p0);
          op.add(c, p1);
        });
  }
  op.end(c, args, i);
}

// ------------------------------------------------------------------------------------------------
// Gather kernel "global" (option gather = 0) — one thread per particle, candidates straight from global memory (L1/L2).
// The simple form: used for A/B comparison (PBF_FLAG_NO_LDS / option gather = 0) and as the
// in-kernel fallback of the brick kernel (pile-ups beyond the LDS tile, diffuse with obstacles).
// ------------------------------------------------------------------------------------------------
template <typename N, typename Op>
__global__ __launch_bounds__(BLOCK) void k_gather_global(StepConsts<N> c, typename Op::Args args,
                                                         const uint32_t *__restrict__ key,
                                                         const uint32_t *__restrict__ table) {
  const uint32_t i = xcd_chunk() * BLOCK + threadIdx.x;
  if (i >= c.n) return;
  gather_one_global<N, Op>(c, args, key, table, i);
}

// ------------------------------------------------------------------------------------------------
// Morton bricks: a Morton-aligned brick of 4 x 4 x BZ cells is 16*BZ consecutive codes, i.e. ONE contiguous run of the
// sorted particles; its 6 x 6 x (BZ+2) halo of cells, staged x-fastest, makes the three x cells of a (dy, dz) row one
// contiguous run.  Used by the per-cell diffusion (k_diffuse_bricks) and the LDS-tile iteration kernels (pbf_tiles.hpp);
// k_brick_list compiles the list of non-empty bricks during the sort stage.  (Round 1's k_gather_bricks — filter and
// pair terms per launch out of such a tile, option gather = 2 — was removed in round 2: superseded by pbf_tiles.hpp.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_brick_list(const uint32_t *__restrict__ table, uint32_t tableN,
                                                      uint32_t home, uint32_t nBricks,
                                                      uint32_t *__restrict__ active,
                                                      uint32_t *__restrict__ nActive) {
  const uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
  if (b >= nBricks) return;
  const uint32_t s = table[b * home], e = table[min((b + 1u) * home, tableN)];
  if (e > s) active[atomicAdd(nActive, 1u)] = b;  // order is arbitrary: it only schedules, never reorders sums
}

template <int BZ> struct Brick {
  static constexpr int HOME = 16 * BZ;            // Morton codes per brick
  static constexpr int HZ = BZ + 2;               // halo depth in z
  static constexpr int HALO = 36 * HZ;            // 6 x 6 x (BZ+2) cells
  static constexpr int HDR = ((2 * HALO + 1) * 4 + 15) / 16 * 16;  // header bytes, keeps the tile 16-B aligned
};

template <int BZ> struct Brick2 : Brick<BZ> {
  static constexpr int HDR2 = ((2 * Brick<BZ>::HALO + 1 + 1) * 4 + 15) / 16 * 16;  // + the ticket word
};

// ------------------------------------------------------------------------------------------------
// Diffuse per CELL (option cell_diffuse, default on).  Diffuse has no distance test: a particle folds in
// EVERY candidate of its 27 cells, in walk order, and only then looks at its own colour — so all the
// particles of one cell accumulate the very same running sums, bit for bit.  One lane per occupied
// cell does the walk once (k_diffuse_bricks; ~7 particles share a cell in the settled dam-break), the
// sums are parked at the cell's first sorted slot, and k_diffuse_apply streams over the particles.
// Particles outside the table (key >= tableN) keep their own walk.
// ------------------------------------------------------------------------------------------------
// The same per-cell sums with the candidates staged through LDS: one workgroup owns a Morton-aligned
// brick of 4 x 4 x 4 cells (64 consecutive codes), copies the colours of its 6 x 6 x 6 halo of cells
// into LDS once (x-fastest, so the three x cells of a row are ONE LDS run) and one
// lane per home cell then folds its 9 runs in the reference's order.  One lane per cell straight from
// global memory touches 64 different cache lines per load (measured: no faster than one lane per
// particle); out of LDS the walk is bound by LDS bandwidth (64 cells x ~180 records x 16 B per brick;
// a smaller tile with twice the workgroups per CU was measured: no faster).  A brick whose halo exceeds
// `cap` records, or any brick when there are obstacles (candidate types), walks globally.
template <typename N> __device__ inline void diffuse_cell_global(const StepConsts<N> &c, uint32_t code,
                                                                 const vec4<N> *__restrict__ colIn,
                                                                 const uint8_t *__restrict__ type,
                                                                 const uint32_t *__restrict__ table,
                                                                 vec4<N> *__restrict__ cellSum,
                                                                 uint32_t *__restrict__ cellCnt) {
  const uint32_t first = table[code];
  if (table[code + 1u] == first) return;  // the TRUE count (the table keeps entry tableN): the last cell's walkers count
  N mx = N(0), my = N(0), mz = N(0), mw = N(0);
  uint32_t nn = 0;
  auto add = [&](uint32_t b) {
    const vec4<N> cb = colIn[b];
    mx += cb.x, my += cb.y, mz += cb.z, mw += cb.w;
    ++nn;
  };
  if (c.hasObstacles) {
    for_each_candidate(code, table, c.tableN, [&](uint32_t b) {
      if (!(type[b] & 1)) add(b);  // obstacles are skipped as candidates (ompsph.hpp:194)
    });
  } else {
    for_each_candidate(code, table, c.tableN, add);
  }
  cellSum[first] = make_vec4<N>(mx, my, mz, mw);
  cellCnt[first] = nn;
}

constexpr int DIFFUSE_BRICK_THREADS = 256;
template <typename N>
__global__ __launch_bounds__(DIFFUSE_BRICK_THREADS) void k_diffuse_bricks(StepConsts<N> c,
                                                                          const vec4<N> *__restrict__ colIn,
                                                                          const uint8_t *__restrict__ type,
                                                                          const uint32_t *__restrict__ table,
                                                                          const uint32_t *__restrict__ active,
                                                                          const uint32_t *__restrict__ nActivePtr,
                                                                          vec4<N> *__restrict__ cellSum,
                                                                          uint32_t *__restrict__ cellCnt, uint32_t cap) {
  using B = Brick<4>;
  constexpr int THREADS = DIFFUSE_BRICK_THREADS;
  static_assert(B::HALO <= THREADS && B::HOME <= THREADS, "one halo cell per thread");
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t *off = reinterpret_cast<uint32_t *>(smem);  // [HALO + 1]
  uint32_t *gstart = off + B::HALO + 1;                // [HALO]
  vec4<N> *tile = reinterpret_cast<vec4<N> *>(smem + B::HDR);
  const uint32_t tid = threadIdx.x;
  const uint32_t nActive = *nActivePtr;
  // persistent workgroups stride over the list of NON-EMPTY bricks (k_brick_list, built by the sort stage): a
  // workgroup per brick of the whole table spends most of the launch retiring empty ones
  for (uint32_t t = blockIdx.x; t < nActive; t += gridDim.x) {
    __syncthreads();  // the previous brick's LDS reads are done before the header / tile are rewritten
    const uint32_t code0 = active[t] * uint32_t(B::HOME);
    // ---- the halo's cell ranges and their exclusive scan --------------------------------------------
    const uint32_t bx = compact10(code0), by = compact10(code0 >> 1), bz = compact10(code0 >> 2);
    uint32_t cnt = 0;
    if (tid < B::HALO) {
      const uint32_t lx = tid % 6, ly = (tid / 6) % 6, lz = tid / 36;
      const uint32_t code = morton_encode((bx + lx - 1u) & 1023u, (by + ly - 1u) & 1023u, (bz + lz - 1u) & 1023u);
      uint32_t s = 0, e = 0;
      if (code < c.tableN) {  // sph.hpp:206-208
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
    const uint32_t code = code0 + tid;
    const bool home = tid < B::HOME && code < c.tableN;
    if (total > cap || c.hasObstacles) {  // uniform
      if (home) diffuse_cell_global<N>(c, code, colIn, type, table, cellSum, cellCnt);
      continue;
    }
    // ---- stage the halo's colours (all threads, record r -> its halo cell by bisection of off[]) -----
    constexpr uint32_t U = 8;  // loads in flight per thread: the whole halo is usually one batch
    for (uint32_t base = tid; base < total; base += THREADS * U) {
      vec4<N> v[U];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint32_t r = base + u * THREADS;
        if (r < total) {
          uint32_t h = 0;
#pragma unroll
          for (uint32_t step = 128; step >= 1; step >>= 1)
            if (h + step <= uint32_t(B::HALO) && off[h + step] <= r) h += step;  // the last cell with off[h] <= r
          v[u] = colIn[gstart[h] + (r - off[h])];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint32_t r = base + u * THREADS;
        if (r < total) tile[r] = v[u];
      }
    }
    __syncthreads();
    // ---- one lane per home cell folds its 9 x-runs out of LDS, in the reference's order --------------
    if (!home) continue;
    const uint32_t first = table[code];
    if (table[code + 1u] == first) continue;  // the TRUE count: the last cell's walkers count too
    const uint32_t hx = (tid & 1u) | ((tid >> 2) & 2u), hy = ((tid >> 1) & 1u) | ((tid >> 3) & 2u),
                   hz = ((tid >> 2) & 1u) | ((tid >> 4) & 2u);
    N mx = N(0), my = N(0), mz = N(0), mw = N(0);
    uint32_t nn = 0;
#pragma unroll 1
    for (uint32_t dz = 0; dz < 3; ++dz)
#pragma unroll 1
      for (uint32_t dy = 0; dy < 3; ++dy) {
        const uint32_t l0 = ((hz + dz) * 6u + (hy + dy)) * 6u + hx;
        const uint32_t s = off[l0], e = off[l0 + 3];
#pragma unroll 8
        for (uint32_t j = s; j < e; ++j) {
          const vec4<N> cb = tile[j];
          mx += cb.x, my += cb.y, mz += cb.z, mw += cb.w;
        }
        nn += e - s;
      }
    cellSum[first] = make_vec4<N>(mx, my, mz, mw);
    cellCnt[first] = nn;
  }
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_diffuse_apply(StepConsts<N> c, typename DiffuseOp<N>::Args args,
                                                         const uint32_t *__restrict__ key,
                                                         const uint32_t *__restrict__ table,
                                                         const vec4<N> *__restrict__ cellSum,
                                                         const uint32_t *__restrict__ cellCnt) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= c.n) return;
  const uint32_t k = key[i];
  if (k >= c.tableN) {  // in no cell (sph.hpp:206): nobody shares this walk
    gather_one_global<N, DiffuseOp<N>>(c, args, key, table, i);
    return;
  }
  DiffuseOp<N> op;
  if (!op.begin(c, args, i)) return;
  const uint32_t first = table[k];
  const vec4<N> m = cellSum[first];
  op.mx = m.x, op.my = m.y, op.mz = m.z, op.mw = m.w;
  op.nn = int(cellCnt[first]);
  op.end(c, args, i);
}

// ------------------------------------------------------------------------------------------------
// Gather kernel "lists" (option gather = 1, the default) — one lane per particle, candidates from
// global memory (L1/L2), two phases:
//   A  every lane walks its 27 cell ranges like the global kernel but only applies the conservative
//      maybe_within_h filter (3 sub + 3 fma + cmp); survivors' global indices go to a per-lane list
//      in LDS ([slot][thread], 4-byte);
//   B  whenever an active lane's list is full — and once at the end — the lanes drain their lists
//      through the exact pair terms (IEEE sqrt / divide), in visiting order.  Rejected candidates
//      contribute exactly +0, so results are bit-identical to the plain walk.
// No tiles, no bricks: occupancy is set by the list alone (LMAX x 1 KiB per 256 threads), and sparse
// splash regions cost the same per particle as the dense column.
// ------------------------------------------------------------------------------------------------
// Neighbour lists kept in HBM between the lambda and the delta launch of ONE solver iteration: both
// see the same pStar, hence the same filtered candidates in the same order, so delta can skip the
// 27-cell walk and the filter altogether.  Layout [block][slot][thread]: the reader's loads are
// coalesced 1 KiB rows (measured: [particle][slot] rows cost +50 % in the reader and more in the
// writer; whole padded rows overflow — lanes fill at different times); at most NBR_CAP slots per
// particle, a particle with more survivors is marked NBR_OVERFLOW and walks.
// Two tiers (round 3; the settled dam-break's lists: mean 31, p99 41-44, 1.3 % (1 M) / 4 % (4 M) of the particles above 40
// — profiles/r03_list_hist.json): the [block][slot][thread] rows hold the first NBR_ROWS = 40 slots of every particle, a
// particle with more takes ONE chunk of NBR_EXTRA = 120 further slots from a pool (an atomic ticket per such particle and
// launch; pool = capacity / 8 chunks), 160 slots in all.  224 instead of 260 bytes of list per particle (-14 %);
// a particle beyond 160, or one that finds the pool empty, is marked NBR_OVERFLOW and walks its cells in the readers.
// (The chunk was 24 slots — 64 in all — until the trace showed the SECOND delta-p launch of every step taking 100 us against the
// others' 57: after the first correction 40 of a million particles sit in transient clumps of 65+ neighbours, each walked
// its ~200 candidates through the exact pair terms alone, and its wave kept the launch open.  At 4 M particles the deeper
// column packs up to 160 — hence 120 slots, of which a list touches only what it fills.  tools/iter_probe.py.)
constexpr uint32_t NBR_ROWS = 40, NBR_EXTRA = 120, NBR_CAP = NBR_ROWS + NBR_EXTRA;
constexpr uint32_t NBR_OVERFLOW = 0xFFFFFFFFu;
constexpr uint32_t NBR_NO_CHUNK = 0xFFFFFFFFu;
static_assert(NBR_ROWS % 4 == 0 && NBR_EXTRA % 4 == 0 && NBR_CAP < 256, "readers take 4 entries per trip; the length lives in 8 bits");
struct NbrLists {
  uint32_t *rows;    // [block][NBR_ROWS][BLOCK], and behind them, in the SAME allocation (so that a lane addresses both
                     // tiers with one base and a 32-bit offset), the pool: [chunk][NBR_EXTRA] from word `extraAt` on
  uint32_t *count;   // per particle: length | chunk << 8 (chunk only meaningful when length > NBR_ROWS), or NBR_OVERFLOW
  uint32_t *ticket;  // this launch's chunk allocator: a word that is zero when the launch starts
  uint64_t extraAt;  // first word of the pool
  uint32_t chunks;   // chunks in the pool
};
// One particle's list as its lane writes it, slot by slot in walk order.  The hot loops of the list builds store a staged
// batch at a time: reserve(first, count) BEFORE the batch takes the chunk when the batch reaches beyond the rows (one
// test per batch, an atomic for the few lanes that need one), put() is then a branch-free predicated store.
struct NbrWriter {
  uint32_t *row;          // this lane's column of its block's rows
  int32_t extraDelta = 0;  // word offset from `row` to this lane's chunk (valid once chunk != NBR_NO_CHUNK and in the pool)
  uint32_t chunk = NBR_NO_CHUNK;
  bool pooled = false;
  __device__ NbrWriter(const NbrLists &l, uint32_t block, uint32_t tid) : row(l.rows + size_t(block) * NBR_ROWS * BLOCK + tid) {}
  __device__ void reserve(const NbrLists &l, uint32_t first, uint32_t count) {
    if (first + count > NBR_ROWS && count != 0u && chunk == NBR_NO_CHUNK) {
      chunk = atomicAdd(l.ticket, 1u);
      pooled = chunk < l.chunks;
      extraDelta = int32_t(int64_t(l.extraAt + uint64_t(chunk) * NBR_EXTRA) - int64_t(row - l.rows)) - int32_t(NBR_ROWS);
    }
  }
  __device__ void put(uint32_t slot, uint32_t b, bool valid) const {
    const bool inRows = slot < NBR_ROWS;
    const int32_t at = inRows ? int32_t(slot * BLOCK) : extraDelta + int32_t(slot);
    if (valid && (inRows || (pooled && slot < NBR_CAP))) row[at] = b;
  }
  __device__ uint32_t finish(uint32_t written) const {
    if (written <= NBR_ROWS) return written;
    return (written <= NBR_CAP && pooled) ? (written | (chunk << 8)) : NBR_OVERFLOW;
  }
  // A staged batch [first, first + count) stored with as little per-entry arithmetic as the layout allows: everything that
  // depends on the lane alone is worked out once per batch (byte offsets from the block's first row word, how many of the
  // batch's entries land in the rows, how many are stored at all); entry k (wave-uniform) is then one compare, one add and a
  // store while the whole wave's batch stays inside the rows (the common case: one uniform test per batch), two compares,
  // two adds and a select otherwise.  put() — slot by slot, 9 VALU per entry with its 64-bit address — was 8 % of the list
  // build's instructions.
  struct Batch {
    uint32_t rowByte0, kRows, kStore;
    uint64_t chunkByte0;  // (the pool may lie more than 4 GB behind the block's rows: beyond ~19 M particles)
  };
  __device__ Batch begin_batch(const NbrLists &l, uint32_t tid, uint32_t first, uint32_t count) {
    reserve(l, first, count);
    Batch b;
    b.rowByte0 = (tid + first * BLOCK) * 4u;
    b.kRows = first < NBR_ROWS ? NBR_ROWS - first : 0u;
    const uint32_t capLeft = pooled ? (first < NBR_CAP ? NBR_CAP - first : 0u) : b.kRows;
    b.kStore = min(count, capLeft);
    b.chunkByte0 = uint64_t(int64_t(tid) + int64_t(extraDelta) + int64_t(first)) * 4u;  // (row[extraDelta + slot] seen from the block's first word)
    return b;
  }
  // blockRows = row - tid (wave-uniform): the stores take a scalar base and a 32-bit lane offset
  __device__ static void put_rows(uint32_t *blockRows, const Batch &b, uint32_t k, uint32_t value) {
    if (k < b.kStore) *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(blockRows) + (b.rowByte0 + k * (BLOCK * 4u))) = value;
  }
  __device__ static void put_any(uint32_t *blockRows, const Batch &b, uint32_t k, uint32_t value) {
    if (k < b.kStore) {
      const uint64_t off = k < b.kRows ? uint64_t(b.rowByte0 + k * (BLOCK * 4u)) : b.chunkByte0 + k * 4u;
      *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(blockRows) + off) = value;
    }
  }
};
// ... and as a reader sees it
struct NbrReader {
  const uint32_t *row, *extra;
  uint32_t cnt;  // NBR_OVERFLOW: walk
  __device__ NbrReader(const NbrLists &l, uint32_t i) {
    const uint32_t raw = l.count[i];
    row = l.rows + size_t(i / BLOCK) * NBR_ROWS * BLOCK + (i % BLOCK);
    cnt = raw == NBR_OVERFLOW ? raw : (raw & 0xFFu);
    extra = l.rows + l.extraAt + size_t(raw == NBR_OVERFLOW || (raw & 0xFFu) <= NBR_ROWS ? 0u : raw >> 8) * NBR_EXTRA;
  }
  // any slot (the pipelined / cooperative readers); slots past the list are clamped into it and their entries discarded
  __device__ uint32_t entry(uint32_t slot) const {
    slot = min(slot, NBR_CAP - 1u);
    return slot < NBR_ROWS ? row[slot * BLOCK] : extra[slot - NBR_ROWS];
  }
};

// A second, unfiltered op can ride along on the same walk (FUSE_DIFFUSE: the colour diffusion needs
// exactly the candidates the first lambda launch of a step visits, in the same order).
struct NoExtra {
  struct Args {};
};
template <typename N, typename Op, int LMAX, bool SAVE = false, typename Extra = NoExtra>
__global__ __launch_bounds__(BLOCK) void k_gather_lists(StepConsts<N> c, typename Op::Args args,
                                                        const uint32_t *__restrict__ key,
                                                        const uint32_t *__restrict__ table,
                                                        NbrLists lists,
                                                        typename Extra::Args xargs = {}) {
  constexpr bool FUSED = !std::is_same<Extra, NoExtra>::value;
  __shared__ uint32_t list[(Op::kFilter ? LMAX + 4 : 1) * BLOCK];  // +4: a trip appends up to WAYS past LMAX - 1
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = xcd_chunk();
  const uint32_t i = chunk * BLOCK + tid;
  if (i >= c.n) return;
  if constexpr (!Op::kFilter) {  // diffuse has no distance test: nothing to filter, plain (2-way) walk
    gather_one_global<N, Op>(c, args, key, table, i);
    return;
  }
  Op op;
  Extra extra;
  if constexpr (FUSED) extra.begin(c, xargs, i);  // skips exactly when op.begin() does (type != 0) and copies through
  if (!op.begin(c, args, i)) {
    if (SAVE) lists.count[i] = 0;
    return;
  }
  // From here on control flow is WAVE-UNIFORM over the lanes that are left (loop conditions are
  // __any votes): every drain is executed by all of them together.
  uint32_t nl = 0, written = 0;
  NbrWriter wr(lists, chunk, tid);
  auto drain = [&]() {
    if (SAVE) wr.reserve(lists, written, nl);
#pragma unroll 2
    for (uint32_t q = 0; __any(q < nl); ++q) {
      const bool valid = q < nl;
      const uint32_t b = valid ? list[q * BLOCK + tid] : i;
      if (SAVE) wr.put(written + q, b, valid);
      op.add_bf(c, Op::load(args, b), valid);
    }
    written += nl;
    nl = 0;
  };
  const Neigh nb = neigh_codes(key[i]);
#pragma unroll 1
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll 1
    for (int dy = 0; dy < 3; ++dy) {
      const uint32_t yz = nb.ys[dy] | nb.zs[dz];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        // (merging the x-adjacent pair 2m, 2m+1 into one range was measured: slower — odd and even x
        // lanes of a wave then walk ranges of different length and fall out of lockstep)
        const uint32_t code = nb.xs[dx] | yz;
        uint32_t start = 0, len = 0;
        if (code < c.tableN) {  // sph.hpp:206-208
          start = table[code];
          len = ((code + 1u) < c.tableN ? table[code + 1u] : start) - start;
        }
        // phase A: filter only, WAYS candidates per trip (their loads are in flight together); lanes of
        // one cell stay in lockstep, so their candidate loads coalesce
        constexpr uint32_t WAYS = 4;
        for (uint32_t t = 0; __any(t < len); t += WAYS) {
          if (t < len) {
            uint32_t b[WAYS];
            typename Op::Src cnd[WAYS];
#pragma unroll
            for (uint32_t w = 0; w < WAYS; ++w) {
              b[w] = start + min(t + w, len - 1u);  // a tail slot re-reads the last candidate and is masked
              cnd[w] = Op::load(args, b[w]);
            }
#pragma unroll
            for (uint32_t w = 0; w < WAYS; ++w) {
              const bool hit = (t + w < len) && op.near(c, cnd[w]);
              list[nl * BLOCK + tid] = b[w];  // branch-free append: the slot is kept only on a hit
              nl += hit ? 1u : 0u;
            }
            if constexpr (FUSED) {
#pragma unroll
              for (uint32_t w = 0; w < WAYS; ++w)
                if (t + w < len && !(c.hasObstacles && (xargs.type[b[w]] & 1))) extra.add(c, Extra::load(xargs, b[w]));
            }
          }
          if (__any(nl >= uint32_t(LMAX))) drain();  // phase B: exact pair terms for the survivors, in order
        }
      }
    }
  drain();
  if (SAVE) lists.count[i] = wr.finish(written);
  op.end(c, args, i);
  if constexpr (FUSED) extra.end(c, xargs, i);
}

// Neighbour-list build on its own (no pair terms), on 8-byte quantised positions -------------------------------
// The build is bound by the texture-address path (64 lanes x 16 B per candidate load), so the test
// runs on a compact copy of pStar: per axis the LOW 16 bits of floor((p - gridMin) * 2048 / h).
// Differences are taken modulo 2^16 (v_pk_sub_i16), i.e. exact whenever the true separation is below
// 16 h, and squared-summed with two v_dot2 — 5 VALU per candidate, half the bytes.  The result is only
// ever a SUPERSET of the particles within h (rounding is covered by the threshold, wrap-around and
// overflow can only add far candidates); the list-driven ops apply the exact tests, so the physics
// stays bit-identical.  A walker whose own coordinates are unusable (outside +-2^22 units) takes all.
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_quantise(StepConsts<N> c, const vec4<N> *__restrict__ pstar,
                                                    uint2 *__restrict__ qpos) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= c.n) return;
  bool usable;
  qpos[i] = quantise_position<N>(c, pstar[i], &usable);
}

// Two quantised candidates per load: the L1 works per cache line touched, not per byte, so the run is
// walked in pairs (b, b + 1) fetched as one 16-byte load — a quarter of the load instructions of
// the fp32 walk.  (qpos carries one spare entry: the pair of a run's last candidate may read past it.)
struct __attribute__((aligned(8))) QPair {
  uint32_t ax, ay, bx, by;  // two consecutive qpos entries
};

struct __attribute__((aligned(4))) TablePair {
  uint32_t t0, t1;  // table[code], table[code + 1]
};
struct __attribute__((aligned(4))) TableTriple {
  uint32_t t0, t1, t2;  // table[code .. code + 2]
};
// |d|^2 of two packed int16 pairs: v_dot2_i32_i16 with an inline-zero / register addend (the builtin
// lowers to the accumulate-in-place form and spends a v_mov on the zero)
__device__ inline int qdot2(uint32_t d) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(r) : "v"(d));
  return r;
}
__device__ inline int qdot2(uint32_t d, int acc) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %1, %2" : "=v"(r) : "v"(d), "v"(acc));
  return r;
}
constexpr uint32_t QPOS_PAD = 64;  // spare qpos entries: tail pairs of a run read (and mask) what follows it

template <typename N, int W, int LMAX = 16>
__global__ __launch_bounds__(BLOCK) void k_build_lists_q(StepConsts<N> c, const vec4<N> *__restrict__ pstar,
                                                         const uint2 *__restrict__ qpos,
                                                         const uint8_t *__restrict__ type,
                                                         const uint32_t *__restrict__ key,
                                                         const uint32_t *__restrict__ table,
                                                         NbrLists lists) {
  static_assert(4 * W + 2 <= QPOS_PAD, "qpos padding");
  __shared__ uint32_t list[(LMAX + 2 * W) * BLOCK];  // per-lane staging: a trip appends up to 2 W past LMAX - 1
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = xcd_chunk();
  const uint32_t i = chunk * BLOCK + tid;
  if (i >= c.n) return;
  if (c.hasObstacles && type[i] != 0) {
    lists.count[i] = 0;
    return;
  }
  // 32-bit byte offsets from uniform bases: one shift per address
  const char *qbase = reinterpret_cast<const char *>(qpos), *tbase = reinterpret_cast<const char *>(table);
  bool usable;
  const uint2 qa = quantise_position<N>(c, pstar[i], &usable);
  const qpair axy = __builtin_bit_cast(qpair, qa.x), azw = __builtin_bit_cast(qpair, qa.y);
  const uint32_t t2 = usable ? QPOS_T * QPOS_T : 0xFFFFFFFFu;
  auto within = [&](uint32_t qx, uint32_t qy) {
    const qpair dxy = __builtin_bit_cast(qpair, qx) - axy, dzw = __builtin_bit_cast(qpair, qy) - azw;
    return uint32_t(qdot2(__builtin_bit_cast(uint32_t, dzw), qdot2(__builtin_bit_cast(uint32_t, dxy)))) <= t2;
  };
  NbrWriter wr(lists, chunk, tid);
  uint32_t written = 0, nl = 0;
  auto flush = [&]() {
    wr.reserve(lists, written, nl);
    for (uint32_t q = 0; __any(q < nl); ++q) wr.put(written + q, list[q * BLOCK + tid], q < nl);
    written += nl;
    nl = 0;
  };
  // A (dy, dz) row of three x cells is TWO runs of the sorted array: the cells (2m, 2m + 1) have
  // adjacent codes, so for an odd own x the row is [x-1, x] + [x+1], for an even one [x-1] + [x, x+1].
  // Two loads per row: table[pair .. pair + 2] and table[single .. single + 1] (the table keeps entries
  // up to tableN + 1).  A cell outside the table, and the table's last cell, are empty (sph.hpp:206-208).
  const uint32_t k0 = key[i];
  const Neigh nb = neigh_codes(k0);
  const bool odd = (k0 & 1u) != 0u;
  const uint32_t xPair = odd ? nb.xs[0] : nb.xs[1], xSingle = odd ? nb.xs[2] : nb.xs[0];
  struct Row {
    uint32_t sA, lA, sB, lB;
  };
  auto load_row = [&](int r) {
    const uint32_t yz = nb.ys[r % 3] | nb.zs[r / 3];
    const uint32_t cP = xPair | yz, cS = xSingle | yz;
    const TableTriple tp = *reinterpret_cast<const TableTriple *>(tbase + min(cP, c.tableN) * 4u);
    const TablePair ts = *reinterpret_cast<const TablePair *>(tbase + min(cS, c.tableN) * 4u);
    const uint32_t lP0 = (cP + 1u) < c.tableN ? tp.t1 - tp.t0 : 0u, lP1 = (cP + 2u) < c.tableN ? tp.t2 - tp.t1 : 0u;
    const uint32_t sP = lP0 ? tp.t0 : tp.t1, lP = lP0 + lP1;
    const uint32_t lS = (cS + 1u) < c.tableN ? ts.t1 - ts.t0 : 0u;
    Row row;
    row.sA = odd ? sP : ts.t0, row.lA = odd ? lP : lS;
    row.sB = odd ? ts.t0 : sP, row.lB = odd ? lS : lP;
    return row;
  };
  Row next = load_row(0);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const Row row = next;
    if (r < 8) next = load_row(r + 1);
    // One slot sequence for the row, walked in PAIRS (slot 2j, 2j + 1 -> one 16-byte load): run A is
    // padded to an even length so that no pair straddles the two runs; the pad slot is masked.
    const uint32_t lA = row.lA, lAe = (lA + 1u) & ~1u, L = lAe + row.lB, oB = row.sB - lAe;
    for (uint32_t t = 0; __any(t < L); t += 2 * W) {
      if (t < L) {
        uint32_t b[W], lim[W];
        QPair cnd[W];
#pragma unroll
        for (uint32_t w = 0; w < W; ++w) {
          const uint32_t sl = t + 2 * w;  // slots past L read what follows run B (QPOS_PAD) and are masked
          const bool inA = sl < lAe;
          b[w] = sl + (inA ? row.sA : oB);
          lim[w] = inA ? lA : L;
          cnd[w] = *reinterpret_cast<const QPair *>(qbase + b[w] * 8u);
        }
#pragma unroll
        for (uint32_t w = 0; w < W; ++w) {
          const bool hit0 = (t + 2 * w < lim[w]) & within(cnd[w].ax, cnd[w].ay);
          list[nl * BLOCK + tid] = b[w];  // branch-free append: the slot is kept only on a hit
          nl += hit0 ? 1u : 0u;
          const bool hit1 = (t + 2 * w + 1 < lim[w]) & within(cnd[w].bx, cnd[w].by);
          list[nl * BLOCK + tid] = b[w] + 1u;
          nl += hit1 ? 1u : 0u;
        }
      }
      if (__any(nl >= uint32_t(LMAX))) flush();
    }
  }
  flush();
  lists.count[i] = wr.finish(written);
}

// The same build with an op riding on it (option split_build = 8): the survivors staged in LDS are not only flushed to
// their row but folded through the op's exact pair terms on the way — lambda needs no launch and no list read of its own,
// and its arithmetic runs in the issue slots the build (bound by the texture-address path, 4 waves per SIMD by its LDS)
// leaves idle.  Same candidates in the same order as the list-driven reader: the same bits.
template <typename N, typename Op, int W, int LMAX = 32, int FW = 4>
__global__ __launch_bounds__(BLOCK) void k_build_lists_op(StepConsts<N> c, typename Op::Args args, const vec4<N> *__restrict__ pstar,
                                                         const uint2 *__restrict__ qpos,
                                                         const uint8_t *__restrict__ type,
                                                         const uint32_t *__restrict__ key,
                                                         const uint32_t *__restrict__ table,
                                                         NbrLists lists) {
  static_assert(4 * W + 2 <= QPOS_PAD, "qpos padding");
  __shared__ uint32_t list[(LMAX + 2 * W) * BLOCK];  // per-lane staging: a trip appends up to 2 W past LMAX - 1
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = xcd_chunk();
  const uint32_t i = chunk * BLOCK + tid;
  if (i >= c.n) return;
  Op op;
  if (!op.begin(c, args, i)) {  // (obstacles, ghosts: the op has stored what it owes them)
    lists.count[i] = 0;
    return;
  }
  // 32-bit byte offsets from uniform bases: one shift per address
  const char *qbase = reinterpret_cast<const char *>(qpos), *tbase = reinterpret_cast<const char *>(table);
  bool usable;
  const uint2 qa = quantise_position<N>(c, pstar[i], &usable);
  const qpair axy = __builtin_bit_cast(qpair, qa.x), azw = __builtin_bit_cast(qpair, qa.y);
  const uint32_t t2 = usable ? QPOS_T * QPOS_T : 0xFFFFFFFFu;
  auto within = [&](uint32_t qx, uint32_t qy) {
    const qpair dxy = __builtin_bit_cast(qpair, qx) - axy, dzw = __builtin_bit_cast(qpair, qy) - azw;
    return uint32_t(qdot2(__builtin_bit_cast(uint32_t, dzw), qdot2(__builtin_bit_cast(uint32_t, dxy)))) <= t2;
  };
  NbrWriter wr(lists, chunk, tid);
  uint32_t written = 0, nl = 0;
  auto flush = [&]() {
    wr.reserve(lists, written, nl);
    for (uint32_t q = 0; __any(q < nl); q += FW) {  // FW survivors per trip: their gathers and pair terms interleave
      uint32_t b[FW];
      typename Op::Src cnd[FW];
#pragma unroll
      for (uint32_t w = 0; w < FW; ++w) b[w] = q + w < nl ? list[(q + w) * BLOCK + tid] : i;
#pragma unroll
      for (uint32_t w = 0; w < FW; ++w) cnd[w] = Op::load(args, b[w]);
#pragma unroll
      for (uint32_t w = 0; w < FW; ++w)
        wr.put(written + q + w, b[w], q + w < nl);
#pragma unroll
      for (uint32_t w = 0; w < FW; ++w) op.add_bf(c, cnd[w], q + w < nl);
    }
    written += nl;
    nl = 0;
  };
  // A (dy, dz) row of three x cells is TWO runs of the sorted array: the cells (2m, 2m + 1) have
  // adjacent codes, so for an odd own x the row is [x-1, x] + [x+1], for an even one [x-1] + [x, x+1].
  // Two loads per row: table[pair .. pair + 2] and table[single .. single + 1] (the table keeps entries
  // up to tableN + 1).  A cell outside the table, and the table's last cell, are empty (sph.hpp:206-208).
  const uint32_t k0 = key[i];
  const Neigh nb = neigh_codes(k0);
  const bool odd = (k0 & 1u) != 0u;
  const uint32_t xPair = odd ? nb.xs[0] : nb.xs[1], xSingle = odd ? nb.xs[2] : nb.xs[0];
  struct Row {
    uint32_t sA, lA, sB, lB;
  };
  auto load_row = [&](int r) {
    const uint32_t yz = nb.ys[r % 3] | nb.zs[r / 3];
    const uint32_t cP = xPair | yz, cS = xSingle | yz;
    const TableTriple tp = *reinterpret_cast<const TableTriple *>(tbase + min(cP, c.tableN) * 4u);
    const TablePair ts = *reinterpret_cast<const TablePair *>(tbase + min(cS, c.tableN) * 4u);
    const uint32_t lP0 = (cP + 1u) < c.tableN ? tp.t1 - tp.t0 : 0u, lP1 = (cP + 2u) < c.tableN ? tp.t2 - tp.t1 : 0u;
    const uint32_t sP = lP0 ? tp.t0 : tp.t1, lP = lP0 + lP1;
    const uint32_t lS = (cS + 1u) < c.tableN ? ts.t1 - ts.t0 : 0u;
    Row row;
    row.sA = odd ? sP : ts.t0, row.lA = odd ? lP : lS;
    row.sB = odd ? ts.t0 : sP, row.lB = odd ? lS : lP;
    return row;
  };
  Row next = load_row(0);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const Row row = next;
    if (r < 8) next = load_row(r + 1);
    // One slot sequence for the row, walked in PAIRS (slot 2j, 2j + 1 -> one 16-byte load): run A is
    // padded to an even length so that no pair straddles the two runs; the pad slot is masked.
    const uint32_t lA = row.lA, lAe = (lA + 1u) & ~1u, L = lAe + row.lB, oB = row.sB - lAe;
    for (uint32_t t = 0; __any(t < L); t += 2 * W) {
      if (t < L) {
        uint32_t b[W], lim[W];
        QPair cnd[W];
#pragma unroll
        for (uint32_t w = 0; w < W; ++w) {
          const uint32_t sl = t + 2 * w;  // slots past L read what follows run B (QPOS_PAD) and are masked
          const bool inA = sl < lAe;
          b[w] = sl + (inA ? row.sA : oB);
          lim[w] = inA ? lA : L;
          cnd[w] = *reinterpret_cast<const QPair *>(qbase + b[w] * 8u);
        }
#pragma unroll
        for (uint32_t w = 0; w < W; ++w) {
          const bool hit0 = (t + 2 * w < lim[w]) & within(cnd[w].ax, cnd[w].ay);
          list[nl * BLOCK + tid] = b[w];  // branch-free append: the slot is kept only on a hit
          nl += hit0 ? 1u : 0u;
          const bool hit1 = (t + 2 * w + 1 < lim[w]) & within(cnd[w].bx, cnd[w].by);
          list[nl * BLOCK + tid] = b[w] + 1u;
          nl += hit1 ? 1u : 0u;
        }
      }
      if (__any(nl >= uint32_t(LMAX))) flush();
    }
  }
  flush();
  lists.count[i] = wr.finish(written);
  op.end(c, args, i);
}

// ---- the row-major iteration kernels (RowArrays above) ---------------------------------------------------------------
struct RowWalk {
  const uint32_t *lintab, *xyz, *slotOf, *mtable;
  uint32_t tableN, pshift;
};
// the run of row r (0..8: dy = r % 3 - 1, dz = r / 3 - 1) of the walker in cell (x, y, z): the cells x - 1 .. x + 1 of that
// row, clamped at the cube's faces; a row outside the cube is empty
struct RowRun {
  uint32_t s, e;
};
__device__ inline RowRun row_run(const RowWalk &w, uint32_t x, uint32_t y, uint32_t z, uint32_t r) {
  const uint32_t P = 1u << w.pshift, yy = y + r % 3u - 1u, zz = z + r / 3u - 1u;
  RowRun run{0u, 0u};
  if (yy < P && zz < P) {  // (unsigned: -1 wraps far beyond P)
    const uint32_t base = ((zz << w.pshift) | yy) << w.pshift;
    run.s = w.lintab[base + (x ? x - 1u : 0u)];
    run.e = w.lintab[base + min(x + 2u, P)];  // (base + P = the next row's first cell = this row's end)
  }
  return run;
}
// every candidate of row-slot i in the reference's order, as a row slot: the serial form (walkers that take the Morton
// table, rows that overflowed)
template <typename F> __device__ inline void row_for_each_candidate(const RowWalk &w, uint32_t i, F &&f) {
  const uint32_t c = w.xyz[i];
  const uint32_t x = c & 1023u, y = (c >> 10) & 1023u, z = (c >> 20) & 1023u;
  if (c & ROW_FALLBACK) {
    for_each_candidate(morton_encode(x, y, z), w.mtable, w.tableN, [&](uint32_t b) { f(w.slotOf[b]); });
    return;
  }
#pragma unroll 1
  for (uint32_t r = 0; r < 9; ++r) {
    const RowRun run = row_run(w, x, y, z, r);
    for (uint32_t b = run.s; b < run.e; ++b) f(b);
  }
}

// ---- diffusion on the row-major copy ------------------------------------------------------------------------------------
// Diffuse has no distance test, so every particle of a cell folds the very same candidates in the very same order (see
// k_diffuse_bricks): one lane per CELL does the walk.  In the row-major copy a wave takes a SEGMENT of up to 64 consecutive
// x cells of one (y, z) row: for each of the nine (dy, dz) rows, the candidates of all its cells are ONE contiguous run of
// the colour copy — cells x0 - 1 .. x0 + 64 — which the wave copies into LDS with fully coalesced loads (a few records per
// lane) and each lane then folds its own three-cell sub-run out of it, in the reference's order.  The wave finally applies the
// sums to the segment's own particles (again a contiguous run) and writes the new colours to their Morton-sorted places:
// no per-cell sums in memory, no second pass, no halo gather.  k_row_segments lists the segments that hold a particle
// (persistent workgroups stride over that list).  Cells whose walkers take the Morton table (P = 1024 faces) and the
// particles in no cell walk serially, as they do in the iteration kernels.
constexpr int DIFFUSE_ROW_THREADS = 64;
// LDS-DMA (global_load_lds_dwordx4): every active lane's 16 bytes at `src` (per lane) land at `ldsBase` (wave-uniform) + 16 x
// lane — no VGPR destination; completion is counted on vmcnt
__device__ inline void lds_dma16(const void *src, void *ldsBase) {
  using G = const __attribute__((address_space(1))) void *;
  using L = __attribute__((address_space(3))) void *;
  __builtin_amdgcn_global_load_lds((G)(uintptr_t)src, (L)(uint32_t)(uintptr_t)ldsBase, 16, 0, 0);
}
__global__ __launch_bounds__(BLOCK) void k_row_segments(uint32_t pshift, uint32_t segShift, const uint32_t *__restrict__ lintab,
                                                        uint32_t *__restrict__ segs, uint32_t *__restrict__ nSegs) {
  const uint32_t sg = blockIdx.x * BLOCK + threadIdx.x;
  if (sg >= (1u << (3u * pshift - segShift))) return;
  const uint32_t c0 = sg << segShift;
  if (lintab[c0 + (1u << segShift)] != lintab[c0]) segs[atomicAdd(nSegs, 1u)] = sg;  // (any order: segments are independent)
}

template <typename N>
__global__ __launch_bounds__(DIFFUSE_ROW_THREADS) void k_diffuse_rows(StepConsts<N> c, RowWalk rw, const vec4<N> *__restrict__ rowCol,
                                                                      const uint8_t *__restrict__ rowType,
                                                                      const uint32_t *__restrict__ mortonOf,
                                                                      const uint32_t *__restrict__ segs,
                                                                      const uint32_t *__restrict__ nSegsPtr, uint32_t segShift,
                                                                      vec4<N> *__restrict__ colOut, uint32_t cap, uint32_t nSlots) {
  constexpr uint32_t T = DIFFUSE_ROW_THREADS;
  constexpr uint32_t UNITS = sizeof(vec4<N>) / 16u;  // 16-byte pieces per record (what one lane of an LDS-DMA moves)
  extern __shared__ __align__(16) unsigned char smem[];
  vec4<N> *tile = reinterpret_cast<vec4<N> *>(smem);                     // [cap] one row's run of colours
  N *sums = reinterpret_cast<N *>(smem + size_t(cap + 8u) * sizeof(vec4<N>));  // [4][T] the cells' sums for the apply pass (behind the tile and its 8 records of read-ahead padding)
  uint32_t *cnts = reinterpret_cast<uint32_t *>(sums + 4 * T);            // [T]
  uint8_t *flag = reinterpret_cast<uint8_t *>(cnts + T);                  // [cap] candidate types (only with obstacles)
  const uint32_t lane = threadIdx.x, P = 1u << rw.pshift, X = 1u << segShift;
  const uint32_t *__restrict__ lintab = rw.lintab;
  const uint32_t nSegs = *nSegsPtr;
  typename DiffuseOp<N>::Args out{nullptr, colOut, nullptr};
  // typed: some particle is not plain fluid (an obstacle, or — slab mode — a ghost copy, possibly OF an obstacle): walkers
  // look at their own type, and the candidates' types are staged beside their colours so that obstacles can be skipped
  const bool typed = c.hasObstacles != 0u, sift = typed;
  // workgroup -> segment: each XCD (workgroups are dealt to the 8 XCDs round robin) takes a contiguous eighth of the list,
  // so the rows a segment shares with its y / z neighbours are found in that XCD's L2
  const uint32_t xcd = blockIdx.x & 7u, perXcd = gridDim.x >> 3, segsPerXcd = (nSegs + 7u) >> 3;
  const uint32_t tEnd = min(nSegs, (xcd + 1u) * segsPerXcd);
  for (uint32_t t = xcd * segsPerXcd + (blockIdx.x >> 3); t < tEnd; t += perXcd) {
    const uint32_t c0 = segs[t] << segShift;
    const uint32_t x0 = c0 & (P - 1u), y = (c0 >> rw.pshift) & (P - 1u), z = c0 >> (2u * rw.pshift);
    const uint32_t x = x0 + lane;
    uint32_t ownS = 0, ownE = 0;
    if (lane < X) ownS = lintab[c0 + lane], ownE = lintab[c0 + lane + 1u];
    const bool live = ownE > ownS;
    // (a per-cell property, P = 1024 only: the walkers of the cube's face cells take the Morton table)
    const bool serial = rw.pshift == 10u && live && (rw.xyz[ownS] & ROW_FALLBACK) != 0u;
    N mx = N(0), my = N(0), mz = N(0), mw = N(0);
    uint32_t nn = 0;
    auto add = [&](const vec4<N> &cb) { mx += cb.x, my += cb.y, mz += cb.z, mw += cb.w, ++nn; };
    // the segment's own particles [hs, he), lane l takes hs + l, hs + l + 64, ...: what the apply pass at the end needs of the
    // first OWN x 64 of them is requested now and arrives while the planes are folded
#ifndef PBF_DIFF_OWN
#define PBF_DIFF_OWN 8
#endif
    constexpr uint32_t OWN = PBF_DIFF_OWN;
    const uint32_t hs = __shfl(ownS, 0), he = __shfl(ownE, int(X) - 1);
    vec4<N> ownCol[OWN];
    uint32_t ownDst[OWN], ownXyz[OWN];
    uint8_t ownTy[OWN];
#pragma unroll
    for (uint32_t k = 0; k < OWN; ++k) {
      const uint32_t j = min(hs + lane + k * T, nSlots - 1u);
      ownCol[k] = rowCol[j], ownDst[k] = mortonOf[j], ownXyz[k] = rw.xyz[j];
      ownTy[k] = typed ? rowType[j] : uint8_t(0);
    }
    if (__any(serial)) {
      if (serial)
        for_each_candidate(morton_encode(x, y, z), rw.mtable, rw.tableN, [&](uint32_t b) {
          const uint32_t sl = rw.slotOf[b];
          if (!(sift && (rowType[sl] & 1))) add(rowCol[sl]);
        });
    }
    // every table entry of the nine rows first (one round trip): the segment's run [rs, rs + len) and this lane's sub-run
    uint32_t rs[9], len[9], ls[9], le[9];
#pragma unroll
    for (uint32_t r = 0; r < 9; ++r) {
      const uint32_t yy = y + r % 3u - 1u, zz = z + r / 3u - 1u;
      rs[r] = len[r] = ls[r] = le[r] = 0u;
      if (yy < P && zz < P) {  // (uniform; unsigned: -1 wraps far beyond P) a row outside the cube is empty
        const uint32_t base = ((zz << rw.pshift) | yy) << rw.pshift;
        rs[r] = lintab[base + (x0 ? x0 - 1u : 0u)];
        len[r] = lintab[base + min(x0 + X + 1u, P)] - rs[r];
        if (live && !serial) ls[r] = lintab[base + (x ? x - 1u : 0u)], le[r] = lintab[base + min(x + 2u, P)];  // = row_run()
      }
    }
    // one row per phase: its run goes to LDS by LDS-DMA — every piece in flight at once, no registers — and is folded once
    // it has landed.  (A small tile: a dozen single-wave workgroups per CU hide each other's round trips; three rows per
    // phase in a 32-KiB tile — 4 workgroups per CU, one wave per SIMD — measured 4 x slower: the walk is bound by the issue
    // rate of ONE wave.)
#pragma unroll
    for (uint32_t r = 0; r < 9; ++r) {
      if (len[r] == 0u) continue;  // (uniform)
      if (len[r] <= cap) {
        __syncthreads();  // the previous row's folds are done with the tile
        const unsigned char *src = reinterpret_cast<const unsigned char *>(rowCol + rs[r]);
        unsigned char *dst = reinterpret_cast<unsigned char *>(tile);
        const uint32_t units = len[r] * UNITS;
        for (uint32_t u0 = 0; u0 < units; u0 += T)  // 1 KiB per wave-instruction: lane l's 16 bytes land at dst + 16 l
          if (u0 + lane < units) lds_dma16(src + size_t(u0 + lane) * 16u, dst + size_t(u0) * 16u);
        if (sift) {  // the candidates' types: every byte of the run requested before the first is stored
          constexpr uint32_t FB = 8;
          for (uint32_t k0 = lane; k0 < len[r]; k0 += FB * T) {
            uint8_t f[FB];
#pragma unroll
            for (uint32_t u = 0; u < FB; ++u) f[u] = rowType[rs[r] + min(k0 + u * T, len[r] - 1u)];
#pragma unroll
            for (uint32_t u = 0; u < FB; ++u)
              if (k0 + u * T < len[r]) flag[k0 + u * T] = f[u];
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every piece has landed
        __syncthreads();
        const uint32_t s = ls[r] - rs[r], e = le[r] - rs[r];
        if (sift) {
          // four records and their types per trip (reads past the sub-run stay inside the padding and are never added)
          const uint32_t cnt = e - s;
          for (uint32_t done = 0; done < cnt; done += 4u) {
            vec4<N> v4[4];
            uint8_t f4[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) v4[k] = tile[s + done + k], f4[k] = flag[s + done + k];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
              if (done + k < cnt && !(f4[k] & 1)) add(v4[k]);  // obstacles are skipped as candidates (ompsph.hpp:194)
          }
        } else {
          // four records per trip, the next four requested before these are added (reads past the sub-run stay inside the
          // tile's padding and are never added): one LDS latency per row instead of one per trip, no remainder loop
          const uint32_t cnt = e - s;  // (a lane without a sub-run: 0 — its s is meaningless)
          if (cnt) {
            vec4<N> cur[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) cur[k] = tile[s + k];
            uint32_t done = 0;
            for (; done + 4u <= cnt; done += 4u) {
              vec4<N> nxt[4];
#pragma unroll
              for (uint32_t k = 0; k < 4; ++k) nxt[k] = tile[s + done + 4u + k];
#pragma unroll
              for (uint32_t k = 0; k < 4; ++k) mx += cur[k].x, my += cur[k].y, mz += cur[k].z, mw += cur[k].w;
#pragma unroll
              for (uint32_t k = 0; k < 4; ++k) cur[k] = nxt[k];
            }
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k)
              if (done + k < cnt) mx += cur[k].x, my += cur[k].y, mz += cur[k].z, mw += cur[k].w;
          }
          nn += e - s;
        }
      } else {  // a run longer than the tile (a pile-up): straight from memory
        for (uint32_t j = ls[r]; j < le[r]; ++j)
          if (!(sift && (rowType[j] & 1))) add(rowCol[j]);
      }
    }
    __syncthreads();
    {  // what every particle of the cell would compute from the sums (DiffuseOp::end: y = (m / n) * 1.33), once per cell
      const N fn = N(nn ? nn : 1u);
      sums[lane] = (mx / fn) * N(1.33), sums[T + lane] = (my / fn) * N(1.33), sums[2 * T + lane] = (mz / fn) * N(1.33),
      sums[3 * T + lane] = (mw / fn) * N(1.33), cnts[lane] = nn;
    }
    __syncthreads();
    auto apply = [&](const vec4<N> &ca, uint32_t dst, uint32_t xyz, uint8_t ty) {
      const uint32_t cx = (xyz & 1023u) - x0;
      if ((typed && ty != 0) || cnts[cx] == 0u) {  // obstacle, or a ghost copy owned by the neighbouring slab; nobody around: unchanged
        colOut[dst] = ca;
        return;
      }
      const N t = c.diffuseT;
      auto one = [&](N x, N y) {  // (= DiffuseOp::end)
        const N o = x * (N(1) - t) + y * t;
        return min(max(o, N(0.03)), N(1.0));
      };
      colOut[dst] = make_vec4<N>(one(ca.x, sums[cx]), one(ca.y, sums[T + cx]), one(ca.z, sums[2 * T + cx]), one(ca.w, sums[3 * T + cx]));
    };
#pragma unroll
    for (uint32_t k = 0; k < OWN; ++k)
      if (hs + lane + k * T < he) apply(ownCol[k], ownDst[k], ownXyz[k], ownTy[k]);
    for (uint32_t j = hs + lane + OWN * T; j < he; j += T)  // (a segment of more than OWN x 64 particles)
      apply(rowCol[j], mortonOf[j], rw.xyz[j], typed ? rowType[j] : uint8_t(0));
  }
  // the particles in no cell (behind the last cell): one lane each, the serial walk
  for (uint32_t j = lintab[1u << (3u * rw.pshift)] + blockIdx.x * T + lane; j < nSlots; j += gridDim.x * T) {
    DiffuseOp<N> op;
    op.ca = rowCol[j];
    op.mx = op.my = op.mz = op.mw = N(0), op.nn = 0;
    const uint32_t dst = mortonOf[j];
    if (typed && rowType[j] != 0) {
      colOut[dst] = op.ca;
      continue;
    }
    row_for_each_candidate(rw, j, [&](uint32_t b) {
      if (!(sift && (rowType[b] & 1))) op.add(c, rowCol[b]);
    });
    op.end(c, out, dst);
  }
}

// The list build with an op riding on it, on the row-major copy: lane = row slot.  A (dy, dz) row of three x cells is ONE
// run [lintab[c - 1], lintab[c + 2]) — walked in pairs (one 16-byte load = two quantised candidates) from a per-lane byte
// offset that advances by a constant per trip: no run select, no padding slot, one limit compare per slot.
template <typename N, typename Op, int W, int LMAX = 32, int FW = 4>
__global__ __launch_bounds__(BLOCK) void k_build_rows_op(StepConsts<N> c, typename Op::Args args, const uint2 *__restrict__ qpos,
                                                         RowWalk rw, NbrLists lists) {
  static_assert(4 * W + 2 <= QPOS_PAD, "qpos padding");
  __shared__ uint32_t list[(LMAX + 2 * W) * BLOCK];  // per-lane staging: a trip appends up to 2 W past LMAX - 1
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = xcd_chunk();
  const uint32_t i = chunk * BLOCK + tid;
  if (i >= c.n) return;
  Op op;
  if (!op.begin(c, args, i)) {  // (obstacles, ghosts: the op has stored what it owes them)
    lists.count[i] = 0;
    return;
  }
  const char *qbase = reinterpret_cast<const char *>(qpos);
  const uint2 qa = qpos[i];
  // (the walker's own coordinates may be unusable — beyond +-2^22 units: it then takes everything; recomputed like the Morton build)
  bool usable;
  (void)quantise_position<N>(c, Op::load(args, i), &usable);
  const qpair axy = __builtin_bit_cast(qpair, qa.x), azw = __builtin_bit_cast(qpair, qa.y);
  const uint32_t t2 = usable ? QPOS_T * QPOS_T : 0xFFFFFFFFu;
  auto within = [&](uint32_t qx, uint32_t qy) {
    const qpair dxy = __builtin_bit_cast(qpair, qx) - axy, dzw = __builtin_bit_cast(qpair, qy) - azw;
    return uint32_t(qdot2(__builtin_bit_cast(uint32_t, dzw), qdot2(__builtin_bit_cast(uint32_t, dxy)))) <= t2;
  };
  NbrWriter wr(lists, chunk, tid);
  uint32_t *const blockRows = lists.rows + size_t(chunk) * NBR_ROWS * BLOCK;  // (wave-uniform; wr.row = blockRows + tid)
  // the staging list's write cursor as an LDS ADDRESS (slot s of this lane at lbase + s * SLOT): appending is one store through
  // the cursor and "cursor += hit ? SLOT : 0" — a select and a full-rate add, no address arithmetic per candidate.  (The
  // cursor is a 32-bit LDS address kept opaque to the compiler, which otherwise rewrites it as base + offset and adds the
  // base again before every store.)
  constexpr uint32_t SLOT = BLOCK * 4u;
  typedef __attribute__((address_space(3))) uint32_t *LdsWord;
  unsigned char *const lbasePtr = reinterpret_cast<unsigned char *>(list) + tid * 4u;
  const uint32_t lbase = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)lbasePtr));
  const uint32_t lfull = lbase + uint32_t(LMAX) * SLOT;  // the cursor of a lane whose LMAX slots are taken
  auto staged = [&](uint32_t byteOff) -> uint32_t & { return *reinterpret_cast<uint32_t *>(lbasePtr + byteOff); };
  auto append = [&](uint32_t &cursor, uint32_t value, bool keep) {
    *(LdsWord)(uintptr_t)cursor = value;
    cursor += keep ? SLOT : 0u;
    asm volatile("" : "+v"(cursor));
  };
  uint32_t written = 0;
  uint32_t cur = lbase;
  auto flush = [&]() {
    const uint32_t nl = uint32_t(cur - lbase) / SLOT;
    const NbrWriter::Batch bt = wr.begin_batch(lists, tid, written, nl);
    const bool rowsOnly = !__any(written + nl > NBR_ROWS);  // (wave-uniform: nobody's batch reaches beyond the rows)
    for (uint32_t q = 0; __any(q < nl); q += FW) {  // FW survivors per trip: their gathers and pair terms interleave
      uint32_t b[FW];
      typename Op::Src cnd[FW];
#pragma unroll
      for (uint32_t w = 0; w < FW; ++w) b[w] = q + w < nl ? staged((q + w) * SLOT) : i;
#pragma unroll
      for (uint32_t w = 0; w < FW; ++w) cnd[w] = Op::load(args, b[w]);
      if (rowsOnly) {
#pragma unroll
        for (uint32_t w = 0; w < FW; ++w) NbrWriter::put_rows(blockRows, bt, q + w, b[w]);
      } else {
#pragma unroll
        for (uint32_t w = 0; w < FW; ++w) NbrWriter::put_any(blockRows, bt, q + w, b[w]);
      }
#pragma unroll
      for (uint32_t w = 0; w < FW; ++w) op.add_bf(c, cnd[w], q + w < nl);
    }
    written += nl;
    cur = lbase;
  };
  const uint32_t cell = rw.xyz[i];
  const bool slow = (cell & ROW_FALLBACK) != 0u;
  if (__any(slow)) {
    // the few walkers in no cell: the serial walk over the Morton table, every candidate tested
    if (slow)
      row_for_each_candidate(rw, i, [&](uint32_t b) {
        const uint2 q = qpos[b];
        if (within(q.x, q.y)) {
          if (cur >= lfull) {  // (lane-private drain: the others of the wave are not here)
            const uint32_t nl = uint32_t(cur - lbase) / SLOT;
            wr.reserve(lists, written, nl);
            for (uint32_t k = 0; k < nl; ++k) {
              const uint32_t e = staged(k * SLOT);
              wr.put(written + k, e, true);
              op.add_bf(c, Op::load(args, e), true);
            }
            written += nl, cur = lbase;
          }
          append(cur, b, true);
        }
      });
  }
  const uint32_t x = cell & 1023u, y = (cell >> 10) & 1023u, z = (cell >> 20) & 1023u;
  auto load_run = [&](uint32_t r) { return slow ? RowRun{0u, 0u} : row_run(rw, x, y, z, r); };
  RowRun next = load_run(0);
#pragma unroll
  for (uint32_t r = 0; r < 9; ++r) {
    const RowRun run = next;
    if (r < 8) next = load_run(r + 1);
    const uint32_t L = run.e - run.s;
    uint32_t off = run.s * 8u, b0 = run.s;  // byte offset of / row slot at the trip's first candidate
    // (two nested loops: the inner one — trips until a lane's staging list is full or the row is done — touches no state of
    // the drain, so its loop-carried registers are the walk's own four; with the drain inside the trip loop the compiler
    // copied the op's accumulators and the writer's state at the head of every trip: 8 v_mov per trip)
    for (uint32_t t = 0; __any(t < L);) {
      do {
        if (t < L) {
          QPair cnd[W];
#pragma unroll
          for (uint32_t w = 0; w < W; ++w) cnd[w] = *reinterpret_cast<const QPair *>(qbase + off + 16u * w);  // (slots past L: padding / the next run, masked)
#pragma unroll
          for (uint32_t w = 0; w < W; ++w) {
            const bool hit0 = (t + 2 * w < L) & within(cnd[w].ax, cnd[w].ay);
            append(cur, b0 + 2 * w, hit0);  // branch-free: the slot is kept only on a hit
            const bool hit1 = (t + 2 * w + 1 < L) & within(cnd[w].bx, cnd[w].by);
            append(cur, b0 + 2 * w + 1u, hit1);
          }
        }
        off += 16u * W, b0 += 2 * W, t += 2 * W;
      } while (__any(t < L) && !__any(cur >= lfull));
      if (__any(cur >= lfull)) flush();
    }
  }
  flush();
  lists.count[i] = wr.finish(written);
  op.end(c, args, i);
}

// Pins a just-loaded candidate into registers at this point of the program.  Without it LLVM folds the loop-carried
// phi(load in the prologue, load in the loop) back into ONE load at the loop head (InstCombine's phi-of-loads), which
// silently un-pipelines k_gather_from_lists: the pair terms would again wait for gathers issued in the same trip.
__device__ inline void pin_registers(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ inline void pin_registers(double4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <typename N> __device__ inline void pin_registers(PosVel<N> &b) {
  pin_registers(b.p);
  pin_registers(b.v);
}

// List-driven gather: the survivors recorded by the previous list build on the same pStar, visited in the recorded
// (= reference) order; NBR_OVERFLOW particles walk their 27 cells.
// PIPELINED (option "pipeline"; default: fp64 only — measured at 1 M particles it gains 1.3 % in fp64 and LOSES 3 % in
// fp32, whose readers are bound by VALU issue, not by memory latency): W entries per trip; while trip t's pair terms execute,
// trip t+1's candidate gathers and trip t+2's list entries are already in flight, so a wave hides its own two dependent
// latencies (list entry -> candidate) instead of leaning on the other waves of its SIMD.  Slots past the row (the
// prefetch runs up to 3 W - 1 ahead) are clamped to the row's last slot and their entries discarded, never used as
// an address.  Same candidates in the same order: bit-identical to the serial form.
template <typename N, typename Op, bool PIPELINED = true, bool ROWS = false>
__global__ __launch_bounds__(BLOCK) void k_gather_from_lists(StepConsts<N> c, typename Op::Args args,
                                                             const uint32_t *__restrict__ key,
                                                             const uint32_t *__restrict__ table,
                                                             NbrLists lists, RowWalk rw = {}) {
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = xcd_chunk();
  const uint32_t i = chunk * BLOCK + tid;
  if (i >= c.n) return;
  Op op;
  if (!op.begin(c, args, i)) return;
  const NbrReader rd(lists, i);
  const uint32_t cnt = rd.cnt;
  if (cnt == NBR_OVERFLOW) {
    // the walk, alone in its wave: only the candidates that may be within h go through the exact pair terms (the others
    // contribute exactly +0, see maybe_within_h) — 8 VALU per candidate instead of 55
    auto visit = [&](uint32_t b) {
      const typename Op::Src pb = Op::load(args, b);
      if constexpr (Op::kFilter) {
        if (op.near(c, pb)) op.add(c, pb);
      } else {
        op.add(c, pb);
      }
    };
    if constexpr (ROWS) {
      const uint32_t cell = rw.xyz[i];
      if (cell & ROW_FALLBACK) {
        row_for_each_candidate(rw, i, visit);
      } else {
        // nine contiguous runs, eight candidates requested per trip: the walker is alone in its wave, so nobody hides its
        // round trips for it (one candidate at a time, the 40 to 190 walkers of a step held their launch open for 45 us)
        const uint32_t x = cell & 1023u, y = (cell >> 10) & 1023u, z = (cell >> 20) & 1023u;
#pragma unroll 1
        for (uint32_t r = 0; r < 9; ++r) {
          const RowRun run = row_run(rw, x, y, z, r);
          for (uint32_t b = run.s; b < run.e; b += 8u) {
            typename Op::Src pb[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) pb[k] = Op::load(args, min(b + k, run.e - 1u));
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
              if (b + k < run.e) {
                if constexpr (Op::kFilter) {
                  if (op.near(c, pb[k])) op.add(c, pb[k]);
                } else {
                  op.add(c, pb[k]);
                }
              }
            }
          }
        }
      }
    } else {
      for_each_candidate(key[i], table, c.tableN, visit);
    }
  } else if constexpr (PIPELINED) {
    constexpr uint32_t W = sizeof(N) == 4 ? 4 : 2;
    auto entry = [&](uint32_t slot) { return rd.entry(slot); };
    uint32_t e1[W];           // raw list entries of trip t + 1
    typename Op::Src cur[W];  // candidates of trip t
#pragma unroll
    for (uint32_t w = 0; w < W; ++w) e1[w] = entry(w);
#pragma unroll
    for (uint32_t w = 0; w < W; ++w) cur[w] = Op::load(args, w < cnt ? e1[w] : i);
#pragma unroll
    for (uint32_t w = 0; w < W; ++w) e1[w] = entry(W + w);
    for (uint32_t q = 0; q < cnt; q += W) {
      typename Op::Src nxt[W];
#pragma unroll
      for (uint32_t w = 0; w < W; ++w) nxt[w] = Op::load(args, q + W + w < cnt ? e1[w] : i);
#pragma unroll
      for (uint32_t w = 0; w < W; ++w) e1[w] = entry(q + 2 * W + w);
      __builtin_amdgcn_sched_barrier(0);  // the loads above are ISSUED before the pair terms below (the scheduler
                                          // otherwise sinks them to their first use to save registers)
#pragma unroll
      for (uint32_t w = 0; w < W; ++w) op.add_bf(c, cur[w], q + w < cnt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (uint32_t w = 0; w < W; ++w) {
        pin_registers(nxt[w]);  // (waits for trip t + 1's gathers here, a whole trip of pair terms after their issue)
        cur[w] = nxt[w];
      }
    }
  } else {
    // the rows' part of the list, then — for the few particles that have one — the chunk's (same order: slots 0 .. cnt - 1)
    const uint32_t head = min(cnt, NBR_ROWS), tail = cnt - head;
    // RW list entries and their candidates in flight per trip.  fp32: 8 (measured at 1 M: 2 -> 73 us per launch, 4 -> 61,
    // 8 -> 58.5, 12 -> 66: more padded slots for the longest lane past 8); fp64 keeps 4 (its candidates are 32 bytes)
#ifndef PBF_READER_W
#define PBF_READER_W 8
#endif
    constexpr uint32_t RW = sizeof(N) == 4 ? PBF_READER_W : 4;
    for (uint32_t q = 0; q < head; q += RW) {
      uint32_t b[RW];
      typename Op::Src cnd[RW];
#pragma unroll
      for (uint32_t w = 0; w < RW; ++w) b[w] = q + w < head ? rd.row[(q + w) * BLOCK] : i;
#pragma unroll
      for (uint32_t w = 0; w < RW; ++w) cnd[w] = Op::load(args, b[w]);
#pragma unroll
      for (uint32_t w = 0; w < RW; ++w) op.add_bf(c, cnd[w], q + w < head);
    }
    for (uint32_t q = 0; q < tail; q += 4) {
      uint32_t b[4];
      typename Op::Src cnd[4];
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) b[w] = q + w < tail ? rd.extra[q + w] : i;
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) cnd[w] = Op::load(args, b[w]);
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) op.add_bf(c, cnd[w], q + w < tail);
    }
  }
  op.end(c, args, i);
}

// Wave-cooperative list-driven gather (option "coop", opt-in): COOP lanes share one particle — lane j of the group
// takes the list entries j, j + COOP, j + 2 COOP, ... — and the per-particle kernel sums are reduced across the group
// with wave shuffles (__shfl_xor) before one lane writes the result.  Lists of very different length then cost a wave
// max(ceil(len / COOP)) trips over 64 / COOP particles instead of max(len) over 64, and every lane's loop is COOP
// times shorter.  The summation ORDER differs from the reference walk (COOP interleaved partial sums, then a tree),
// so results agree with the oracle to rounding, not bit for bit: tests hold it to the stated tolerance
// (<= 1e-3 world units per step).  Same [block][slot][thread] lists as k_gather_from_lists.
template <typename N, typename Op, int COOP>
__global__ __launch_bounds__(BLOCK) void k_gather_from_lists_coop(StepConsts<N> c, typename Op::Args args,
                                                                  const uint32_t *__restrict__ key,
                                                                  const uint32_t *__restrict__ table,
                                                                  NbrLists lists) {
  static_assert(COOP == 2 || COOP == 4 || COOP == 8, "group size");
  constexpr uint32_t PER_BLOCK = BLOCK / COOP;  // particles per workgroup
  const uint32_t sub = threadIdx.x % COOP;
  const uint32_t i = xcd_chunk() * PER_BLOCK + threadIdx.x / COOP;
  // (no early return: every lane of a group takes part in the shuffles; a group past the end only idles)
  const bool live = i < c.n;
  Op op;
  bool active = false;
  if (live) active = op.begin(c, args, i);  // every lane of the group loads the particle (same addresses: one request)
  if (active) {
    const NbrReader rd(lists, i);
    const uint32_t cnt = rd.cnt;
    if (cnt == NBR_OVERFLOW) {
      if (sub == 0) for_each_candidate(key[i], table, c.tableN, [&](uint32_t b) { op.add(c, Op::load(args, b)); });
    } else {
      for (uint32_t q = sub; q < cnt; q += 2 * COOP) {  // two entries of this lane's share in flight per trip
        const uint32_t q1 = q + COOP;
        const uint32_t b0 = rd.entry(q), b1 = q1 < cnt ? rd.entry(q1) : i;
        const typename Op::Src c0 = Op::load(args, b0), c1 = Op::load(args, b1);
        op.add_bf(c, c0, true);
        op.add_bf(c, c1, q1 < cnt);
      }
    }
  }
  // butterfly over the group: after log2(COOP) steps every lane holds the group's sums
  op.combine([&](N v) {
#pragma unroll
    for (int m = 1; m < COOP; m <<= 1) v += __shfl_xor(active ? v : N(0), m, 64);
    return v;
  });
  if (active && sub == 0) op.end(c, args, i);
}

// ------------------------------------------------------------------------------------------------
// finalise (ompsph.hpp:256-264): pure stream, 48 B in / 32 B out per particle (fp32)
// ------------------------------------------------------------------------------------------------
template <typename N> __device__ inline void finalise_one(const StepConsts<N> &c, const vec4<N> &ps, vec4<N> &p, vec4<N> &v) {
  const N dxx = ps.x - p.x / c.scale, dyy = ps.y - p.y / c.scale, dzz = ps.z - p.z / c.scale;
  const N invdt = N(1) / c.dt;
  p.x = ps.x * c.scale, p.y = ps.y * c.scale, p.z = ps.z * c.scale;
  v.x = (dxx * invdt + v.x) * N(VD), v.y = (dyy * invdt + v.y) * N(VD), v.z = (dzz * invdt + v.z) * N(VD);
}

// (rowSlotOf != NULL: pstar is the iterations' row-major copy, particle a's record sits at rowSlotOf[a])
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_finalise(StepConsts<N> c, const uint8_t *__restrict__ type,
                                                    const vec4<N> *__restrict__ pstar, vec4<N> *__restrict__ pos4,
                                                    vec4<N> *__restrict__ vel4, const uint32_t *__restrict__ rowSlotOf) {
  const uint32_t a = blockIdx.x * BLOCK + threadIdx.x;
  if (a >= c.n) return;
  if (c.hasObstacles && type[a] != 0) return;  // obstacles and ghosts do not move here
  vec4<N> p = pos4[a], v = vel4[a];
  finalise_one<N>(c, pstar[rowSlotOf ? rowSlotOf[a] : a], p, v);
  pos4[a] = p;
  vel4[a] = v;
}

// finalise(t) and predict(t + 1) of one particle in one pass (pbf_steps, same parameters for both steps, nothing between
// them): the finalised position and velocity stay in registers, pStar is overwritten in place by the next prediction —
// one launch and 48 bytes per particle of re-reads less.  `cn` carries step t + 1's constants (the same, but for the
// frame-dependent ones a caller may have changed — the host only fuses when they are equal).
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_finalise_predict(StepConsts<N> c, const uint8_t *__restrict__ type,
                                                            vec4<N> *__restrict__ pstar, vec4<N> *__restrict__ pos4,
                                                            vec4<N> *__restrict__ vel4, const N *__restrict__ wells,
                                                            uint32_t *__restrict__ key, uint32_t *__restrict__ count,
                                                            const vec4<N> *__restrict__ rowPstar,
                                                            const uint32_t *__restrict__ rowSlotOf) {
  const uint32_t a = blockIdx.x * BLOCK + threadIdx.x;
  if (a >= c.n) return;
  const uint8_t ty = c.hasObstacles ? type[a] : uint8_t(0);
  if (c.slabOn && (ty & TYPE_GHOST)) {  // (see k_predict)
    key[a] = DEAD_KEY;
    return;
  }
  vec4<N> p = pos4[a], v = vel4[a];
  if (ty == 0) {
    finalise_one<N>(c, rowSlotOf ? rowPstar[rowSlotOf[a]] : pstar[a], p, v);
    pos4[a] = p;
  }
  uint32_t k;
  pstar[a] = predict_one<N>(c, p, v, ty, wells, k);
  if (!(ty & 1)) vel4[a] = v;
  key[a] = k;
  if (!slab_stays(c, k)) return;
  uint32_t before, rank;
  wave_bucket_atomic(min(k, c.tableN), [&](uint32_t b, uint32_t cnt) { return atomicAdd(&count[b], cnt); }, before, rank);
}

// the iterations' row-major {pStar, lambda} back into the Morton-sorted array (stage-level read-backs, the opt-in extras)
template <typename N>
__global__ __launch_bounds__(BLOCK) void k_rows_to_morton(uint32_t n, const vec4<N> *__restrict__ rowPstar,
                                                          const uint32_t *__restrict__ rowSlotOf, vec4<N> *__restrict__ pstar) {
  const uint32_t a = blockIdx.x * BLOCK + threadIdx.x;
  if (a < n) pstar[a] = rowPstar[rowSlotOf[a]];
}

// ------------------------------------------------------------------------------------------------
// AoS <-> SoA on the device for the advance() shim (std::vector<Particle>, sph.hpp:36-54)
// ------------------------------------------------------------------------------------------------
struct AosLayout {
  uint32_t stride, off_id, off_type, off_mass, off_pos, off_vel, off_colour;
};

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_unpack_aos(uint32_t n, const uint8_t *__restrict__ aos, AosLayout l,
                                                      ParticleArrays<N> dst, uint32_t *__restrict__ anyObstacle) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint8_t *p = aos + size_t(i) * l.stride;
  if (p[l.off_type] == 1) *anyObstacle = 1u;  // (sph::Type::Obstacle; found here instead of by a strided host scan)
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

// Exhaustive check of the trimmed sqrt / divides against hipcc's IEEE forms (pbf_selftest_math):
//   bad[0]  sqrt_rsq(x) vs sqrtf(x) over every fp32 x >= 2^-75;
//   bad[1]  div_seeded((h - r)^2, r, y) vs the IEEE quotient over every fp32 d2 with r = sqrt(d2) in [1e-8, h], y the
//           by-product of sqrt_rsq(d2) — exactly the operands pair_geom hands it — for four h;
//   bad[2]  div_ranged(x, poly6(0.3 h)) over every fp32 x with 1e-30 <= |x| <= 1e30 or x == 0 (NaNs compare by class);
//   bad[3]  DeltaOp's x / RHO — div_ranged where div_ranged_ok(x), the compiler's divide otherwise — over EVERY fp32 x.
__global__ __launch_bounds__(BLOCK) void k_selftest_math(unsigned long long *__restrict__ bad, float divisorA,
                                                         float divisorB, float hOwn) {
  unsigned long long badSqrt = 0, badDiv = 0, badA = 0, badB = 0;
  const float hs[4] = {hOwn, 0.05f, 0.2f, 0.0999999f};  // the context's own h (0.1 in every shipped configuration) + three more
  for (uint64_t v = uint64_t(blockIdx.x) * BLOCK + threadIdx.x; v < (1ull << 32); v += uint64_t(gridDim.x) * BLOCK) {
    const float x = __int_as_float(int(uint32_t(v)));
    if (x >= 0x1p-75f && x <= 3.0e38f) {
      float y;
      const float r = sqrt_rsq(x, y), ref = sqrtf(x);
      badSqrt += __float_as_int(r) != __float_as_int(ref);
      if (ref >= 1e-8f && ref <= fmaxf(0.2f, hOwn)) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ref <= hs[k]) {
            const float hr = hs[k] - ref, num = hr * hr;
            badDiv += __float_as_int(div_seeded(num, ref, y)) != __float_as_int(num / ref);
          }
      }
    }
    {
      const float qa = div_ranged(x, divisorA), ra = x / divisorA;
      const bool mid = x == 0.f || (fabsf(x) >= 1e-30f && fabsf(x) <= 1e30f);
      badA += mid && ((qa != qa) ? !(ra != ra) : __float_as_int(qa) != __float_as_int(ra));
      const float rb = x / divisorB, qb = div_ranged_ok(x) ? div_ranged(x, divisorB) : rb;
      badB += (qb != qb) ? !(rb != rb) : __float_as_int(qb) != __float_as_int(rb);
    }
  }
  if (badSqrt) atomicAdd(&bad[0], badSqrt);
  if (badDiv) atomicAdd(&bad[1], badDiv);
  if (badA) atomicAdd(&bad[2], badA);
  if (badB) atomicAdd(&bad[3], badB);
}

// The fp64 counterpart: an exhaustive sweep is impossible (2^64 operands), so `rounds` x gridDim x BLOCK pseudo-random
// operands per category (splitmix64 of a global counter: reproducible), drawn so that every binade of the stated range is hit
// equally often and the pair terms' own operand ranges densely:
//   bad[0]  sqrt_rsq(x) vs sqrt(x): x = 2^e m, e uniform in [-540, 500] (half of the draws) or d2 of a pair: r uniform in
//           (0, 2 h) (the other half);
//   bad[1]  div_seeded((h - r)^2, r, y) vs the IEEE quotient, r = sqrt(d2) in [1e-8, h] (r uniform, and log-uniform), y the
//           by-product of sqrt_rsq(d2) — exactly the operands pair_geom hands it — for the context's h and three more;
//   bad[2]  div_ranged(x, poly6(0.3 h)) vs x / poly6(0.3 h): |x| log-uniform in [1e-60, 1e30], both signs;
//   bad[3]  DeltaOp's x / RHO — div_ranged where div_ranged_ok(x), the compiler's divide otherwise — |x| = 2^e m with e
//           uniform over [-1070, 1000] (denormals included), both signs.
__device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline double unit53(uint64_t u) { return double(u >> 11) * 0x1p-53; }                 // [0, 1)
__device__ inline double pow2_times_mantissa(uint64_t u, int elo, int ehi) {                       // 2^e m, m in [1, 2)
  const int e = elo + int((u >> 52) % uint64_t(ehi - elo + 1));
  return ldexp(1.0 + double(u & 0xFFFFFFFFFFFFFull) * 0x1p-52, e);
}
__global__ __launch_bounds__(BLOCK) void k_selftest_math64(unsigned long long *__restrict__ bad, double divisorA,
                                                           double divisorB, double hOwn, uint32_t rounds) {
  unsigned long long badSqrt = 0, badDiv = 0, badA = 0, badB = 0;
  const double hs[4] = {hOwn, 0.05, 0.2, 0.0999999};
  const uint64_t tid = uint64_t(blockIdx.x) * BLOCK + threadIdx.x, stride = uint64_t(gridDim.x) * BLOCK;
  for (uint32_t k = 0; k < rounds; ++k) {
    const uint64_t u0 = splitmix64((tid + k * stride) * 4u), u1 = splitmix64(u0), u2 = splitmix64(u1), u3 = splitmix64(u2);
    {
      double x;
      if (u0 & 1u) x = pow2_times_mantissa(u1, -540, 500);
      else {
        const double r = unit53(u1) * 2.0 * hOwn;
        x = r * r;
      }
      double y;
      badSqrt += __double_as_longlong(sqrt_rsq(x, y)) != __double_as_longlong(sqrt(x));
    }
    {
      const double hk = hs[(u0 >> 1) & 3u];
      double r = (u0 & 8u) ? unit53(u2) * hk : exp2(-26.5 + unit53(u2) * 26.5) * hk;  // uniform / log-uniform in (~1e-8 h, h)
      const double d2 = r * r;
      double y;
      const double root = sqrt_rsq(d2, y);
      if (root >= 1e-8 && root <= hk) {
        const double hr = hk - root, num = hr * hr;
        badDiv += __double_as_longlong(div_seeded(num, root, y)) != __double_as_longlong(num / root);
      }
    }
    {
      double x = exp2(-199.3 + unit53(u3) * 299.0);  // 1e-60 .. 1e30
      if (u3 & 1u) x = -x;
      const double qa = div_ranged(x, divisorA), ra = x / divisorA;
      badA += __double_as_longlong(qa) != __double_as_longlong(ra);
      double z = pow2_times_mantissa(splitmix64(u3), -1070, 1000);
      if (u3 & 2u) z = -z;
      const double rb = z / divisorB, qb = div_ranged_ok(z) ? div_ranged(z, divisorB) : rb;
      badB += __double_as_longlong(qb) != __double_as_longlong(rb);
    }
  }
  if (badSqrt) atomicAdd(&bad[0], badSqrt);
  if (badDiv) atomicAdd(&bad[1], badDiv);
  if (badA) atomicAdd(&bad[2], badA);
  if (badB) atomicAdd(&bad[3], badB);
}

}  // namespace pbf
