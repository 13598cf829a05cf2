// pbf_hip.hip — context, launch sequence and C ABI (include/pbf_hip.h) of the gfx950 PBF-SPH step.
//
// The product path: there is NO CPU fallback anywhere in this file.  If no HIP device is usable
// pbf_create fails with PBF_ERR_NO_DEVICE and every other entry point needs a ctx.
#include "pbf_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "pbf_kernels.hpp"
#include "pbf_slab.hpp"
#include "pbf_tiles.hpp"
#include "pbf_mc.hpp"
#include "pbf_comm.hpp"

using namespace pbf;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

constexpr uint32_t kTickets = 254;  // (kTickets + 2 words = 1024 bytes: ONE fill kernel per step; 1028 bytes took two)
constexpr int kBrickZ = 4;

enum Stage { ST_PREDICT = 0, ST_SORT, ST_DIFFUSE, ST_LAMBDA, ST_DELTA, ST_FINALISE, ST_BUILD, ST_COUNT };
// names follow the reference's Stopwatch entries (ompsph.hpp:130,157,161,188,209,252)
const char *kStageNames[ST_COUNT] = {"advect+zindex",      "sortz+gridtable", "sph-diffuse",
                                     "sph-lambda",         "sph-delta",       "sph-finalise",
                                     "sph-lambda/list-build"};  // (a sub-interval of sph-lambda: one kernel)

struct EventPair {
  hipEvent_t a, b;
  int stage;
};

}  // namespace

// hipGraph replay of the resident step (option "graph"): everything host-side that one step reads AND leaves behind —
// which physical buffer plays which role, the validity flags — so that a replayed graph can put the context into the
// state the captured step left it in.  Also the key under which a captured step is found again.
struct StepState {
  int cur, pcur;
  uint32_t flags;  // sorted | counted << 1 | nbrValid << 2 | qposValid << 3 | hasObstacles << 4
  uint32_t tableN, countedTableN, gatherSeq;
  uint64_t extent[3];
  double minExtent[3];
  void *bufs[15];  // pos4[2] vel4[2] col4[2] id[2] type[2] key[2] pstar[3]
  size_t caps[15];
  bool operator<(const StepState &o) const { return std::memcmp(this, &o, sizeof(StepState)) < 0; }
};
struct GraphEntry {
  hipGraphExec_t exec = nullptr;
  StepState after;
};
struct GraphKey {
  StepState before;
  unsigned char params[sizeof(double) * 11 + 32];  // dt, scale, force, bounds, iteration, xsph, vorticity
  uint64_t n;
  int options[10];
  bool operator<(const GraphKey &o) const { return std::memcmp(this, &o, sizeof(GraphKey)) < 0; }
};

struct pbf_ctx {
  pbf_desc desc{};
  int device = 0;
  bool fp64 = false;
  bool fast = false;
  hipStream_t stream = nullptr;
  bool ownStream = false;
  std::string err;

  size_t n = 0;
  size_t cap = 0;        // particle capacity of the SoA buffers
  bool hasObstacles = false;
  bool sorted = false;   // keys/table valid for the current arrays
  bool counted = false;  // cell histogram of the current keys is in `count` (set by predict, consumed by sort)
  int cur = 0;           // which of the two particle-array sets is live
  int pcur = 0;          // which pstar buffer is live
  // two sets (sort scatters from one into the other)
  DevBuf pos4[2], vel4[2], col4[2], id[2], type[2], key[2];
  DevBuf pstar[3];       // [0],[1]: sort ping-pong partner of set 0/1 ; [2]: Jacobi partner
  DevBuf count, table, blockSums, permTmp, wells, staging;
  // slab decomposition (pbf_slab_*): bookkeeping of what was sent / received this step
  DevBuf slotOf, selCounts, selTotals, ghostSrcL, ghostSrcR, colHist;
  bool slabActive = false, realObstacles = false;
  bool ghostsPending = false;  // pbf_slab_step leaves the copies in place (the next step's migration select drops them);
                               // anything that looks at the particle arrays from outside drops them first (drop_ghosts)
  bool slabConfigured = false;  // pbf_slab_configure: rank-local x frame for the keys (compact table per rank)
  pbf_slab_cut slabCut{0, 0, 0, 0};
  uint32_t xoff = 0;
  int32_t shiftL = 0, shiftR = 0;
  // pbf_slab_attach: the whole step incl. the exchanges runs inside the library (pbf_slab_step)
  pbf_comm *comm = nullptr;  // not owned
  std::vector<uint32_t> cuts;
  uint32_t capMig = 0, capGhost = 0;  // records in the FIRST message of an assembly round (the rest follows when needed)
  uint32_t wireCap = 0;               // records the wire buffers hold per neighbour
  DevBuf wireSend[2], wireRecv[2];
  DevBuf wireGhost[2];               // the ghost round's send buffers (packed by the same select pass as the migrants')
  bool slabStepMode = false;         // inside pbf_slab_step: predict classifies (StepConsts::slabOn)
  size_t sortLive = 0;               // pbf_slab_step: particles that take part in this step's sort (the others are dead slots)
  uint32_t ghostAt = 0;              // pre-sort index of the first copy received this step (k_unpack_field)
  uint32_t slabSeq = 0;              // sequence number of the next read-back (k_slab_counts -> pinned host word)
  uint64_t slabHostSyncs = 0;        // host read-backs inside pbf_slab_step so far (2 per step: the two assembly rounds)
  uint32_t *hostCounts = nullptr;  // pinned: read-back of the assembly rounds' counts
  size_t reserve = 0;        // pbf_reserve: capacity kept for migrants and ghost copies
  uint32_t nOwned = 0, sentL = 0, sentR = 0, gotL = 0, gotR = 0;
  // marching cubes (pbf_surface)
  DevBuf latticePN, latticeC, mcCounts, mcOffsets, mcSums, meshV, meshN, meshC, mcNear;
  void *meshHost = nullptr;  // pinned staging of the last mesh (pbf_map_mesh)
  size_t meshHostCap = 0;
  bool meshStaged = false;
  uint64_t mcSample[3] = {0, 0, 0};
  uint64_t mcTriangles = 0;
  DevBuf qpos;               // 8-byte quantised pStar for the list build (k_build_lists_q)
  DevBuf nbrList, nbrCount;  // neighbour lists handed from the lambda launch to the delta launch: NBR_ROWS slots per particle
  // option "row_major": the iterations' working set also laid out cell-row-major (csrc/pbf_kernels.hpp RowArrays)
  int rowMajor = 1;           // default ON since round 3: -3 % per step at 1 M, -10 % at 4 M (profiles/r03_matrix.txt)
  DevBuf rowPstar[2], rowMass, rowQpos, rowXYZ, rowType, rowSlotOf, linCount, linTable, linSums;
  DevBuf rowCol, rowMortonOf, rowSegs;  // k_diffuse_rows: the colours in row order, slot -> Morton index, non-empty segments
  uint32_t rowSegShift = 0;
  uint32_t diffuseCap = 0;   // option "diffuse_cap" (diagnostic): records in k_diffuse_rows' tile, 0 = default
  int rowDiffuse = 1;        // option "row_diffuse": the diffusion runs on the row-major copy (k_diffuse_rows)
  bool bricksValid = false;  // ctx->bricks lists the non-empty bricks of the current table
  bool rowColValid = false;  // rowCol holds the colours of col4[cur] (set by the sort, consumed by the diffusion)
  int rcur = 0;               // which rowPstar buffer is live
  bool rowsValid = false;     // the row arrays describe this step's sorted set (built by the sort)
  bool nbrRows = false;       // the current neighbour lists hold ROW slots (built by k_build_rows_op)
  bool pstarInRows = false;   // the current {pStar, lambda} live in rowPstar[rcur] ONLY: pstar[pcur] is stale (materialise_pstar)
  bool rowsCurrent = false;   // rowPstar[rcur] holds the current {pStar, lambda} (false once a Morton-path stage has moved on)
  uint32_t rowShift = 0;      // the row grid is the cube [0, 1 << rowShift)^3
  uint64_t nbrExtraAt = 0;   // ... + behind them a pool of NBR_EXTRA-slot chunks for the particles that need more (NbrLists)
  uint32_t nbrChunks = 0, nbrChunksOpt = 0;
  bool nbrValid = false;     // the lists describe pstar[pcur] as it is now
  bool omegaValid = false;   // pstar[2] holds the vorticity of the last extras pass (PBF_BUF_OMEGA), same order as the arrays
  // advance() path: the caller's std::vector<Particle> buffer, page-locked in place (hipHostRegister) so the per-frame
  // 56-byte-per-particle upload and download are plain DMA instead of the runtime's pageable staging
  void *regPtr = nullptr;
  size_t regBytes = 0;
  size_t stagedBytes = 0;  // bytes of `staging` that hold a defined AoS image (padding bytes of a download come from it)
  // option "graph" (default OFF): pbf_steps replays each distinct step as a captured hipGraph (one graph launch instead
  // of ~25 kernel launches); stage timing, wells, growing buffers or a box that moves every frame fall back to eager.
  // Measured on MI355X / ROCm 7 (tools/graph_ab.sh): replay is SLOWER than the eager launches — 0.294 vs 0.269 ms/step
  // at 16 K particles, 0.512 vs 0.479 at 256 K, 1.724 vs 1.709 at 1 M: the step is never host-launch bound (the host
  // enqueues a step in ~90 us) and a graph does not shorten the dependent-kernel boundary.
  int graphMode = 0;
  std::map<GraphKey, GraphEntry> graphs;
  uint64_t allocEpoch = 0;     // bumped by every hipMalloc of ensure(): a step that allocated is not captured
  uint32_t graphMisses = 0;    // captures without a replay in between: a scene that never repeats turns graphs off
  uint64_t graphReplays = 0, graphCaptures = 0;
  pbf_params lastParams{};   // the params of the last stage call (entry points without a params argument derive their consts from it)
  bool haveParams = false;
  bool qposValid = false;    // qpos is the quantised copy of pstar[pcur] (written by the sort, delta-p and the slab refresh)
  bool reuseLists = true;    // option "reuse_lists"
  // option "split_build": 0 = lambda builds the neighbour lists while it gathers on fp32 candidates (k_gather_lists<SAVE>);
  // 4 / 5 = the build is a launch of its own (k_build_lists_q on quantised pairs, 2 / 4 pair loads per trip) followed by a
  // list-driven lambda; 8 (default) = the quantised build with lambda riding on its flushes (k_build_lists_op: one launch,
  // no list read for lambda).  (Round 1's intermediate build kernels, values 1-3, are gone.)
  int splitBuild = 8;
  int pipeline = -1;         // option "pipeline": software-pipelined list readers (bit-identical either way); -1 = auto = off
                             // (measured at 1 M, round 3: fp32 +3 %, fp64 +2.4 % per step with it)
  int coop = 0;              // option "coop": 0 = one lane per particle (bit-exact), 2 / 4 / 8 = lanes sharing a particle's
                             // list with a wave-shuffle reduction (k_gather_from_lists_coop; rounding-level differences)
  bool fusePredict = true;   // option "fuse_predict": pbf_steps runs finalise(t) + predict(t + 1) as one kernel
  bool cellDiffuse = true;   // option "cell_diffuse": one walk per occupied cell instead of one per particle
  // option "overlap_diffuse" (default on): inside pbf_step the colour diffusion — memory-latency bound, 85 % of its wave
  // time parked — runs on a side stream beside the VALU-bound solver iterations (nothing else touches colours)
  bool overlapDiffuse = true;
  bool overlapDiffuseForced = false;  // option / env set explicitly: no size heuristic
  hipStream_t sideStream = nullptr;
  hipStream_t copyStream = nullptr;  // pbf_download_aos_begin's DMA
  hipEvent_t evPacked = nullptr;
  bool downloadPending = false;
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  bool diffusePending = false;
  DevBuf diffSum, diffCnt;   // the overlapped diffusion's own per-cell scratch
  bool fuseDiffuse = false;  // option "fuse_diffuse": pbf_step folds the diffuse walk into the first lambda launch
                             // (bit-identical; measured 2 % SLOWER at 1 M — the colour loads stall the filter loop — so off)
  bool fuseDiffuseNow = false;
  // pbf_steps: the NEXT step's predict rides on this step's finalise (k_finalise_predict) / has already been done
  bool fuseNextPredict = false, prePredicted = false;
  DevBuf bricks, brickCtl;  // non-empty brick list; brickCtl = {nActive, ticket[kTickets], nBigCells}
  DevBuf bigCells;          // cells with more than BIG_CELL members this step (k_sort_big_cells)
  uint32_t gatherSeq = 0;   // which ticket word the next persistent gather launch uses
  int numCUs = 256;
  uint32_t timingMask = 0xFFFFFFFFu;  // option "timing_mask": which stages PBF_FLAG_STAGE_TIMING brackets with events
  uint32_t padLds = 0;      // option "pad_lds": occupancy limiter for k_gather_global
  int gatherKind = 1;       // 0 = global walk (k_gather_global), 1 = neighbour lists (default), 3 = LDS tiles per brick (pbf_tiles.hpp)
  uint32_t tileCap = 0, listMax = 0;  // 0 = defaults (env PBF_TILE_CAP / PBF_LIST_MAX override)
  size_t tableCap = 0;   // entries allocated in count/table
  uint32_t tableN = 0;
  uint32_t countedTableN = 0;
  uint64_t extent[3] = {0, 0, 0};
  double minExtent[3] = {0, 0, 0};

  // stage timing
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> freeEvents;
  double stageMs[ST_COUNT] = {0};
  uint64_t stageCalls[ST_COUNT] = {0};
};

namespace {

#define HIPCHK(ctx, expr)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
      return PBF_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

int fail(pbf_ctx *ctx, int code, const std::string &msg) {
  ctx->err = msg;
  return code;
}

int ensure(pbf_ctx *ctx, DevBuf &b, size_t bytes, bool zero = false) {
  if (b.cap >= bytes) return PBF_OK;
  void *np = nullptr;
  const size_t want = bytes + bytes / 4 + 256;
  HIPCHK(ctx, hipMalloc(&np, want));
  ctx->allocEpoch++;
  if (zero) HIPCHK(ctx, hipMemsetAsync(np, 0, want, ctx->stream));
  if (b.p) {
    // contents are never carried across a grow: callers refill
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(b.p));
  }
  b.p = np;
  b.cap = want;
  return PBF_OK;
}

template <typename N> size_t vsz() { return sizeof(vec4<N>); }

int ensure_particles(pbf_ctx *ctx, size_t n) {
  n = std::max(n, ctx->reserve);
  if (ctx->cap >= n) return PBF_OK;
  // slab mode (a reserve was asked for): pbf_slab_step leaves the slots of the particles that left — migrants, last step's
  // copies — in place until the step's sort drops them, so a step transiently needs more slots than live particles
  if (ctx->reserve) n += n / 4 + 65536;
  const size_t v = ctx->fp64 ? sizeof(double4) : sizeof(float4);
  for (int s = 0; s < 2; ++s) {
    if (int rc = ensure(ctx, ctx->pos4[s], n * v)) return rc;
    if (int rc = ensure(ctx, ctx->vel4[s], n * v)) return rc;
    if (int rc = ensure(ctx, ctx->col4[s], n * v)) return rc;
    if (int rc = ensure(ctx, ctx->id[s], n * 8)) return rc;
    if (int rc = ensure(ctx, ctx->type[s], n)) return rc;
    if (int rc = ensure(ctx, ctx->key[s], n * 4)) return rc;
  }
  for (int s = 0; s < 3; ++s)
    if (int rc = ensure(ctx, ctx->pstar[s], n * v)) return rc;
  if (int rc = ensure(ctx, ctx->permTmp, n * 4)) return rc;
  if (int rc = ensure(ctx, ctx->slotOf, n * 4)) return rc;
  if (int rc = ensure(ctx, ctx->qpos, (n + QPOS_PAD) * 8)) return rc;
  if (int rc = ensure(ctx, ctx->nbrCount, n * 4)) return rc;
  // two-tier lists: NBR_ROWS slots per particle in rows + a pool of NBR_EXTRA-slot chunks for the few longer lists
  // (one allocation: a lane addresses both tiers from its row pointer with a 32-bit word offset — which caps the whole at
  // 2^31 words, i.e. ~45 M particles per GPU; beyond that the pool is left out and longer lists walk)
  const size_t rowWords = ((n + BLOCK - 1) / BLOCK) * size_t(NBR_ROWS) * BLOCK;
  ctx->nbrChunks = ctx->nbrChunksOpt ? ctx->nbrChunksOpt : uint32_t(n / 8 + 1024);  // (option "nbr_chunks": tests shrink the pool)
  if (rowWords + size_t(ctx->nbrChunks) * NBR_EXTRA >= (size_t(1) << 31)) ctx->nbrChunks = 0;
  ctx->nbrExtraAt = rowWords;
  if (int rc = ensure(ctx, ctx->nbrList, (rowWords + size_t(ctx->nbrChunks) * NBR_EXTRA + 64) * 4)) return rc;
  ctx->cap = n;
  return PBF_OK;
}

template <typename N> ParticleArrays<N> arrays(pbf_ctx *ctx, int set, int pstarIdx) {
  ParticleArrays<N> a;
  a.pos4 = ctx->pos4[set].as<vec4<N>>();
  a.vel4 = ctx->vel4[set].as<vec4<N>>();
  a.col4 = ctx->col4[set].as<vec4<N>>();
  a.pstar = ctx->pstar[pstarIdx].as<vec4<N>>();
  a.id = ctx->id[set].as<uint64_t>();
  a.type = ctx->type[set].as<uint8_t>();
  a.key = ctx->key[set].as<uint32_t>();
  return a;
}

// ---- per-step constants, in N, as the reference computes them ---------------------------------
template <typename N> N piN() { return std::acos(-N(1)); }
// sph.hpp:251-253 — std::pow(N, int) promotes to double for N = float; the result is rounded to N
template <typename N> N poly6_factor(N h) { return N(N(315.0) / (N(64.0) * piN<N>() * std::pow(h, 9))); }
template <typename N> N spiky_factor(N h) { return N(-(N(45.0) / (piN<N>() * std::pow(h, 6)))); }

template <typename N> int make_consts(pbf_ctx *ctx, const pbf_params *p, StepConsts<N> &c) {
  const N h = N(ctx->desc.h), scale = N(p->scale);
  c.h = h, c.dt = N(p->dt), c.scale = scale;
  // ompsph.hpp:132-135
  const N padding = h * 2;
  uint64_t ext[3];
  for (int i = 0; i < 3; ++i) {
    c.force[i] = N(p->constant_force[i]);
    c.minB[i] = N(p->min_bound[i]);
    c.maxB[i] = N(p->max_bound[i]);
    const N lo = c.minB[i] / scale - padding;
    const N hi = c.maxB[i] / scale + padding;
    c.minExtent[i] = lo;
    ext[i] = static_cast<uint64_t>((hi - lo) / h);
    ctx->minExtent[i] = double(lo);
  }
  for (int i = 0; i < 3; ++i) {
    if (ext[i] > 1023)
      return fail(ctx, PBF_ERR_INVALID, "grid extent exceeds the 10-bit-per-axis Morton range (curves.h:72-88)");
    ctx->extent[i] = ext[i];
  }
  c.xoff = 0;
  c.slabOn = 0, c.sxlo = 0, c.sxhi = 0xFFFFFFFFu, c.sHasL = c.sHasR = 0;
  if (ctx->slabStepMode && ctx->comm) {  // pbf_slab_step: predict sorts the particles into stayers / leavers / old copies
    const uint32_t xo = ctx->slabConfigured ? ctx->xoff : 0u;
    const int r = ctx->comm->rank, nr = ctx->comm->nranks;
    c.slabOn = 1;
    c.sxlo = ctx->cuts[r] - std::min(ctx->cuts[r], xo), c.sxhi = ctx->cuts[r + 1] - std::min(ctx->cuts[r + 1], xo);
    c.sHasL = r > 0 ? 1u : 0u, c.sHasR = r + 1 < nr ? 1u : 0u;
  }
  if (ctx->slabConfigured) {  // keys live in this rank's x frame: columns [xoff, right ghost column]
    c.xoff = ctx->xoff;
    const uint64_t hi = ctx->slabCut.has_right ? std::min<uint64_t>(ext[0], uint64_t(ctx->slabCut.xhi) + 1) : ext[0];
    ext[0] = hi > ctx->xoff ? hi - ctx->xoff : 1;
  }
  c.tableN = morton_encode(uint32_t(ext[0]), uint32_t(ext[1]), uint32_t(ext[2]));  // sph.hpp:240
  if (c.tableN == 0) return fail(ctx, PBF_ERR_INVALID, "empty grid (max_bound <= min_bound?)");
  c.poly6Factor = poly6_factor<N>(h);
  c.spikyFactor = spiky_factor<N>(h);
  {  // ompsph.hpp:213 with poly6Kernel of ompsph.hpp:67-69
    const N r = N(CorrDeltaQ * h);
    const N d = (h * h) - r * r;
    c.p6DeltaQ = r <= h ? c.poly6Factor * (d * d * d) : N(0);
  }
  c.diffuseT = c.dt / N(750.0);
  c.h2filter = (h * h) * N(1.00001);
  c.n = uint32_t(ctx->n);
  c.nWells = uint32_t(p->n_wells > 0 ? p->n_wells : 0);
  c.hasObstacles = ctx->hasObstacles ? 1u : 0u;
  ctx->tableN = c.tableN;
  return PBF_OK;
}

// A histogram that was built but never consumed (predict without sort) must not leak into the next one.
int drop_histogram(pbf_ctx *ctx) {
  if (ctx->counted && ctx->count.p) HIPCHK(ctx, hipMemsetAsync(ctx->count.p, 0, ctx->count.cap, ctx->stream));
  ctx->counted = false;
  return PBF_OK;
}

int ensure_table(pbf_ctx *ctx, uint32_t tableN) {
  const size_t entries = size_t(tableN) + 2;  // buckets 0..tableN (+1 overflow) and the closing total
  if (ctx->tableCap >= entries) return PBF_OK;
  // count must start (and, by the atomicSub in k_scatter_slots, always returns to) all-zero
  ctx->count.cap = 0;
  if (ctx->count.p) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(ctx->count.p));
    ctx->count.p = nullptr;
  }
  if (int rc = ensure(ctx, ctx->count, (entries + SCAN_TILE) * 4, true)) return rc;
  if (int rc = ensure(ctx, ctx->table, (entries + SCAN_TILE) * 4)) return rc;
  const size_t nb = (entries + SCAN_TILE - 1) / SCAN_TILE + 1;
  if (int rc = ensure(ctx, ctx->blockSums, nb * 4)) return rc;
  if (int rc = ensure(ctx, ctx->bricks, (entries / Brick<kBrickZ>::HOME + 2) * 4)) return rc;
  if (int rc = ensure(ctx, ctx->brickCtl, (kTickets + 2) * 4)) return rc;
  ctx->tableCap = ctx->count.cap / 4 - SCAN_TILE;
  return PBF_OK;
}

// ---- stage timing (Stopwatch analogue, utils.hpp:15-57) ----------------------------------------
hipEvent_t get_event(pbf_ctx *ctx) {
  if (!ctx->freeEvents.empty()) {
    hipEvent_t e = ctx->freeEvents.back();
    ctx->freeEvents.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
int resolve_events(pbf_ctx *ctx) {
  if (ctx->pending.empty()) return PBF_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto &ep : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
      ctx->stageMs[ep.stage] += ms;
      ctx->stageCalls[ep.stage] += 1;
    }
    ctx->freeEvents.push_back(ep.a);
    ctx->freeEvents.push_back(ep.b);
  }
  ctx->pending.clear();
  return PBF_OK;
}
struct StageTimer {
  pbf_ctx *ctx;
  EventPair ep{};
  bool on;
  StageTimer(pbf_ctx *c, int stage)
      : ctx(c), on((c->desc.flags & PBF_FLAG_STAGE_TIMING) != 0 && ((c->timingMask >> stage) & 1u) != 0) {
    if (!on) return;
    ep.a = get_event(ctx), ep.b = get_event(ctx), ep.stage = stage;
    (void)hipEventRecord(ep.a, ctx->stream);
  }
  ~StageTimer() {
    if (!on) return;
    (void)hipEventRecord(ep.b, ctx->stream);
    ctx->pending.push_back(ep);
    if (ctx->pending.size() > 8192) (void)resolve_events(ctx);
  }
};

// (at least one workgroup: a slab may own no particle at all, and every kernel bounds-checks its index)
inline dim3 grid_for(size_t n) { return dim3(unsigned(std::max<size_t>(1, (n + BLOCK - 1) / BLOCK))); }

#define LAUNCH_CHECK(ctx)                                                  \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(e_); \
      return PBF_ERR_HIP;                                                  \
    }                                                                      \
  } while (0)

int upload_wells(pbf_ctx *ctx, const pbf_params *p) {
  if (p->n_wells <= 0) return PBF_OK;
  if (!p->wells) return fail(ctx, PBF_ERR_INVALID, "n_wells > 0 but wells == NULL");
  const size_t cnt = size_t(p->n_wells) * 4;
  const size_t esz = ctx->fp64 ? 8 : 4;
  if (int rc = ensure(ctx, ctx->wells, cnt * esz)) return rc;
  if (ctx->fp64) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->wells.p, p->wells, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
  } else {
    std::vector<float> w(cnt);
    for (size_t i = 0; i < cnt; ++i) w[i] = float(p->wells[i]);
    HIPCHK(ctx, hipMemcpyAsync(ctx->wells.p, w.data(), cnt * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // w is a temporary
  }
  return PBF_OK;
}

// ---- stages ------------------------------------------------------------------------------------
template <typename N> int stage_predict(pbf_ctx *ctx, const pbf_params *p) {
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  if (int rc = ensure_table(ctx, c.tableN)) return rc;
  if (int rc = drop_histogram(ctx)) return rc;
  if (int rc = upload_wells(ctx, p)) return rc;
  StageTimer t(ctx, ST_PREDICT);
  const int s = ctx->cur;
  hipLaunchKernelGGL((k_predict<N>), grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, c,
                     ctx->pos4[s].as<const vec4<N>>(), ctx->vel4[s].as<vec4<N>>(), ctx->type[s].as<const uint8_t>(),
                     ctx->wells.as<const N>(), ctx->pstar[s].as<vec4<N>>(), ctx->key[s].as<uint32_t>(),
                     ctx->count.as<uint32_t>());
  LAUNCH_CHECK(ctx);
  ctx->pcur = s;
  ctx->sorted = false;
  ctx->nbrValid = false;
  ctx->qposValid = false;
  ctx->omegaValid = false;
  ctx->pstarInRows = false, ctx->rowsValid = false, ctx->rowsCurrent = false;
  ctx->counted = true;
  ctx->countedTableN = c.tableN;
  return PBF_OK;
}

bool row_mode(const pbf_ctx *ctx);

// list of the non-empty 4 x 4 x 4 bricks for the persistent brick kernels (k_diffuse_bricks, pbf_tiles.hpp): built by the sort
// stage when one of them is going to run, otherwise by whoever turns out to need it
void brick_list(pbf_ctx *ctx, uint32_t tableN, bool counterIsZero = false) {
  if (ctx->bricksValid) return;
  if (!counterIsZero) (void)hipMemsetAsync(ctx->brickCtl.p, 0, 4, ctx->stream);
  const uint32_t home = Brick<kBrickZ>::HOME, nBricks = (tableN + home - 1) / home;
  hipLaunchKernelGGL(k_brick_list, grid_for(nBricks), dim3(BLOCK), 0, ctx->stream, ctx->table.as<const uint32_t>(), tableN, home,
                     nBricks, ctx->bricks.as<uint32_t>(), ctx->brickCtl.as<uint32_t>());
  ctx->bricksValid = true;
}

// one to two exclusive scans in three launches (k_scan_sums zeroes `nZero` control words on its way)
void launch_scans(pbf_ctx *ctx, const ScanJobs &jobs, int njobs, uint32_t *zero = nullptr, uint32_t nZero = 0) {
  const uint32_t blocks = jobs.nb[0] + (njobs > 1 ? jobs.nb[1] : 0u);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(blocks), dim3(BLOCK), 0, ctx->stream, jobs);
  hipLaunchKernelGGL(k_scan_sums, dim3(uint32_t(njobs)), dim3(BLOCK), 0, ctx->stream, jobs, zero, nZero);
  hipLaunchKernelGGL(k_scan_apply, dim3(blocks), dim3(BLOCK), 0, ctx->stream, jobs);
}
ScanJobs scan_job(const uint32_t *count, uint32_t len, uint32_t *sums, uint32_t *table) {
  ScanJobs j{};
  j.count[0] = count, j.sums[0] = sums, j.table[0] = table, j.len[0] = len, j.nb[0] = (len + SCAN_TILE - 1) / SCAN_TILE;
  return j;
}

template <typename N> int stage_sort(pbf_ctx *ctx, const pbf_params *p) {
  // the scatter consumes the histogram k_predict built (atomicSub back to zero): never run it twice
  if (!ctx->counted) return fail(ctx, PBF_ERR_STATE, "pbf_stage_sort needs pbf_stage_predict first");
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  if (c.tableN != ctx->countedTableN) return fail(ctx, PBF_ERR_STATE, "bounds changed between predict and sort");
  StageTimer t(ctx, ST_SORT);
  const uint32_t len = c.tableN + 2;
  const uint32_t nb = (len + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t *count = ctx->count.as<uint32_t>(), *table = ctx->table.as<uint32_t>(),
           *sums = ctx->blockSums.as<uint32_t>();
  ScanJobs jobs = scan_job(count, len, sums, table);
  int njobs = 1;
  (void)nb;
  const int s = ctx->cur, d = 1 - s;
  // option "row_major": the box cells' populations in cell-row-major order and their scan (the row table), read off the
  // Morton table; k_rank_move below then writes the iterations' row-major copy on its way
  RowArrays<N> row{};
  ctx->rowsValid = false, ctx->pstarInRows = false, ctx->rowsCurrent = false, ctx->rowColValid = false;
  if (row_mode(ctx)) {
    // the cube that holds every cell the Morton table knows: P = 2^(bits of tableN / 3, rounded up)
    uint32_t pshift = 1;
    while (pshift < 10 && (uint64_t(1) << (3 * pshift)) < uint64_t(c.tableN)) ++pshift;
    const size_t ncells = size_t(1) << (3 * pshift), llen = ncells + 1;
    const uint32_t lnb = uint32_t((llen + SCAN_TILE - 1) / SCAN_TILE);
    const size_t v = sizeof(vec4<N>);
    if (int rc = ensure(ctx, ctx->linCount, (ncells + 2 + SCAN_TILE) * 4)) return rc;
    if (int rc = ensure(ctx, ctx->linTable, (ncells + 2 + SCAN_TILE) * 4)) return rc;
    if (int rc = ensure(ctx, ctx->linSums, (size_t(lnb) + 2) * 4)) return rc;
    for (int k = 0; k < 2; ++k)
      if (int rc = ensure(ctx, ctx->rowPstar[k], ctx->cap * v)) return rc;
    if (int rc = ensure(ctx, ctx->rowMass, ctx->cap * sizeof(N))) return rc;
    if (int rc = ensure(ctx, ctx->rowQpos, (ctx->cap + QPOS_PAD) * 8)) return rc;
    if (int rc = ensure(ctx, ctx->rowXYZ, ctx->cap * 4)) return rc;
    if (int rc = ensure(ctx, ctx->rowType, ctx->cap)) return rc;
    if (int rc = ensure(ctx, ctx->rowSlotOf, ctx->cap * 4)) return rc;
    const bool rowDiffuse = ctx->cellDiffuse && ctx->rowDiffuse > 0;
    const uint32_t segShift = std::min<uint32_t>(pshift, 6u);  // segments of 64 x cells (the whole row in a smaller cube)
    const size_t nSegTotal = ncells >> segShift;
    if (rowDiffuse) {
      if (int rc = ensure(ctx, ctx->rowCol, ctx->cap * v)) return rc;
      if (int rc = ensure(ctx, ctx->rowMortonOf, ctx->cap * 4)) return rc;
      if (int rc = ensure(ctx, ctx->rowSegs, (std::min(nSegTotal, ctx->cap) + 1) * 4)) return rc;
    }
    // the cells' populations in row order straight from the histogram, so that both tables' scans share their launches
    hipLaunchKernelGGL(k_lin_count, grid_for(ncells + 1), dim3(BLOCK), 0, ctx->stream, c.tableN, pshift, count,
                       ctx->linCount.as<uint32_t>());
    jobs.count[1] = ctx->linCount.as<const uint32_t>(), jobs.sums[1] = ctx->linSums.as<uint32_t>(), jobs.table[1] = ctx->linTable.as<uint32_t>();
    jobs.len[1] = uint32_t(llen), jobs.nb[1] = lnb;
    njobs = 2;
    // (brickCtl = {nActive bricks, tickets[kTickets], number of big cells}: zeroed once per step, by k_scan_sums on its way)
    launch_scans(ctx, jobs, njobs, ctx->brickCtl.as<uint32_t>(), kTickets + 2);
    if (rowDiffuse)  // the x-segments that hold a particle (at most one per particle), for k_diffuse_rows
      hipLaunchKernelGGL(k_row_segments, grid_for(nSegTotal), dim3(BLOCK), 0, ctx->stream, pshift, segShift,
                         ctx->linTable.as<const uint32_t>(), ctx->rowSegs.as<uint32_t>(), ctx->linCount.as<uint32_t>() + ncells + 2);
    ctx->rcur = 0;
    row = RowArrays<N>{ctx->rowPstar[0].as<vec4<N>>(), ctx->rowMass.as<N>(), ctx->rowQpos.as<uint2>(), ctx->rowXYZ.as<uint32_t>(),
                       ctx->rowType.as<uint8_t>(), ctx->rowSlotOf.as<uint32_t>(),
                       rowDiffuse ? ctx->rowCol.as<vec4<N>>() : nullptr, rowDiffuse ? ctx->rowMortonOf.as<uint32_t>() : nullptr,
                       ctx->linTable.as<const uint32_t>(), ctx->linCount.as<uint32_t>() + ncells + 1, pshift};
    ctx->rowShift = pshift, ctx->rowSegShift = segShift;
    ctx->rowColValid = rowDiffuse;
    ctx->rowsValid = true, ctx->rowsCurrent = true;
  }
  if (njobs == 1) launch_scans(ctx, jobs, 1, ctx->brickCtl.as<uint32_t>(), kTickets + 2);  // (Morton order only)
  uint32_t *nBig = ctx->brickCtl.as<uint32_t>() + kTickets + 1;
  if (int rc = ensure(ctx, ctx->bigCells, (ctx->cap / BIG_CELL + 2) * 4)) return rc;
  hipLaunchKernelGGL(k_scatter_slots, grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, c.n, c.tableN,
                     ctx->key[s].as<const uint32_t>(), table, count, ctx->permTmp.as<uint32_t>(),
                     ctx->bigCells.as<uint32_t>(), nBig);
  // pile-ups only: cells with more than BIG_CELL members get their segment sorted (exits at once when there are none)
  hipLaunchKernelGGL(k_sort_big_cells, dim3(64), dim3(BLOCK), 0, ctx->stream, table, c.tableN,
                     ctx->bigCells.as<const uint32_t>(), nBig, ctx->permTmp.as<uint32_t>(),
                     ctx->key[s].as<const uint32_t>());
  // pbf_slab_step: the arrays still hold the slots of the particles that left (DEAD_KEY: skipped by the scatter); only
  // the live ones arrive in the sorted set
  const size_t nLive = ctx->slabStepMode ? ctx->sortLive : ctx->n;
  hipLaunchKernelGGL((k_rank_move<N>), grid_for(nLive), dim3(BLOCK), 0, ctx->stream, c, uint32_t(nLive), c.tableN,
                     ctx->permTmp.as<const uint32_t>(), table, arrays<N>(ctx, s, s), arrays<N>(ctx, d, d),
                     ctx->slabActive ? ctx->slotOf.as<uint32_t>() : nullptr, ctx->qpos.as<uint2>(), row);
  ctx->n = nLive;
  ctx->gatherSeq = 0;  // (fresh tickets)
  ctx->bricksValid = false;
  if (!ctx->rowColValid) brick_list(ctx, c.tableN, /*counterIsZero=*/true);  // (the row-major diffusion has its own segments)
  LAUNCH_CHECK(ctx);
  ctx->cur = d;
  ctx->pcur = d;
  ctx->sorted = true;
  ctx->nbrValid = false;
  ctx->qposValid = true;
  if (ctx->rowsValid) ctx->pstarInRows = true, ctx->qposValid = false;  // (the sort wrote the row copy only)
  ctx->counted = false;
  return PBF_OK;
}

// the Jacobi partner of pstar[pcur]: any of the three buffers that is neither live nor needed
inline int other_pstar(const pbf_ctx *ctx) { return ctx->pcur == 2 ? ctx->cur : 2; }

// Launch one gather stage.  gatherKind picks the kernel (option "gather" / env PBF_GATHER);
// PBF_FLAG_NO_LDS always forces the plain per-particle global walk.
enum GatherMode { GATHER_PLAIN = 0, GATHER_SAVE_LISTS = 1, GATHER_FROM_LISTS = 2 };

// Row-major iterations apply to the default configuration only (neighbour lists, lambda riding on the build, one lane per
// particle, eager launches); everything else keeps the Morton walk.
bool row_mode(const pbf_ctx *ctx) {
  return ctx->rowMajor > 0 && ctx->gatherKind == 1 && ctx->splitBuild == 8 && ctx->coop == 0 && ctx->reuseLists &&
         !ctx->fuseDiffuse && ctx->graphMode <= 0 &&
         !(ctx->desc.flags & PBF_FLAG_NO_LDS) && uint64_t(ctx->n) * (ctx->fp64 ? 32u : 16u) <= 0xFFFFFFFFull;
}
template <typename N> RowWalk row_walk(pbf_ctx *ctx) {
  return RowWalk{ctx->linTable.as<const uint32_t>(), ctx->rowXYZ.as<const uint32_t>(), ctx->rowSlotOf.as<const uint32_t>(),
                 ctx->table.as<const uint32_t>(), ctx->tableN, ctx->rowShift};
}
// whoever wants {pStar, lambda} in the Morton-sorted array (stage-level read-backs, the extras, the surface) gets it here
template <typename N> int materialise_pstar(pbf_ctx *ctx) {
  if (!ctx->pstarInRows) return PBF_OK;
  hipLaunchKernelGGL((k_rows_to_morton<N>), grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, uint32_t(ctx->n),
                     ctx->rowPstar[ctx->rcur].as<const vec4<N>>(), ctx->rowSlotOf.as<const uint32_t>(),
                     ctx->pstar[ctx->pcur].as<vec4<N>>());
  LAUNCH_CHECK(ctx);
  ctx->pstarInRows = false;
  return PBF_OK;
}

// A zeroed word of brickCtl (the sort stage zeroes them all once per step) for ONE launch: the work ticket of a persistent
// tile kernel, or the chunk allocator of a list build.  More launches than words since the last sort: re-arm.
uint32_t *next_ticket(pbf_ctx *ctx) {
  uint32_t *ctl = ctx->brickCtl.as<uint32_t>();
  if (ctx->gatherSeq >= kTickets) {
    (void)hipMemsetAsync(ctl + 1, 0, kTickets * 4, ctx->stream);
    ctx->gatherSeq = 0;
  }
  return ctl + 1 + ctx->gatherSeq++;
}
// build = this launch WRITES the lists (takes a fresh chunk allocator)
NbrLists nbr_lists(pbf_ctx *ctx, bool build) {
  return NbrLists{ctx->nbrList.as<uint32_t>(), ctx->nbrCount.as<uint32_t>(), build ? next_ticket(ctx) : nullptr,
                  ctx->nbrExtraAt, ctx->nbrChunks};
}

template <typename N, typename Op>
int launch_gather(pbf_ctx *ctx, const StepConsts<N> &c, typename Op::Args args, GatherMode mode = GATHER_PLAIN) {
  const uint32_t *key = ctx->key[ctx->cur].as<const uint32_t>();
  const uint32_t *table = ctx->table.as<const uint32_t>();
  if ((ctx->desc.flags & PBF_FLAG_NO_LDS) || ctx->gatherKind == 0) {
    // `padLds` bytes of unused dynamic LDS cap the workgroups per CU: fewer resident waves keep the
    // 32 KiB L1 from thrashing on the gather's working set (each wave touches ~9 cache lines per load)
    hipLaunchKernelGGL((k_gather_global<N, Op>), grid_for(ctx->n), dim3(BLOCK), ctx->padLds, ctx->stream, c, args, key,
                       table);
    LAUNCH_CHECK(ctx);
    return PBF_OK;
  }
  if constexpr (Op::kTileable && Op::kFilter) {
    // gather = 3: the solver iteration per brick out of LDS tiles (pbf_tiles.hpp); everything else it does not cover
    // (diffuse, the opt-in extras) takes the list path below
    if (ctx->gatherKind == 3 && mode != GATHER_PLAIN && uint64_t(ctx->n) * sizeof(typename Op::Src) <= 0xFFFFFFFFull) {
      using B = TileBrick;
      static_assert(kBrickZ == TILE_BZ, "the sort stage's brick list is the tile kernels' one");
      brick_list(ctx, c.tableN);
      constexpr int WAYS = 4, LMAX = 16;
      uint32_t cap = ctx->tileCap ? std::min(ctx->tileCap, 65535u) : 2048u;  // list entries of a tiled brick are tile slots
      uint32_t *nl = ctx->nbrList.as<uint32_t>(), *nc = ctx->nbrCount.as<uint32_t>();
      uint32_t *ctl = ctx->brickCtl.as<uint32_t>();
      auto ticket = [&]() -> uint32_t * { return next_ticket(ctx); };
      auto launch_dims = [&](const void *kernel, size_t lds, size_t &attrSet) {
        if (lds > attrSet) {
          (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
          attrSet = lds;
        }
        const uint32_t perCU = uint32_t(std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / (lds + 512))));
        return dim3(uint32_t(ctx->numCUs) * perCU);
      };
      if (mode == GATHER_SAVE_LISTS) {
        uint2 *qp = ctx->qpos.as<uint2>();
        if (!ctx->qposValid) {
          hipLaunchKernelGGL((k_quantise<N>), grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, c, Op::src(args), qp);
          ctx->qposValid = true;
        }
        StageTimer tb(ctx, ST_BUILD);
        const size_t lds = B::HDR2 + size_t(cap + WAYS) * sizeof(uint2) + size_t(LMAX + WAYS) * TILE_THREADS * 2;
        if (lds > 160 * 1024 - 64) return fail(ctx, PBF_ERR_INVALID, "tile cap exceeds the CU's 160 KiB LDS");
        auto kernel = k_tile_build<N, WAYS, LMAX>;
        static size_t attrSet = 0;
        const dim3 grid = launch_dims(reinterpret_cast<const void *>(kernel), lds, attrSet);
        hipLaunchKernelGGL(kernel, grid, dim3(TILE_THREADS), lds, ctx->stream, c, Op::src(args), qp, args.type, key, table,
                           ctx->bricks.as<const uint32_t>(), ctl, ticket(), cap, nl, nc);
      }
      const size_t lds = B::HDR2 + size_t(cap) * sizeof(typename Op::Src);
      if (lds > 160 * 1024 - 64) return fail(ctx, PBF_ERR_INVALID, "tile cap exceeds the CU's 160 KiB LDS");
      auto kernel = k_tile_from_lists<N, Op>;
      static size_t attrSet = 0;  // per instantiation
      const dim3 grid = launch_dims(reinterpret_cast<const void *>(kernel), lds, attrSet);
      hipLaunchKernelGGL(kernel, grid, dim3(TILE_THREADS), lds, ctx->stream, c, args, key, table,
                         ctx->bricks.as<const uint32_t>(), ctl, ticket(), cap, nl, nc);
      LAUNCH_CHECK(ctx);
      return PBF_OK;
    }
  }
  if (ctx->gatherKind == 1 || ctx->gatherKind == 3) {
    const dim3 g = grid_for(ctx->n), b(BLOCK);
    auto from_lists = [&]() {
      const NbrLists ls = nbr_lists(ctx, false);  // the list-driven reader: one lane per particle, or (option "coop") a lane group per particle
      if constexpr (Op::kTileable && Op::kFilter) {
        auto coop_grid = [&](int k) { return dim3(unsigned(std::max<size_t>(1, (ctx->n * k + BLOCK - 1) / BLOCK))); };
        switch (ctx->coop) {
          case 2: hipLaunchKernelGGL((k_gather_from_lists_coop<N, Op, 2>), coop_grid(2), b, 0, ctx->stream, c, args, key, table, ls); return;
          case 4: hipLaunchKernelGGL((k_gather_from_lists_coop<N, Op, 4>), coop_grid(4), b, 0, ctx->stream, c, args, key, table, ls); return;
          case 8: hipLaunchKernelGGL((k_gather_from_lists_coop<N, Op, 8>), coop_grid(8), b, 0, ctx->stream, c, args, key, table, ls); return;
          default: break;
        }
      }
      // (auto = off: with round 3's trimmed fp64 sqrt / divides the plain reader wins in fp64 too — 2.05 vs 2.10 ms per step
      // at 1 M; in round 2, on the compiler's IEEE forms, the pipelined one had been 1.3 % ahead there)
      if (ctx->pipeline > 0)
        hipLaunchKernelGGL((k_gather_from_lists<N, Op, true>), g, b, 0, ctx->stream, c, args, key, table, ls);
      else
        hipLaunchKernelGGL((k_gather_from_lists<N, Op, false>), g, b, 0, ctx->stream, c, args, key, table, ls);
    };
    if (mode == GATHER_FROM_LISTS) {
      from_lists();
    } else if (mode == GATHER_SAVE_LISTS && ctx->splitBuild &&
               uint64_t(ctx->n) * sizeof(typename Op::Src) <= 0xFFFFFFFFull) {  // (k_build_lists: 32-bit offsets)  // build the lists, then run the op list-driven
      if constexpr (Op::kTileable && Op::kFilter) {  // (the ops that filter on pStar itself: lambda, delta-p)
        uint2 *qp = ctx->qpos.as<uint2>();
        if (!ctx->qposValid) {  // (only after a stage that moved pStar without refreshing its quantised copy)
          hipLaunchKernelGGL((k_quantise<N>), g, b, 0, ctx->stream, c, Op::src(args), qp);
          ctx->qposValid = true;
        }
        if (ctx->splitBuild == 8) {  // the op rides on the build
          // (staging depth 32 and four survivors per drain trip: measured best of 16 / 24 / 32 x 2 / 4 / 6 / 8)
#ifndef PBF_BUILD_LMAX
#define PBF_BUILD_LMAX 32
#endif
          hipLaunchKernelGGL((k_build_lists_op<N, Op, 4, PBF_BUILD_LMAX, 4>), g, b, 0, ctx->stream, c, args, Op::src(args), qp, args.type, key, table, nbr_lists(ctx, true));
          LAUNCH_CHECK(ctx);
          return PBF_OK;
        }
        StageTimer tb(ctx, ST_BUILD);
        if (ctx->splitBuild == 4)
          hipLaunchKernelGGL((k_build_lists_q<N, 2>), g, b, 0, ctx->stream, c, Op::src(args), qp, args.type, key, table, nbr_lists(ctx, true));
        else  // staging depth (option "list_max", default 32: one flush for most particles beats the two more workgroups per CU that 16 leaves room for: -2 % per step)
          switch (ctx->listMax ? ctx->listMax : 32u) {
            case 16: hipLaunchKernelGGL((k_build_lists_q<N, 4, 16>), g, b, 0, ctx->stream, c, Op::src(args), qp, args.type, key, table, nbr_lists(ctx, true)); break;
            case 24: hipLaunchKernelGGL((k_build_lists_q<N, 4, 24>), g, b, 0, ctx->stream, c, Op::src(args), qp, args.type, key, table, nbr_lists(ctx, true)); break;
            default: hipLaunchKernelGGL((k_build_lists_q<N, 4, 32>), g, b, 0, ctx->stream, c, Op::src(args), qp, args.type, key, table, nbr_lists(ctx, true)); break;
          }
      }
      from_lists();
    } else if (mode == GATHER_SAVE_LISTS) {
      hipLaunchKernelGGL((k_gather_lists<N, Op, 16, true>), g, b, 0, ctx->stream, c, args, key, table, nbr_lists(ctx, true));
    } else {
      switch (ctx->listMax ? ctx->listMax : 16u) {
        case 12: hipLaunchKernelGGL((k_gather_lists<N, Op, 12>), g, b, 0, ctx->stream, c, args, key, table, nbr_lists(ctx, false)); break;
        case 24: hipLaunchKernelGGL((k_gather_lists<N, Op, 24>), g, b, 0, ctx->stream, c, args, key, table, nbr_lists(ctx, false)); break;
        case 32: hipLaunchKernelGGL((k_gather_lists<N, Op, 32>), g, b, 0, ctx->stream, c, args, key, table, nbr_lists(ctx, false)); break;
        default: hipLaunchKernelGGL((k_gather_lists<N, Op, 16>), g, b, 0, ctx->stream, c, args, key, table, nbr_lists(ctx, false)); break;
      }
    }
    LAUNCH_CHECK(ctx);
    return PBF_OK;
  }
  return fail(ctx, PBF_ERR_INVALID, "unknown gather kind");
}

// The overlapped diffusion must be done before anything rewrites what it reads (colours, types, keys, table, brick
// list: the next sort) or reads what it writes (colours: download, surface).  Stream-side wait, no host sync.
int join_diffuse(pbf_ctx *ctx) {
  if (!ctx->diffusePending) return PBF_OK;
  ctx->diffusePending = false;
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->evJoin, 0));
  return PBF_OK;
}

template <typename N> int stage_diffuse(pbf_ctx *ctx, const pbf_params *p, bool overlap = false) {
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  const int s = ctx->cur, d = 1 - s;  // col4[d] is free after the sort
  typename DiffuseOp<N>::Args args{ctx->col4[s].as<const vec4<N>>(), ctx->col4[d].as<vec4<N>>(),
                                   ctx->type[s].as<const uint8_t>()};
  const bool timed = (ctx->desc.flags & PBF_FLAG_STAGE_TIMING) != 0 && ((ctx->timingMask >> ST_DIFFUSE) & 1u) != 0;
  // beside the solver iterations only where that pays: at 1 M particles -2.2 % per step, at 256 K +1.4 % (the side stream's
  // one-workgroup-per-CU launch is then longer than the iteration it hides behind) — measured, profiles/r03_matrix.txt
  overlap = overlap && ctx->overlapDiffuse && ctx->cellDiffuse && !(ctx->desc.flags & PBF_FLAG_NO_LDS) && !timed &&
            (ctx->overlapDiffuseForced || ctx->n >= (size_t(1) << 19));
  ctx->omegaValid = false;  // (the per-cell sums may be parked in pStar's idle Jacobi partner)
  StageTimer t(ctx, ST_DIFFUSE);
  if (ctx->cellDiffuse && ctx->rowColValid && ctx->rowsValid && ctx->rowDiffuse) {
    // the row-major copy: one wave per segment of 64 x cells, runs staged through LDS, sums applied in place (k_diffuse_rows).
    // On the solver's own stream: the launch is short, nothing is gained by hiding it
    const uint32_t ncells = 1u << (3u * ctx->rowShift);
    // records of one row's run in the LDS tile (66 cells; the settled dam-break's are ~460): a longer run walks from memory.
    // A small tile matters more than a roomy one: 13 single-wave workgroups per CU at 640 records (fp32), and the launch time
    // is inversely proportional to that number (measured: 1 024 records 68 us, 768 61 us, 640 51 us, 576 51 us)
    const uint32_t cap = ctx->diffuseCap ? ctx->diffuseCap : 640u;
    const size_t lds = size_t(cap + 8) * sizeof(vec4<N>) + size_t(DIFFUSE_ROW_THREADS) * (4 * sizeof(N) + 4) + cap + 8;  // (+ 8 records / types: the fold reads ahead)
    static size_t attrSet = 0;  // per instantiation
    if (lds > attrSet) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_diffuse_rows<N>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
      attrSet = lds;
    }
    const uint32_t perCU = uint32_t(std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / (lds + 256))));
    const uint32_t blocks = std::max(8u, (uint32_t(ctx->numCUs) * perCU) & ~7u);  // (the kernel deals segments per XCD: a multiple of 8)
    hipLaunchKernelGGL((k_diffuse_rows<N>), dim3(blocks), dim3(DIFFUSE_ROW_THREADS), lds, ctx->stream, c,
                       row_walk<N>(ctx), ctx->rowCol.as<const vec4<N>>(), ctx->rowType.as<const uint8_t>(),
                       ctx->rowMortonOf.as<const uint32_t>(), ctx->rowSegs.as<const uint32_t>(),
                       ctx->linCount.as<const uint32_t>() + ncells + 2, ctx->rowSegShift, args.colOut, cap, uint32_t(ctx->n));
    LAUNCH_CHECK(ctx);
    ctx->rowColValid = false;  // (col4 moves on: the copy is the pre-diffusion state)
  } else if (ctx->cellDiffuse && !(ctx->desc.flags & PBF_FLAG_NO_LDS)) {
    brick_list(ctx, c.tableN);
    // sums per cell, parked in buffers that are idle here (the Jacobi partner of pStar and the list lengths) — or, when
    // the stage runs beside the solver iterations, in scratch of its own
    vec4<N> *cellSum = ctx->pstar[other_pstar(ctx)].as<vec4<N>>();
    uint32_t *cellCnt = ctx->nbrCount.as<uint32_t>();
    hipStream_t st = ctx->stream;
    // records in the LDS tile: 48 KiB (fp32) / 96 KiB (fp64) of colours.  (Round 2 gave fp64 the same BYTES, i.e. half the
    // records — below the p99 of the settled dam-break's halos, 1 800: a third of the bricks fell back to the per-cell
    // global walk and the fp64 diffusion took 0.43 ms against fp32's 0.13.)
    uint32_t cap = 3072u;
    // beside the solver iterations: a 32-KiB tile (fp32) leaves room for THREE of the list build's 40-KiB workgroups on
    // the CU instead of two (measured: step -2.6 %; p99 of the settled dam-break's halos is 1 800 records, a fuller
    // brick walks globally)
    if (overlap) cap = 2048u;  // (fp64: 64 KiB + two of the build's 40-KiB workgroups)
    const size_t lds = Brick<4>::HDR + size_t(cap) * sizeof(vec4<N>);
    uint32_t perCU = uint32_t((160 * 1024) / (lds + 1024));
    if (overlap) {
      if (int rc = ensure(ctx, ctx->diffSum, ctx->cap * sizeof(vec4<N>))) return rc;
      if (int rc = ensure(ctx, ctx->diffCnt, ctx->cap * 4)) return rc;
      if (!ctx->sideStream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->sideStream, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->evJoin, hipEventDisableTiming));
      }
      cellSum = ctx->diffSum.as<vec4<N>>(), cellCnt = ctx->diffCnt.as<uint32_t>();
      st = ctx->sideStream;
      perCU = 1;  // one 48-KiB workgroup per CU: the list build's staging lists (24 KiB per workgroup) keep their room
      HIPCHK(ctx, hipEventRecord(ctx->evFork, ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(st, ctx->evFork, 0));
    }
    const uint32_t *key = ctx->key[s].as<const uint32_t>(), *table = ctx->table.as<const uint32_t>();
    static_assert(kBrickZ == 4, "the sort stage's brick list is the 4 x 4 x 4 one");
    hipLaunchKernelGGL((k_diffuse_bricks<N>), dim3(uint32_t(ctx->numCUs) * perCU), dim3(DIFFUSE_BRICK_THREADS), lds,
                       st, c, args.colIn, args.type, table, ctx->bricks.as<const uint32_t>(),
                       ctx->brickCtl.as<const uint32_t>(), cellSum, cellCnt, cap);
    hipLaunchKernelGGL((k_diffuse_apply<N>), grid_for(ctx->n), dim3(BLOCK), 0, st, c, args, key, table,
                       cellSum, cellCnt);
    LAUNCH_CHECK(ctx);
    if (overlap) {
      HIPCHK(ctx, hipEventRecord(ctx->evJoin, st));
      ctx->diffusePending = true;
    } else {
      ctx->nbrValid = false;
    }
  } else if (int rc = launch_gather<N, DiffuseOp<N>>(ctx, c, args)) {
    return rc;
  }
  std::swap(ctx->col4[s], ctx->col4[d]);
  return PBF_OK;
}

// the row build's shape: pair loads per trip, staging depth, survivors per drain trip (A/B builds override these)
#ifndef PBF_ROWS_W
#define PBF_ROWS_W 4
#endif
#ifndef PBF_ROWS_LMAX
#define PBF_ROWS_LMAX 32
#endif
#ifndef PBF_ROWS_FW
#define PBF_ROWS_FW 4
#endif
template <typename N> int stage_lambda(pbf_ctx *ctx, const pbf_params *p) {
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  StageTimer t(ctx, ST_LAMBDA);
  const int s = ctx->cur;
  // the survivors of lambda's filter are exactly delta's (same pStar): hand them over through HBM
  const bool lists = (ctx->gatherKind == 1 || ctx->gatherKind == 3) && ctx->reuseLists && !(ctx->desc.flags & PBF_FLAG_NO_LDS);
  const GatherMode save = lists ? GATHER_SAVE_LISTS : GATHER_PLAIN;
  ctx->nbrValid = lists;
  if (lists && ctx->rowsValid && ctx->rowsCurrent && row_mode(ctx) && !ctx->fuseDiffuseNow) {  // the build + lambda on the row-major copy
    const dim3 g = grid_for(ctx->n), b(BLOCK);
    const RowWalk rw = row_walk<N>(ctx);
    if (ctx->fast) {
      typename LambdaOp<N, true>::Args a{ctx->rowPstar[ctx->rcur].as<vec4<N>>(), nullptr, ctx->rowType.as<const uint8_t>(), ctx->rowMass.as<const N>()};
      hipLaunchKernelGGL((k_build_rows_op<N, LambdaOp<N, true>, PBF_ROWS_W, PBF_ROWS_LMAX, PBF_ROWS_FW>), g, b, 0, ctx->stream, c, a, ctx->rowQpos.as<const uint2>(), rw, nbr_lists(ctx, true));
    } else {
      typename LambdaOp<N, false>::Args a{ctx->rowPstar[ctx->rcur].as<vec4<N>>(), nullptr, ctx->rowType.as<const uint8_t>(), ctx->rowMass.as<const N>()};
      hipLaunchKernelGGL((k_build_rows_op<N, LambdaOp<N, false>, PBF_ROWS_W, PBF_ROWS_LMAX, PBF_ROWS_FW>), g, b, 0, ctx->stream, c, a, ctx->rowQpos.as<const uint2>(), rw, nbr_lists(ctx, true));
    }
    LAUNCH_CHECK(ctx);
    ctx->pstarInRows = true, ctx->nbrRows = true;
    return PBF_OK;
  }
  if (int rc = materialise_pstar<N>(ctx)) return rc;  // (a Morton-path launch after row-major ones: options changed mid-step)
  ctx->rowsCurrent = false, ctx->nbrRows = false;
  if (lists && ctx->fuseDiffuseNow) {  // pbf_step: the colour diffusion rides on this launch's walk
    ctx->fuseDiffuseNow = false;
    const int d = 1 - s;
    typename DiffuseOp<N>::Args xa{ctx->col4[s].as<const vec4<N>>(), ctx->col4[d].as<vec4<N>>(),
                                   ctx->type[s].as<const uint8_t>()};
    const uint32_t *key = ctx->key[s].as<const uint32_t>(), *table = ctx->table.as<const uint32_t>();
    const NbrLists ls = nbr_lists(ctx, true);
    if (ctx->fast) {
      typename LambdaOp<N, true>::Args a{ctx->pstar[ctx->pcur].as<vec4<N>>(), ctx->pos4[s].as<const vec4<N>>(),
                                         ctx->type[s].as<const uint8_t>()};
      hipLaunchKernelGGL((k_gather_lists<N, LambdaOp<N, true>, 16, true, DiffuseOp<N>>), grid_for(ctx->n), dim3(BLOCK), 0,
                         ctx->stream, c, a, key, table, ls, xa);
    } else {
      typename LambdaOp<N, false>::Args a{ctx->pstar[ctx->pcur].as<vec4<N>>(), ctx->pos4[s].as<const vec4<N>>(),
                                          ctx->type[s].as<const uint8_t>()};
      hipLaunchKernelGGL((k_gather_lists<N, LambdaOp<N, false>, 16, true, DiffuseOp<N>>), grid_for(ctx->n), dim3(BLOCK), 0,
                         ctx->stream, c, a, key, table, ls, xa);
    }
    LAUNCH_CHECK(ctx);
    std::swap(ctx->col4[s], ctx->col4[d]);
    return PBF_OK;
  }
  if (ctx->fast) {
    typename LambdaOp<N, true>::Args args{ctx->pstar[ctx->pcur].as<vec4<N>>(), ctx->pos4[s].as<const vec4<N>>(),
                                          ctx->type[s].as<const uint8_t>()};
    return launch_gather<N, LambdaOp<N, true>>(ctx, c, args, save);
  }
  typename LambdaOp<N, false>::Args args{ctx->pstar[ctx->pcur].as<vec4<N>>(), ctx->pos4[s].as<const vec4<N>>(),
                                         ctx->type[s].as<const uint8_t>()};
  return launch_gather<N, LambdaOp<N, false>>(ctx, c, args, save);
}

template <typename N> int stage_delta(pbf_ctx *ctx, const pbf_params *p) {
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  StageTimer t(ctx, ST_DELTA);
  if (ctx->nbrValid && ctx->nbrRows && ctx->rowsValid && ctx->rowsCurrent) {  // list-driven delta-p on the row-major copy
    ctx->nbrValid = false, ctx->omegaValid = false;
    const dim3 g = grid_for(ctx->n), b(BLOCK);
    const RowWalk rw = row_walk<N>(ctx);
    const int rin = ctx->rcur, rout = 1 - rin;
    const NbrLists ls = nbr_lists(ctx, false);
    const uint32_t *key = ctx->key[ctx->cur].as<const uint32_t>(), *table = ctx->table.as<const uint32_t>();
    if (ctx->fast) {
      typename DeltaOp<N, true>::Args a{ctx->rowPstar[rin].as<const vec4<N>>(), ctx->rowPstar[rout].as<vec4<N>>(), ctx->rowType.as<const uint8_t>(), ctx->rowQpos.as<uint2>()};
      if (ctx->pipeline > 0) hipLaunchKernelGGL((k_gather_from_lists<N, DeltaOp<N, true>, true, true>), g, b, 0, ctx->stream, c, a, key, table, ls, rw);
      else hipLaunchKernelGGL((k_gather_from_lists<N, DeltaOp<N, true>, false, true>), g, b, 0, ctx->stream, c, a, key, table, ls, rw);
    } else {
      typename DeltaOp<N, false>::Args a{ctx->rowPstar[rin].as<const vec4<N>>(), ctx->rowPstar[rout].as<vec4<N>>(), ctx->rowType.as<const uint8_t>(), ctx->rowQpos.as<uint2>()};
      if (ctx->pipeline > 0) hipLaunchKernelGGL((k_gather_from_lists<N, DeltaOp<N, false>, true, true>), g, b, 0, ctx->stream, c, a, key, table, ls, rw);
      else hipLaunchKernelGGL((k_gather_from_lists<N, DeltaOp<N, false>, false, true>), g, b, 0, ctx->stream, c, a, key, table, ls, rw);
    }
    LAUNCH_CHECK(ctx);
    ctx->rcur = rout;
    ctx->pstarInRows = true;
    ctx->qposValid = false;  // (the Morton-order quantised copy is stale now; the row copy is current)
    return PBF_OK;
  }
  if (int rc = materialise_pstar<N>(ctx)) return rc;
  ctx->rowsCurrent = false;
  const int s = ctx->cur, in = ctx->pcur, out = other_pstar(ctx);
  const GatherMode from = (ctx->nbrValid && !ctx->nbrRows) ? GATHER_FROM_LISTS : GATHER_PLAIN;
  ctx->nbrValid = false;  // delta moves pStar: the lists are stale afterwards
  ctx->omegaValid = false;  // (and may write the buffer the last vorticity pass left its result in)
  int rc;
  // delta-p's epilogue also writes the quantised copy of the new pStar: the next iteration's list build needs it
  uint2 *qp = ctx->qposValid ? ctx->qpos.as<uint2>() : nullptr;
  if (ctx->fast) {
    typename DeltaOp<N, true>::Args args{ctx->pstar[in].as<const vec4<N>>(), ctx->pstar[out].as<vec4<N>>(),
                                         ctx->type[s].as<const uint8_t>(), qp};
    rc = launch_gather<N, DeltaOp<N, true>>(ctx, c, args, from);
  } else {
    typename DeltaOp<N, false>::Args args{ctx->pstar[in].as<const vec4<N>>(), ctx->pstar[out].as<vec4<N>>(),
                                          ctx->type[s].as<const uint8_t>(), qp};
    rc = launch_gather<N, DeltaOp<N, false>>(ctx, c, args, from);
  }
  if (rc) return rc;
  ctx->pcur = out;
  return PBF_OK;
}

// Opt-in extras (pbf_params.vorticity / .xsph), absent from the reference: see VorticityOp / XsphOp.
template <typename N, bool FAST> int extras_impl(pbf_ctx *ctx, const pbf_params *p, const StepConsts<N> &c) {
  const int s = ctx->cur, o = 1 - s;
  const uint8_t *type = ctx->type[s].as<const uint8_t>();
  const vec4<N> *ps = ctx->pstar[s].as<const vec4<N>>();
  if (p->vorticity) {
    vec4<N> *omega = ctx->pstar[2].as<vec4<N>>();
    typename VorticityOp<N, FAST>::Args a1{ps, ctx->vel4[s].as<const vec4<N>>(), omega, type};
    if (int rc = launch_gather<N, VorticityOp<N, FAST>>(ctx, c, a1)) return rc;
    typename VorticityForceOp<N, FAST>::Args a2{ps, omega, ctx->vel4[s].as<const vec4<N>>(), ctx->vel4[o].as<vec4<N>>(), type};
    if (int rc = launch_gather<N, VorticityForceOp<N, FAST>>(ctx, c, a2)) return rc;
    std::swap(ctx->vel4[s], ctx->vel4[o]);
    ctx->omegaValid = true;
  }
  if (p->xsph) {
    typename XsphOp<N, FAST>::Args a3{ps, ctx->vel4[s].as<const vec4<N>>(), ctx->vel4[o].as<vec4<N>>(), type};
    if (int rc = launch_gather<N, XsphOp<N, FAST>>(ctx, c, a3)) return rc;
    std::swap(ctx->vel4[s], ctx->vel4[o]);
  }
  return PBF_OK;
}
template <typename N> int extras(pbf_ctx *ctx, const pbf_params *p, const StepConsts<N> &c) {
  if (ctx->slabActive) return PBF_OK;  // slab mode: pbf_slab_step runs them itself, with the ghost refreshes in between
  return ctx->fast ? extras_impl<N, true>(ctx, p, c) : extras_impl<N, false>(ctx, p, c);
}

template <typename N> int stage_finalise(pbf_ctx *ctx, const pbf_params *p) {
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  const int s = ctx->cur;
  if (ctx->fuseNextPredict && !(p->vorticity || p->xsph)) {
    // finalise(t) + predict(t + 1) in one pass: everything stage_predict does, around one launch
    ctx->fuseNextPredict = false;
    if (int rc = ensure_table(ctx, c.tableN)) return rc;
    if (int rc = drop_histogram(ctx)) return rc;
    if (int rc = upload_wells(ctx, p)) return rc;
    if (int rc = join_diffuse(ctx)) return rc;  // (the side stream reads keys and types of this step)
    StageTimer t(ctx, ST_FINALISE);
    if (ctx->pcur != s) {  // pStar is overwritten in place: make the live buffer pstar[cur] first
      std::swap(ctx->pstar[ctx->pcur], ctx->pstar[s]);
      ctx->pcur = s;
    }
    hipLaunchKernelGGL((k_finalise_predict<N>), grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, c,
                       ctx->type[s].as<const uint8_t>(), ctx->pstar[s].as<vec4<N>>(), ctx->pos4[s].as<vec4<N>>(),
                       ctx->vel4[s].as<vec4<N>>(), ctx->wells.as<const N>(), ctx->key[s].as<uint32_t>(),
                       ctx->count.as<uint32_t>(), ctx->rowPstar[ctx->rcur].as<const vec4<N>>(),
                       ctx->pstarInRows ? ctx->rowSlotOf.as<const uint32_t>() : nullptr);
    ctx->pstarInRows = false, ctx->rowsValid = false, ctx->rowsCurrent = false;
    LAUNCH_CHECK(ctx);
    ctx->sorted = false, ctx->nbrValid = false, ctx->qposValid = false, ctx->omegaValid = false;
    ctx->counted = true, ctx->countedTableN = c.tableN;
    ctx->prePredicted = true;
    return PBF_OK;
  }
  ctx->fuseNextPredict = false;
  if (ctx->pstarInRows && (p->vorticity || p->xsph))  // the extras read the Morton-sorted pStar
    if (int rc = materialise_pstar<N>(ctx)) return rc;
  StageTimer t(ctx, ST_FINALISE);
  if (ctx->pstarInRows)
    hipLaunchKernelGGL((k_finalise<N>), grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, c, ctx->type[s].as<const uint8_t>(),
                       ctx->rowPstar[ctx->rcur].as<const vec4<N>>(), ctx->pos4[s].as<vec4<N>>(), ctx->vel4[s].as<vec4<N>>(),
                       ctx->rowSlotOf.as<const uint32_t>());
  else
    hipLaunchKernelGGL((k_finalise<N>), grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, c, ctx->type[s].as<const uint8_t>(),
                       ctx->pstar[ctx->pcur].as<const vec4<N>>(), ctx->pos4[s].as<vec4<N>>(), ctx->vel4[s].as<vec4<N>>(),
                       static_cast<const uint32_t *>(nullptr));
  LAUNCH_CHECK(ctx);
  // keep pstar[cur] as the live buffer so that a later sort scatters pstar[cur] -> pstar[1-cur]
  if (ctx->pcur != s) {
    std::swap(ctx->pstar[ctx->pcur], ctx->pstar[s]);
    ctx->pcur = s;
  }
  ctx->nbrValid = false;
  if (p->vorticity || p->xsph) return extras<N>(ctx, p, c);
  return PBF_OK;
}

template <typename N> int step_impl(pbf_ctx *ctx, const pbf_params *p) {
  if (ctx->n == 0) return PBF_OK;  // "Particles depleted" (ompsph.hpp:122-126)
  if (ctx->prePredicted) {
    ctx->prePredicted = false;  // the previous step of this pbf_steps call has predicted already (k_finalise_predict)
  } else if (int rc = stage_predict<N>(ctx, p)) {
    return rc;
  }
  if (int rc = stage_sort<N>(ctx, p)) return rc;
  // diffuse (ompsph.hpp:188-207) visits exactly the candidates of the first lambda launch: fuse the two walks
  const bool fuse = ctx->fuseDiffuse && p->iteration > 0 && ctx->gatherKind == 1 && ctx->reuseLists &&
                    !(ctx->desc.flags & PBF_FLAG_NO_LDS);
  if (!fuse) {
    if (int rc = stage_diffuse<N>(ctx, p, /*overlap=*/p->iteration > 0)) return rc;
  }
  ctx->fuseDiffuseNow = fuse;
  for (uint64_t it = 0; it < p->iteration; ++it) {
    if (int rc = stage_lambda<N>(ctx, p)) return rc;
    if (int rc = stage_delta<N>(ctx, p)) return rc;
  }
  if (int rc = stage_finalise<N>(ctx, p)) return rc;
  return join_diffuse(ctx);  // the step is complete on ctx->stream only once the colours are
}

int check(pbf_ctx *ctx, const pbf_params *p, bool needSorted) {
  if (!ctx) return PBF_ERR_INVALID;
  if (!p) return fail(ctx, PBF_ERR_INVALID, "params == NULL");
  if (!(p->scale > 0) || !(p->dt > 0)) return fail(ctx, PBF_ERR_INVALID, "dt and scale must be > 0");
  if (needSorted && !ctx->sorted) return fail(ctx, PBF_ERR_STATE, "stage needs pbf_stage_sort first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->lastParams = *p;
  ctx->haveParams = true;
  return PBF_OK;
}

int drop_ghosts(pbf_ctx *ctx);  // (defined with the slab code)

template <typename N>
int upload_impl(pbf_ctx *ctx, size_t n, const uint64_t *id, const uint8_t *type, const N *mass, const N *pos,
                const N *vel, const N *colour) {
  if (ctx->downloadPending) (void)pbf_download_aos_end(ctx);  // (a download left open: finish it before the state changes)
  if (int rc = ensure_particles(ctx, n)) return rc;
  if (int rc = drop_histogram(ctx)) return rc;
  ctx->cur = 0, ctx->pcur = 0, ctx->sorted = false;
  ctx->ghostsPending = false, ctx->slabActive = false, ctx->prePredicted = false;
  ctx->n = n;
  ctx->hasObstacles = false;
  if (n == 0) return PBF_OK;
  std::vector<vec4<N>> P(n), V(n), C(n);
  for (size_t i = 0; i < n; ++i) {
    P[i] = make_vec4<N>(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], mass[i]);
    V[i] = make_vec4<N>(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2], N(0));
    C[i] = make_vec4<N>(colour[4 * i], colour[4 * i + 1], colour[4 * i + 2], colour[4 * i + 3]);
    if (type[i] & PBF_TYPE_OBSTACLE) ctx->hasObstacles = true;
  }
  ctx->realObstacles = ctx->hasObstacles;
  HIPCHK(ctx, hipMemcpyAsync(ctx->pos4[0].p, P.data(), n * sizeof(vec4<N>), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->vel4[0].p, V.data(), n * sizeof(vec4<N>), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->col4[0].p, C.data(), n * sizeof(vec4<N>), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->id[0].p, id, n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->type[0].p, type, n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // P, V, C are temporaries
  return PBF_OK;
}

template <typename N>
int download_impl(pbf_ctx *ctx, uint64_t *id, uint8_t *type, N *mass, N *pos, N *vel, N *colour) {
  if (int rc = drop_ghosts(ctx)) return rc;
  const size_t n = ctx->n;
  if (n == 0) return PBF_OK;
  const int s = ctx->cur;
  std::vector<vec4<N>> tmp(n);
  if (mass || pos) {
    HIPCHK(ctx, hipMemcpyAsync(tmp.data(), ctx->pos4[s].p, n * sizeof(vec4<N>), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i) {
      if (pos) pos[3 * i] = tmp[i].x, pos[3 * i + 1] = tmp[i].y, pos[3 * i + 2] = tmp[i].z;
      if (mass) mass[i] = tmp[i].w;
    }
  }
  if (vel) {
    HIPCHK(ctx, hipMemcpyAsync(tmp.data(), ctx->vel4[s].p, n * sizeof(vec4<N>), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i) vel[3 * i] = tmp[i].x, vel[3 * i + 1] = tmp[i].y, vel[3 * i + 2] = tmp[i].z;
  }
  if (colour) {
    HIPCHK(ctx, hipMemcpyAsync(colour, ctx->col4[s].p, n * sizeof(vec4<N>), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (id) HIPCHK(ctx, hipMemcpyAsync(id, ctx->id[s].p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (type) HIPCHK(ctx, hipMemcpyAsync(type, ctx->type[s].p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PBF_OK;
}

}  // namespace

#define DISPATCH(ctx, fn, ...) ((ctx)->fp64 ? fn<double>(__VA_ARGS__) : fn<float>(__VA_ARGS__))

extern "C" {

int pbf_abi_version(void) { return PBF_ABI_VERSION; }

int pbf_set_option(pbf_ctx *ctx, const char *name, int64_t value) {
  if (!ctx || !name) return PBF_ERR_INVALID;
  const std::string n(name);
  if (n == "list_max") ctx->listMax = uint32_t(value);
  else if (n == "gather") {
    if (value != 0 && value != 1 && value != 3) return fail(ctx, PBF_ERR_INVALID, "gather: 0, 1 or 3 (2, round 1's brick kernel, was removed)");
    ctx->gatherKind = int(value);
  }
  else if (n == "tile_cap") ctx->tileCap = uint32_t(value);
  else if (n == "reuse_lists") ctx->reuseLists = value != 0;
  else if (n == "fuse_diffuse") ctx->fuseDiffuse = value != 0;
  else if (n == "cell_diffuse") ctx->cellDiffuse = value != 0;
  else if (n == "pipeline") ctx->pipeline = int(value);
  else if (n == "graph") ctx->graphMode = int(value);
  else if (n == "overlap_diffuse") ctx->overlapDiffuse = value != 0, ctx->overlapDiffuseForced = true;
  else if (n == "coop") {
    if (value != 0 && value != 2 && value != 4 && value != 8) return fail(ctx, PBF_ERR_INVALID, "coop must be 0, 2, 4 or 8");
    ctx->coop = int(value);
  }
  else if (n == "split_build") ctx->splitBuild = int(value);
  else if (n == "fuse_predict") ctx->fusePredict = value != 0;
  else if (n == "timing_mask") ctx->timingMask = uint32_t(value);
  else if (n == "pad_lds") ctx->padLds = uint32_t(value);
  else if (n == "row_major") ctx->rowMajor = int(value);
  else if (n == "row_diffuse") ctx->rowDiffuse = int(value);
  else if (n == "diffuse_cap") {
    if (value < 0 || value > 4096) return fail(ctx, PBF_ERR_INVALID, "diffuse_cap: 0 (default) .. 4096 records");
    ctx->diffuseCap = uint32_t(value);
  }
  else if (n == "nbr_chunks") {  // diagnostic: size of the lists' second tier (before the first upload; 0 = capacity / 8 + 1024)
    if (ctx->cap) return fail(ctx, PBF_ERR_STATE, "nbr_chunks must be set before the first upload");
    ctx->nbrChunksOpt = uint32_t(value);
  }
  else return fail(ctx, PBF_ERR_INVALID, "unknown option " + n);
  return PBF_OK;
}

int pbf_create(const pbf_desc *desc, pbf_ctx **out) {
  if (!desc || !out) {
    g_create_error = "pbf_create: NULL argument";
    return PBF_ERR_INVALID;
  }
  *out = nullptr;
  if (desc->abi_version != PBF_ABI_VERSION) {
    g_create_error = "pbf_create: abi_version mismatch";
    return PBF_ERR_INVALID;
  }
  if (!(desc->h > 0)) {
    g_create_error = "pbf_create: h must be > 0";
    return PBF_ERR_INVALID;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("pbf_create: no usable HIP device (") + hipGetErrorString(e) +
                     "); this library has no CPU fallback";
    return PBF_ERR_NO_DEVICE;
  }
  if (desc->device < 0 || desc->device >= count) {
    g_create_error = "pbf_create: device ordinal out of range";
    return PBF_ERR_INVALID;
  }
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, desc->device)) != hipSuccess) {
    g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
    return PBF_ERR_HIP;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("pbf_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return PBF_ERR_NO_DEVICE;
  }
  auto *ctx = new pbf_ctx();
  ctx->desc = *desc;
  ctx->device = desc->device;
  ctx->fp64 = desc->fp64 != 0;
  ctx->fast = (desc->flags & PBF_FLAG_FAST_MATH) != 0;
  ctx->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char *e = std::getenv("PBF_TILE_CAP")) ctx->tileCap = uint32_t(std::atoi(e));
  if (const char *e = std::getenv("PBF_GATHER")) {
    const int g = std::atoi(e);
    if (g == 0 || g == 1 || g == 3) ctx->gatherKind = g;  // (2, round 1's brick kernel, was removed)
  }
  if (const char *e = std::getenv("PBF_REUSE_LISTS")) ctx->reuseLists = std::atoi(e) != 0;
  if (const char *e = std::getenv("PBF_SPLIT_BUILD")) ctx->splitBuild = std::atoi(e);
  if (const char *e = std::getenv("PBF_LIST_MAX")) ctx->listMax = uint32_t(std::atoi(e));
  if (const char *e = std::getenv("PBF_OVERLAP_DIFFUSE")) ctx->overlapDiffuse = std::atoi(e) != 0, ctx->overlapDiffuseForced = true;
  if (const char *e = std::getenv("PBF_PIPELINE")) ctx->pipeline = std::atoi(e);
  if (const char *e = std::getenv("PBF_ROW_MAJOR")) ctx->rowMajor = std::atoi(e);
  if (const char *e = std::getenv("PBF_GRAPH")) ctx->graphMode = std::atoi(e);
  if (const char *e = std::getenv("PBF_COOP")) {
    const int v = std::atoi(e);
    if (v == 0 || v == 2 || v == 4 || v == 8) ctx->coop = v;
  }
  if ((e = hipSetDevice(ctx->device)) != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
    delete ctx;
    return PBF_ERR_HIP;
  }
  if (desc->stream) {
    ctx->stream = static_cast<hipStream_t>(desc->stream);
  } else {
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
      g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
      delete ctx;
      return PBF_ERR_HIP;
    }
    ctx->ownStream = true;
  }
  *out = ctx;
  return PBF_OK;
}

void pbf_destroy(pbf_ctx *ctx) {
  if (ctx && ctx->downloadPending) (void)pbf_download_aos_end(ctx);
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &ep : ctx->pending) {
    (void)hipEventDestroy(ep.a);
    (void)hipEventDestroy(ep.b);
  }
  for (auto ev : ctx->freeEvents) (void)hipEventDestroy(ev);
  DevBuf *all[] = {&ctx->pos4[0], &ctx->pos4[1], &ctx->vel4[0], &ctx->vel4[1], &ctx->col4[0],  &ctx->col4[1],
                   &ctx->id[0],   &ctx->id[1],   &ctx->type[0], &ctx->type[1], &ctx->key[0],   &ctx->key[1],
                   &ctx->pstar[0], &ctx->pstar[1], &ctx->pstar[2], &ctx->count, &ctx->table,   &ctx->blockSums,
                   &ctx->permTmp, &ctx->wells,   &ctx->staging, &ctx->bricks, &ctx->brickCtl, &ctx->bigCells,
                   &ctx->latticePN, &ctx->latticeC, &ctx->mcCounts, &ctx->mcOffsets, &ctx->mcSums, &ctx->mcNear, &ctx->meshV, &ctx->meshN,
                   &ctx->meshC, &ctx->qpos, &ctx->nbrList, &ctx->nbrCount, &ctx->rowPstar[0], &ctx->rowPstar[1], &ctx->rowMass, &ctx->rowQpos, &ctx->rowXYZ, &ctx->rowType, &ctx->rowSlotOf, &ctx->rowCol, &ctx->rowMortonOf, &ctx->rowSegs, &ctx->linCount, &ctx->linTable, &ctx->linSums, &ctx->slotOf,  &ctx->selCounts, &ctx->selTotals, &ctx->ghostSrcL, &ctx->ghostSrcR, &ctx->colHist, &ctx->wireSend[0], &ctx->wireSend[1], &ctx->wireRecv[0], &ctx->wireRecv[1], &ctx->wireGhost[0], &ctx->wireGhost[1], &ctx->diffSum, &ctx->diffCnt};
  for (DevBuf *b : all)
    if (b->p) (void)hipFree(b->p);
  for (auto &g : ctx->graphs)
    if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
  if (ctx->evFork) (void)hipEventDestroy(ctx->evFork);
  if (ctx->evJoin) (void)hipEventDestroy(ctx->evJoin);
  if (ctx->sideStream) (void)hipStreamDestroy(ctx->sideStream);
  if (ctx->copyStream) (void)hipStreamDestroy(ctx->copyStream);
  if (ctx->evPacked) (void)hipEventDestroy(ctx->evPacked);
  if (ctx->hostCounts) (void)hipHostFree(ctx->hostCounts);
  if (ctx->meshHost) (void)hipHostFree(ctx->meshHost);
  if (ctx->regPtr) (void)hipHostUnregister(ctx->regPtr);
  if (ctx->ownStream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *pbf_last_error(const pbf_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pbf_upload(pbf_ctx *ctx, size_t n, const uint64_t *id, const uint8_t *type, const void *mass, const void *pos,
               const void *vel, const void *colour) {
  if (!ctx) return PBF_ERR_INVALID;
  if (n && (!id || !type || !mass || !pos || !vel || !colour)) return fail(ctx, PBF_ERR_INVALID, "NULL array");
  if (n >= (size_t(1) << 31)) return fail(ctx, PBF_ERR_INVALID, "n must be < 2^31 (ompsph.hpp:41 iterates with int)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->fp64)
    return upload_impl<double>(ctx, n, id, type, (const double *)mass, (const double *)pos, (const double *)vel,
                               (const double *)colour);
  return upload_impl<float>(ctx, n, id, type, (const float *)mass, (const float *)pos, (const float *)vel,
                            (const float *)colour);
}

int pbf_download(pbf_ctx *ctx, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel, void *colour) {
  if (!ctx) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->fp64)
    return download_impl<double>(ctx, id, type, (double *)mass, (double *)pos, (double *)vel, (double *)colour);
  return download_impl<float>(ctx, id, type, (float *)mass, (float *)pos, (float *)vel, (float *)colour);
}

size_t pbf_count(const pbf_ctx *ctx) { return ctx ? (ctx->ghostsPending ? ctx->nOwned : ctx->n) : 0; }

namespace {
// Page-lock the caller's AoS buffer once and keep the registration while the same buffer comes back every frame
// (benchmark.cpp:33,47 calls advance() on one vector).  Failure is not an error: the copy then takes the pageable path.
void pin_user_buffer(pbf_ctx *ctx, const void *ptr, size_t bytes) {
  if (ctx->regPtr == ptr && ctx->regBytes >= bytes) return;
  if (ctx->regPtr) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipHostUnregister(ctx->regPtr);
    ctx->regPtr = nullptr, ctx->regBytes = 0;
  }
  if (bytes < (1u << 20)) return;  // small scenes: registration costs more than it saves
  if (hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault) == hipSuccess)
    ctx->regPtr = const_cast<void *>(ptr), ctx->regBytes = bytes;
  else
    (void)hipGetLastError();
}
}  // namespace

int pbf_upload_aos(pbf_ctx *ctx, size_t n, const void *particles, const pbf_aos_layout *l) {
  if (ctx && ctx->downloadPending) (void)pbf_download_aos_end(ctx);  // (an open download owns the staging buffer)
  if (!ctx) return PBF_ERR_INVALID;
  if (!l || (n && !particles)) return fail(ctx, PBF_ERR_INVALID, "NULL argument");
  if (n >= (size_t(1) << 31)) return fail(ctx, PBF_ERR_INVALID, "n must be < 2^31");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (int rc = ensure_particles(ctx, n)) return rc;
  if (int rc = drop_histogram(ctx)) return rc;
  ctx->cur = 0, ctx->pcur = 0, ctx->sorted = false, ctx->n = n;
  ctx->ghostsPending = false, ctx->slabActive = false;
  ctx->hasObstacles = false;
  if (n == 0) return PBF_OK;
  if (int rc = ensure(ctx, ctx->staging, n * l->stride)) return rc;
  if (int rc = ensure(ctx, ctx->selTotals, 16)) return rc;
  pin_user_buffer(ctx, particles, n * l->stride);
  HIPCHK(ctx, hipMemcpyAsync(ctx->staging.p, particles, n * l->stride, hipMemcpyHostToDevice, ctx->stream));
  ctx->stagedBytes = n * l->stride;
  // "are there obstacle particles" is answered by the unpack kernel (a strided host scan over the 56-byte structs cost
  // more than the copy itself: 2 of 3 ms at 1 M particles)
  uint32_t *flag = ctx->selTotals.as<uint32_t>() + 3;
  HIPCHK(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
  AosLayout L{l->stride, l->off_id, l->off_type, l->off_mass, l->off_pos, l->off_vel, l->off_colour};
  if (ctx->fp64)
    hipLaunchKernelGGL((k_unpack_aos<double>), grid_for(n), dim3(BLOCK), 0, ctx->stream, uint32_t(n),
                       ctx->staging.as<const uint8_t>(), L, arrays<double>(ctx, 0, 0), flag);
  else
    hipLaunchKernelGGL((k_unpack_aos<float>), grid_for(n), dim3(BLOCK), 0, ctx->stream, uint32_t(n),
                       ctx->staging.as<const uint8_t>(), L, arrays<float>(ctx, 0, 0), flag);
  LAUNCH_CHECK(ctx);
  uint32_t any = 0;
  HIPCHK(ctx, hipMemcpyAsync(&any, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  // the caller owns `particles` and may change or free it as soon as we return: from page-locked memory the copy
  // above is a genuinely asynchronous DMA, so wait for it (the pageable path used to block inside the runtime)
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->hasObstacles = any != 0;
  ctx->realObstacles = ctx->hasObstacles;
  return PBF_OK;
}

// The download in two halves: _begin packs the AoS image on the solver's stream and starts its DMA on a copy stream of
// its own, _end waits for it — so that whatever the caller enqueues in between (the surface kernels) runs WHILE the
// particles travel.  The image is packed before _begin returns control to the stream, so later launches cannot change
// what travels; the uploads (which reuse the staging buffer) and pbf_destroy end an open download themselves.
int pbf_download_aos_begin(pbf_ctx *ctx, void *particles, const pbf_aos_layout *l) {
  if (!ctx) return PBF_ERR_INVALID;
  if (!l || (ctx->n && !particles)) return fail(ctx, PBF_ERR_INVALID, "NULL argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->downloadPending) return fail(ctx, PBF_ERR_STATE, "pbf_download_aos_begin: the previous one has not been ended");
  if (int rc = drop_ghosts(ctx)) return rc;
  const size_t n = ctx->n;
  if (n == 0) return PBF_OK;
  const size_t before = ctx->staging.cap;
  if (int rc = ensure(ctx, ctx->staging, n * l->stride)) return rc;
  if (ctx->staging.cap != before) ctx->stagedBytes = 0;  // (a grown buffer starts undefined)
  // Only the fields are written below.  The struct's padding bytes are no part of the reference's contract (its
  // write-back copies whole structs whose padding is unspecified, ompsph.hpp:479-481): they keep the bytes of the
  // last uploaded image at that slot — all zero / all equal in practice — instead of costing a second full
  // host-to-device copy per frame as in round 1; slots never uploaded are zeroed.
  if (ctx->stagedBytes < n * l->stride) {
    HIPCHK(ctx, hipMemsetAsync(ctx->staging.as<uint8_t>() + ctx->stagedBytes, 0, n * l->stride - ctx->stagedBytes, ctx->stream));
    ctx->stagedBytes = n * l->stride;
  }
  pin_user_buffer(ctx, particles, n * l->stride);
  AosLayout L{l->stride, l->off_id, l->off_type, l->off_mass, l->off_pos, l->off_vel, l->off_colour};
  if (ctx->fp64)
    hipLaunchKernelGGL((k_pack_aos<double>), grid_for(n), dim3(BLOCK), 0, ctx->stream, uint32_t(n),
                       ctx->staging.as<uint8_t>(), L, arrays<double>(ctx, ctx->cur, ctx->pcur));
  else
    hipLaunchKernelGGL((k_pack_aos<float>), grid_for(n), dim3(BLOCK), 0, ctx->stream, uint32_t(n),
                       ctx->staging.as<uint8_t>(), L, arrays<float>(ctx, ctx->cur, ctx->pcur));
  LAUNCH_CHECK(ctx);
  if (!ctx->copyStream) {
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->evPacked, hipEventDisableTiming));
  }
  HIPCHK(ctx, hipEventRecord(ctx->evPacked, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->copyStream, ctx->evPacked, 0));
  HIPCHK(ctx, hipMemcpyAsync(particles, ctx->staging.p, n * l->stride, hipMemcpyDeviceToHost, ctx->copyStream));
  ctx->downloadPending = true;
  return PBF_OK;
}
int pbf_download_aos_end(pbf_ctx *ctx) {
  if (!ctx) return PBF_ERR_INVALID;
  if (!ctx->downloadPending) return PBF_OK;
  ctx->downloadPending = false;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->copyStream));
  return PBF_OK;
}
int pbf_download_aos(pbf_ctx *ctx, void *particles, const pbf_aos_layout *l) {
  if (int rc = pbf_download_aos_begin(ctx, particles, l)) return rc;
  return pbf_download_aos_end(ctx);
}

int pbf_step(pbf_ctx *ctx, const pbf_params *p) {
  if (int rc = check(ctx, p, false)) return rc;
  if (int rc = drop_ghosts(ctx)) return rc;
  return DISPATCH(ctx, step_impl, ctx, p);
}
namespace {

DevBuf *role_buffers(pbf_ctx *ctx, DevBuf *out[15]) {
  DevBuf *all[15] = {&ctx->pos4[0], &ctx->pos4[1], &ctx->vel4[0], &ctx->vel4[1], &ctx->col4[0], &ctx->col4[1],
                     &ctx->id[0],   &ctx->id[1],   &ctx->type[0], &ctx->type[1], &ctx->key[0],  &ctx->key[1],
                     &ctx->pstar[0], &ctx->pstar[1], &ctx->pstar[2]};
  for (int k = 0; k < 15; ++k) out[k] = all[k];
  return nullptr;
}
StepState snapshot(pbf_ctx *ctx) {
  StepState s;
  std::memset(&s, 0, sizeof(s));  // (padding bytes take part in the comparison)
  s.cur = ctx->cur, s.pcur = ctx->pcur;
  s.flags = uint32_t(ctx->sorted) | uint32_t(ctx->counted) << 1 | uint32_t(ctx->nbrValid) << 2 | uint32_t(ctx->qposValid) << 3 |
            uint32_t(ctx->hasObstacles) << 4;
  s.tableN = ctx->tableN, s.countedTableN = ctx->countedTableN, s.gatherSeq = ctx->gatherSeq;
  DevBuf *b[15];
  role_buffers(ctx, b);
  for (int k = 0; k < 3; ++k) s.extent[k] = ctx->extent[k], s.minExtent[k] = ctx->minExtent[k];
  for (int k = 0; k < 15; ++k) s.bufs[k] = b[k]->p, s.caps[k] = b[k]->cap;
  return s;
}
void restore(pbf_ctx *ctx, const StepState &s) {
  ctx->cur = s.cur, ctx->pcur = s.pcur;
  ctx->sorted = s.flags & 1u, ctx->counted = (s.flags >> 1) & 1u, ctx->nbrValid = (s.flags >> 2) & 1u;
  ctx->qposValid = (s.flags >> 3) & 1u, ctx->hasObstacles = (s.flags >> 4) & 1u;
  ctx->tableN = s.tableN, ctx->countedTableN = s.countedTableN, ctx->gatherSeq = s.gatherSeq;
  DevBuf *b[15];
  role_buffers(ctx, b);
  for (int k = 0; k < 3; ++k) ctx->extent[k] = s.extent[k], ctx->minExtent[k] = s.minExtent[k];
  for (int k = 0; k < 15; ++k) b[k]->p = s.bufs[k], b[k]->cap = s.caps[k];
}
GraphKey graph_key(pbf_ctx *ctx, const pbf_params *p) {
  GraphKey k;
  std::memset(&k, 0, sizeof(k));
  k.before = snapshot(ctx);
  double v[11] = {p->dt, p->scale, p->constant_force[0], p->constant_force[1], p->constant_force[2], p->min_bound[0],
                  p->min_bound[1], p->min_bound[2], p->max_bound[0], p->max_bound[1], p->max_bound[2]};
  std::memcpy(k.params, v, sizeof(v));
  const uint64_t it = p->iteration;
  const int32_t xv[2] = {p->xsph, p->vorticity};
  std::memcpy(k.params + sizeof(v), &it, 8);
  std::memcpy(k.params + sizeof(v) + 8, xv, 8);
  k.n = ctx->n;
  const int opt[10] = {ctx->gatherKind, ctx->splitBuild, ctx->coop, ctx->pipeline, int(ctx->cellDiffuse), int(ctx->overlapDiffuse),
                       int(ctx->fuseDiffuse), int(ctx->reuseLists), int(ctx->listMax), int(ctx->tileCap)};
  std::memcpy(k.options, opt, sizeof(opt));
  return k;
}

// One step of pbf_steps: replayed from a captured graph when this exact step (same buffer roles, same parameters) has been
// seen before, captured the first time it comes by, eager whenever capturing is not possible.
int step_maybe_graphed(pbf_ctx *ctx, const pbf_params *p) {
  const bool timing = (ctx->desc.flags & PBF_FLAG_STAGE_TIMING) != 0 && ctx->timingMask != 0;
  if (ctx->graphMode <= 0 || timing || p->n_wells > 0 || ctx->n == 0 || ctx->slabConfigured)
    return DISPATCH(ctx, step_impl, ctx, p);
  const GraphKey key = graph_key(ctx, p);
  auto it = ctx->graphs.find(key);
  if (it != ctx->graphs.end()) {
    HIPCHK(ctx, hipGraphLaunch(it->second.exec, ctx->stream));
    restore(ctx, it->second.after);
    ctx->graphMisses = 0;
    ctx->graphReplays++;
    return PBF_OK;
  }
  if (ctx->graphs.size() >= 32 || ctx->graphMisses >= 8) {  // the steps never repeat (a box that moves every frame)
    ctx->graphMode = 0;
    return DISPATCH(ctx, step_impl, ctx, p);
  }
  // capture only a step that allocates nothing: run it eagerly once more if the last one still grew a buffer
  const uint64_t epoch = ctx->allocEpoch;
  if (ctx->graphCaptures == 0 && ctx->graphReplays == 0 && ctx->graphMisses == 0) {
    ctx->graphMisses = 1;  // the very first step after an upload allocates (scratch, side stream): eager
    return DISPATCH(ctx, step_impl, ctx, p);
  }
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    ctx->graphMode = 0;
    return DISPATCH(ctx, step_impl, ctx, p);
  }
  const int rc = DISPATCH(ctx, step_impl, ctx, p);
  const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
  if (rc != PBF_OK || e != hipSuccess || !graph || ctx->allocEpoch != epoch) {
    // could not be captured (an allocation, a call that is illegal while capturing): nothing ran — redo it eagerly
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    ctx->graphMode = 0;
    restore(ctx, key.before);
    ctx->diffusePending = false;
    return DISPATCH(ctx, step_impl, ctx, p);
  }
  GraphEntry entry;
  const hipError_t ei = hipGraphInstantiate(&entry.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess) {
    (void)hipGetLastError();
    ctx->graphMode = 0;
    restore(ctx, key.before);
    ctx->diffusePending = false;
    return DISPATCH(ctx, step_impl, ctx, p);
  }
  entry.after = snapshot(ctx);
  ctx->graphs.emplace(key, entry);
  ctx->graphCaptures++, ctx->graphMisses++;
  HIPCHK(ctx, hipGraphLaunch(entry.exec, ctx->stream));  // the capture recorded the step; this runs it
  return PBF_OK;
}

}  // namespace

int pbf_steps(pbf_ctx *ctx, const pbf_params *p, uint32_t count) {
  if (int rc = check(ctx, p, false)) return rc;
  if (int rc = drop_ghosts(ctx)) return rc;
  // finalise(t) + predict(t + 1) as one kernel between two steps of THIS call (same parameters, nothing observes the
  // state in between).  Not with stage timing on either entry, hipGraph replay, slabs or a step that may stop early.
  const bool timed = (ctx->desc.flags & PBF_FLAG_STAGE_TIMING) != 0 &&
                     (((ctx->timingMask >> ST_PREDICT) & 1u) != 0 || ((ctx->timingMask >> ST_FINALISE) & 1u) != 0);
  const bool fusable = ctx->fusePredict && !timed && ctx->graphMode <= 0 && !ctx->slabActive && ctx->n != 0;
  for (uint32_t i = 0; i < count; ++i) {
    ctx->fuseNextPredict = fusable && i + 1 < count;
    if (int rc = step_maybe_graphed(ctx, p)) {
      ctx->fuseNextPredict = ctx->prePredicted = false;
      return rc;
    }
  }
  ctx->fuseNextPredict = false;
  return PBF_OK;
}
int pbf_graph_stats(const pbf_ctx *ctx, uint64_t out[3]) {
  if (!ctx || !out) return PBF_ERR_INVALID;
  out[0] = ctx->graphCaptures, out[1] = ctx->graphReplays, out[2] = uint64_t(ctx->graphMode > 0);
  return PBF_OK;
}
int pbf_sync(pbf_ctx *ctx) {
  if (!ctx) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PBF_OK;
}

int pbf_stage_predict(pbf_ctx *ctx, const pbf_params *p) {
  if (int rc = check(ctx, p, false)) return rc;
  if (ctx->n == 0) return PBF_OK;
  return DISPATCH(ctx, stage_predict, ctx, p);
}
int pbf_stage_sort(pbf_ctx *ctx, const pbf_params *p) {
  if (int rc = check(ctx, p, false)) return rc;
  if (ctx->n == 0) return PBF_OK;
  return DISPATCH(ctx, stage_sort, ctx, p);
}
int pbf_stage_diffuse(pbf_ctx *ctx, const pbf_params *p) {
  if (int rc = check(ctx, p, true)) return rc;
  return DISPATCH(ctx, stage_diffuse, ctx, p);
}
int pbf_stage_lambda(pbf_ctx *ctx, const pbf_params *p) {
  if (int rc = check(ctx, p, true)) return rc;
  return DISPATCH(ctx, stage_lambda, ctx, p);
}
int pbf_stage_delta(pbf_ctx *ctx, const pbf_params *p) {
  if (int rc = check(ctx, p, true)) return rc;
  return DISPATCH(ctx, stage_delta, ctx, p);
}
int pbf_stage_finalise(pbf_ctx *ctx, const pbf_params *p) {
  if (int rc = check(ctx, p, true)) return rc;
  return DISPATCH(ctx, stage_finalise, ctx, p);
}

int pbf_read_buffer(pbf_ctx *ctx, int which, void *host, size_t bytes) {
  if (!ctx || !host) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (which != PBF_BUF_NBR_COUNT)  // (the list lengths are indexed like the arrays the last build saw)
    if (int rc = drop_ghosts(ctx)) return rc;
  const void *src = nullptr;
  size_t avail = 0;
  const size_t v = ctx->fp64 ? sizeof(double4) : sizeof(float4);
  switch (which) {
    case PBF_BUF_KEYS: src = ctx->key[ctx->cur].p, avail = ctx->n * 4; break;
    case PBF_BUF_TABLE:
      if (!ctx->sorted) return fail(ctx, PBF_ERR_STATE, "table not built yet");
      src = ctx->table.p, avail = size_t(ctx->tableN) * 4;
      break;
    case PBF_BUF_PSTAR:
      if (int rc = ctx->fp64 ? materialise_pstar<double>(ctx) : materialise_pstar<float>(ctx)) return rc;
      src = ctx->pstar[ctx->pcur].p, avail = ctx->n * v;
      break;
    case PBF_BUF_NBR_COUNT: src = ctx->nbrCount.p, avail = (ctx->ghostsPending ? ctx->nOwned : ctx->n) * 4; break;
    case PBF_BUF_OMEGA:
      if (!ctx->omegaValid) return fail(ctx, PBF_ERR_STATE, "no vorticity pass since the arrays last changed (pbf_params.vorticity)");
      src = ctx->pstar[2].p, avail = ctx->n * v;
      break;
    default: return fail(ctx, PBF_ERR_INVALID, "unknown buffer");
  }
  if (bytes > avail) return fail(ctx, PBF_ERR_INVALID, "read beyond buffer");
  if (!src) return fail(ctx, PBF_ERR_STATE, "buffer not allocated yet");
  HIPCHK(ctx, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PBF_OK;
}

int pbf_selftest_math(pbf_ctx *ctx, uint64_t mismatches[4]) {
  if (!ctx || !mismatches) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (int rc = ensure(ctx, ctx->selTotals, 64)) return rc;
  HIPCHK(ctx, hipMemsetAsync(ctx->selTotals.p, 0, 32, ctx->stream));
  if (ctx->fp64) {  // 1.07e10 pseudo-random operands per category and run (exhaustive is impossible in fp64): 2^20 threads x 10240
    const double h = ctx->desc.h;
    const double p6 = poly6_factor<double>(h), r = double(CorrDeltaQ * h), d = (h * h) - r * r;
    const uint32_t rounds = 10240, blocks = 4096;
    hipLaunchKernelGGL(k_selftest_math64, dim3(blocks), dim3(BLOCK), 0, ctx->stream, ctx->selTotals.as<unsigned long long>(),
                       p6 * (d * d * d), double(RHO), h, rounds);
    LAUNCH_CHECK(ctx);
    unsigned long long hst[4] = {0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(hst, ctx->selTotals.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; ++k) mismatches[k] = hst[k];
    return PBF_OK;
  }
  // the two per-launch constant divisors of delta-p for this context's h: poly6(0.3 h) and the reference density
  const float h = float(ctx->desc.h);
  const float p6 = poly6_factor<float>(h), r = float(CorrDeltaQ * h), d = (h * h) - r * r;
  const float p6DeltaQ = p6 * (d * d * d);
  hipLaunchKernelGGL(k_selftest_math, dim3(uint32_t(ctx->numCUs) * 8), dim3(BLOCK), 0, ctx->stream,
                     ctx->selTotals.as<unsigned long long>(), p6DeltaQ, float(RHO), h);
  LAUNCH_CHECK(ctx);
  unsigned long long hst[4] = {0, 0, 0, 0};
  HIPCHK(ctx, hipMemcpyAsync(hst, ctx->selTotals.p, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 4; ++k) mismatches[k] = hst[k];
  return PBF_OK;
}

size_t pbf_table_size(const pbf_ctx *ctx) { return ctx ? ctx->tableN : 0; }
int pbf_grid_extent(const pbf_ctx *ctx, uint64_t extent[3], double min_extent[3]) {
  if (!ctx) return PBF_ERR_INVALID;
  for (int i = 0; i < 3; ++i) {
    if (extent) extent[i] = ctx->extent[i];
    if (min_extent) min_extent[i] = ctx->minExtent[i];
  }
  return PBF_OK;
}

int pbf_stage_times(pbf_ctx *ctx, const char **names, double *mean_ms, uint64_t *calls, int cap) {
  if (!ctx) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (int rc = resolve_events(ctx)) return rc;
  int k = 0;
  for (; k < ST_COUNT && k < cap; ++k) {
    if (names) names[k] = kStageNames[k];
    if (mean_ms) mean_ms[k] = ctx->stageCalls[k] ? ctx->stageMs[k] / double(ctx->stageCalls[k]) : 0.0;
    if (calls) calls[k] = ctx->stageCalls[k];
  }
  return k;
}
int pbf_reset_stage_times(pbf_ctx *ctx) {
  if (!ctx) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (int rc = resolve_events(ctx)) return rc;
  for (int k = 0; k < ST_COUNT; ++k) ctx->stageMs[k] = 0, ctx->stageCalls[k] = 0;
  return PBF_OK;
}

}  // extern "C"

// ================================================================================================
// Slab decomposition entry points (include/pbf_hip.h "multi-GPU").  The caller (one process per
// GPU) owns the wire buffers and moves them with RCCL (torch.distributed "nccl" send/recv over
// xGMI); this library only selects, packs, appends and unpacks on the device.
// ================================================================================================
namespace {

template <typename N, int MODE>
int run_select(pbf_ctx *ctx, const pbf_slab_cut *cut, void *sendL, void *sendR, uint32_t cap, uint32_t totals[3],
               bool readback = true) {
  const uint32_t n = uint32_t(ctx->n);
  const uint32_t nb = (n + SEL_TILE - 1) / SEL_TILE;
  if (int rc = ensure(ctx, ctx->selCounts, size_t(3) * std::max(nb, 1u) * 4)) return rc;
  if (int rc = ensure(ctx, ctx->selTotals, 16)) return rc;
  if (int rc = ensure(ctx, ctx->ghostSrcL, size_t(std::max(cap, 1u)) * 4)) return rc;
  if (int rc = ensure(ctx, ctx->ghostSrcR, size_t(std::max(cap, 1u)) * 4)) return rc;
  totals[0] = totals[1] = totals[2] = 0;
  if (n == 0) return PBF_OK;
  const uint32_t xo = ctx->slabConfigured ? ctx->xoff : 0u;  // the keys' x frame
  SlabCut s{cut->xlo - std::min(cut->xlo, xo), cut->xhi == 0xFFFFFFFFu ? cut->xhi : cut->xhi - std::min(cut->xhi, xo),
            cut->has_left ? 1u : 0u, cut->has_right ? 1u : 0u};
  const int a = ctx->cur, b = 1 - a;
  uint32_t *counts = ctx->selCounts.as<uint32_t>(), *tot = ctx->selTotals.as<uint32_t>();
  hipLaunchKernelGGL((k_sel_count<MODE>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n, s, ctx->key[a].as<const uint32_t>(),
                     ctx->type[a].as<const uint8_t>(), nb, counts);
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(BLOCK), 0, ctx->stream, nb, counts, tot);
  hipLaunchKernelGGL((k_sel_emit<N, MODE>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n, s, arrays<N>(ctx, a, ctx->pcur),
                     arrays<N>(ctx, b, b), nb, counts, sendL, sendR, cap, ctx->ghostSrcL.as<uint32_t>(),
                     ctx->ghostSrcR.as<uint32_t>());
  LAUNCH_CHECK(ctx);
  if (!readback) return PBF_OK;  // (pbf_slab_step reads selTotals back together with the received headers)
  HIPCHK(ctx, hipMemcpyAsync(totals, tot, 12, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the caller needs the counts to size its sends
  return PBF_OK;
}

template <typename N> int slab_migrate(pbf_ctx *ctx, const pbf_slab_cut *cut, void *sL, void *sR, uint32_t cap, uint32_t out[2]) {
  if (int rc = drop_histogram(ctx)) return rc;  // predict's histogram describes the pre-migration set
  uint32_t t[3];
  if (int rc = run_select<N, SEL_MIGRATE>(ctx, cut, sL, sR, cap, t)) return rc;
  if (t[1] > cap || t[2] > cap) return fail(ctx, PBF_ERR_INVALID, "migrant buffer too small");
  if (ctx->n) {
    ctx->cur = 1 - ctx->cur;  // the keeps were compacted into the other array set
    ctx->pcur = ctx->cur;
  }
  ctx->n = t[0];
  ctx->nOwned = t[0];
  ctx->sorted = false;
  out[0] = t[1], out[1] = t[2];
  return PBF_OK;
}

template <typename N> int slab_add_migrants(pbf_ctx *ctx, const void *rL, uint32_t nL, const void *rR, uint32_t nR) {
  if (ctx->n + nL + nR > ctx->cap) return fail(ctx, PBF_ERR_INVALID, "particle capacity exceeded: call pbf_reserve");
  if (nL + nR)
    hipLaunchKernelGGL((k_append_migrants<N>), grid_for(nL + nR), dim3(BLOCK), 0, ctx->stream, uint32_t(ctx->n),
                       static_cast<const MigrantRec<N> *>(rL), nL, static_cast<const MigrantRec<N> *>(rR), nR,
                       ctx->shiftL, ctx->shiftR, arrays<N>(ctx, ctx->cur, ctx->pcur));
  LAUNCH_CHECK(ctx);
  ctx->n += nL + nR;
  ctx->nOwned = uint32_t(ctx->n);
  return PBF_OK;
}

template <typename N> int slab_ghosts(pbf_ctx *ctx, const pbf_slab_cut *cut, void *sL, void *sR, uint32_t cap, uint32_t out[2]) {
  uint32_t t[3];
  if (int rc = run_select<N, SEL_GHOST>(ctx, cut, sL, sR, cap, t)) return rc;
  if (t[0] > cap || t[1] > cap) return fail(ctx, PBF_ERR_INVALID, "ghost buffer too small");
  ctx->sentL = t[0], ctx->sentR = t[1];
  out[0] = t[0], out[1] = t[1];
  return PBF_OK;
}

template <typename N> int slab_add_ghosts(pbf_ctx *ctx, const void *rL, uint32_t nL, const void *rR, uint32_t nR) {
  if (ctx->n + nL + nR > ctx->cap) return fail(ctx, PBF_ERR_INVALID, "particle capacity exceeded: call pbf_reserve");
  if (nL + nR)
    hipLaunchKernelGGL((k_append_ghosts<N>), grid_for(nL + nR), dim3(BLOCK), 0, ctx->stream, uint32_t(ctx->n),
                       static_cast<const GhostRec<N> *>(rL), nL, static_cast<const GhostRec<N> *>(rR), nR,
                       ctx->shiftL, ctx->shiftR, arrays<N>(ctx, ctx->cur, ctx->pcur));
  ctx->gotL = nL, ctx->gotR = nR;
  ctx->ghostAt = uint32_t(ctx->n);
  ctx->n += nL + nR;
  if (nL + nR) ctx->hasObstacles = true;  // "special" particles exist: the kernels must look at type[]
  // histogram of the re-assembled set (owned + copies) for the sort
  if (ctx->n)
    hipLaunchKernelGGL(k_count_keys, grid_for(ctx->n), dim3(BLOCK), 0, ctx->stream, uint32_t(ctx->n), ctx->tableN,
                       ctx->key[ctx->cur].as<const uint32_t>(), ctx->count.as<uint32_t>());
  LAUNCH_CHECK(ctx);
  ctx->counted = true;
  ctx->countedTableN = ctx->tableN;
  ctx->slabActive = true;
  return PBF_OK;
}

template <typename N> int slab_pack(pbf_ctx *ctx, void *sL, void *sR, const void *field = nullptr) {
  const uint32_t m = ctx->sentL + ctx->sentR;
  // {pStar, lambda}: from wherever it currently lives — the row-major copy while the iterations run on it
  const bool rows = !field && ctx->pstarInRows && ctx->rowsCurrent;
  if (!field && ctx->pstarInRows && !rows) return fail(ctx, PBF_ERR_STATE, "slab pack: no current pStar");
  const vec4<N> *src = field ? static_cast<const vec4<N> *>(field)
                             : rows ? ctx->rowPstar[ctx->rcur].as<const vec4<N>>() : ctx->pstar[ctx->pcur].as<const vec4<N>>();
  if (m)
    hipLaunchKernelGGL((k_pack_field<N>), grid_for(m), dim3(BLOCK), 0, ctx->stream, ctx->sentL, ctx->sentR,
                       ctx->ghostSrcL.as<const uint32_t>(), ctx->ghostSrcR.as<const uint32_t>(),
                       ctx->slotOf.as<const uint32_t>(), src,
                       static_cast<vec4<N> *>(sL), static_cast<vec4<N> *>(sR), rows ? ctx->rowSlotOf.as<const uint32_t>() : nullptr);
  LAUNCH_CHECK(ctx);
  return PBF_OK;
}

template <typename N> int slab_unpack(pbf_ctx *ctx, const void *rL, const void *rR, void *field = nullptr) {
  const uint32_t m = ctx->gotL + ctx->gotR;
  StepConsts<N> c;
  if (!ctx->haveParams) return fail(ctx, PBF_ERR_STATE, "pbf_slab_unpack before any stage");
  if (int rc = make_consts<N>(ctx, &ctx->lastParams, c)) return rc;
  const bool rows = !field && ctx->pstarInRows && ctx->rowsCurrent;
  if (m)
    hipLaunchKernelGGL((k_unpack_field<N>), grid_for(m), dim3(BLOCK), 0, ctx->stream, c, ctx->ghostAt, ctx->gotL, ctx->gotR,
                       static_cast<const vec4<N> *>(rL), static_cast<const vec4<N> *>(rR),
                       ctx->slotOf.as<const uint32_t>(),
                       field ? static_cast<vec4<N> *>(field) : rows ? ctx->rowPstar[ctx->rcur].as<vec4<N>>() : ctx->pstar[ctx->pcur].as<vec4<N>>(),
                       field ? nullptr : rows ? ctx->rowQpos.as<uint2>() : ctx->qpos.as<uint2>(),
                       rows ? ctx->rowSlotOf.as<const uint32_t>() : nullptr);
  LAUNCH_CHECK(ctx);
  return PBF_OK;
}

template <typename N> int slab_finish(pbf_ctx *ctx, bool knownOwned = false) {
  // drop the copies: the MIGRATE select with no neighbours keeps exactly the non-ghost particles
  pbf_slab_cut none{0, 0xFFFFFFFFu, 0, 0};
  uint32_t t[3];
  const bool wasSorted = ctx->sorted;
  // (pbf_slab_step knows the count — every non-copy is owned — and skips the synchronising read-back)
  if (int rc = run_select<N, SEL_MIGRATE>(ctx, &none, nullptr, nullptr, 0, t, !knownOwned)) return rc;
  if (knownOwned) t[0] = ctx->n ? ctx->nOwned : 0;
  if (ctx->n) {
    ctx->cur = 1 - ctx->cur;
    ctx->pcur = ctx->cur;
  }
  ctx->n = t[0];
  ctx->nOwned = t[0];
  ctx->sorted = false;
  (void)wasSorted;
  ctx->hasObstacles = ctx->realObstacles;
  ctx->slabActive = false;
  ctx->sentL = ctx->sentR = ctx->gotL = ctx->gotR = 0;
  return PBF_OK;
}

int slab_check(pbf_ctx *ctx) {
  if (!ctx) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return PBF_OK;
}

int drop_ghosts(pbf_ctx *ctx) {
  if (!ctx->ghostsPending) return PBF_OK;
  ctx->ghostsPending = false;
  return DISPATCH(ctx, slab_finish, ctx, /*knownOwned=*/true);
}

}  // namespace

extern "C" {

int pbf_reserve(pbf_ctx *ctx, size_t capacity) {
  if (!ctx) return PBF_ERR_INVALID;
  if (ctx->n && capacity > ctx->cap) return fail(ctx, PBF_ERR_STATE, "pbf_reserve must precede pbf_upload");
  ctx->reserve = capacity;
  return PBF_OK;
}
int pbf_slab_configure(pbf_ctx *ctx, const pbf_slab_cut *cut, uint32_t left_xlo, uint32_t right_xlo) {
  if (!ctx) return PBF_ERR_INVALID;
  if (!cut) {  // back to the global x frame
    ctx->slabConfigured = false, ctx->xoff = 0, ctx->shiftL = ctx->shiftR = 0;
    return PBF_OK;
  }
  if (cut->xhi <= cut->xlo) return fail(ctx, PBF_ERR_INVALID, "empty slab");
  // frame origin = PBF_SLAB_FRAME_MARGIN columns left of the first owned column (the left ghost column is the one
  // next to it); rank 0 keeps the padding columns.  The margin keeps local x coordinates non-negative — a negative one
  // would wrap in the 10-bit Morton field and send the particle to the wrong neighbour — for every particle this rank
  // can hold at predict time: owned ones that moved left (< 2 columns per step) and, after a re-cut, the owners of
  // columns just handed to the left neighbour (a cut moves by <= 2 columns).
  auto origin = [](uint32_t xlo, bool hasLeft) {
    return hasLeft && xlo > 0 ? xlo - std::min<uint32_t>(xlo, PBF_SLAB_FRAME_MARGIN) : 0u;
  };
  ctx->slabCut = *cut;
  ctx->xoff = origin(cut->xlo, cut->has_left != 0);
  // a neighbour's records arrive keyed in ITS frame: x_mine = x_theirs + their_origin - my_origin
  ctx->shiftL = int32_t(origin(left_xlo, left_xlo > 0)) - int32_t(ctx->xoff);
  ctx->shiftR = int32_t(origin(right_xlo, true)) - int32_t(ctx->xoff);
  ctx->slabConfigured = true;
  ctx->sorted = false;
  return PBF_OK;
}
size_t pbf_slab_record_bytes(const pbf_ctx *ctx, int kind) {
  if (!ctx) return 0;
  switch (kind) {
    case PBF_REC_MIGRANT: return ctx->fp64 ? sizeof(MigrantRec<double>) : sizeof(MigrantRec<float>);
    case PBF_REC_GHOST: return ctx->fp64 ? sizeof(GhostRec<double>) : sizeof(GhostRec<float>);
    case PBF_REC_FIELD: return ctx->fp64 ? sizeof(double4) : sizeof(float4);
    default: return 0;
  }
}
int pbf_slab_migrate(pbf_ctx *ctx, const pbf_slab_cut *cut, void *send_left, void *send_right, uint32_t cap_records,
                     uint32_t counts[2]) {
  if (int rc = slab_check(ctx)) return rc;
  if (!cut || !counts) return fail(ctx, PBF_ERR_INVALID, "NULL argument");
  return DISPATCH(ctx, slab_migrate, ctx, cut, send_left, send_right, cap_records, counts);
}
int pbf_slab_add_migrants(pbf_ctx *ctx, const void *recv_left, uint32_t n_left, const void *recv_right, uint32_t n_right) {
  if (int rc = slab_check(ctx)) return rc;
  return DISPATCH(ctx, slab_add_migrants, ctx, recv_left, n_left, recv_right, n_right);
}
int pbf_slab_ghosts(pbf_ctx *ctx, const pbf_slab_cut *cut, void *send_left, void *send_right, uint32_t cap_records,
                    uint32_t counts[2]) {
  if (int rc = slab_check(ctx)) return rc;
  if (!cut || !counts) return fail(ctx, PBF_ERR_INVALID, "NULL argument");
  return DISPATCH(ctx, slab_ghosts, ctx, cut, send_left, send_right, cap_records, counts);
}
int pbf_slab_add_ghosts(pbf_ctx *ctx, const void *recv_left, uint32_t n_left, const void *recv_right, uint32_t n_right) {
  if (int rc = slab_check(ctx)) return rc;
  return DISPATCH(ctx, slab_add_ghosts, ctx, recv_left, n_left, recv_right, n_right);
}
int pbf_slab_pack(pbf_ctx *ctx, void *send_left, void *send_right) {
  if (int rc = slab_check(ctx)) return rc;
  if (!ctx->sorted) return fail(ctx, PBF_ERR_STATE, "pbf_slab_pack needs pbf_stage_sort first");
  return DISPATCH(ctx, slab_pack, ctx, send_left, send_right);
}
int pbf_slab_unpack(pbf_ctx *ctx, const void *recv_left, const void *recv_right) {
  if (int rc = slab_check(ctx)) return rc;
  if (!ctx->sorted) return fail(ctx, PBF_ERR_STATE, "pbf_slab_unpack needs pbf_stage_sort first");
  return DISPATCH(ctx, slab_unpack, ctx, recv_left, recv_right);
}
int pbf_slab_finish(pbf_ctx *ctx) {
  if (int rc = slab_check(ctx)) return rc;
  ctx->ghostsPending = false;
  return DISPATCH(ctx, slab_finish, ctx);
}
size_t pbf_owned_count(const pbf_ctx *ctx) { return ctx ? (ctx->slabActive ? ctx->nOwned : ctx->n) : 0; }

int pbf_slab_column_histogram(pbf_ctx *ctx, uint32_t out[1024]) {
  if (int rc = slab_check(ctx)) return rc;
  if (!out) return fail(ctx, PBF_ERR_INVALID, "NULL argument");
  if (int rc = ensure(ctx, ctx->selTotals, 16)) return rc;
  if (int rc = ensure(ctx, ctx->colHist, 1024 * 4)) return rc;
  HIPCHK(ctx, hipMemsetAsync(ctx->colHist.p, 0, 1024 * 4, ctx->stream));
  if (ctx->n) {
    const uint32_t blocks = uint32_t(std::min<size_t>((ctx->n + BLOCK - 1) / BLOCK, size_t(ctx->numCUs) * 4));
    hipLaunchKernelGGL(k_column_histogram, dim3(blocks), dim3(BLOCK), 0, ctx->stream, uint32_t(ctx->n),
                       ctx->slabConfigured ? ctx->xoff : 0u, ctx->key[ctx->cur].as<const uint32_t>(),
                       ctx->type[ctx->cur].as<const uint8_t>(), ctx->colHist.as<uint32_t>());
    LAUNCH_CHECK(ctx);
  }
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->colHist.p, 1024 * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PBF_OK;
}

}  // extern "C"

// ================================================================================================
// Communicator + the whole slab step inside the library (include/pbf_hip.h "communicator")
// ================================================================================================
namespace {

thread_local std::string g_comm_error;

pbf_slab_cut cut_of(const pbf_ctx *ctx) {
  const int r = ctx->comm->rank, n = ctx->comm->nranks;
  return pbf_slab_cut{ctx->cuts[r], ctx->cuts[r + 1], r > 0 ? 1 : 0, r + 1 < n ? 1 : 0};
}

int apply_cuts(pbf_ctx *ctx) {
  const int r = ctx->comm->rank, n = ctx->comm->nranks;
  if (n == 1) return pbf_slab_configure(ctx, nullptr, 0, 0);
  const pbf_slab_cut c = cut_of(ctx);
  return pbf_slab_configure(ctx, &c, r > 0 ? ctx->cuts[r - 1] : 0, r + 1 < n ? ctx->cuts[r + 1] : 0);
}

int exchange(pbf_ctx *ctx, size_t nSL, size_t nSR, size_t nRL, size_t nRR, size_t offset = 0) {
  const int rc = comm_exchange(ctx->comm, ctx->stream, ctx->wireSend[0].as<uint8_t>() + offset, nSL,
                               ctx->wireSend[1].as<uint8_t>() + offset, nSR, ctx->wireRecv[0].as<uint8_t>() + offset, nRL,
                               ctx->wireRecv[1].as<uint8_t>() + offset, nRR);
  if (rc) ctx->err = "slab exchange: " + ctx->comm->err;
  return rc;
}

// The opt-in extras in slab mode: the same three gather ops as extras_impl, with the owners refreshing their copies'
// velocity / vorticity before each op that reads them (the copies' velocities are not part of the ghost records).
template <typename N, bool FAST> int slab_extras_impl(pbf_ctx *ctx, const pbf_params *p) {
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  const uint8_t *type = ctx->type[ctx->cur].as<const uint8_t>();
  const vec4<N> *ps = ctx->pstar[ctx->cur].as<const vec4<N>>();
  const size_t fb = sizeof(vec4<N>);
  auto refresh = [&](void *field) -> int {
    if (int rc = slab_pack<N>(ctx, ctx->wireSend[0].p, ctx->wireSend[1].p, field)) return rc;
    if (int rc = exchange(ctx, ctx->sentL * fb, ctx->sentR * fb, ctx->gotL * fb, ctx->gotR * fb)) return rc;
    return slab_unpack<N>(ctx, ctx->wireRecv[0].p, ctx->wireRecv[1].p, field);
  };
  auto vel = [&](int k) { return ctx->vel4[k].as<vec4<N>>(); };
  int s = ctx->cur, o = 1 - s;
  if (int rc = refresh(vel(s))) return rc;
  if (p->vorticity) {
    vec4<N> *omega = ctx->pstar[2].as<vec4<N>>();
    typename VorticityOp<N, FAST>::Args a1{ps, vel(s), omega, type};
    if (int rc = launch_gather<N, VorticityOp<N, FAST>>(ctx, c, a1)) return rc;
    if (int rc = refresh(omega)) return rc;
    typename VorticityForceOp<N, FAST>::Args a2{ps, omega, vel(s), vel(o), type};
    if (int rc = launch_gather<N, VorticityForceOp<N, FAST>>(ctx, c, a2)) return rc;
    std::swap(ctx->vel4[s], ctx->vel4[o]);
    if (p->xsph)
      if (int rc = refresh(vel(s))) return rc;
  }
  if (p->xsph) {
    typename XsphOp<N, FAST>::Args a3{ps, vel(s), vel(o), type};
    if (int rc = launch_gather<N, XsphOp<N, FAST>>(ctx, c, a3)) return rc;
    std::swap(ctx->vel4[s], ctx->vel4[o]);
  }
  return PBF_OK;
}

// Spin on the pinned word k_slab_counts writes last.  A kernel fault or a lost device would leave it unwritten for ever:
// every ~2 ms of spinning the stream is queried, and an error (or an idle stream without the word) ends the wait.
int wait_for_counts(pbf_ctx *ctx, uint32_t seq) {
  volatile uint32_t *h = ctx->hostCounts;
  for (uint64_t spin = 1;; ++spin) {
    if (h[7] == seq) return PBF_OK;
    if ((spin & 0xFFFFu) == 0) {
      const hipError_t e = hipStreamQuery(ctx->stream);
      if (e == hipSuccess) {
        if (h[7] == seq) return PBF_OK;
        return fail(ctx, PBF_ERR_HIP, "slab read-back: the stream went idle without delivering the counts");
      }
      if (e != hipErrorNotReady) {
        ctx->err = std::string("slab read-back: ") + hipGetErrorString(e);
        return PBF_ERR_HIP;
      }
    }
  }
}

// One step of the slab protocol, everything on the solver's stream (round 3: ONE select pass, no compaction, no
// re-histogram, finalise + predict fused between the steps of one pbf_slab_steps call):
//   predict (classifies: stayers into the histogram, leavers and last step's copies not)
//   select: leavers -> migrant wire (their slots die), copies of the boundary stayers -> ghost wire       [3 small kernels]
//   [migrants]  exchange, read-back #1 (how many arrived), append + histogram, copies of boundary arrivals -> ghost wire
//   [copies]    exchange, read-back #2 (how many copies each way), append + histogram
//   sort (skips the dead slots: this is where the leavers and the old copies disappear) -> diffuse -> K x { lambda ->
//   [field] -> delta -> [field] } -> finalise (+ next predict)
template <typename N> int slab_step_impl(pbf_ctx *ctx, const pbf_params *p) {
  struct ModeGuard {
    pbf_ctx *c;
    ~ModeGuard() { c->slabStepMode = false; }
  } guard{ctx};
  ctx->slabStepMode = true;
  if (ctx->prePredicted) {
    ctx->prePredicted = false;  // the previous step of this pbf_slab_steps call has predicted already (k_finalise_predict)
  } else if (int rc = stage_predict<N>(ctx, p)) {
    return rc;
  }
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  const uint32_t nPrev = uint32_t(ctx->n), oldCopies = ctx->ghostsPending ? ctx->gotL + ctx->gotR : 0u;
  const pbf_slab_cut cut = cut_of(ctx);
  const SlabCut sc{c.sxlo, c.sxhi, c.sHasL, c.sHasR};
  (void)cut;
  uint8_t *mL = ctx->wireSend[0].as<uint8_t>(), *mR = ctx->wireSend[1].as<uint8_t>();
  uint8_t *gL = ctx->wireGhost[0].as<uint8_t>(), *gR = ctx->wireGhost[1].as<uint8_t>();
  uint8_t *rL = ctx->wireRecv[0].as<uint8_t>(), *rR = ctx->wireRecv[1].as<uint8_t>();
  const int a = ctx->cur;
  const uint32_t nb = std::max(1u, (nPrev + SEL_TILE - 1) / SEL_TILE);
  if (int rc = ensure(ctx, ctx->selCounts, size_t(4) * nb * 4)) return rc;
  if (int rc = ensure(ctx, ctx->selTotals, 16)) return rc;
  if (int rc = ensure(ctx, ctx->ghostSrcL, size_t(ctx->wireCap) * 4)) return rc;
  if (int rc = ensure(ctx, ctx->ghostSrcR, size_t(ctx->wireCap) * 4)) return rc;
  uint32_t *counts = ctx->selCounts.as<uint32_t>(), *tot = ctx->selTotals.as<uint32_t>();
  volatile uint32_t *h = ctx->hostCounts;
  // ---- the select: one pass for both rounds --------------------------------------------------------------------
  hipLaunchKernelGGL(k_slab_count, dim3(nb), dim3(BLOCK), 0, ctx->stream, nPrev, sc, ctx->key[a].as<const uint32_t>(),
                     ctx->type[a].as<const uint8_t>(), nb, counts);
  hipLaunchKernelGGL(k_slab_scan, dim3(1), dim3(BLOCK), 0, ctx->stream, nb, counts, tot, reinterpret_cast<uint32_t *>(mL),
                     reinterpret_cast<uint32_t *>(mR), reinterpret_cast<uint32_t *>(gL), reinterpret_cast<uint32_t *>(gR),
                     reinterpret_cast<uint32_t *>(rL), reinterpret_cast<uint32_t *>(rR));
  hipLaunchKernelGGL((k_slab_emit<N>), dim3(nb), dim3(BLOCK), 0, ctx->stream, nPrev, sc, arrays<N>(ctx, a, ctx->pcur), nb, counts,
                     reinterpret_cast<MigrantRec<N> *>(mL + WIRE_HDR), reinterpret_cast<MigrantRec<N> *>(mR + WIRE_HDR),
                     reinterpret_cast<GhostRec<N> *>(gL + WIRE_HDR), reinterpret_cast<GhostRec<N> *>(gR + WIRE_HDR), ctx->wireCap,
                     ctx->ghostSrcL.as<uint32_t>(), ctx->ghostSrcR.as<uint32_t>());
  LAUNCH_CHECK(ctx);
  // an exchange round: fixed-size first message {header | first `chunk` records}; the rest, when a side holds more
  // (a re-cut hands whole columns over), in a second, exactly sized exchange once both ends know the counts
  auto round = [&](uint8_t *sL, uint8_t *sR, uint32_t chunk, size_t recBytes, int idxL, int idxR, uint32_t got[2]) -> int {
    const size_t first = WIRE_HDR + size_t(chunk) * recBytes;
    if (int rc = comm_exchange(ctx->comm, ctx->stream, sL, first, sR, first, rL, first, rR, first)) {
      ctx->err = "slab exchange: " + ctx->comm->err;
      return rc;
    }
    const uint32_t seq = ++ctx->slabSeq;
    hipLaunchKernelGGL(k_slab_counts, dim3(1), dim3(64), 0, ctx->stream, tot, reinterpret_cast<uint32_t *>(rL),
                       reinterpret_cast<uint32_t *>(rR), h, seq);
    LAUNCH_CHECK(ctx);
    // the one read-back of this round (the counts size the launches below): the host POLLS the pinned sequence word the
    // kernel writes last — a few microseconds after the kernel retires, where hipStreamSynchronize took 15-30
    if (int rc = wait_for_counts(ctx, seq)) return rc;
    ctx->slabHostSyncs++;
    got[0] = h[4], got[1] = h[5];
    const uint32_t ownL = h[idxL], ownR = h[idxR];
    const uint32_t most = std::max(std::max(ownL, ownR), std::max(got[0], got[1]));
    if (most > ctx->wireCap)
      return fail(ctx, PBF_ERR_COMM, "slab wire buffers too small (" + std::to_string(most) + " records > " +
                                         std::to_string(ctx->wireCap) + "): pbf_reserve a larger particle capacity before attaching");
    auto rest = [&](uint32_t count) { return count > chunk ? size_t(count - chunk) * recBytes : size_t(0); };
    if (most > chunk)
      if (int rc = comm_exchange(ctx->comm, ctx->stream, sL + first, rest(ownL), sR + first, rest(ownR), rL + first,
                                 rest(got[0]), rR + first, rest(got[1]))) {
        ctx->err = "slab exchange: " + ctx->comm->err;
        return rc;
      }
    return PBF_OK;
  };
  // ---- round 1: particles whose cell column left the slab move to the neighbour -------------------------------
  uint32_t got[2];
  if (int rc = round(mL, mR, ctx->capMig, sizeof(MigrantRec<N>), 0, 1, got)) return rc;
  const uint32_t leavers = h[0] + h[1], arrived = got[0] + got[1];
  if (size_t(nPrev) + arrived > ctx->cap) return fail(ctx, PBF_ERR_INVALID, "particle capacity exceeded: call pbf_reserve");
  if (arrived) {
    hipLaunchKernelGGL((k_append_migrants_h<N>), grid_for(arrived), dim3(BLOCK), 0, ctx->stream, nPrev,
                       reinterpret_cast<const MigrantRec<N> *>(rL + WIRE_HDR), got[0],
                       reinterpret_cast<const MigrantRec<N> *>(rR + WIRE_HDR), got[1], ctx->shiftL, ctx->shiftR,
                       arrays<N>(ctx, a, ctx->pcur), c.tableN, ctx->count.as<uint32_t>());
    hipLaunchKernelGGL((k_slab_arrival_ghosts<N>), dim3(1), dim3(BLOCK), 0, ctx->stream, nPrev, arrived, sc,
                       arrays<N>(ctx, a, ctx->pcur), tot, reinterpret_cast<GhostRec<N> *>(gL + WIRE_HDR),
                       reinterpret_cast<GhostRec<N> *>(gR + WIRE_HDR), reinterpret_cast<uint32_t *>(gL),
                       reinterpret_cast<uint32_t *>(gR), ctx->wireCap, ctx->ghostSrcL.as<uint32_t>(), ctx->ghostSrcR.as<uint32_t>());
    LAUNCH_CHECK(ctx);
  }
  // ---- round 2: copies of the boundary columns -------------------------------------------------------------------
  if (int rc = round(gL, gR, ctx->capGhost, sizeof(GhostRec<N>), 2, 3, got)) return rc;
  ctx->sentL = h[2], ctx->sentR = h[3];
  const uint32_t copies = got[0] + got[1];
  if (size_t(nPrev) + arrived + copies > ctx->cap) return fail(ctx, PBF_ERR_INVALID, "particle capacity exceeded: call pbf_reserve");
  ctx->ghostAt = nPrev + arrived;
  if (copies) {
    hipLaunchKernelGGL((k_append_ghosts_h<N>), grid_for(copies), dim3(BLOCK), 0, ctx->stream, ctx->ghostAt,
                       reinterpret_cast<const GhostRec<N> *>(rL + WIRE_HDR), got[0],
                       reinterpret_cast<const GhostRec<N> *>(rR + WIRE_HDR), got[1], ctx->shiftL, ctx->shiftR,
                       arrays<N>(ctx, a, ctx->pcur), c.tableN, ctx->count.as<uint32_t>());
    LAUNCH_CHECK(ctx);
    ctx->hasObstacles = true;  // "special" particles exist: the kernels must look at type[]
  }
  ctx->gotL = got[0], ctx->gotR = got[1];
  ctx->n = size_t(nPrev) + arrived + copies;                      // slots, dead ones included
  ctx->sortLive = ctx->n - oldCopies - leavers;                   // what the sort keeps
  ctx->nOwned = uint32_t(ctx->sortLive - copies);
  ctx->slabActive = true;
  ctx->counted = true, ctx->countedTableN = c.tableN;
  if (int rc = stage_sort<N>(ctx, p)) return rc;
  if (int rc = stage_diffuse<N>(ctx, p, /*overlap=*/p->iteration > 0)) return rc;  // beside the iterations, like pbf_step
  // ---- K x { lambda, delta-p }, each followed by the owners refreshing their copies' {pStar, lambda} ------
  const size_t fb = sizeof(vec4<N>);
  auto refresh = [&]() -> int {
    if (int rc = slab_pack<N>(ctx, ctx->wireSend[0].p, ctx->wireSend[1].p)) return rc;
    if (int rc = exchange(ctx, ctx->sentL * fb, ctx->sentR * fb, ctx->gotL * fb, ctx->gotR * fb)) return rc;
    return slab_unpack<N>(ctx, ctx->wireRecv[0].p, ctx->wireRecv[1].p);
  };
  for (uint64_t it = 0; it < p->iteration; ++it) {
    if (int rc = stage_lambda<N>(ctx, p)) return rc;
    if (int rc = refresh()) return rc;
    if (int rc = stage_delta<N>(ctx, p)) return rc;
    if (int rc = refresh()) return rc;
  }
  if (int rc = stage_finalise<N>(ctx, p)) return rc;
  if (p->vorticity || p->xsph)
    if (int rc = ctx->fast ? slab_extras_impl<N, true>(ctx, p) : slab_extras_impl<N, false>(ctx, p)) return rc;
  if (int rc = join_diffuse(ctx)) return rc;
  // The copies stay where they are: the next step's predict marks them dead and its sort drops them; whoever looks at
  // the arrays from outside calls drop_ghosts() first.
  ctx->ghostsPending = true;
  return PBF_OK;
}

}  // namespace

extern "C" {

const char *pbf_comm_last_error(const pbf_comm *comm) { return comm ? comm->err.c_str() : g_comm_error.c_str(); }
uint64_t pbf_comm_rounds(const pbf_comm *comm) { return comm ? comm->rounds : 0; }

int pbf_comm_unique_id(void *id128) {
  if (!id128) return PBF_ERR_INVALID;
  pbf_comm tmp;
  if (!comm_load_rccl(&tmp)) {
    g_comm_error = tmp.err;
    return PBF_ERR_COMM;
  }
  ncclUniqueId id;
  const ncclResult_t r = tmp.fGetUniqueId(&id);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclGetUniqueId: ") + tmp.fErrorString(r);
    return PBF_ERR_COMM;
  }
  static_assert(sizeof(id) == PBF_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof(id));
  return PBF_OK;
}

int pbf_comm_create_rccl(const void *id128, int nranks, int rank, int device, pbf_comm **out) {
  if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) {
    g_comm_error = "pbf_comm_create_rccl: bad argument";
    return PBF_ERR_INVALID;
  }
  *out = nullptr;
  auto *c = new pbf_comm();
  c->nranks = nranks, c->rank = rank, c->device = device;
  if (!comm_load_rccl(c)) {
    g_comm_error = c->err;
    delete c;
    return PBF_ERR_COMM;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_comm_error = "hipSetDevice failed";
    delete c;
    return PBF_ERR_HIP;
  }
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = c->fCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclCommInitRank: ") + c->fErrorString(r);
    delete c;
    return PBF_ERR_COMM;
  }
  *out = c;
  return PBF_OK;
}

int pbf_comm_create_host_callback(pbf_exchange_fn fn, void *user, int nranks, int rank, pbf_comm **out) {
  if (!fn || !out || nranks < 1 || rank < 0 || rank >= nranks) {
    g_comm_error = "pbf_comm_create_host_callback: bad argument";
    return PBF_ERR_INVALID;
  }
  auto *c = new pbf_comm();
  c->nranks = nranks, c->rank = rank, c->fn = fn, c->user = user;
  *out = c;
  return PBF_OK;
}

void pbf_comm_destroy(pbf_comm *c) {
  if (!c) return;
  if (c->comm && c->fCommDestroy) (void)c->fCommDestroy(c->comm);
  for (void *h : c->host)
    if (h) (void)hipHostFree(h);
  delete c;  // (the dlopen handle stays: other communicators / the hosting process may use librccl)
}

int pbf_comm_allreduce_u32(pbf_comm *c, void *device_u32, size_t count, void *stream) {
  if (!c || !device_u32) return PBF_ERR_INVALID;
  if (!c->comm) {
    c->err = "pbf_comm_allreduce_u32 needs an RCCL communicator";
    return PBF_ERR_COMM;
  }
  const ncclResult_t r = c->fAllReduce(device_u32, device_u32, count, ncclUint32, ncclSum, c->comm, static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? PBF_OK : comm_fail(c, "ncclAllReduce", r);
}

int pbf_slab_set_cuts(pbf_ctx *ctx, const uint32_t *cuts) {
  if (int rc = slab_check(ctx)) return rc;
  if (!ctx->comm || !cuts) return fail(ctx, PBF_ERR_STATE, "pbf_slab_set_cuts needs pbf_slab_attach first");
  const int n = ctx->comm->nranks;
  for (int g = 0; g < n; ++g)
    if (cuts[g + 1] <= cuts[g]) return fail(ctx, PBF_ERR_INVALID, "cuts must be strictly increasing");
  ctx->cuts.assign(cuts, cuts + n + 1);
  return apply_cuts(ctx);
}

int pbf_slab_attach(pbf_ctx *ctx, pbf_comm *comm, const uint32_t *cuts, uint32_t cap_migrants, uint32_t cap_ghosts) {
  if (int rc = slab_check(ctx)) return rc;
  if (!comm || !cuts || !cap_migrants || !cap_ghosts) return fail(ctx, PBF_ERR_INVALID, "NULL / zero argument");
  ctx->comm = comm;
  ctx->capMig = cap_migrants, ctx->capGhost = cap_ghosts;
  const size_t mig = ctx->fp64 ? sizeof(MigrantRec<double>) : sizeof(MigrantRec<float>);
  const size_t gho = ctx->fp64 ? sizeof(GhostRec<double>) : sizeof(GhostRec<float>);
  // the buffers hold far more than the first message: half the particle capacity per neighbour (whole columns change
  // hands after a re-cut) — a few hundred MB at most, nothing against 288 GB of HBM
  ctx->wireCap = uint32_t(std::max<size_t>(std::max<size_t>(ctx->cap, ctx->reserve) / 2, 4 * size_t(std::max(cap_migrants, cap_ghosts))));
  const size_t bytes = WIRE_HDR + size_t(ctx->wireCap) * std::max(mig, gho);
  for (int k = 0; k < 2; ++k) {
    if (int rc = ensure(ctx, ctx->wireSend[k], bytes)) return rc;
    if (int rc = ensure(ctx, ctx->wireRecv[k], bytes)) return rc;
    if (int rc = ensure(ctx, ctx->wireGhost[k], bytes)) return rc;
  }
  if (!ctx->hostCounts) {
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->hostCounts), 64, hipHostMallocDefault));
    std::memset(ctx->hostCounts, 0, 64);
  }
  return pbf_slab_set_cuts(ctx, cuts);
}

int pbf_slab_step(pbf_ctx *ctx, const pbf_params *p) { return pbf_slab_steps(ctx, p, 1); }
int pbf_slab_steps(pbf_ctx *ctx, const pbf_params *p, uint32_t count) {
  if (int rc = check(ctx, p, false)) return rc;
  if (!ctx->comm) return fail(ctx, PBF_ERR_STATE, "pbf_slab_step needs pbf_slab_attach first");
  // finalise(t) + predict(t + 1) as one kernel between two steps of THIS call (same parameters, same cuts, nothing looks
  // at the state in between) — like pbf_steps
  const bool timed = (ctx->desc.flags & PBF_FLAG_STAGE_TIMING) != 0 &&
                     (((ctx->timingMask >> ST_PREDICT) & 1u) != 0 || ((ctx->timingMask >> ST_FINALISE) & 1u) != 0);
  const bool fusable = ctx->fusePredict && !timed && !(p->vorticity || p->xsph);
  for (uint32_t i = 0; i < count; ++i) {
    ctx->fuseNextPredict = fusable && i + 1 < count;
    if (int rc = DISPATCH(ctx, slab_step_impl, ctx, p)) {
      ctx->fuseNextPredict = ctx->prePredicted = false;
      return rc;
    }
  }
  ctx->fuseNextPredict = false;
  return PBF_OK;
}
uint64_t pbf_slab_host_syncs(const pbf_ctx *ctx) { return ctx ? ctx->slabHostSyncs : 0; }

}  // extern "C"

// ================================================================================================
// Marching-cubes surface (pbf_surface / pbf_download_mesh): reference src/omp/ompsph.hpp:277-477
// ================================================================================================
namespace {

template <typename N> int surface_impl(pbf_ctx *ctx, const pbf_params *p, const pbf_mc_params *mp, uint64_t *nTriangles) {
  StepConsts<N> c;
  if (int rc = make_consts<N>(ctx, p, c)) return rc;
  // Slab mode (after pbf_slab_step; the copies of the neighbours' boundary columns are still in the arrays): every rank
  // extracts the part of the GLOBAL lattice whose nodes lie in its own cell columns — the ghost layer supplies exactly the
  // 27-cell neighbourhoods those nodes need — after refreshing the copies' colours (the owners diffused them during the
  // step), receives the one node plane its last cubes share with the right-hand neighbour, and emits its cubes'
  // triangles: the ranks' meshes, concatenated in rank order, ARE the single-device mesh's cube order (x-major).
  if (int rc = materialise_pstar<N>(ctx)) return rc;  // (slab mode reads the copies' pStar)
  const bool slab = ctx->slabConfigured && ctx->comm && ctx->comm->nranks > 1;
  if (ctx->slabConfigured && !slab) return fail(ctx, PBF_ERR_STATE, "pbf_surface in slab mode needs pbf_slab_attach");
  if (slab && !ctx->ghostsPending) return fail(ctx, PBF_ERR_STATE, "pbf_surface in slab mode must follow pbf_slab_step directly");
  McConsts<N> m;
  m.scale = c.scale, m.res = N(mp->resolution), m.isolevel = N(mp->isolevel), m.particleSize = N(mp->particle_size);
  m.particleInfluence = N(mp->particle_influence);
  m.step = c.h / m.res;           // ompsph.hpp:291
  m.threshold = c.h * c.scale * 1;  // ompsph.hpp:293
  uint64_t sampleG[3];
  for (int k = 0; k < 3; ++k) {
    m.minExtent[k] = c.minExtent[k];
    m.extent[k] = uint32_t(ctx->extent[k]);   // (global: make_consts leaves ctx->extent untouched by the slab frame)
    sampleG[k] = uint64_t(std::floor(N(ctx->extent[k]) * m.res)) + 1;  // ompsph.hpp:283-284
    m.sample[k] = uint32_t(sampleG[k]);
  }
  m.xoff = 0, m.nodeX0 = 0, m.planes = m.sample[0];
  bool hasRight = false;
  if (slab) {
    const int r = ctx->comm->rank, nr = ctx->comm->nranks;
    // first global node x whose cell column floor(x / res) is >= col: the smallest integer x with x / res >= col (in N,
    // like the kernel's own division)
    auto first_node = [&](uint32_t col) {
      uint64_t x = uint64_t(std::ceil(N(col) * m.res));
      while (x > 0 && uint64_t(N(x - 1) / m.res) >= col) --x;
      while (uint64_t(N(x) / m.res) < col) ++x;
      return std::min<uint64_t>(x, sampleG[0]);
    };
    const uint64_t x0 = r > 0 ? first_node(ctx->cuts[r]) : 0, x1 = r + 1 < nr ? first_node(ctx->cuts[r + 1]) : sampleG[0];
    hasRight = r + 1 < nr && x1 < sampleG[0];
    m.xoff = ctx->xoff, m.nodeX0 = uint32_t(x0), m.planes = uint32_t(x1 - x0);
    m.sample[0] = m.planes + (hasRight ? 1u : 0u);
    // the owners' diffused colours -> their copies on the neighbours (one more field round; the copies' pStar is current)
    const size_t fb = sizeof(vec4<N>);
    vec4<N> *col = ctx->col4[ctx->cur].as<vec4<N>>();
    if (int rc = slab_pack<N>(ctx, ctx->wireSend[0].p, ctx->wireSend[1].p, col)) return rc;
    if (int rc = exchange(ctx, ctx->sentL * fb, ctx->sentR * fb, ctx->gotL * fb, ctx->gotR * fb)) return rc;
    if (int rc = slab_unpack<N>(ctx, ctx->wireRecv[0].p, ctx->wireRecv[1].p, col)) return rc;
  }
  for (int k = 0; k < 3; ++k) ctx->mcSample[k] = m.sample[k];
  const uint64_t planeN = uint64_t(m.sample[1]) * m.sample[2];
  const uint64_t latticeN = uint64_t(m.sample[0]) * planeN;
  if (latticeN >= (uint64_t(1) << 31)) return fail(ctx, PBF_ERR_INVALID, "surface lattice too large (resolution x extent)");
  m.tableN = c.tableN, m.hasObstacles = c.hasObstacles;
  const int s = ctx->cur;
  if (int rc = ensure(ctx, ctx->latticePN, (latticeN + 1) * sizeof(vec4<N>))) return rc;
  if (int rc = ensure(ctx, ctx->latticeC, (latticeN + 1) * sizeof(vec4<N>))) return rc;
  *nTriangles = 0;
  ctx->mcTriangles = 0;
  ctx->meshStaged = false;
  if (m.planes) {
    // which of a cell's 27 slots hold particles (most lattice nodes sit in empty space and skip their gather)
    if (int rc = ensure(ctx, ctx->mcNear, (size_t(c.tableN) + 64) * 4)) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->mcNear.p, 0, size_t(c.tableN) * 4, ctx->stream));
    hipLaunchKernelGGL(k_mc_mark_near, grid_for(c.tableN), dim3(BLOCK), 0, ctx->stream, c.tableN,
                       make_uint3(m.extent[0], m.extent[1], m.extent[2]), m.xoff, ctx->table.as<const uint32_t>(),
                       ctx->mcNear.as<uint32_t>());
    const uint64_t nodeBlocks = uint64_t((m.planes + 3) / 4) * ((m.sample[1] + 3) / 4) * ((m.sample[2] + 3) / 4);
    hipLaunchKernelGGL((k_mc_field<N>), grid_for(nodeBlocks * 64), dim3(BLOCK), 0, ctx->stream, m, ctx->table.as<const uint32_t>(),
                       ctx->pos4[s].as<const vec4<N>>(), ctx->pstar[ctx->pcur].as<const vec4<N>>(), ctx->col4[s].as<const vec4<N>>(),
                       ctx->type[s].as<const uint8_t>(), ctx->mcNear.as<const uint32_t>(), ctx->latticePN.as<vec4<N>>(),
                       ctx->latticeC.as<vec4<N>>());
    LAUNCH_CHECK(ctx);
  }
  if (slab) {  // my first node plane -> the left neighbour's extra plane (two rounds: {v, normal} and colours)
    const size_t pb = size_t(planeN) * sizeof(vec4<N>);
    const bool sendLeft = ctx->comm->rank > 0 && m.planes > 0;
    for (DevBuf *lat : {&ctx->latticePN, &ctx->latticeC}) {
      uint8_t *base = lat->as<uint8_t>();
      if (int rc = comm_exchange(ctx->comm, ctx->stream, base, sendLeft ? pb : 0, nullptr, 0, nullptr, 0,
                                 base + size_t(m.planes) * pb, hasRight ? pb : 0)) {
        ctx->err = "slab surface exchange: " + ctx->comm->err;
        return rc;
      }
    }
  }
  if (m.sample[0] < 2 || m.sample[1] < 2 || m.sample[2] < 2) return PBF_OK;
  const uint64_t march64 = uint64_t(m.sample[0] - 1) * (m.sample[1] - 1) * (m.sample[2] - 1);
  const uint32_t marchVolume = uint32_t(march64), len = marchVolume + 1;
  const uint32_t nb = (len + SCAN_TILE - 1) / SCAN_TILE;
  if (int rc = ensure(ctx, ctx->mcCounts, (size_t(len) + SCAN_TILE) * 4)) return rc;
  if (int rc = ensure(ctx, ctx->mcOffsets, (size_t(len) + SCAN_TILE) * 4)) return rc;
  if (int rc = ensure(ctx, ctx->mcSums, (size_t(nb) + 1) * 4)) return rc;
  uint32_t *counts = ctx->mcCounts.as<uint32_t>(), *offsets = ctx->mcOffsets.as<uint32_t>(), *sums = ctx->mcSums.as<uint32_t>();
  HIPCHK(ctx, hipMemsetAsync(counts + marchVolume, 0, 4, ctx->stream));  // closing sentinel: offsets[marchVolume] = total
  hipLaunchKernelGGL((k_mc_count<N>), grid_for(marchVolume), dim3(BLOCK), 0, ctx->stream, m, marchVolume,
                     ctx->latticePN.as<const vec4<N>>(), counts);
  launch_scans(ctx, scan_job(counts, len, sums, offsets), 1);
  LAUNCH_CHECK(ctx);
  uint32_t total = 0;
  HIPCHK(ctx, hipMemcpyAsync(&total, offsets + marchVolume, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the mesh buffers are sized from the count
  ctx->mcTriangles = total;
  *nTriangles = total;
  if (total == 0) return PBF_OK;
  if (int rc = ensure(ctx, ctx->meshV, size_t(total) * 9 * sizeof(N))) return rc;
  if (int rc = ensure(ctx, ctx->meshN, size_t(total) * 9 * sizeof(N))) return rc;
  if (int rc = ensure(ctx, ctx->meshC, size_t(total) * 12 * sizeof(N))) return rc;
  hipLaunchKernelGGL((k_mc_emit<N>), grid_for(marchVolume), dim3(BLOCK), 0, ctx->stream, m, marchVolume,
                     ctx->latticePN.as<const vec4<N>>(), ctx->latticeC.as<const vec4<N>>(), offsets, ctx->meshV.as<N>(),
                     ctx->meshN.as<N>(), ctx->meshC.as<N>());
  LAUNCH_CHECK(ctx);
  return PBF_OK;
}

}  // namespace

extern "C" {

int pbf_surface(pbf_ctx *ctx, const pbf_params *params, const pbf_mc_params *mc, uint64_t *n_triangles) {
  if (int rc = check(ctx, params, true)) return rc;

  if (!mc || !n_triangles) return fail(ctx, PBF_ERR_INVALID, "NULL argument");
  if (!(mc->resolution > 0)) return fail(ctx, PBF_ERR_INVALID, "resolution must be > 0");
  return DISPATCH(ctx, surface_impl, ctx, params, mc, n_triangles);
}

int pbf_download_mesh(pbf_ctx *ctx, void *vs, void *ns, void *cs) {
  if (!ctx) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = ctx->mcTriangles, e = ctx->fp64 ? 8 : 4;
  if (n == 0) return PBF_OK;
  if (vs) HIPCHK(ctx, hipMemcpyAsync(vs, ctx->meshV.p, n * 9 * e, hipMemcpyDeviceToHost, ctx->stream));
  if (ns) HIPCHK(ctx, hipMemcpyAsync(ns, ctx->meshN.p, n * 9 * e, hipMemcpyDeviceToHost, ctx->stream));
  if (cs) HIPCHK(ctx, hipMemcpyAsync(cs, ctx->meshC.p, n * 12 * e, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PBF_OK;
}

int pbf_map_mesh(pbf_ctx *ctx, const void **vs, const void **ns, const void **cs) {
  if (!ctx || !vs || !ns || !cs) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = ctx->mcTriangles, e = ctx->fp64 ? 8 : 4;
  *vs = *ns = *cs = nullptr;
  if (n == 0) return PBF_OK;
  const size_t bv = n * 9 * e, bc = n * 12 * e, total = 2 * bv + bc;
  if (ctx->meshHostCap < total) {
    if (ctx->meshHost) (void)hipHostFree(ctx->meshHost);
    ctx->meshHost = nullptr, ctx->meshHostCap = 0;
    const size_t want = total + total / 4 + 4096;
    HIPCHK(ctx, hipHostMalloc(&ctx->meshHost, want, hipHostMallocDefault));
    ctx->meshHostCap = want;
    ctx->meshStaged = false;
  }
  char *h = static_cast<char *>(ctx->meshHost);
  if (!ctx->meshStaged) {
    // (one wait behind all three: handing the arrays out one by one, an event behind each, so that the caller's copy of the
    // vertices overlaps the normals' DMA was measured — no faster at 0.75 M vertices, 11 % slower at 1.35 M: the host copies
    // and the DMA share the memory system)
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->meshV.p, bv, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h + bv, ctx->meshN.p, bv, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h + 2 * bv, ctx->meshC.p, bc, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->meshStaged = true;
  }
  *vs = h, *ns = h + bv, *cs = h + 2 * bv;
  return PBF_OK;
}

int pbf_read_lattice(pbf_ctx *ctx, uint64_t sample[3], void *pn, void *c) {
  if (!ctx || !sample) return PBF_ERR_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t n = 1;
  for (int k = 0; k < 3; ++k) sample[k] = ctx->mcSample[k], n *= ctx->mcSample[k];
  const size_t v = ctx->fp64 ? sizeof(double4) : sizeof(float4);
  if (n && pn) HIPCHK(ctx, hipMemcpyAsync(pn, ctx->latticePN.p, n * v, hipMemcpyDeviceToHost, ctx->stream));
  if (n && c) HIPCHK(ctx, hipMemcpyAsync(c, ctx->latticeC.p, n * v, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PBF_OK;
}

}  // extern "C"
