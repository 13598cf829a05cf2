/*
 * pbf_hip.h — C ABI of libpbf_hip.so: the MI355X (gfx950) implementation of the PBF-SPH
 * per-step hot path behind the reference's Solver::advance() surface.
 *
 * The reference has no FFI: its backends are C++ classes chosen by a switch
 * (src/benchmark.cpp:105-172) behind
 *     sph::Solver<T,N,V>::advance(const SphParams&, const Scene&, std::vector<Particle>&)
 * (src/sph.hpp:119-125).  This header is the boundary a maintainer binds a new backend to;
 * pbf-sph_amd/host/hipsph.hpp is that binding (sph::hip_impl::Solver<T,N>, a thin shim), and
 * INTEGRATION.md shows the lines to add to the reference's benchmark.cpp / args.hpp.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returns 0 on
 * success or a negative pbf_status; pbf_last_error() gives the text.  Nothing throws across
 * the ABI (reference: exceptions, src/benchmark.cpp:32-37).  One ctx = one caller thread at a
 * time (the reference's advance() has no thread-safety contract either, src/visualise.cpp:85-109).
 * Arrays are caller-owned SoA: pos/vel 3 per particle, colour 4 per particle, element type
 * float (fp64 = 0) or double (fp64 = 1) as chosen in pbf_desc (reference: template parameter N,
 * src/specialisation.cpp:13-14).  ids are uint64 (reference T = size_t).
 */
#ifndef PBF_HIP_H
#define PBF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBF_ABI_VERSION 1

typedef struct pbf_ctx pbf_ctx;

typedef enum pbf_status {
  PBF_OK = 0,
  PBF_ERR_INVALID = -1,   /* bad argument */
  PBF_ERR_HIP = -2,       /* a HIP runtime call failed (text in pbf_last_error) */
  PBF_ERR_NO_DEVICE = -3, /* no gfx950 device / HIP runtime unusable: the product path never falls back to CPU */
  PBF_ERR_STATE = -4,     /* call order violated (e.g. stage before upload) */
  PBF_ERR_COMM = -5,      /* RCCL failure */
} pbf_status;

/* sph::Type (src/sph.hpp:15) */
enum { PBF_TYPE_FLUID = 0, PBF_TYPE_OBSTACLE = 1 };

enum {
  PBF_FLAG_STAGE_TIMING = 1u << 0, /* record hipEvents around every stage (Stopwatch analogue, src/utils.hpp:15-57) */
  PBF_FLAG_FAST_MATH = 1u << 1,    /* v_rsq/v_rcp in the pair kernels (the reference itself ships -Ofast,
                                      CMakeLists.txt:136, and native_divide/fast_distance in OpenCL) */
  PBF_FLAG_NO_LDS = 1u << 2,       /* force the per-particle global-memory gather kernels (debug / A-B) */
};

/* Replaces the ctor  omp_impl::Solver<T,N>(N h)  (src/omp/ompsph.hpp:83). */
typedef struct pbf_desc {
  uint32_t abi_version; /* PBF_ABI_VERSION */
  int32_t fp64;         /* 0: N = float, 1: N = double (benchmark --fp64, src/args.cpp:34-37) */
  int32_t device;       /* HIP device ordinal (benchmark -d, src/args.cpp:20-24) */
  uint32_t flags;       /* PBF_FLAG_* */
  double h;             /* smoothing length; benchmark.cpp:160-163 passes 0.1 */
  void *stream;         /* hipStream_t to launch on; NULL = a stream owned by the ctx */
} pbf_desc;

/* Replaces sph::SphParams (src/sph.hpp:97-103) + sph::Scene::wells (src/sph.hpp:56-60,76).
 * SphParams::h is dead in the reference (src/sph.hpp:98, never read) and therefore absent. */
typedef struct pbf_params {
  double dt, scale;
  uint64_t iteration;       /* solver iterations K */
  double constant_force[3];
  double min_bound[3], max_bound[3];
  int32_t n_wells;
  const double *wells;      /* n_wells x {cx, cy, cz, force}, host */
  int32_t xsph, vorticity;  /* opt-in, NOT in the reference (only constants survive, src/sph_constants.h:13-14) */
} pbf_params;

/* ---- lifetime -------------------------------------------------------------------------- */
int pbf_create(const pbf_desc *desc, pbf_ctx **out);
void pbf_destroy(pbf_ctx *ctx);
/* ctx may be NULL: last error of a failed pbf_create on this thread */
const char *pbf_last_error(const pbf_ctx *ctx);
int pbf_abi_version(void);
/* Tuning / diagnostic knobs (no reference counterpart): "gather" (0 global walk, 1 neighbour lists = default, 3 LDS tiles
 * per brick), "list_max", "tile_cap", "reuse_lists", "split_build" (0 = lambda builds the lists while it gathers; 4 / 5 = a
 * list-build launch of its own with 2 / 4 pair loads per trip, then a list-driven lambda; 8 = DEFAULT: the quantised list
 * build with lambda riding on its flushes, one launch), "coop" (0 = one lane per particle in the list-driven lambda /
 * delta-p, bit-exact, default; 2 / 4 / 8 = that many lanes share a particle's list and reduce the kernel sums with wave
 * shuffles: rounding-level differences), "cell_diffuse" (one colour walk per occupied cell, default 1), "fuse_diffuse",
 * "overlap_diffuse", "fuse_predict", "pipeline", "graph", "pad_lds", "timing_mask" (bit i = stage i of pbf_stage_times is
 * bracketed with events), "row_major" (DEFAULT 1: the solver iterations run on a cell-ROW-major copy of {pStar, lambda,
 * quantised position, mass, type} — one contiguous run per (dy, dz) row of the 27-cell stencil, lists of row slots; 0 =
 * everything in Morton order), "nbr_chunks" (size of the pool of 120-slot overflow chunks of the two-tier neighbour lists;
 * 0 = sized from the particle count), "row_diffuse" (DEFAULT 1, with row_major: the colour diffusion walks the row-major copy, one
 * wave per segment of 64 x cells, runs staged by LDS-DMA, sums applied in place; 0 = per-cell sums on the Morton order, beside
 * the iterations), "diffuse_cap" (diagnostic: records in that kernel's LDS tile, 0 = default 640).  Every setting of these
 * is bit-identical.  Unknown names return PBF_ERR_INVALID. */
int pbf_set_option(pbf_ctx *ctx, const char *name, int64_t value);

/* ---- particle state (replaces the std::vector<Particle>& in/out argument, src/sph.hpp:124) */
int pbf_upload(pbf_ctx *ctx, size_t n, const uint64_t *id, const uint8_t *type, const void *mass, const void *pos,
               const void *vel, const void *colour);
/* Device order = Z-sorted after a step, exactly like the reference's write-back
 * (src/omp/ompsph.hpp:479-481).  Any pointer may be NULL. */
int pbf_download(pbf_ctx *ctx, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel, void *colour);
size_t pbf_count(const pbf_ctx *ctx);

/* AoS path for the C++ shim: `particles` is the caller's std::vector<Particle>::data()
 * (src/sph.hpp:36-54).  Unpacked / repacked on the device; offsets in bytes.  The buffer is page-locked in place
 * (hipHostRegister, kept while the same pointer comes back: benchmark.cpp:33,47 passes one vector every frame) so both
 * copies are plain DMA.  pbf_download_aos writes the FIELDS; struct padding bytes (no part of the reference's contract:
 * its write-back copies whole structs, ompsph.hpp:479-481) receive the bytes the last uploaded image held at that slot. */
typedef struct pbf_aos_layout {
  uint32_t stride, off_id, off_type, off_mass, off_pos, off_vel, off_colour;
} pbf_aos_layout;
int pbf_upload_aos(pbf_ctx *ctx, size_t n, const void *particles, const pbf_aos_layout *layout);
int pbf_download_aos(pbf_ctx *ctx, void *particles, const pbf_aos_layout *layout);
/* The same download in two halves: _begin packs the image and starts its DMA on a copy stream of its own, _end waits for it.
 * Work enqueued in between that only READS the particle state — pbf_surface, in the shim's advance() — runs while the
 * particles travel over PCIe (1 M particles: 1.1 ms hidden behind 1.0 ms of surface kernels).  `particles` must stay valid
 * until _end; no step, upload or stage call in between. */
int pbf_download_aos_begin(pbf_ctx *ctx, void *particles, const pbf_aos_layout *layout);
int pbf_download_aos_end(pbf_ctx *ctx);

/* ---- the hot path ---------------------------------------------------------------------- */
/* One advance() on device-resident state (src/omp/ompsph.hpp:128-271 + 479-481), asynchronous
 * on the ctx stream: predict+key -> counting sort + cell table -> diffuse -> K x (lambda, delta)
 * -> finalise.  The two loops the reference updates in place (racy with >1 thread,
 * src/omp/ompsph.hpp:188-207,235-248) are double-buffered (Jacobi) here. */
int pbf_step(pbf_ctx *ctx, const pbf_params *params);
/* count x pbf_step, no host sync.  Option "graph" = 1: each distinct step (same buffer roles, same parameters) is captured
 * into a hipGraph the first time it comes by and replayed afterwards — one graph launch instead of ~25 kernel launches;
 * stage timing, wells, slab mode, a step that still allocates and scenes whose parameters change every frame (the stock
 * moving box) run eagerly; results are identical either way.  Default 0: measured 1-9 % SLOWER than eager launches on
 * MI355X / ROCm 7 at 16 K - 1 M particles (the step is never host-launch bound). */
int pbf_steps(pbf_ctx *ctx, const pbf_params *params, uint32_t count);
int pbf_graph_stats(const pbf_ctx *ctx, uint64_t out[3]); /* {graphs captured, graph replays, graphs still enabled} */
int pbf_sync(pbf_ctx *ctx);

/* Stage-level entry points: exactly the launches pbf_step makes, exposed so that each kernel
 * can be checked against the oracle stage by stage.  Order as in pbf_step. */
int pbf_stage_predict(pbf_ctx *ctx, const pbf_params *params);  /* ompsph.hpp:132-154 */
int pbf_stage_sort(pbf_ctx *ctx, const pbf_params *params);     /* ompsph.hpp:157-165, sph.hpp:238-250 */
int pbf_stage_diffuse(pbf_ctx *ctx, const pbf_params *params);  /* ompsph.hpp:188-207 */
int pbf_stage_lambda(pbf_ctx *ctx, const pbf_params *params);   /* ompsph.hpp:217-232 */
int pbf_stage_delta(pbf_ctx *ctx, const pbf_params *params);    /* ompsph.hpp:235-248 */
int pbf_stage_finalise(pbf_ctx *ctx, const pbf_params *params); /* ompsph.hpp:256-264 */

/* ---- introspection (parity tests, Stopwatch analogue) ------------------------------------ */
enum pbf_buffer {
  PBF_BUF_KEYS = 0,   /* uint32[n]  Morton cell key per particle, current device order */
  PBF_BUF_TABLE = 1,  /* uint32[table_size] = the reference's gridTable (sph.hpp:238-250) */
  PBF_BUF_PSTAR = 2,  /* N[4n]: pStar.xyz, lambda */
  PBF_BUF_NBR_COUNT = 3, /* uint32[n]: neighbour list of each particle after the last list build: low 8 bits = length (<= 160),
                            upper 24 bits = the pool chunk holding its slots 40..63 when the length exceeds 40 (two-tier
                            lists, csrc/pbf_kernels.hpp NbrLists); 0xFFFFFFFF = more than 160 survivors (or the chunk pool ran
                            dry): the particle walks its cells; diagnostic, list gather only */
  PBF_BUF_OMEGA = 4,  /* N[4n]: {omega.xyz, 0}, the vorticity estimate of the last step run with pbf_params.vorticity
                         (opt-in extra, absent from the reference), device order; PBF_ERR_STATE when there is none */
  PBF_BUF_COUNT_ = 5,
};
int pbf_read_buffer(pbf_ctx *ctx, int which, void *host, size_t bytes);
size_t pbf_table_size(const pbf_ctx *ctx);                /* Morton(extent), sph.hpp:240 */
int pbf_grid_extent(const pbf_ctx *ctx, uint64_t extent[3], double min_extent[3]); /* ompsph.hpp:132-135 */

/* Device self-test of the trimmed exact sqrt / divides the precise pair terms use (csrc/pbf_kernels.hpp sqrt_rsq /
 * div_seeded / div_ranged) against the compiler's full IEEE forms, exhaustively: mismatches[0] sqrt over EVERY fp32 value
 * >= 2^-75; mismatches[1] (h - r)^2 / r over every fp32 d2 whose root lies in [1e-8, h], for the context's own h and
 * three more; mismatches[2] x / poly6(0.3 h) over every fp32 x with 1e-30 <= |x| <= 1e30 or x == 0 (one mismatch is the
 * sign of a zero quotient, which is only ever squared); mismatches[3] delta-p's x / RHO — trimmed where the numerator is
 * in range, the compiler's divide otherwise — over EVERY fp32 x.  [0], [1], [3] must be 0, [2] <= 1.
 * On an fp64 context the same four categories are swept for the fp64 forms (v_rsq_f64-seeded sqrt = the compiler's own
 * sequence without its rescale wrappers; Newton divides with one exact-residual correction) over 1.07e10 pseudo-random
 * operands EACH (an exhaustive fp64 sweep is impossible): every exponent of the stated ranges equally often plus the pair
 * terms' own operand ranges densely (csrc/pbf_kernels.hpp k_selftest_math64); all four must be 0. */
int pbf_selftest_math(pbf_ctx *ctx, uint64_t mismatches[4]);

/* Mean milliseconds per call of each stage since the last pbf_reset_stage_times (needs
 * PBF_FLAG_STAGE_TIMING).  names[i] points at static strings that follow the reference's Stopwatch
 * entries (ompsph.hpp:130,157,161,188,209,252); an entry named "stage/part" is a sub-interval of "stage"
 * (e.g. "sph-lambda/list-build": the neighbour-list kernel inside the lambda stage) and must not be added
 * to it.  Returns the number of entries (<= cap). */
int pbf_stage_times(pbf_ctx *ctx, const char **names, double *mean_ms, uint64_t *calls, int cap);
int pbf_reset_stage_times(pbf_ctx *ctx);

/* ---- marching-cubes surface (reference: config.surface, src/sph.hpp:82-95,102; src/omp/ompsph.hpp:277-477) ---
 * Runs on the state the last pbf_step left (its grid table is still valid): scalar field on a lattice of
 * floor(extent * resolution) + 1 nodes per axis, triangles per cube, emission in cube order.
 * pbf_surface returns the triangle count; pbf_download_mesh copies 3 vertices per triangle:
 * vs / ns = 9 values of N per triangle, cs = 12 (the reference's ColouredMesh, src/sph.hpp:105-112). */
typedef struct pbf_mc_params {
  double resolution, isolevel, particle_size, particle_influence; /* sph::McParams */
} pbf_mc_params;
int pbf_surface(pbf_ctx *ctx, const pbf_params *params, const pbf_mc_params *mc, uint64_t *n_triangles);
int pbf_download_mesh(pbf_ctx *ctx, void *vs, void *ns, void *cs);
/* The same mesh in PAGE-LOCKED host memory owned by the ctx (one DMA at PCIe speed instead of three pageable copies into
 * freshly allocated vectors: 54 MB at 1 M particles); the pointers stay valid until the next pbf_surface / pbf_destroy.
 * The C++ shim builds Result::mesh's vectors from them (range construction: no zero fill, the three copies in parallel). */
int pbf_map_mesh(pbf_ctx *ctx, const void **vs, const void **ns, const void **cs);
/* the lattice of the last pbf_surface: sample[3] nodes per axis, 4 + 4 values of N per node {v, normal} {colour} */
int pbf_read_lattice(pbf_ctx *ctx, uint64_t sample[3], void *pn, void *c);

/* ---- multi-GPU: slab decomposition along x (no reference counterpart: it is single-device) ------
 * One process per GPU.  Every rank uses the GLOBAL grid (same pbf_params bounds), owns the cell columns
 * [xlo, xhi) and keeps a one-cell layer of COPIES ("ghosts", type bit PBF_TYPE_GHOST) of its x-neighbours'
 * boundary columns.  The product path is pbf_slab_attach + pbf_slab_step further down: the whole step including the
 * RCCL exchanges runs inside the library.  The calls of THIS section are the same step cut into its stages (select /
 * pack / append / unpack on the device, the caller moves the wire buffers, which are device pointers) — kept so that
 * tests can check every stage against the oracle and drive several ranks on one GPU; see pbf-sph_amd/slab.py:
 *   predict -> migrate -> add_migrants -> ghosts -> add_ghosts -> sort -> diffuse ->
 *   K x { lambda -> pack/exchange/unpack -> delta -> pack/exchange/unpack } -> finalise -> finish */
enum { PBF_TYPE_GHOST = 2 };
enum { PBF_REC_MIGRANT = 0, PBF_REC_GHOST = 1, PBF_REC_FIELD = 2 };
typedef struct pbf_slab_cut {
  uint32_t xlo, xhi;           /* owned cell columns [xlo, xhi), grid coordinates of pbf_grid_extent */
  int32_t has_left, has_right; /* is there a rank on that side */
} pbf_slab_cut;
int pbf_reserve(pbf_ctx *ctx, size_t capacity); /* room for migrants + copies; call before pbf_upload */
/* Optional, before the first step: key the particles in a rank-LOCAL x frame (origin = PBF_SLAB_FRAME_MARGIN columns left of this slab's first column), so the grid table covers only the slab + ghost columns instead of Morton(global extent) — on an
 * elongated N-slab box that is 2 M entries per rank instead of 138 M at N = 8.  left_xlo / right_xlo = the
 * xlo of the neighbouring slabs (0 for rank 0 / unused without that neighbour); records are re-keyed on arrival.
 * cut = NULL returns to global keys. */
#define PBF_SLAB_FRAME_MARGIN 6u /* columns between the local frame's origin and the first owned column */
int pbf_slab_configure(pbf_ctx *ctx, const pbf_slab_cut *cut, uint32_t left_xlo, uint32_t right_xlo);
size_t pbf_slab_record_bytes(const pbf_ctx *ctx, int kind);
/* after pbf_stage_predict: compact the particles that stay, pack the leavers; counts[2] = records for left / right */
int pbf_slab_migrate(pbf_ctx *ctx, const pbf_slab_cut *cut, void *send_left, void *send_right, uint32_t cap_records,
                     uint32_t counts[2]);
int pbf_slab_add_migrants(pbf_ctx *ctx, const void *recv_left, uint32_t n_left, const void *recv_right, uint32_t n_right);
/* pack copies of the first / last owned column; remembers the sources for pbf_slab_pack */
int pbf_slab_ghosts(pbf_ctx *ctx, const pbf_slab_cut *cut, void *send_left, void *send_right, uint32_t cap_records,
                    uint32_t counts[2]);
/* append the neighbours' copies and rebuild the cell histogram; pbf_stage_sort comes next */
int pbf_slab_add_ghosts(pbf_ctx *ctx, const void *recv_left, uint32_t n_left, const void *recv_right, uint32_t n_right);
/* after each lambda / delta launch: owners' {pStar, lambda} -> wire, wire -> copies (PBF_REC_FIELD records) */
int pbf_slab_pack(pbf_ctx *ctx, void *send_left, void *send_right);
int pbf_slab_unpack(pbf_ctx *ctx, const void *recv_left, const void *recv_right);
int pbf_slab_finish(pbf_ctx *ctx); /* after finalise: drop the copies */
size_t pbf_owned_count(const pbf_ctx *ctx);
/* load balance: owned particles per GLOBAL grid column (keys of the last pbf_stage_predict), 1024 bins; synchronises.
 * The driver all-reduces the histograms and moves the cuts (pbf-sph_amd/slab.py recut()). */
int pbf_slab_column_histogram(pbf_ctx *ctx, uint32_t out[1024]);

/* ---- communicator + whole slab step behind the C ABI ------------------------------------------------------
 * (no reference counterpart).  A pbf_comm links this rank with its left (rank - 1) and right (rank + 1) slab:
 *   RCCL over xGMI  pbf_comm_unique_id on rank 0, the 128 bytes are broadcast by the caller over any channel, then
 *                   pbf_comm_create_rccl on every rank = ncclCommInitRank; the exchanges are
 *                   ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the solver's stream (no host sync);
 *   host callback   bring-up and tests (several ranks sharing ONE GPU under gloo): the library stages the wire
 *                   buffers through pinned host memory and calls `fn` to move the bytes (return 0 = ok).
 * pbf_slab_attach hands the communicator, the cuts (nranks + 1 column boundaries, identical on every rank) and the
 * wire capacities to the ctx; pbf_slab_step then runs ONE step including every exchange:
 *   predict -> [migrants] -> [ghost copies] -> sort -> diffuse -> K x { lambda -> [field] -> delta -> [field] }
 *   -> finalise [-> xsph / vorticity extras with 3 more field rounds when requested]
 *                              ([..] = one exchange round: 2 + 2K rounds per step)
 * The two assembly rounds carry their record counts in the header of a fixed-size first message {header | first
 * cap_* records} (no count round trip); the host reads those counts once per assembly round (2 small synchronising
 * read-backs per step), the 2K field rounds need none.  Only when a side holds more records than the first message
 * takes (a re-cut hands whole columns over) a second, exactly sized exchange follows.  Running out of wire buffer
 * (half the particle capacity per neighbour) is an error (PBF_ERR_COMM), never silent loss. */
typedef struct pbf_comm pbf_comm;
#define PBF_COMM_ID_BYTES 128
typedef int (*pbf_exchange_fn)(void *user, const void *send_left, size_t send_left_bytes, const void *send_right,
                               size_t send_right_bytes, void *recv_left, size_t recv_left_bytes, void *recv_right,
                               size_t recv_right_bytes); /* host pointers */
int pbf_comm_unique_id(void *id128);
int pbf_comm_create_rccl(const void *id128, int nranks, int rank, int device, pbf_comm **out);
int pbf_comm_create_host_callback(pbf_exchange_fn fn, void *user, int nranks, int rank, pbf_comm **out);
void pbf_comm_destroy(pbf_comm *comm);
const char *pbf_comm_last_error(const pbf_comm *comm); /* comm may be NULL: last failed create on this thread */
uint64_t pbf_comm_rounds(const pbf_comm *comm);        /* exchange rounds so far */
/* in-place sum over all ranks of `count` uint32 in DEVICE memory (load balance: column histograms); RCCL
 * communicators only (a host-callback communicator returns PBF_ERR_COMM: its caller reduces on the host) */
int pbf_comm_allreduce_u32(pbf_comm *comm, void *device_u32, size_t count, void *stream);
/* cuts[nranks + 1]; cap_migrants / cap_ghosts = records per neighbour in the FIRST message of the two assembly rounds
 * (size them for an ordinary step: a fraction of / one boundary column).  The ctx keeps the pointer
 * to comm (not owned).  Switches the ctx to the rank-local key frame (pbf_slab_configure). */
int pbf_slab_attach(pbf_ctx *ctx, pbf_comm *comm, const uint32_t *cuts, uint32_t cap_migrants, uint32_t cap_ghosts);
int pbf_slab_set_cuts(pbf_ctx *ctx, const uint32_t *cuts); /* load balance: new cuts (same on every rank) */
int pbf_slab_step(pbf_ctx *ctx, const pbf_params *params);
int pbf_slab_steps(pbf_ctx *ctx, const pbf_params *params, uint32_t count);
/* host read-backs pbf_slab_step has made so far: exactly 2 per step — the counts of the two assembly rounds, which size the
 * append launches and the new particle count (the 2K field rounds need none).  No hipStreamSynchronize: a one-wave kernel
 * writes the six words and then a sequence number into pinned host memory, the host polls that word. */
uint64_t pbf_slab_host_syncs(const pbf_ctx *ctx);

/* ---- scene factory (sph.hpp:127-186; dam-break: SURVEY.md §8d) — host only, no GPU needed -- */
size_t pbf_scene_cubes(int fp64, size_t count, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel,
                       void *colour);
size_t pbf_scene_dambreak(int fp64, size_t nominal, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel,
                          void *colour, double *box_side);
/* applyMotionSinXCosZ (sph.hpp:147-158): writes min/max bound of `base` shifted for `frame` into `out` */
void pbf_apply_motion(int fp64, const pbf_params *base, uint64_t frame, pbf_params *out);
/* simpleConfigWith2Cubes' SphParams (sph.hpp:168-175) with K = iteration */
void pbf_default_params(uint64_t iteration, double box_side, pbf_params *out);

#ifdef __cplusplus
}
#endif
#endif
