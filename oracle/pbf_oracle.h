/*
 * pbf_oracle.h — C interface of the CPU oracle for the PBF-SPH per-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the *checker*: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (libpbf_hip.so, the C++ host,
 * the benchmark CLI) must never link or call it.
 *
 * PARITY UNPINNED (arithmetic):  the reference ships no tests, fixtures or golden vectors, and
 * its OpenMP backend (src/omp/ompsph.hpp) cannot be compiled in this image (it needs glm, a
 * network FetchContent dependency, CMakeLists.txt:26-30; stand-ins are not allowed).  The
 * floating-point stages below are therefore a line-by-line restatement checked by reading only.
 * The integer stages (Morton encode/decode, grid table, 27-cell walk, scene factory) ARE pinned:
 * oracle/ref_grid.cpp compiles the reference's own glm-free headers (src/curves.h, src/sph.hpp)
 * into oracle/_ref/libref_grid.so and tests/test_oracle_vs_ref.py + tests/golden/ compare them.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root).  T (id type) = uint64, N = float or double selected at create time.
 */
#ifndef PBF_ORACLE_H
#define PBF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pbf_oracle pbf_oracle;

/* update-order semantics of the two racy loops (SURVEY.md finding 2, Appendix A 1-2) */
enum {
  PBF_ORACLE_GS = 0,     /* reference, 1 thread: in-place, ascending sorted index (ompsph.hpp:188-207,234-248) */
  PBF_ORACLE_JACOBI = 1, /* double-buffered pStar / colour: what the HIP path computes */
};

enum {
  PBF_ORACLE_SORT_STD = 0,    /* std::sort, key-only comparator (ompsph.hpp:158): libstdc++ tie permutation */
  PBF_ORACLE_SORT_STABLE = 1, /* std::stable_sort: ties keep previous order (what the HIP path does) */
};

typedef struct pbf_oracle_params {
  double h;                   /* solver ctor argument (ompsph.hpp:83; benchmark.cpp:160-163 passes 0.1) */
  double dt, scale;           /* sph.hpp:98 */
  uint64_t iteration;         /* sph.hpp:99 */
  double constant_force[3];   /* sph.hpp:100 */
  double min_bound[3];
  double max_bound[3];
  int32_t mode;               /* PBF_ORACLE_GS / PBF_ORACLE_JACOBI */
  int32_t sort;               /* PBF_ORACLE_SORT_* */
  int32_t threads;            /* OpenMP threads for the race-free loops; <=0: library default */
  int32_t xsph;               /* opt-in extras absent from the reference (SURVEY finding 3); 0 = reference */
  int32_t vorticity;
  int32_t n_wells;            /* ompsph.hpp:141-148 */
  const double *wells;        /* n_wells x {cx, cy, cz, force} */
} pbf_oracle_params;

/* fp64 = 0: N = float; 1: N = double */
pbf_oracle *pbf_oracle_create(int fp64);
void pbf_oracle_destroy(pbf_oracle *);

/* Particle arrays are SoA, n-long, N = float or double as selected at create time
 * (pos, vel: 3 per particle; colour: 4 per particle).  Copies in / out. */
int pbf_oracle_set_particles(pbf_oracle *, size_t n, const uint64_t *id, const uint8_t *type, const void *mass,
                             const void *pos, const void *vel, const void *colour);
size_t pbf_oracle_count(const pbf_oracle *);
int pbf_oracle_get_particles(const pbf_oracle *, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel,
                             void *colour);

/* One full advance() (ompsph.hpp:85-485 minus source/drain/query/MC): particles are left in
 * Z-sorted order exactly like the reference's write-back (ompsph.hpp:479-481). */
int pbf_oracle_step(pbf_oracle *, const pbf_oracle_params *);

/* Stage-level entry points (same code pbf_oracle_step runs), for per-kernel parity tests.
 * Order: predict -> sort -> grid_table -> diffuse -> iteration x (lambda, delta) -> finalise. */
int pbf_oracle_predict(pbf_oracle *, const pbf_oracle_params *);   /* ompsph.hpp:132-154 */
int pbf_oracle_sort(pbf_oracle *, const pbf_oracle_params *);      /* ompsph.hpp:157-159 */
int pbf_oracle_grid_table(pbf_oracle *, const pbf_oracle_params *);/* sph.hpp:238-250 */
int pbf_oracle_diffuse(pbf_oracle *, const pbf_oracle_params *);   /* ompsph.hpp:188-207 */
int pbf_oracle_lambda(pbf_oracle *, const pbf_oracle_params *);    /* ompsph.hpp:217-232 */
int pbf_oracle_delta(pbf_oracle *, const pbf_oracle_params *);     /* ompsph.hpp:235-248 */
int pbf_oracle_finalise(pbf_oracle *, const pbf_oracle_params *);  /* ompsph.hpp:256-264 */

/* Scratch state after the last stage call (sorted order once sort ran). */
int pbf_oracle_get_keys(const pbf_oracle *, uint64_t *keys);            /* zIndex, n */
int pbf_oracle_get_pstar(const pbf_oracle *, void *pstar);              /* 3n of N */
int pbf_oracle_get_lambda(const pbf_oracle *, void *lambda);            /* n of N */
size_t pbf_oracle_table_size(const pbf_oracle *);
int pbf_oracle_get_table(const pbf_oracle *, uint64_t *table);
int pbf_oracle_get_extent(const pbf_oracle *, uint64_t extent[3], void *min_extent /* 3 of N */);
/* mean / max number of 27-cell candidates per particle for the current table (for §8d pair rate) */
int pbf_oracle_candidate_stats(const pbf_oracle *, double *mean, uint64_t *max, double *mean_within_h);

/* Free-standing integer helpers (restated from src/curves.h) */
uint64_t pbf_oracle_morton_encode(uint64_t x, uint64_t y, uint64_t z);  /* curves.h:72-88 */
uint64_t pbf_oracle_morton_decode(uint64_t code, int axis);             /* curves.h:46-65 */
/* 27 neighbour codes of a cell in the reference's order (sph.hpp:217-234) */
void pbf_oracle_neighbour_codes(uint64_t zindex, uint64_t out[27]);
/* kernel factors as the reference computes them (sph.hpp:251-253), in N then widened to double */
double pbf_oracle_poly6_factor(int fp64, double h);
double pbf_oracle_spiky_factor(int fp64, double h);

/* Scene factories (restated from sph.hpp:127-186 and SURVEY.md §8d "dam-break").
 * Return the particle count; arrays may be NULL to query the count only. pos is world space. */
size_t pbf_oracle_scene_cubes(int fp64, size_t count, uint64_t *id, void *mass, void *pos, void *vel, void *colour);
size_t pbf_oracle_scene_dambreak(int fp64, size_t nominal, uint64_t *id, void *mass, void *pos, void *vel,
                                 void *colour, double *box_side);
/* applyMotionSinXCosZ (sph.hpp:147-158): offset added to min/max bound at a frame, computed in float */
void pbf_oracle_motion_offset(int fp64, uint64_t frame, double out[3]);

/* Marching-cubes surface of the CURRENT state (ompsph.hpp:277-477; run after pbf_oracle_step, which leaves
 * the sorted particles, their predict-time cells and the grid table in place).  McParams = sph.hpp:82-95.
 * Mesh: 3 vertices per triangle, in cube order. */
typedef struct pbf_oracle_mc {
  double resolution, isolevel, particle_size, particle_influence;
} pbf_oracle_mc;
int pbf_oracle_surface(pbf_oracle *, const pbf_oracle_params *, const pbf_oracle_mc *, uint64_t *n_triangles);
/* emit stage only, from a given lattice (sample[0]*sample[1]*sample[2] nodes, 4 + 4 values each) */
int pbf_oracle_surface_from_lattice(pbf_oracle *, const pbf_oracle_params *, const pbf_oracle_mc *,
                                    const uint64_t sample[3], const void *pn, const void *c, uint64_t *n_triangles);
int pbf_oracle_get_lattice(const pbf_oracle *, uint64_t sample[3], void *pn, void *c);
int pbf_oracle_get_mesh(const pbf_oracle *, void *vs /* 9n */, void *ns /* 9n */, void *cs /* 12n */);

/* Overwrite the scratch state (same order as the particles); NULL = leave.  Used by the tests of the
 * slab driver, which re-assemble owned particles + ghost copies (type bit 1 = ghost: a candidate that is
 * never updated locally) between stages. */
int pbf_oracle_set_scratch(pbf_oracle *, const uint64_t *keys, const void *pstar, const void *lambda);

/* Sensitivity probe (tests only): evaluate pow(q, CorrN) as (q*q)*(q*q), the device's form, instead
 * of the reference's std::pow (ompsph.hpp:240).  Default off = reference semantics. */
void pbf_oracle_set_pow4(pbf_oracle *, int on);

const char *pbf_oracle_last_error(void);

/* Scene dynamics on the host side of advance() (ompsph.hpp:91-126, 167-186): sources emit a floor(sqrt(rate)) x
 * ceil(sqrt(rate)) sheet at spacing h*scale/2, drains erase fluid closer than `width`, a query lists the fluid ids
 * in the cell of a point (after sort + grid_table).  sources[k] = {centre.xyz, velocity.xyz, colour.rgba, rate},
 * drains[k] = {centre.xyz, width}. */
int pbf_oracle_scene_emit(pbf_oracle *, double h, double scale, size_t n_sources, const uint64_t *tags,
                          const double *sources);
int pbf_oracle_scene_drain(pbf_oracle *, size_t n_drains, const double *drains);
size_t pbf_oracle_query(const pbf_oracle *, const pbf_oracle_params *, const double point[3], uint64_t *out, size_t cap);

/* The opt-in extras (absent from the reference) as stages of their own, run after pbf_oracle_finalise with xsph =
 * vorticity = 0: vorticity -> vorticity_force -> xsph is what finalise does itself when the flags are set.  The slab
 * twin refreshes the ghost copies' velocity (which = 0) / vorticity (which = 1) between them. */
int pbf_oracle_vorticity(pbf_oracle *, const pbf_oracle_params *);
int pbf_oracle_vorticity_force(pbf_oracle *, const pbf_oracle_params *);
int pbf_oracle_xsph(pbf_oracle *, const pbf_oracle_params *);
int pbf_oracle_get_vec(const pbf_oracle *, int which, void *out);
int pbf_oracle_set_vec(pbf_oracle *, int which, const void *in);

#ifdef __cplusplus
}
#endif
#endif
