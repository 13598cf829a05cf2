"""ctypes binding of oracle/libpbf_oracle.so and oracle/_ref/libref_grid.so.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product path (pbf-sph_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libpbf_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_grid.so")

GS, JACOBI = 0, 1
SORT_STD, SORT_STABLE = 0, 1


class OracleMc(C.Structure):
    """sph::McParams (sph.hpp:82-95); defaults = simpleConfigWith2Cubes' (sph.hpp:179-184)."""
    _fields_ = [("resolution", C.c_double), ("isolevel", C.c_double), ("particle_size", C.c_double),
                ("particle_influence", C.c_double)]

    def __init__(self, resolution=2.0, isolevel=100.0, particle_size=25.0, particle_influence=0.5):
        super().__init__(resolution, isolevel, particle_size, particle_influence)


class OracleParams(C.Structure):
    _fields_ = [
        ("h", C.c_double),
        ("dt", C.c_double),
        ("scale", C.c_double),
        ("iteration", C.c_uint64),
        ("constant_force", C.c_double * 3),
        ("min_bound", C.c_double * 3),
        ("max_bound", C.c_double * 3),
        ("mode", C.c_int32),
        ("sort", C.c_int32),
        ("threads", C.c_int32),
        ("xsph", C.c_int32),
        ("vorticity", C.c_int32),
        ("n_wells", C.c_int32),
        ("wells", C.POINTER(C.c_double)),
    ]


def build(force=False):
    """Compile the checker (make -C oracle). Building the checker is not using it."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.exists("/root/reference/src/sph.hpp") and not os.path.exists(REF_SO)
    ):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


_lib = None
_libs = {}
_ref = None
NATIVE_DIR = os.path.join(ORACLE_DIR, "_native")


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build_native():
    """The cpu_baseline builds BASELINE.md §4 promises: the same restatement at -O3 -march=native (no fast-math)
    and at -Ofast -march=native (the reference's Release flags, CMakeLists.txt:136).  -march=native binds the
    binary to the machine it was compiled on, so these are (re)built on the box that times them (make stamps the
    CPU model).  Returns {"O3": path, "Ofast": path}; raises if the compiler is missing."""
    subprocess.run(["make", "-C", ORACLE_DIR, "native"], check=True, capture_output=True)
    return {k: os.path.join(NATIVE_DIR, f"libpbf_oracle_{k}.so") for k in ("O3", "Ofast")}


def lib(path=None):
    """path=None: the checker build (-O2, no fast-math, no contraction) every parity test uses."""
    global _lib
    if path is not None:
        if path not in _libs:
            _libs[path] = _bind(C.CDLL(path))
        return _libs[path]
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(ORACLE_SO))
    return _lib


def _bind(L):
    L.pbf_oracle_create.restype = C.c_void_p
    L.pbf_oracle_create.argtypes = [C.c_int]
    L.pbf_oracle_destroy.argtypes = [C.c_void_p]
    L.pbf_oracle_set_particles.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6
    L.pbf_oracle_count.restype = C.c_size_t
    L.pbf_oracle_count.argtypes = [C.c_void_p]
    L.pbf_oracle_get_particles.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    for name in ("step", "predict", "sort", "grid_table", "diffuse", "lambda", "delta", "finalise"):
        f = getattr(L, "pbf_oracle_" + name)
        f.argtypes = [C.c_void_p, C.POINTER(OracleParams)]
        f.restype = C.c_int
    for name in ("get_keys", "get_pstar", "get_lambda", "get_table"):
        getattr(L, "pbf_oracle_" + name).argtypes = [C.c_void_p, C.c_void_p]
    L.pbf_oracle_table_size.restype = C.c_size_t
    L.pbf_oracle_table_size.argtypes = [C.c_void_p]
    L.pbf_oracle_get_extent.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pbf_oracle_candidate_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_double)]
    L.pbf_oracle_morton_encode.restype = C.c_uint64
    L.pbf_oracle_morton_encode.argtypes = [C.c_uint64] * 3
    L.pbf_oracle_morton_decode.restype = C.c_uint64
    L.pbf_oracle_morton_decode.argtypes = [C.c_uint64, C.c_int]
    L.pbf_oracle_neighbour_codes.argtypes = [C.c_uint64, C.c_void_p]
    L.pbf_oracle_poly6_factor.restype = C.c_double
    L.pbf_oracle_poly6_factor.argtypes = [C.c_int, C.c_double]
    L.pbf_oracle_spiky_factor.restype = C.c_double
    L.pbf_oracle_spiky_factor.argtypes = [C.c_int, C.c_double]
    L.pbf_oracle_scene_cubes.restype = C.c_size_t
    L.pbf_oracle_scene_cubes.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 5
    L.pbf_oracle_scene_dambreak.restype = C.c_size_t
    L.pbf_oracle_scene_dambreak.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 5 + [C.POINTER(C.c_double)]
    L.pbf_oracle_motion_offset.argtypes = [C.c_int, C.c_uint64, C.c_void_p]
    L.pbf_oracle_set_pow4.argtypes = [C.c_void_p, C.c_int]
    L.pbf_oracle_set_scratch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pbf_oracle_surface.argtypes = [C.c_void_p, C.POINTER(OracleParams), C.POINTER(OracleMc), C.POINTER(C.c_uint64)]
    L.pbf_oracle_surface_from_lattice.argtypes = [C.c_void_p, C.POINTER(OracleParams), C.POINTER(OracleMc), C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.pbf_oracle_get_lattice.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pbf_oracle_get_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for name in ("vorticity", "vorticity_force", "xsph"):
        f = getattr(L, "pbf_oracle_" + name)
        f.argtypes = [C.c_void_p, C.POINTER(OracleParams)]
        f.restype = C.c_int
    L.pbf_oracle_get_vec.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.pbf_oracle_set_vec.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.pbf_oracle_scene_emit.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_size_t, C.c_void_p, C.c_void_p]
    L.pbf_oracle_scene_drain.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.pbf_oracle_query.restype = C.c_size_t
    L.pbf_oracle_query.argtypes = [C.c_void_p, C.POINTER(OracleParams), C.c_void_p, C.c_void_p, C.c_size_t]
    return L


def ref():
    """The reference's own glm-free headers, compiled (oracle/_ref). None if unavailable."""
    global _ref
    if _ref is None:
        build()
        if not os.path.exists(REF_SO):
            return None
        R = C.CDLL(REF_SO)
        R.ref_morton_encode.restype = C.c_uint64
        R.ref_morton_encode.argtypes = [C.c_uint64] * 3
        R.ref_morton_decode.restype = C.c_uint64
        R.ref_morton_decode.argtypes = [C.c_uint64, C.c_int]
        R.ref_grid_index_at_f32.restype = C.c_uint64
        R.ref_grid_index_at_f32.argtypes = [C.c_float] * 4
        R.ref_grid_index_at_f64.restype = C.c_uint64
        R.ref_grid_index_at_f64.argtypes = [C.c_double] * 4
        R.ref_make_grid_table.restype = C.c_uint64
        R.ref_make_grid_table.argtypes = [C.c_uint64] * 4 + [C.c_void_p, C.c_void_p]
        R.ref_foreach_grid.restype = C.c_uint64
        R.ref_foreach_grid.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        for nm in ("ref_poly6_factor_f32", "ref_spiky_factor_f32"):
            getattr(R, nm).restype = C.c_double
            getattr(R, nm).argtypes = [C.c_float]
        for nm in ("ref_poly6_factor_f64", "ref_spiky_factor_f64"):
            getattr(R, nm).restype = C.c_double
            getattr(R, nm).argtypes = [C.c_double]
        for nm in ("ref_scene_cubes_f32", "ref_scene_cubes_f64"):
            getattr(R, nm).restype = C.c_uint64
            getattr(R, nm).argtypes = [C.c_uint64, C.c_uint64, C.c_double] + [C.c_void_p] * 7
        R.ref_motion_f32.argtypes = [C.c_uint64, C.c_void_p]
        R.ref_motion_f64.argtypes = [C.c_uint64, C.c_void_p]
        R.ref_sizeof_partially_advected_f32.restype = C.c_uint64
        _ref = R
    return _ref


def make_params(h=0.1, dt=0.0083 * 1.5, scale=500.0, iteration=4, force=(0.0, 9.8, 0.0), min_bound=(0, 0, 0),
                max_bound=(1000, 1000, 1000), mode=JACOBI, sort=SORT_STABLE, threads=0, xsph=0, vorticity=0,
                wells=None):
    """Defaults restate simpleConfigWith2Cubes (sph.hpp:168-175) with K=4 and h=0.1 (benchmark.cpp:160)."""
    p = OracleParams()
    p.h, p.dt, p.scale, p.iteration = h, dt, scale, iteration
    p.constant_force[:] = force
    p.min_bound[:] = [float(v) for v in min_bound]
    p.max_bound[:] = [float(v) for v in max_bound]
    p.mode, p.sort, p.threads, p.xsph, p.vorticity = mode, sort, threads, xsph, vorticity
    if wells is not None and len(wells):
        w = np.ascontiguousarray(wells, dtype=np.float64).reshape(-1, 4)
        p._wells_keepalive = w
        p.n_wells = w.shape[0]
        p.wells = w.ctypes.data_as(C.POINTER(C.c_double))
    else:
        p.n_wells = 0
        p.wells = None
    return p


class Oracle:
    """Stateful CPU oracle (one per precision)."""

    def __init__(self, fp64=False, device_pow=False, so_path=None):
        """device_pow=True evaluates pow(q, 4) as (q*q)*(q*q) like the HIP kernels do — the ONE
        arithmetic substitution of the device path; default False = std::pow as in ompsph.hpp:240.
        so_path: another build of the same source (build_native(): timing only, never a checker)."""
        self.fp64 = bool(fp64)
        self.dtype = np.float64 if fp64 else np.float32
        self.L = lib(so_path)
        self.h = C.c_void_p(self.L.pbf_oracle_create(int(self.fp64)))
        if device_pow:
            self.L.pbf_oracle_set_pow4(self.h, 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pbf_oracle_destroy(self.h)
            self.h = None

    def set_particles(self, id, type, mass, pos, vel, colour):
        n = len(id)
        id = np.ascontiguousarray(id, np.uint64)
        type = np.ascontiguousarray(type, np.uint8)
        mass = np.ascontiguousarray(mass, self.dtype)
        pos = np.ascontiguousarray(pos, self.dtype).reshape(n, 3)
        vel = np.ascontiguousarray(vel, self.dtype).reshape(n, 3)
        colour = np.ascontiguousarray(colour, self.dtype).reshape(n, 4)
        self.L.pbf_oracle_set_particles(self.h, n, _vp(id), _vp(type), _vp(mass), _vp(pos), _vp(vel), _vp(colour))

    @property
    def n(self):
        return self.L.pbf_oracle_count(self.h)

    def get_particles(self):
        n = self.n
        out = dict(id=np.empty(n, np.uint64), type=np.empty(n, np.uint8), mass=np.empty(n, self.dtype),
                   pos=np.empty((n, 3), self.dtype), vel=np.empty((n, 3), self.dtype),
                   colour=np.empty((n, 4), self.dtype))
        self.L.pbf_oracle_get_particles(self.h, _vp(out["id"]), _vp(out["type"]), _vp(out["mass"]), _vp(out["pos"]),
                                        _vp(out["vel"]), _vp(out["colour"]))
        return out

    def _stage(self, name, p):
        rc = getattr(self.L, "pbf_oracle_" + name)(self.h, C.byref(p))
        assert rc == 0, name
        return self

    def step(self, p):
        return self._stage("step", p)

    def predict(self, p):
        return self._stage("predict", p)

    def sort(self, p):
        return self._stage("sort", p)

    def grid_table(self, p):
        return self._stage("grid_table", p)

    def diffuse(self, p):
        return self._stage("diffuse", p)

    def lambda_(self, p):
        return self._stage("lambda", p)

    def delta(self, p):
        return self._stage("delta", p)

    def finalise(self, p):
        return self._stage("finalise", p)

    def keys(self):
        k = np.empty(self.n, np.uint64)
        self.L.pbf_oracle_get_keys(self.h, _vp(k))
        return k

    def pstar(self):
        a = np.empty((self.n, 3), self.dtype)
        self.L.pbf_oracle_get_pstar(self.h, _vp(a))
        return a

    def lambdas(self):
        a = np.empty(self.n, self.dtype)
        self.L.pbf_oracle_get_lambda(self.h, _vp(a))
        return a

    def set_scratch(self, keys=None, pstar=None, lambdas=None):
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        ps = None if pstar is None else np.ascontiguousarray(pstar, self.dtype)
        la = None if lambdas is None else np.ascontiguousarray(lambdas, self.dtype)
        self.L.pbf_oracle_set_scratch(self.h, _vp(k), _vp(ps), _vp(la))

    def surface(self, p, mc=None, lattice=None):
        """Marching cubes on the current state -> dict(vs, ns, cs, sample, pn, c).  lattice=(sample, pn, c) runs
        the emit stage only on a given lattice."""
        mc = mc or OracleMc()
        nt = C.c_uint64()
        if lattice is None:
            self.L.pbf_oracle_surface(self.h, C.byref(p), C.byref(mc), C.byref(nt))
        else:
            smp = np.ascontiguousarray(lattice[0], np.uint64)
            pn = np.ascontiguousarray(lattice[1], self.dtype)
            cc = np.ascontiguousarray(lattice[2], self.dtype)
            self.L.pbf_oracle_surface_from_lattice(self.h, C.byref(p), C.byref(mc), _vp(smp), _vp(pn), _vp(cc), C.byref(nt))
        n = nt.value
        vs, ns, cs = np.empty((3 * n, 3), self.dtype), np.empty((3 * n, 3), self.dtype), np.empty((3 * n, 4), self.dtype)
        self.L.pbf_oracle_get_mesh(self.h, _vp(vs), _vp(ns), _vp(cs))
        smp = np.zeros(3, np.uint64)
        self.L.pbf_oracle_get_lattice(self.h, _vp(smp), None, None)
        nn = int(smp.prod())
        pn, cc = np.empty((nn, 4), self.dtype), np.empty((nn, 4), self.dtype)
        self.L.pbf_oracle_get_lattice(self.h, _vp(smp), _vp(pn), _vp(cc))
        return dict(vs=vs, ns=ns, cs=cs, sample=smp, pn=pn, c=cc)

    # -- the opt-in extras as stages of their own (slab twin) --
    def vorticity(self, p):
        return self._stage("vorticity", p)

    def vorticity_force(self, p):
        return self._stage("vorticity_force", p)

    def xsph(self, p):
        return self._stage("xsph", p)

    def get_vec(self, which):
        a = np.empty((self.n, 3), self.dtype)
        assert self.L.pbf_oracle_get_vec(self.h, which, _vp(a)) == 0
        return a

    def set_vec(self, which, a):
        a = np.ascontiguousarray(a, self.dtype).reshape(self.n, 3)
        self.L.pbf_oracle_set_vec(self.h, which, _vp(a))

    # -- scene dynamics on the host side of advance() (ompsph.hpp:91-126, 167-186) --
    def emit(self, sources, h=0.1, scale=500.0):
        """sources: list of (tag, centre3, velocity3, colour4, rate)"""
        tags = np.array([s[0] for s in sources], np.uint64)
        rows = np.array([list(s[1]) + list(s[2]) + list(s[3]) + [s[4]] for s in sources], np.float64)
        self.L.pbf_oracle_scene_emit(self.h, h, scale, len(sources), _vp(tags), _vp(rows))
        return self

    def drain(self, drains):
        """drains: list of (centre3, width)"""
        rows = np.array([list(d[0]) + [d[1]] for d in drains], np.float64)
        self.L.pbf_oracle_scene_drain(self.h, len(drains), _vp(rows))
        return self

    def query(self, p, point):
        pt = np.array(point, np.float64)
        out = np.empty(4096, np.uint64)
        k = self.L.pbf_oracle_query(self.h, C.byref(p), _vp(pt), _vp(out), len(out))
        return out[:k].copy()

    def table(self):
        t = np.empty(self.L.pbf_oracle_table_size(self.h), np.uint64)
        self.L.pbf_oracle_get_table(self.h, _vp(t))
        return t

    def extent(self):
        e = np.zeros(3, np.uint64)
        m = np.zeros(3, self.dtype)
        self.L.pbf_oracle_get_extent(self.h, _vp(e), _vp(m))
        return e, m

    def candidate_stats(self):
        mean, mx, within = C.c_double(), C.c_uint64(), C.c_double()
        self.L.pbf_oracle_candidate_stats(self.h, C.byref(mean), C.byref(mx), C.byref(within))
        return mean.value, mx.value, within.value


def scene_cubes(count, fp64=False):
    L = lib()
    dt = np.float64 if fp64 else np.float32
    n = L.pbf_oracle_scene_cubes(int(fp64), count, None, None, None, None, None)
    out = dict(id=np.empty(n, np.uint64), type=np.zeros(n, np.uint8), mass=np.empty(n, dt), pos=np.empty((n, 3), dt),
               vel=np.empty((n, 3), dt), colour=np.empty((n, 4), dt))
    L.pbf_oracle_scene_cubes(int(fp64), count, _vp(out["id"]), _vp(out["mass"]), _vp(out["pos"]), _vp(out["vel"]),
                             _vp(out["colour"]))
    return out


def scene_dambreak(nominal, fp64=False):
    L = lib()
    dt = np.float64 if fp64 else np.float32
    side = C.c_double()
    n = L.pbf_oracle_scene_dambreak(int(fp64), nominal, None, None, None, None, None, C.byref(side))
    out = dict(id=np.empty(n, np.uint64), type=np.zeros(n, np.uint8), mass=np.empty(n, dt), pos=np.empty((n, 3), dt),
               vel=np.empty((n, 3), dt), colour=np.empty((n, 4), dt))
    L.pbf_oracle_scene_dambreak(int(fp64), nominal, _vp(out["id"]), _vp(out["mass"]), _vp(out["pos"]),
                                _vp(out["vel"]), _vp(out["colour"]), C.byref(side))
    return out, side.value


def motion_offset(frame, fp64=False):
    o = np.zeros(3)
    lib().pbf_oracle_motion_offset(int(fp64), frame, _vp(o))
    return o
