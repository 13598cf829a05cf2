"""ctypes binding of include/pbf_hip.h (libpbf_hip.so).  Plumbing only: no numerics here."""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# PBF_HIP_LIB: A/B another build of the same sources (diagnostic; tools/ab_flags.sh)
LIB_PATH = os.environ.get("PBF_HIP_LIB") or os.path.join(PKG_DIR, "libpbf_hip.so")

ABI_VERSION = 1
FLAG_STAGE_TIMING = 1 << 0
FLAG_FAST_MATH = 1 << 1
FLAG_NO_LDS = 1 << 2

BUF_KEYS, BUF_TABLE, BUF_PSTAR = 0, 1, 2
BUF_NBR_COUNT = 3
BUF_OMEGA = 4


class PbfError(RuntimeError):
    pass


class Desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("fp64", C.c_int32),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
        ("h", C.c_double),
        ("stream", C.c_void_p),
    ]


class Params(C.Structure):
    _fields_ = [
        ("dt", C.c_double),
        ("scale", C.c_double),
        ("iteration", C.c_uint64),
        ("constant_force", C.c_double * 3),
        ("min_bound", C.c_double * 3),
        ("max_bound", C.c_double * 3),
        ("n_wells", C.c_int32),
        ("wells", C.POINTER(C.c_double)),
        ("xsph", C.c_int32),
        ("vorticity", C.c_int32),
    ]

    def copy(self):
        p = Params()
        C.memmove(C.byref(p), C.byref(self), C.sizeof(Params))
        if hasattr(self, "_wells_keepalive"):
            p._wells_keepalive = self._wells_keepalive
        return p

    def set_wells(self, wells):
        if wells is None or len(wells) == 0:
            self.n_wells, self.wells = 0, None
            return self
        w = np.ascontiguousarray(wells, dtype=np.float64).reshape(-1, 4)
        self._wells_keepalive = w
        self.n_wells = w.shape[0]
        self.wells = w.ctypes.data_as(C.POINTER(C.c_double))
        return self


class McParams(C.Structure):
    """sph::McParams (sph.hpp:82-95); defaults = simpleConfigWith2Cubes' (sph.hpp:179-184)."""
    _fields_ = [("resolution", C.c_double), ("isolevel", C.c_double), ("particle_size", C.c_double),
                ("particle_influence", C.c_double)]

    def __init__(self, resolution=2.0, isolevel=100.0, particle_size=25.0, particle_influence=0.5):
        super().__init__(resolution, isolevel, particle_size, particle_influence)


class SlabCut(C.Structure):
    _fields_ = [("xlo", C.c_uint32), ("xhi", C.c_uint32), ("has_left", C.c_int32), ("has_right", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                          C.c_void_p, C.c_size_t)


class AosLayout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("stride", "off_id", "off_type", "off_mass", "off_pos", "off_vel",
                                          "off_colour")]


def build(force=False):
    """Compile libpbf_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        r = subprocess.run(["make", "-C", PKG_DIR], capture_output=True, text=True)
        if r.returncode != 0:
            raise PbfError("building libpbf_hip.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None

_SIGS = {
    "pbf_abi_version": (C.c_int, []),
    "pbf_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pbf_create": (C.c_int, [C.POINTER(Desc), C.POINTER(C.c_void_p)]),
    "pbf_destroy": (None, [C.c_void_p]),
    "pbf_last_error": (C.c_char_p, [C.c_void_p]),
    "pbf_upload": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6),
    "pbf_download": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "pbf_count": (C.c_size_t, [C.c_void_p]),
    "pbf_upload_aos": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(AosLayout)]),
    "pbf_download_aos": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(AosLayout)]),
    "pbf_download_aos_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(AosLayout)]),
    "pbf_download_aos_end": (C.c_int, [C.c_void_p]),
    "pbf_step": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_steps": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_uint32]),
    "pbf_sync": (C.c_int, [C.c_void_p]),
    "pbf_graph_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pbf_stage_predict": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_stage_sort": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_stage_diffuse": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_stage_lambda": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_stage_delta": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_stage_finalise": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_read_buffer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "pbf_table_size": (C.c_size_t, [C.c_void_p]),
    "pbf_selftest_math": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pbf_grid_extent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pbf_stage_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                  C.c_int]),
    "pbf_reset_stage_times": (C.c_int, [C.c_void_p]),
    "pbf_surface": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(C.c_uint64)]),
    "pbf_download_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pbf_map_mesh": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "pbf_read_lattice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pbf_reserve": (C.c_int, [C.c_void_p, C.c_size_t]),
    "pbf_slab_configure": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "pbf_slab_record_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "pbf_slab_migrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "pbf_slab_add_migrants": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "pbf_slab_ghosts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "pbf_slab_add_ghosts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "pbf_slab_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pbf_slab_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pbf_slab_finish": (C.c_int, [C.c_void_p]),
    "pbf_owned_count": (C.c_size_t, [C.c_void_p]),
    "pbf_slab_column_histogram": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pbf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pbf_comm_create_rccl": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pbf_comm_create_host_callback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pbf_comm_destroy": (None, [C.c_void_p]),
    "pbf_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "pbf_comm_rounds": (C.c_uint64, [C.c_void_p]),
    "pbf_comm_allreduce_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pbf_slab_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "pbf_slab_set_cuts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pbf_slab_step": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pbf_slab_steps": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_uint32]),
    "pbf_slab_host_syncs": (C.c_uint64, [C.c_void_p]),
    "pbf_scene_cubes": (C.c_size_t, [C.c_int, C.c_size_t] + [C.c_void_p] * 6),
    "pbf_scene_dambreak": (C.c_size_t, [C.c_int, C.c_size_t] + [C.c_void_p] * 6 + [C.POINTER(C.c_double)]),
    "pbf_apply_motion": (None, [C.c_int, C.POINTER(Params), C.c_uint64, C.POINTER(Params)]),
    "pbf_default_params": (None, [C.c_uint64, C.c_double, C.POINTER(Params)]),
}


def exported_symbols():
    """Every entry point include/pbf_hip.h declares (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PbfError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        if L.pbf_abi_version() != ABI_VERSION:
            raise PbfError("libpbf_hip.so ABI version mismatch")
        _lib = L
    return _lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params(iteration=4, box_side=1000.0):
    p = Params()
    lib().pbf_default_params(iteration, box_side, C.byref(p))
    return p


def apply_motion(base, frame, fp64=False):
    out = Params()
    lib().pbf_apply_motion(int(fp64), C.byref(base), frame, C.byref(out))
    return out


def _scene(fn, fp64, *args):
    dt = np.float64 if fp64 else np.float32
    n = fn(int(fp64), *args, None, None, None, None, None, None)
    out = dict(id=np.empty(n, np.uint64), type=np.empty(n, np.uint8), mass=np.empty(n, dt), pos=np.empty((n, 3), dt),
               vel=np.empty((n, 3), dt), colour=np.empty((n, 4), dt))
    return n, out


def scene_cubes(count, fp64=False):
    """simpleConfigWith2Cubes particles (sph.hpp:160-166)."""
    L = lib()
    n, o = _scene(lambda f, c, *a: L.pbf_scene_cubes(f, c, *a), fp64, count)
    L.pbf_scene_cubes(int(fp64), count, _vp(o["id"]), _vp(o["type"]), _vp(o["mass"]), _vp(o["pos"]), _vp(o["vel"]),
                      _vp(o["colour"]))
    return o


def scene_dambreak(nominal, fp64=False):
    """Dam-break column. Returns (particles, box_side)."""
    L = lib()
    side = C.c_double()
    n, o = _scene(lambda f, c, *a: L.pbf_scene_dambreak(f, c, *a, C.byref(side)), fp64, nominal)
    L.pbf_scene_dambreak(int(fp64), nominal, _vp(o["id"]), _vp(o["type"]), _vp(o["mass"]), _vp(o["pos"]),
                         _vp(o["vel"]), _vp(o["colour"]), C.byref(side))
    return o, side.value


class Solver:
    """Mirror of sph::hip_impl::Solver<T,N> (host/hipsph.hpp) over the C ABI: ctor takes h like
    omp_impl::Solver(N h) (ompsph.hpp:83); state is device-resident between steps."""

    def __init__(self, h=0.1, fp64=False, device=0, flags=0, stream=None):
        self.L = lib()
        self.fp64 = bool(fp64)
        self.dtype = np.float64 if fp64 else np.float32
        d = Desc(ABI_VERSION, int(self.fp64), device, flags, h, stream)
        self.ctx = C.c_void_p()
        rc = self.L.pbf_create(C.byref(d), C.byref(self.ctx))
        if rc != 0:
            msg = self.L.pbf_last_error(None)
            self.ctx = None
            raise PbfError(f"pbf_create failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "ctx", None):
            self.L.pbf_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            msg = self.L.pbf_last_error(self.ctx)
            raise PbfError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    @property
    def n(self):
        return self.L.pbf_count(self.ctx)

    def upload(self, id, type, mass, pos, vel, colour):
        n = len(id)
        a = [np.ascontiguousarray(id, np.uint64), np.ascontiguousarray(type, np.uint8),
             np.ascontiguousarray(mass, self.dtype), np.ascontiguousarray(pos, self.dtype).reshape(n, 3),
             np.ascontiguousarray(vel, self.dtype).reshape(n, 3),
             np.ascontiguousarray(colour, self.dtype).reshape(n, 4)]
        self._chk(self.L.pbf_upload(self.ctx, n, *[_vp(x) for x in a]), "pbf_upload")
        return self

    def download(self):
        n = self.n
        o = dict(id=np.empty(n, np.uint64), type=np.empty(n, np.uint8), mass=np.empty(n, self.dtype),
                 pos=np.empty((n, 3), self.dtype), vel=np.empty((n, 3), self.dtype),
                 colour=np.empty((n, 4), self.dtype))
        self._chk(self.L.pbf_download(self.ctx, _vp(o["id"]), _vp(o["type"]), _vp(o["mass"]), _vp(o["pos"]),
                                      _vp(o["vel"]), _vp(o["colour"])), "pbf_download")
        return o

    def step(self, p):
        self._chk(self.L.pbf_step(self.ctx, C.byref(p)), "pbf_step")
        return self

    def steps(self, p, count):
        self._chk(self.L.pbf_steps(self.ctx, C.byref(p), count), "pbf_steps")
        return self

    def set_option(self, name, value):
        self._chk(self.L.pbf_set_option(self.ctx, name.encode(), int(value)), "pbf_set_option")
        return self

    def sync(self):
        self._chk(self.L.pbf_sync(self.ctx), "pbf_sync")
        return self

    def graph_stats(self):
        """(graphs captured, graph replays, graphs still enabled) of pbf_steps' hipGraph path"""
        o = np.zeros(3, np.uint64)
        self._chk(self.L.pbf_graph_stats(self.ctx, _vp(o)), "pbf_graph_stats")
        return int(o[0]), int(o[1]), bool(o[2])

    def stage(self, name, p):
        self._chk(getattr(self.L, "pbf_stage_" + name)(self.ctx, C.byref(p)), "pbf_stage_" + name)
        return self

    def nbr_counts(self):
        """Neighbour-list length per particle of the last list build (0xFFFFFFFF: overflowed row)."""
        k = np.empty(self.n, np.uint32)
        self._chk(self.L.pbf_read_buffer(self.ctx, BUF_NBR_COUNT, _vp(k), k.nbytes), "read neighbour counts")
        # raw word = length | chunk << 8 (the chunk of the list's slots beyond the 40 in the rows), or 0xFFFFFFFF
        return np.where(k == 0xFFFFFFFF, k, k & 0xFF).astype(np.uint32)

    def keys(self):
        k = np.empty(self.n, np.uint32)
        self._chk(self.L.pbf_read_buffer(self.ctx, BUF_KEYS, _vp(k), k.nbytes), "read keys")
        return k

    def table(self):
        t = np.empty(self.L.pbf_table_size(self.ctx), np.uint32)
        self._chk(self.L.pbf_read_buffer(self.ctx, BUF_TABLE, _vp(t), t.nbytes), "read table")
        return t

    def pstar(self):
        """(n,4): pStar.xyz, lambda"""
        a = np.empty((self.n, 4), self.dtype)
        self._chk(self.L.pbf_read_buffer(self.ctx, BUF_PSTAR, _vp(a), a.nbytes), "read pstar")
        return a

    def omega(self):
        """(n,3): vorticity estimate of the last step run with p.vorticity (opt-in extra), device order"""
        a = np.empty((self.n, 4), self.dtype)
        self._chk(self.L.pbf_read_buffer(self.ctx, BUF_OMEGA, _vp(a), a.nbytes), "read omega")
        return a[:, :3].copy()

    def surface(self, p, mc=None):
        """Marching cubes on the state the last step left -> dict(vs, ns, cs, sample, pn, c)."""
        mc = mc or McParams()
        nt = C.c_uint64()
        self._chk(self.L.pbf_surface(self.ctx, C.byref(p), C.byref(mc), C.byref(nt)), "pbf_surface")
        n = nt.value
        vs, ns, cs = np.empty((3 * n, 3), self.dtype), np.empty((3 * n, 3), self.dtype), np.empty((3 * n, 4), self.dtype)
        self._chk(self.L.pbf_download_mesh(self.ctx, _vp(vs), _vp(ns), _vp(cs)), "pbf_download_mesh")
        smp = np.zeros(3, np.uint64)
        self._chk(self.L.pbf_read_lattice(self.ctx, _vp(smp), None, None), "pbf_read_lattice")
        nn = int(smp.prod())
        pn, cc = np.empty((nn, 4), self.dtype), np.empty((nn, 4), self.dtype)
        self._chk(self.L.pbf_read_lattice(self.ctx, _vp(smp), _vp(pn), _vp(cc)), "pbf_read_lattice")
        return dict(vs=vs, ns=ns, cs=cs, sample=smp, pn=pn, c=cc)

    def extent(self):
        e = np.zeros(3, np.uint64)
        m = np.zeros(3, np.float64)
        self._chk(self.L.pbf_grid_extent(self.ctx, _vp(e), _vp(m)), "grid_extent")
        return e, m

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        calls = (C.c_uint64 * 16)()
        k = self._chk(self.L.pbf_stage_times(self.ctx, names, ms, calls, 16), "pbf_stage_times")
        return {names[i].decode(): (ms[i], calls[i]) for i in range(k)}

    def reset_stage_times(self):
        self._chk(self.L.pbf_reset_stage_times(self.ctx), "pbf_reset_stage_times")
