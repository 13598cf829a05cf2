"""pbf-sph_amd — MI355X-native PBF-SPH hot path (HIP kernels behind a C ABI).

The product is native: `libpbf_hip.so` (csrc/, include/pbf_hip.h) plus the C++17 host shim in
host/ that mirrors the reference's `sph::Solver<T,N>::advance()` surface and benchmark CLI.
This Python module is only the ctypes plumbing tests/ and bench.py use to drive the C ABI; it
contains no numerics and no CPU fallback — if the HIP library or a gfx950 device is missing,
`Solver()` raises.

The directory name has a hyphen (it mirrors the reference repo's name), so import it with
`load_package()` from tests/conftest.py / bench.py, or via importlib by path.
"""
from .capi import (  # noqa: F401
    FLAG_FAST_MATH,
    FLAG_NO_LDS,
    FLAG_STAGE_TIMING,
    LIB_PATH,
    McParams,
    Params,
    PbfError,
    SlabCut,
    Solver,
    apply_motion,
    build,
    default_params,
    lib,
    scene_cubes,
    scene_dambreak,
)
