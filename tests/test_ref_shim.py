"""The product's C++ shim compiled against the REFERENCE's own src/sph.hpp (INTEGRATION.md's recipe
"copy hipsph.hpp + pbf_hip.h next to sph.hpp, add a case" done for real): oracle/ref_shim.cpp -> oracle/_ref/ref_shim_check
instantiates sph::hip_impl::Solver<size_t, float|double, V> behind the reference's abstract sph::Solver<T, N, V>
(src/sph.hpp:119-125) and drives it the way runN does (src/benchmark.cpp:22-58) with the reference's own scene factory
and box motion.  Built here (needs /root/reference); the binary travels to the GPU box like the other oracle/_ref files."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_shim_check")
REF_HEADER = "/root/reference/src/sph.hpp"


def test_shim_compiles_and_links_against_the_reference_header(pkg):
    if not os.path.exists(REF_HEADER):
        pytest.skip("needs /root/reference (this container); the built binary is what travels")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/ref_shim_check"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(BIN), r.stdout + r.stderr
    # built from the reference's header, not ours: the TU #errors when PBF_SPH_HAS_VEC (host/sph.hpp) is visible
    assert os.path.getmtime(BIN) >= os.path.getmtime(os.path.join(ROOT, "pbf-sph_amd", "host", "hipsph.hpp"))
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run is covered by the gpu-marked test")
    r = subprocess.run([BIN, "0", "1", "1024", "0"], capture_output=True, text=True, timeout=120)
    # no GPU here: the reference-side caller sees the loud failure, never a CPU fallback
    assert r.returncode == 3 and "no CPU fallback" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("fp64,surface", [(0, 0), (1, 0), (0, 1)])
def test_reference_side_caller_gets_the_capi_bits(pkg, tmp_path, fp64, surface):
    """advance() through the reference's interface == the C-ABI path from Python, bit for bit, after three frames of the
    stock moving-box scene; with config.surface set like benchmark.cpp:29 the mesh comes back through Result::mesh."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/ref_shim_check was not built (needs /root/reference at build time)")
    dump = str(tmp_path / "dump.bin")
    r = subprocess.run([BIN, str(fp64), "3", "2048", str(surface), dump], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok n=2000"), r.stdout + r.stderr
    dt = np.float64 if fp64 else np.float32
    raw = open(dump, "rb").read()
    n, vertices = (int(v) for v in np.frombuffer(raw[:16], np.uint64))
    rec = np.dtype([("id", "<u8"), ("pos", dt, 3), ("vel", dt, 3), ("colour", dt, 4)])
    g = np.frombuffer(raw[16:], rec)
    assert n == 2000 and len(g) == n
    sc = pkg.scene_cubes(2048, bool(fp64))
    s = pkg.Solver(h=0.1, fp64=bool(fp64))
    s.upload(**sc)
    base = pkg.default_params(4, 1000.0)
    for frame in range(3):
        p = pkg.apply_motion(base, frame, bool(fp64))
        s.step(p)
    w = s.download()
    for k in ("id", "pos", "vel", "colour"):
        assert np.array_equal(g[k], w[k]), k
    if surface:
        m = s.surface(p)
        assert vertices == len(m["vs"]) > 0
    else:
        assert vertices == 0
