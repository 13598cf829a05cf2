"""Slab decomposition on the GPU: several ranks share cuda:0 (gloo, host-staged wire buffers — the
production transport is RCCL, exercised by bench.py --gpus N on a multi-GPU node).
  engine "hipc" = the product path: pbf_slab_step inside libpbf_hip.so (whole step + exchanges behind the C ABI,
                  host-callback transport here); engine "hip" = the same kernels sequenced by slab.py's Python driver.
  * HIP slabs == oracle-engine slabs, bit for bit (same protocol, same ordering, same arithmetic);
  * HIP slabs == single-GPU run to summation-order noise; nothing lost or duplicated;
  * 2 + 2K exchange rounds per step; load-balance re-cuts agree with the CPU twin; RCCL communicator set-up."""
import json
import os

import numpy as np
import pytest

from test_slab_cpu import launch, merged

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("engine,world,cuts", [("hipc", 2, "x:210"), ("hipc", 3, "x:210,700"), ("hip", 2, "x:210")])
def test_hip_slabs_bit_exact_vs_oracle_slabs(pkg, tmp_path, engine, world, cuts):
    args = ("--scene", "cubes2048", "--steps", "5", "--cuts", cuts)
    hip = launch(world, str(tmp_path / "hip"), "--engine", engine, *args)
    ora = launch(world, str(tmp_path / "ora"), "--engine", "oracle", *args)
    for r in range(world):
        for k in ("id", "pos", "vel", "colour", "type"):
            assert np.array_equal(hip[r][k], ora[r][k]), (r, k)
        if engine == "hip":
            assert int(hip[r]["ghosts"]) == int(ora[r]["ghosts"]) > 0
        else:  # inside the library: 2 assembly rounds + 2K field refreshes per step, nothing else
            assert int(hip[r]["exchanges"]) == 5 * (2 + 2 * 4), int(hip[r]["exchanges"])
    # against ONE solver on the whole scene
    sc = pkg.scene_cubes(2048)
    s = pkg.Solver(h=0.1)
    s.upload(**sc)
    s.steps(pkg.default_params(4, 1000.0), 5)
    w = s.download()
    wo = np.argsort(w["id"], kind="stable")
    got = merged(hip)
    assert np.array_equal(got["id"], w["id"][wo])
    d = np.linalg.norm(got["pos"].astype(np.float64) - w["pos"][wo], axis=1)
    assert d.max() <= 2e-2 and d.mean() <= 1e-4, (d.max(), d.mean())


def test_hip_slabs_small_first_message(pkg, tmp_path):
    """First assembly message of only 16 records: the ghost copies (hundreds per boundary column) do not fit, so every
    step takes the second, exactly sized exchange as well — same particles, same bits as the CPU twin."""
    args = ("--scene", "cubes2048", "--steps", "4", "--cuts", "x:210,700")
    hip = launch(3, str(tmp_path / "hip"), "--engine", "hipc", "--chunk", "16", *args)
    ora = launch(3, str(tmp_path / "ora"), "--engine", "oracle", *args)
    for r in range(3):
        for k in ("id", "pos", "vel", "colour", "type"):
            assert np.array_equal(hip[r][k], ora[r][k]), (r, k)
    assert int(hip[1]["exchanges"]) > 4 * (2 + 2 * 4)   # the remainder rounds happened


@pytest.mark.parametrize("fp64", [False, True])
def test_hip_slabs_xsph_vorticity_bit_exact(pkg, tmp_path, fp64):
    """XSPH + vorticity confinement in slab mode (pbf_slab_step: three more field refreshes — velocity, vorticity,
    velocity — around the three gather ops): the same bits as the CPU twin, 3 ranks, and 2 + 2K + 3 rounds per step."""
    args = ("--scene", "cubes2048", "--steps", "4", "--cuts", "x:210,700", "--xsph", "1", "--vorticity", "1") + \
        (("--fp64",) if fp64 else ())
    hip = launch(3, str(tmp_path / "hip"), "--engine", "hipc", *args)
    ora = launch(3, str(tmp_path / "ora"), "--engine", "oracle", *args)
    for r in range(3):
        for k in ("id", "pos", "vel", "colour", "type"):
            assert np.array_equal(hip[r][k], ora[r][k]), (r, k)
    assert int(hip[1]["exchanges"]) == 4 * (2 + 2 * 4 + 3)


def test_hip_slabs_fp64_bit_exact(pkg, tmp_path):
    args = ("--scene", "cubes2048", "--steps", "4", "--cuts", "x:210", "--fp64")
    hip = launch(2, str(tmp_path / "hip"), "--engine", "hipc", *args)
    ora = launch(2, str(tmp_path / "ora"), "--engine", "oracle", *args)
    for r in range(2):
        assert hip[r]["pos"].dtype == np.float64
        for k in ("id", "pos", "vel", "colour", "type"):
            assert np.array_equal(hip[r][k], ora[r][k]), (r, k)


@pytest.mark.parametrize("engine", ["hipc", "hip"])
def test_hip_slabs_migration(pkg, tmp_path, engine):
    args = ("--scene", "dam8192", "--steps", "30", "--iteration", "2", "--cuts", "x:150")
    hip = launch(2, str(tmp_path / "hip"), "--engine", engine, *args)
    got = merged(hip)
    sc, side = pkg.scene_dambreak(8192)
    assert np.array_equal(got["id"], np.sort(sc["id"]))
    assert np.isfinite(got["pos"]).all() and got["pos"].min() >= 0 and got["pos"].max() <= side
    # same protocol on the CPU engine: bit-exact, migrants included (a free-running comparison with ONE
    # solver is meaningless after 30 violent steps: summation-order noise grows chaotically)
    ora = launch(2, str(tmp_path / "ora"), "--engine", "oracle", *args)
    for r in range(2):
        for k in ("id", "pos", "vel", "colour"):
            assert np.array_equal(hip[r][k], ora[r][k]), (r, k)
        if engine == "hip":
            assert int(hip[r]["migrated"]) == int(ora[r]["migrated"]) > 0
    assert sum(int(p["migrated"]) for p in ora) > 0


def test_hip_slabs_rebalance_matches_cpu_twin(pkg, tmp_path):
    """Load balance: cuts start badly placed and are moved every 2 steps from the all-reduced column histogram —
    the library path re-cuts exactly like the CPU twin (same cuts, same particles, same bits)."""
    args = ("--scene", "dam8192", "--steps", "12", "--iteration", "2", "--cuts", "x:400,700", "--rebalance", "2")
    hip = launch(3, str(tmp_path / "hip"), "--engine", "hipc", *args)
    ora = launch(3, str(tmp_path / "ora"), "--engine", "oracle", *args)
    assert int(hip[0]["recuts"]) > 0
    for r in range(3):
        assert np.array_equal(hip[r]["cuts"], ora[r]["cuts"])
        for k in ("id", "pos", "vel", "colour"):
            assert np.array_equal(hip[r][k], ora[r][k]), (r, k)
    sc, _ = pkg.scene_dambreak(8192)
    assert np.array_equal(merged(hip)["id"], np.sort(sc["id"]))


def test_bench_strong_scaling_rehearsal(tmp_path):
    """bench.py --gpus 4 as the driver would start it without torchrun (it launches its own ranks), strong scaling =
    ONE column over 4 particle-balanced slabs with re-cuts; the 4 ranks share this GPU through the host-callback
    transport (PBF_BENCH_BACKEND=gloo).  The JSON line is the only thing on stdout; nothing lost; load balanced."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "10", "--warmup", "5",
                        "--settle-to", "60", "--particles", "262144", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, PBF_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["particles"] == 250000
    assert "max rank load 1.0" in d["config"]["parallelism"] or "max rank load 1.1" in d["config"]["parallelism"]
    assert "10 exchange rounds per step" in d["config"]["parallelism"] or "11 exchange rounds" in d["config"]["parallelism"]


RCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.path.join({root!r}, "tests"))
if {torch_first}:
    import torch            # like bench.py: PyTorch (its bundled HIP runtime and librccl) is in the process first
    torch.zeros(1, device="cuda")
import numpy as np
from conftest import load_package
pkg = load_package()
from pbf_sph_amd import slab
sc, side = pkg.scene_dambreak(8192)
p = pkg.default_params(4, side)
a = pkg.Solver(h=0.1)
a.upload(**sc)
drv = slab.CSlabSolver(a, None, None, 0, 1, [0, 1024], 1024, 1024, transport="rccl")
drv.steps(p, 3)
b = pkg.Solver(h=0.1)
b.upload(**sc)
b.steps(p, 3)
ga, gb = a.download(), b.download()
assert all(np.array_equal(ga[k], gb[k]) for k in ("id", "pos", "vel", "colour"))
assert drv.rounds == 0
drv.close()
print("RCCL-OK")
"""


@pytest.mark.parametrize("torch_first", [True, False])
def test_rccl_communicator_single_rank(torch_first):
    """The RCCL transport's set-up path on this one-GPU box, in a fresh process: librccl resolves (PyTorch's copy when
    PyTorch is already in the process — bench.py's situation — else ROCm's: the C++ CLI's), ncclCommInitRank(1 rank)
    succeeds, a slab step through it equals a plain step (no neighbours => no exchange rounds).  (Loading
    libpbf_hip.so BEFORE PyTorch would bind PyTorch's librccl to ROCm's HIP runtime — two ROCm versions in one
    process; bench.py imports torch first.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", RCCL_SCRIPT.format(root=root, torch_first=torch_first)], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, NCCL_DEBUG="WARN"))
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


BAD_ORDER_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import load_package
pkg = load_package()
a = pkg.Solver(h=0.1)          # libpbf_hip.so (and ROCm's HIP runtime) first ...
import torch                   # ... then PyTorch with the HIP runtime and librccl it bundles (in this order PyTorch's own
                               # runtime finds no device any more — "No HIP GPUs are available" — so nothing of torch's is used)
from pbf_sph_amd import slab
sc, side = pkg.scene_dambreak(8192)
a.upload(**sc)
try:
    slab.CSlabSolver(a, None, None, 0, 1, [0, 1024], 1024, 1024, transport="rccl")
    print("CREATED")
except RuntimeError as e:
    print("REFUSED:", e)
"""


def test_rccl_refuses_a_second_hip_runtime():
    """libpbf_hip.so loaded BEFORE PyTorch leaves two HIP runtimes in the process and RCCL then fails
    somewhere inside ncclCommInitRank.  The library now checks that the librccl it binds drives the same libamdhip64 as
    its own HIP calls and refuses with the reason (or, where the image's PyTorch shares ROCm's runtime, simply works)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", BAD_ORDER_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "CREATED" in r.stdout or ("REFUSED:" in r.stdout and "two HIP runtimes" in r.stdout), out[-3000:]
    assert "unhandled cuda error" not in out


def test_rccl_send_recv_between_two_gpus():
    """The ncclSend / ncclRecv path itself needs two GPUs (RCCL refuses two ranks on one device): bench.py --gpus 2 — its
    own two ranks, one per GPU, ONE 256 K column cut into two slabs, ghost exchange over xGMI inside pbf_slab_step — and
    its built-in conservation check (no particle lost or duplicated across the cut).  Skips itself on a one-GPU box."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the driver's multi-GPU node); this box has %d" % torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "5",
                        "--settle-to", "40", "--particles", "262144", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert d["n_gpus"] == 2 and d["config"]["particles"] == 250000
    assert "ncclSend/ncclRecv" in d["config"]["parallelism"] and "10 exchange rounds per step" in d["config"]["parallelism"]
