"""Marching-cubes surface (reference src/omp/ompsph.hpp:277-477).

CPU: the oracle's restatement produces a closed, consistently oriented mesh.
GPU: field kernel vs oracle within libm noise (pow), count + emit kernels bit-exact against the oracle's emit
stage fed with the DEVICE lattice, triangle counts of the two full pipelines agree, fp32 and fp64.
Parity vs the reference binary is unpinned (no goldens, OpenMP backend unbuildable), and the case tables are
our own (tools/gen_mc_tables.py), so triangle counts need not equal the reference's."""
import numpy as np
import pytest


def mesh_is_closed_and_oriented(vs, quantum):
    q = np.round(vs.astype(np.float64) / quantum).astype(np.int64)
    keys = (q[:, 0] << 42) + (q[:, 1] << 21) + q[:, 2]
    tri = keys.reshape(-1, 3)
    directed = {}
    for a, b, c in tri:
        if a == b or b == c or a == c:
            continue  # degenerate sliver (an intersection exactly on a lattice node)
        for e in ((a, b), (b, c), (c, a)):
            directed[e] = directed.get(e, 0) + 1
    bad = sum(1 for (a, b), n in directed.items() if directed.get((b, a), 0) != n)
    return bad, len(directed)


def test_oracle_surface_closed(oracle):
    s = oracle.scene_cubes(2048)
    o = oracle.Oracle(False)
    o.set_particles(**s)
    p = oracle.make_params(threads=4)
    for _ in range(3):
        o.step(p)
    m = o.surface(p)
    assert list(m["sample"]) == [49, 49, 49]  # floor(24 * 2) + 1
    assert len(m["vs"]) % 3 == 0 and len(m["vs"]) // 3 > 2000
    assert np.isfinite(m["vs"]).all() and np.isfinite(m["ns"]).all() and np.isfinite(m["cs"]).all()
    assert m["vs"].min() >= -100 and m["vs"].max() <= 1100
    bad, total = mesh_is_closed_and_oriented(m["vs"], 0.05)
    assert bad <= total * 0.002, (bad, total)  # fp: the same lattice edge is interpolated from either end
    # vertex normals are interpolated between unit node normals: never longer than 1
    ln = np.linalg.norm(m["ns"].astype(np.float64), axis=1)
    assert ln.max() <= 1 + 1e-5 and ln.min() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("fp64", [False, True])
def test_surface_gpu_vs_oracle(pkg, oracle, fp64):
    sc = pkg.scene_cubes(8192, fp64)
    s = pkg.Solver(h=0.1, fp64=fp64)
    s.upload(**sc)
    o = oracle.Oracle(fp64, device_pow=True)
    o.set_particles(**sc)
    p = pkg.default_params(4, 1000.0)
    q = oracle.make_params(mode=oracle.JACOBI, sort=oracle.SORT_STABLE, threads=4)
    for _ in range(4):
        s.step(p)
        o.step(q)
    g = s.surface(p, pkg.McParams())
    w = o.surface(q, oracle.OracleMc())
    assert np.array_equal(g["sample"], w["sample"])
    # field: same arithmetic except libm pow(len, 0.5) -> relative noise only; NaN (0/0 colours) in the same places
    tol = 2e-6 if not fp64 else 1e-13
    np.testing.assert_allclose(g["pn"][:, 0], w["pn"][:, 0], rtol=tol, atol=tol * 100)
    assert np.array_equal(np.isnan(g["c"]), np.isnan(w["c"]))
    np.testing.assert_allclose(np.nan_to_num(g["pn"]), np.nan_to_num(w["pn"]), rtol=1e-4, atol=1e-4)
    # count + emit: bit-exact against the oracle's emit stage on the device lattice
    e = o.surface(q, oracle.OracleMc(), lattice=(g["sample"], g["pn"], g["c"]))
    assert len(e["vs"]) == len(g["vs"]) > 3 * 4000
    for k in ("vs", "ns", "cs"):
        assert np.array_equal(g[k], e[k], equal_nan=True), k
    # the two full pipelines agree on the triangle count (a node within 1e-6 of the isolevel may flip a case)
    assert abs(len(g["vs"]) - len(w["vs"])) <= 0.005 * len(w["vs"]) + 30
    bad, total = mesh_is_closed_and_oriented(g["vs"], 0.05)
    assert bad <= total * 0.002, (bad, total)


@pytest.mark.gpu
def test_surface_through_cli(tmp_path):
    """The stock driver runs with the surface on (benchmark.cpp:29) and prints its vertex count."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "pbf-sph_amd", "benchmark"), "-n", "3", "-w", "2", "-o", str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    n = int(re.search(r"Final Vertex count   : (\d+)", r.stdout).group(1))
    assert n > 3000 and n % 3 == 0
    obj = open(tmp_path / "out" / "mesh.obj").read().split("\n")
    assert sum(1 for l in obj if l.startswith("v ")) == n and sum(1 for l in obj if l.startswith("f ")) == n // 3
    r2 = subprocess.run([os.path.join(root, "pbf-sph_amd", "benchmark"), "-n", "2", "-w", "1", "--no-surface", "-o", ""],
                        capture_output=True, text=True, timeout=300)
    assert "Final Vertex count   : 0" in r2.stdout
