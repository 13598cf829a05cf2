"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle in Jacobi /
stable-sort mode on the same inputs, stage by stage and end to end.

Bars (north_star: "within a stated float tolerance"; integer work bit-exact):
  * keys, cell table, sorted permutation: bit-exact;
  * predict / finalise streams: bit-exact (contraction is off, divide and sqrt are IEEE);
  * lambda: |d lambda| <= 2e-6 * max|lambda| (fp32), 1e-13 (fp64) — the pair sums run in the
    oracle's candidate order; only pow(x,4) vs (x*x)*(x*x) and libm differ;
  * positions after 1 step: <= 1e-3 world units fp32 (box = 1000; SURVEY §8c allows 5e-3),
    <= 1e-9 fp64; after 3 steps: <= 2e-2 fp32 (chaotic growth), 1e-7 fp64.
Parity vs the *reference* is unpinned for these stages (see oracle/pbf_oracle.h).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def by_id(d):
    o = np.argsort(d["id"], kind="stable")
    return {k: v[o] for k, v in d.items()}


def mk(pkg, oracle, scene, fp64, flags=0, **kw):
    s = pkg.Solver(h=0.1, fp64=fp64, flags=flags)
    s.upload(**scene)
    o = oracle.Oracle(fp64)
    o.set_particles(**scene)
    return s, o


def params_pair(pkg, oracle, iteration=4, side=1000.0, wells=None):
    p = pkg.default_params(iteration, side)
    if wells is not None:
        p.set_wells(wells)
    q = oracle.make_params(iteration=iteration, max_bound=(side, side, side), mode=oracle.JACOBI,
                           sort=oracle.SORT_STABLE, wells=wells)
    return p, q


SCENES = ["cubes8192", "dam8192"]


def get_scene(pkg, name, fp64):
    if name == "cubes8192":
        return pkg.scene_cubes(8192, fp64), 1000.0
    if name == "cubes1024":
        return pkg.scene_cubes(1024, fp64), 1000.0
    if name == "dam8192":
        return pkg.scene_dambreak(8192, fp64)
    raise KeyError(name)


@pytest.mark.parametrize("fp64", [False, True])
@pytest.mark.parametrize("scene", SCENES)
def test_predict_sort_table_exact(pkg, oracle, scene, fp64):
    sc, side = get_scene(pkg, scene, fp64)
    s, o = mk(pkg, oracle, sc, fp64)
    p, q = params_pair(pkg, oracle, side=side)
    # run two warm frames first so that the state is irregular (not a lattice)
    for _ in range(2):
        s.step(p)
        o.step(q)
    # re-seed the oracle from the GPU state so that the comparison isolates this stage
    st = s.download()
    o.set_particles(**st)
    s.stage("predict", p)
    o.predict(q)
    assert np.array_equal(s.keys().astype(np.uint64), o.keys())
    assert np.array_equal(s.pstar()[:, :3], o.pstar())
    s.stage("sort", p)
    o.sort(q).grid_table(q)
    assert np.array_equal(s.keys().astype(np.uint64), o.keys())
    assert np.all(np.diff(s.keys().astype(np.int64)) >= 0)
    t = s.table()
    assert len(t) == len(o.table())
    assert np.array_equal(t.astype(np.uint64), o.table())
    e, m = s.extent()
    eo, mo = o.extent()
    assert np.array_equal(e, eo) and np.array_equal(m, mo.astype(np.float64))
    g = s.download()
    w = o.get_particles()
    assert np.array_equal(g["id"], w["id"])  # stable sort => identical permutation
    for k in ("pos", "vel", "colour", "mass", "type"):
        assert np.array_equal(g[k], w[k]), k
    assert np.array_equal(s.pstar()[:, :3], o.pstar())


LAM_TOL = {False: 2e-6, True: 1e-13}
POS1_TOL = {False: 1e-3, True: 1e-9}
POS3_TOL = {False: 2e-2, True: 1e-7}


@pytest.mark.parametrize("fp64", [False, True])
@pytest.mark.parametrize("scene", SCENES)
def test_diffuse_lambda_delta_stages(pkg, oracle, scene, fp64):
    sc, side = get_scene(pkg, scene, fp64)
    s, o = mk(pkg, oracle, sc, fp64)
    p, q = params_pair(pkg, oracle, side=side)
    for _ in range(3):
        s.step(p)
    o.set_particles(**s.download())
    s.stage("predict", p).stage("sort", p)
    o.predict(q).sort(q).grid_table(q)
    s.stage("diffuse", p)
    o.diffuse(q)
    ctol = 1e-6 if not fp64 else 1e-14
    np.testing.assert_allclose(s.download()["colour"], o.get_particles()["colour"], rtol=ctol, atol=ctol)
    for it in range(2):
        s.stage("lambda", p)
        o.lambda_(q)
        lg, lo = s.pstar()[:, 3], o.lambdas()
        scale = np.abs(lo).max()
        assert scale > 0
        assert np.abs(lg - lo).max() <= LAM_TOL[fp64] * scale, (it, np.abs(lg - lo).max(), scale)
        s.stage("delta", p)
        o.delta(q)
        d = np.abs(s.pstar()[:, :3].astype(np.float64) - o.pstar()) * 500.0  # world units
        assert d.max() <= POS1_TOL[fp64], (it, d.max())
    s.stage("finalise", p)
    o.finalise(q)
    g, w = s.download(), o.get_particles()
    assert np.abs(g["pos"].astype(np.float64) - w["pos"]).max() <= POS1_TOL[fp64]
    vt = 1e-3 if not fp64 else 1e-9
    assert np.abs(g["vel"].astype(np.float64) - w["vel"]).max() <= vt


@pytest.mark.parametrize("fp64", [False, True])
@pytest.mark.parametrize("scene", SCENES)
def test_full_steps_vs_oracle(pkg, oracle, scene, fp64):
    sc, side = get_scene(pkg, scene, fp64)
    s, o = mk(pkg, oracle, sc, fp64)
    p, q = params_pair(pkg, oracle, side=side)
    for frame in (1, 2, 3):
        s.step(p)
        o.step(q)
        g, w = by_id(s.download()), by_id(o.get_particles())
        assert np.array_equal(g["id"], w["id"])
        d = np.linalg.norm(g["pos"].astype(np.float64) - w["pos"], axis=1)
        tol = POS1_TOL[fp64] if frame == 1 else POS3_TOL[fp64]
        assert d.max() <= tol, (frame, d.max(), d.mean())
        assert np.isfinite(g["vel"]).all()


@pytest.mark.parametrize("nm,fp64", [("f32", False), ("f64", True)])
def test_against_committed_golden(pkg, golden_dir, nm, fp64):
    """Same comparison against tests/golden/oracle_selfcheck.npz (oracle outputs committed as data)."""
    S = np.load(os.path.join(golden_dir, "oracle_selfcheck.npz"))
    sc, side = get_scene(pkg, "cubes1024", fp64)
    s = pkg.Solver(h=0.1, fp64=fp64)
    s.upload(**sc)
    p = pkg.default_params(4, side)
    s.stage("predict", p).stage("sort", p)
    assert np.array_equal(s.keys().astype(np.uint64), S[f"cubes1024_{nm}_keys"])
    assert np.array_equal(s.download()["id"], S[f"cubes1024_{nm}_sorted_ids"])
    s.stage("diffuse", p).stage("lambda", p)
    lo = S[f"cubes1024_{nm}_lambda1"]
    assert np.abs(s.pstar()[:, 3] - lo).max() <= LAM_TOL[fp64] * np.abs(lo).max()
    s.stage("delta", p)
    assert (np.abs(s.pstar()[:, :3].astype(np.float64) - S[f"cubes1024_{nm}_pstar1"]) * 500).max() <= POS1_TOL[fp64]
    s.upload(**sc)
    for frame in (1, 2, 3):
        s.step(p)
        if frame in (1, 3):
            g = by_id(s.download())
            d = np.linalg.norm(g["pos"].astype(np.float64) - S[f"cubes1024_{nm}_jacobi_f{frame}_pos"], axis=1)
            assert d.max() <= (POS1_TOL if frame == 1 else POS3_TOL)[fp64], (frame, d.max())
            dc = np.abs(g["colour"].astype(np.float64) - S[f"cubes1024_{nm}_jacobi_f{frame}_colour"]).max()
            assert dc <= (1e-5 if not fp64 else 1e-12)


def test_fast_math_within_reference_noise_floor(pkg, oracle):
    """PBF_FLAG_FAST_MATH (v_rsq, contracted FMAs) — the analogue of the reference's own -Ofast /
    native_divide builds: <= 5e-3 world units after one step (SURVEY §8c link-1 tolerance)."""
    sc, side = get_scene(pkg, "dam8192", False)
    s, o = mk(pkg, oracle, sc, False, flags=pkg.FLAG_FAST_MATH)
    p, q = params_pair(pkg, oracle, side=side)
    for _ in range(3):
        s.step(p)
    o.set_particles(**s.download())
    s.step(p)
    o.step(q)
    g, w = by_id(s.download()), by_id(o.get_particles())
    d = np.linalg.norm(g["pos"].astype(np.float64) - w["pos"], axis=1)
    assert d.max() <= 5e-3, d.max()


@pytest.mark.parametrize("fp64", [False, True])
def test_run_to_run_determinism(pkg, fp64):
    sc, side = get_scene(pkg, "dam8192", fp64)
    outs = []
    for _ in range(2):
        s = pkg.Solver(h=0.1, fp64=fp64)
        s.upload(**sc)
        p = pkg.default_params(4, side)
        s.steps(p, 12)
        outs.append(s.download())
    for k in ("id", "pos", "vel", "colour"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_moving_box_frames(pkg, oracle):
    """benchmark.cpp:33,47: every frame runs with applyMotionSinXCosZ(param, frame)."""
    sc, side = get_scene(pkg, "cubes8192", False)
    s, o = mk(pkg, oracle, sc, False)
    base, q = params_pair(pkg, oracle, side=side)
    for frame in range(4):
        p = pkg.apply_motion(base, frame, False)
        off = oracle.motion_offset(frame, False)
        q.min_bound[:] = [float(np.float32(0) + np.float32(v)) for v in off]
        q.max_bound[:] = [float(np.float32(1000) + np.float32(v)) for v in off]
        assert list(p.min_bound) == list(q.min_bound)
        o.set_particles(**s.download())
        s.step(p)
        o.step(q)
        assert len(s.table()) == len(o.table())
        g, w = by_id(s.download()), by_id(o.get_particles())
        assert np.linalg.norm(g["pos"].astype(np.float64) - w["pos"], axis=1).max() <= POS1_TOL[False]


def test_edge_cases(pkg, oracle):
    p, q = params_pair(pkg, oracle)
    s = pkg.Solver(h=0.1)
    # empty: "Particles depleted" (ompsph.hpp:122-126) — a no-op, not an error
    z = dict(id=np.zeros(0, np.uint64), type=np.zeros(0, np.uint8), mass=np.zeros(0, np.float32),
             pos=np.zeros((0, 3), np.float32), vel=np.zeros((0, 3), np.float32), colour=np.zeros((0, 4), np.float32))
    s.upload(**z).step(p).sync()
    assert s.n == 0 and len(s.download()["id"]) == 0
    # single particle: free fall
    one = dict(id=[7], type=[0], mass=[1.0], pos=[[500, 500, 500]], vel=[[0, 0, 0]], colour=[[0.5, 0.5, 0.5, 1]])
    s.upload(**one).step(p)
    o = oracle.Oracle(False)
    o.set_particles(**one)
    o.step(q)
    g, w = s.download(), o.get_particles()
    assert g["id"][0] == 7 and np.array_equal(g["pos"], w["pos"]) and np.array_equal(g["vel"], w["vel"])
    # ragged: a pile of 300 particles in ONE cell + particles far outside the grid (in no cell,
    # sph.hpp:206) + particles sitting exactly on the bounds
    rng = np.random.default_rng(5)
    pile = (rng.random((300, 3)) * 40 + 480).astype(np.float32)
    outside = np.array([[5000, 500, 500], [500, -4000, 500], [900, 900, 30000]], np.float32)
    onb = np.array([[0, 0, 0], [1000, 1000, 1000], [0, 1000, 500]], np.float32)
    pos = np.concatenate([pile, outside, onb])
    n = len(pos)
    rag = dict(id=np.arange(n)[::-1].copy(), type=np.zeros(n, np.uint8), mass=np.ones(n, np.float32), pos=pos,
               vel=(rng.random((n, 3)).astype(np.float32) - 0.5), colour=rng.random((n, 4)).astype(np.float32))
    s.upload(**rag)
    o.set_particles(**rag)
    s.stage("predict", p).stage("sort", p)
    o.predict(q).sort(q).grid_table(q)
    assert np.array_equal(s.keys().astype(np.uint64), o.keys())
    assert np.array_equal(s.table().astype(np.uint64), o.table())
    assert np.array_equal(s.download()["id"], o.get_particles()["id"])
    s.stage("diffuse", p).stage("lambda", p)
    o.diffuse(q).lambda_(q)
    lo = o.lambdas()
    assert np.abs(s.pstar()[:, 3] - lo).max() <= 1e-5 * np.abs(lo).max()
    s.stage("delta", p).stage("finalise", p)
    o.delta(q).finalise(q)
    g, w = by_id(s.download()), by_id(o.get_particles())
    assert np.isfinite(g["pos"]).all()
    # a 300-particle cell is ~50x rest density: deltaP is huge, compare relatively
    d = np.abs(g["pos"].astype(np.float64) - w["pos"]).max()
    assert d <= 1e-3 * max(1.0, np.abs(w["pos"]).max() / 1000.0), d


def test_obstacles_and_wells(pkg, oracle):
    """Obstacles follow the OpenCL backend (ocl/oclsph.cpp:66-69): fixed, lambda = 0, still
    neighbours.  Wells: ompsph.hpp:141-148."""
    sc, side = get_scene(pkg, "cubes1024", False)
    sc = {k: v.copy() for k, v in sc.items()}
    sc["type"][::7] = 1
    wells = [[300.0, 100.0, 300.0, 5000.0], [700.0, 50.0, 650.0, -2000.0]]
    s, o = mk(pkg, oracle, sc, False)
    p, q = params_pair(pkg, oracle, side=side, wells=wells)
    for frame in range(2):
        s.step(p)
        o.step(q)
    g, w = by_id(s.download()), by_id(o.get_particles())
    obs = g["type"] == 1
    assert obs.sum() > 100
    orig = by_id(sc)
    assert np.array_equal(g["pos"][obs], orig["pos"][obs]) and np.array_equal(g["vel"][obs], orig["vel"][obs])
    assert np.linalg.norm(g["pos"].astype(np.float64) - w["pos"], axis=1).max() <= POS3_TOL[False]


def test_aos_roundtrip_and_step(pkg):
    """pbf_upload_aos / pbf_download_aos with the reference's Particle<size_t,float> layout (56 B)."""
    import ctypes as C
    from pbf_sph_amd import capi
    sc, side = get_scene(pkg, "cubes1024", False)
    n = len(sc["id"])
    dt = np.dtype([("id", "<u8"), ("type", "u1"), ("_pad", "u1", 3), ("mass", "<f4"), ("pos", "<f4", 3),
                   ("vel", "<f4", 3), ("colour", "<f4", 4)])
    assert dt.itemsize == 56
    a = np.zeros(n, dt)
    for k in ("id", "type", "mass", "pos", "vel", "colour"):
        a[k] = sc[k]
    a["_pad"] = 0xAB
    lay = capi.AosLayout(56, 0, 8, 12, 16, 28, 40)
    s = pkg.Solver(h=0.1)
    L = s.L
    assert L.pbf_upload_aos(s.ctx, n, a.ctypes.data_as(C.c_void_p), C.byref(lay)) == 0
    g = s.download()
    for k in ("id", "type", "mass", "pos", "vel", "colour"):
        assert np.array_equal(g[k], sc[k])
    p = pkg.default_params(4, side)
    s.step(p)
    b = a.copy()
    assert L.pbf_download_aos(s.ctx, b.ctypes.data_as(C.c_void_p), C.byref(lay)) == 0
    g = s.download()
    for k in ("id", "type", "mass", "pos", "vel", "colour"):
        assert np.array_equal(b[k], g[k])
    assert np.all(b["_pad"] == 0xAB)
    s2 = pkg.Solver(h=0.1)
    s2.upload(**sc).step(p)
    assert np.array_equal(s2.download()["pos"], g["pos"])


def test_error_paths(pkg):
    s = pkg.Solver(h=0.1)
    p = pkg.default_params(4, 1000.0)
    sc, _ = get_scene(pkg, "cubes1024", False)
    s.upload(**sc)
    with pytest.raises(pkg.PbfError):
        s.stage("lambda", p)  # needs sort first
    bad = pkg.default_params(4, 1000.0)
    bad.dt = 0.0
    with pytest.raises(pkg.PbfError):
        s.step(bad)
    huge = pkg.default_params(4, 1000.0)
    huge.max_bound[0] = 1e6  # > 1023 cells per axis: beyond the 10-bit Morton range (curves.h:72-88)
    with pytest.raises(pkg.PbfError):
        s.step(huge)
    s.step(p).sync()  # still usable afterwards


@pytest.mark.parametrize("nominal", [262144, 1048576])
def test_full_size_properties(pkg, nominal):
    """BASELINE.json sizes (256 K, 1 M): size-independent properties instead of the oracle."""
    sc, side = pkg.scene_dambreak(nominal, False)
    n = len(sc["id"])
    s = pkg.Solver(h=0.1)
    s.upload(**sc)
    p = pkg.default_params(4, side)
    s.steps(p, 5)
    s.stage("predict", p).stage("sort", p)
    keys, table = s.keys().astype(np.int64), s.table().astype(np.int64)
    assert np.all(np.diff(keys) >= 0)                                   # sortedness
    tn = len(table)
    inside = keys[keys < tn]
    cnt = np.bincount(inside, minlength=tn)
    assert np.array_equal(table, np.concatenate([[0], np.cumsum(cnt)[:-1]]))  # table = exclusive scan of counts
    g = s.download()
    assert np.array_equal(np.sort(g["id"]), np.arange(n, dtype=np.uint64))    # a permutation: nothing lost
    s.stage("diffuse", p)
    for _ in range(4):
        s.stage("lambda", p).stage("delta", p)
    s.stage("finalise", p)
    g = s.download()
    assert np.isfinite(g["pos"]).all() and np.isfinite(g["vel"]).all()
    assert g["pos"].min() >= 0 and g["pos"].max() <= side                     # clamp (ompsph.hpp:246)
    assert np.all((g["colour"] >= 0.03 - 1e-7) & (g["colour"] <= 1.0))        # clamp (ompsph.hpp:204)
    # idempotence of the sort: sorting sorted data changes nothing
    ids1 = g["id"].copy()
    s.stage("predict", p).stage("sort", p)
    k2 = s.keys()
    with pytest.raises(pkg.PbfError):
        s.stage("sort", p)  # the histogram was consumed: a second sort needs a new predict
    assert np.all(np.diff(k2.astype(np.int64)) >= 0)
    assert len(ids1) == n
