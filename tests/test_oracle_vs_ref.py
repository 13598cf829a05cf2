"""Oracle restatement vs the reference's own headers compiled live (oracle/_ref/libref_grid.so).

Wider, randomised version of test_oracle_golden.py for machines where oracle/_ref exists (the
build container builds it from /root/reference; the GPU box receives the prebuilt .so).
"""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def R(oracle):
    r = oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref/libref_grid.so not available (no /root/reference here)")
    return r


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_morton_exhaustive_slices(oracle, R):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    for x, y, z in rng.integers(0, 2048, size=(4000, 3)):
        assert L.pbf_oracle_morton_encode(int(x), int(y), int(z)) == R.ref_morton_encode(int(x), int(y), int(z))
    for c in rng.integers(0, 2 ** 31, size=4000):
        for a in range(3):
            assert L.pbf_oracle_morton_decode(int(c), a) == R.ref_morton_decode(int(c), a)


def test_neighbour_codes_match_foreach_grid(oracle, R):
    """The oracle's 27 codes visit exactly what sph::foreach_grid visits on an identity table."""
    L = oracle.lib()
    tn = R.ref_morton_encode(24, 24, 24)
    # table[c] = c: every cell holds exactly one "particle" whose index is the cell code
    table = np.arange(tn, dtype=np.uint64)
    rng = np.random.default_rng(2)
    homes = np.concatenate([np.array([0, 1, 2, 3, tn - 1, tn - 2, tn, tn + 9], np.uint64),
                            rng.integers(0, tn + 100, 300, dtype=np.uint64)])
    buf = np.empty(64, np.uint64)
    codes = np.empty(27, np.uint64)
    for h in homes:
        k = R.ref_foreach_grid(int(h), vp(table), tn, vp(buf), 64)
        L.pbf_oracle_neighbour_codes(int(h), vp(codes))
        want = [int(c) for c in codes if c < tn and c + 1 < tn]  # last entry: empty range
        assert list(buf[:k]) == want


@pytest.mark.parametrize("fp64", [False, True])
def test_oracle_table_equals_reference_table_on_a_real_frame(oracle, R, fp64):
    s = oracle.scene_cubes(8192, fp64)
    o = oracle.Oracle(fp64)
    o.set_particles(**s)
    p = oracle.make_params()
    for _ in range(3):
        o.step(p)
    o.predict(p).sort(p).grid_table(p)
    keys = o.keys()
    e, _ = o.extent()
    tn = R.ref_make_grid_table(int(e[0]), int(e[1]), int(e[2]), 0, None, None)
    t = np.empty(tn, np.uint64)
    R.ref_make_grid_table(int(e[0]), int(e[1]), int(e[2]), len(keys), vp(keys), vp(t))
    assert np.array_equal(t, o.table())
    # keys recomputed by the reference's zCurveGridIndexAtCoordAt from the oracle's pStar
    ps = o.pstar()
    _, m = o.extent()
    f = R.ref_grid_index_at_f64 if fp64 else R.ref_grid_index_at_f32
    dt = np.float64 if fp64 else np.float32
    h = dt(0.1)
    for i in range(0, len(keys), 37):
        d = (ps[i] - m).astype(dt)
        assert f(float(d[0]), float(d[1]), float(d[2]), float(h)) == keys[i]


@pytest.mark.parametrize("fp64", [False, True])
def test_scene_matches_reference_factory(oracle, R, fp64):
    nm, dt = ("f64", np.float64) if fp64 else ("f32", np.float32)
    fn = getattr(R, "ref_scene_cubes_" + nm)
    for count in (1024, 8192, 9000):
        n = fn(count, 4, 500.0, None, None, None, None, None, None, None)
        ids, ty = np.empty(n, np.uint64), np.empty(n, np.uint8)
        mass, pos, vel, col = np.empty(n, dt), np.empty((n, 3), dt), np.empty((n, 3), dt), np.empty((n, 4), dt)
        fn(count, 4, 500.0, vp(ids), vp(ty), vp(mass), vp(pos), vp(vel), vp(col), None)
        s = oracle.scene_cubes(count, fp64)
        assert np.array_equal(s["id"], ids) and np.array_equal(s["pos"], pos) and np.array_equal(s["colour"], col)
        assert np.array_equal(s["mass"], mass) and np.array_equal(s["vel"], vel)
