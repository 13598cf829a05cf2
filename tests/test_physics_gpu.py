"""GPU checks of the parity-unpinned floating-point stages that do NOT go through oracle/pbf_oracle.cpp
(VERDICT r02 "next round" #1): the HIP path against

  * tests/nversion.py — a grid-free, sort-free, all-pairs float64 evaluation of lambda / delta-p / finalise written
    from ompsph.hpp:215-264 alone (fp64 kernels: relative 1e-12; fp32 kernels: the fp32 rounding of ~30-term sums);
  * physical invariants no restatement can share a misreading with: a lattice at rest spacing has rho ~ rho0; the
    pairwise-antisymmetric delta-p sums to zero over an unclamped blob; a rigid rotation about +z has its vorticity
    along +z and vorticity confinement ADDS angular momentum (rounds 1-2 had the sign flipped, in the oracle and in the
    kernel alike, and every bit-exact test was green); XSPH conserves momentum and lowers the velocity variance.
"""
import numpy as np
import pytest

import nversion as NV
from test_nversion_cpu import angular_momentum_z, rotating_blob, scene

pytestmark = pytest.mark.gpu


def sorted_state(pkg, sc, fp64, p, warm=0):
    s = pkg.Solver(h=0.1, fp64=fp64)
    s.upload(**sc)
    if warm:
        s.steps(p, warm)
    s.stage("predict", p).stage("sort", p)
    return s


@pytest.mark.parametrize("fp64", [True, False])
@pytest.mark.parametrize("name", ["cubes1024", "cloud", "obstacles"])
def test_hip_lambda_delta_finalise_equal_all_pairs_evaluation(pkg, name, fp64):
    sc = scene(name)
    p = pkg.default_params(2, 1000.0)
    s = sorted_state(pkg, sc, fp64, p, warm=3 if name == "cubes1024" else 0)
    st = s.download()                                   # sorted; vel = predicted velocity
    ps = s.pstar()[:, :3].astype(np.float64)
    mass, obstacle = st["mass"].astype(np.float64), st["type"] == 1
    assert s.keys().max() < len(s.table())
    cells = NV.predict_cells(ps, 0.1, p.scale, list(p.min_bound))
    # fp64: summation-order noise only.  fp32: every pair term carries ~1e-7, the density constraint C = rho/rho0 - 1
    # cancels one digit, the gradient sum two => absolute bars relative to the largest value of the field.
    lam_tol, move_tol = (1e-12, 1e-12) if fp64 else (3e-5, 2e-4)
    for it in range(2):
        cm = None if it == 0 else cells
        s.stage("lambda", p)
        lam, _ = NV.lambdas(ps, mass, 0.1, obstacle, cm)
        got = s.pstar()[:, 3].astype(np.float64)
        assert np.abs(got - lam).max() <= lam_tol * np.abs(lam).max(), (name, it, np.abs(got - lam).max() / np.abs(lam).max())
        s.stage("delta", p)
        # delta-p from the DEVICE's lambda (fp32: lambda's own rounding is not delta-p's error)
        ps_new, _ = NV.delta(ps, got, 0.1, p.scale, list(p.min_bound), list(p.max_bound), obstacle, cm)
        new = s.pstar()[:, :3].astype(np.float64)
        move_g, move_n = new - ps, ps_new - ps
        err = np.abs(move_g - move_n).max()
        assert err <= move_tol * np.abs(move_n).max() + (4e-16 if fp64 else 2.5e-7), (name, it, err, np.abs(move_n).max())
        ps = new
    s.stage("finalise", p)
    pos, vel = NV.finalise(ps, st["pos"].astype(np.float64), st["vel"].astype(np.float64), p.dt, p.scale)
    g = s.download()
    keep = ~obstacle
    rel = 1e-12 if fp64 else 1e-6
    assert np.abs(g["pos"][keep] - pos[keep]).max() <= rel * 1000.0
    assert np.abs(g["vel"][keep] - vel[keep]).max() <= (1e-11 if fp64 else 2e-5) * max(np.abs(vel).max(), 1.0)


@pytest.mark.parametrize("fp64", [False, True])
def test_rest_lattice_has_rest_density(pkg, fp64):
    """rho0 = 6378 particles of mass 1 per unit volume <=> a cubic lattice of spacing 6378^(-1/3) = 26.96 world units:
    the poly6 sum over it must return rho0 to the kernel's discretisation error (0.8 % at h / spacing = 1.85), i.e.
    lambda = -C / (|sum grad|^2 + 600) with a vanishing gradient sum gives C = -600 lambda ~ 0.  Pins poly6Factor,
    the r <= h support and the density normalisation without any restatement; the reference's own start lattice
    (spacing 22, sph.hpp:165) is 1.84x over-dense: C = +0.859."""
    dt = np.float64 if fp64 else np.float32
    for spacing, want, tol in ((6378.0 ** (-1.0 / 3.0) * 500.0, 0.0, 0.012), (22.0, 0.8587, 2e-3)):
        ax = (np.arange(11) - 5) * spacing + 500.0
        g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
        n = len(g)
        sc = dict(id=np.arange(n, dtype=np.uint64), type=np.zeros(n, np.uint8), mass=np.ones(n, dt), pos=g.astype(dt),
                  vel=np.zeros((n, 3), dt), colour=np.full((n, 4), 0.5, dt))
        p = pkg.default_params(1, 1000.0)
        p.constant_force[1] = 0.0
        s = sorted_state(pkg, sc, fp64, p)
        s.stage("lambda", p)
        st = s.download()
        inner = np.abs(st["pos"].astype(np.float64) - 500.0).max(1) <= 2.5 * spacing   # full support inside the block
        c = -600.0 * s.pstar()[inner, 3].astype(np.float64)
        assert inner.sum() >= 100 and np.abs(c - want).max() <= tol, (spacing, c.min(), c.max())


@pytest.mark.parametrize("fp64", [False, True])
def test_delta_p_conserves_momentum_in_an_unclamped_blob(pkg, fp64):
    """delta-p_a = sum_b grad W_ab (lambda_a + lambda_b + corr_ab) / rho0 with grad W_ab = -grad W_ba and a symmetric
    factor: the moves of an isolated blob away from every wall sum to zero (ompsph.hpp:235-248) — per iteration, 8 000
    particles, over-dense start (large moves)."""
    sc, _ = rotating_blob(n_side=20, spacing=22.0, omega_z=0.0, fp64=fp64)
    p = pkg.default_params(4, 1000.0)
    p.constant_force[1] = 0.0
    s = sorted_state(pkg, sc, fp64, p)
    for it in range(4):
        s.stage("lambda", p)
        before = s.pstar()[:, :3].astype(np.float64)
        s.stage("delta", p)
        move = s.pstar()[:, :3].astype(np.float64) - before
        assert np.abs(move).max() > 1e-4                                   # it does move
        assert np.abs(move.sum(0)).max() <= (1e-11 if fp64 else 2e-5) * np.abs(move).sum(), (it, move.sum(0))
    g = s.download()
    assert g["pos"].min() > 1.0 and g["pos"].max() < 999.0                 # nothing reached a wall: no clamp took part


@pytest.mark.parametrize("fp64", [False, True])
def test_rigid_rotation_vorticity_sign_and_confinement(pkg, fp64):
    """curl(Omega z x r) = 2 Omega z: the vorticity estimate (Macklin & Mueller 2013 eq. 15) of a block in rigid +z
    rotation points along +z at every particle with full kernel support, and the confinement force (eq. 16) adds
    angular momentum about z instead of removing it.  Fails on rounds 1-2's kernels (omega_z < 0, L_z decreasing)."""
    sc, _ = rotating_blob(fp64=fp64)
    centre = (500.0, 500.0, 500.0)
    p0 = pkg.default_params(0, 1000.0)
    p0.constant_force[1] = 0.0
    p1 = p0.copy()
    p1.vorticity = 1
    a = pkg.Solver(h=0.1, fp64=fp64)
    a.upload(**sc).step(p0)                    # K = 0, no gravity: advect, v <- 0.98 v
    b = pkg.Solver(h=0.1, fp64=fp64)
    b.upload(**sc).step(p1)                    # the same + vorticity confinement
    ga, gb = a.download(), b.download()
    assert np.array_equal(ga["id"], gb["id"]) and np.array_equal(ga["pos"], gb["pos"])
    w = b.omega().astype(np.float64)
    r = np.linalg.norm(ga["pos"].astype(np.float64) - np.asarray(centre), axis=1)
    inner = r < 100.0
    assert inner.sum() > 100
    assert np.all(w[inner, 2] > 0), "omega must point along +z for a +z rotation"
    assert np.abs(w[inner, :2]).max() < (1e-6 if fp64 else 2e-3) * w[inner, 2].min()
    # magnitude: eq. 15 carries no volume weights, so omega = rho0 x curl v up to the kernel's discretisation error
    curl = 2.0 * 3.0 * 0.98
    assert np.all(np.abs(w[inner, 2] / NV.RHO / curl - 1.0) < 0.15)
    la, lb = angular_momentum_z(ga["pos"], ga["vel"], centre), angular_momentum_z(gb["pos"], gb["vel"], centre)
    assert la > 0 and lb > la, ("vorticity confinement must add angular momentum to a vortex", la, lb)
    # and the device's omega / force equal the all-pairs evaluation of the paper's equations
    ps = ga["pos"].astype(np.float64) / 500.0
    wn = NV.vorticity(ps, ga["vel"].astype(np.float64), 0.1)
    assert np.abs(w - wn).max() <= (1e-10 if fp64 else 2e-4) * np.abs(wn).max()
    dv, eta = NV.vorticity_force_dv(ps, wn, 0.1, p0.dt, return_eta=True)
    got = gb["vel"].astype(np.float64) - ga["vel"].astype(np.float64)
    # (N = eta / |eta| is ill-conditioned where the |omega| field is flat, i.e. eta = grad |omega| ~ 0 — the block's
    # interior and a few symmetric spots of its faces: compare where the direction is well defined)
    steep = eta > 0.05 * eta.max()
    assert steep.sum() > 500
    assert np.abs(got[steep] - dv[steep]).max() <= (1e-8 if fp64 else 5e-3) * np.abs(dv).max()


@pytest.mark.parametrize("fp64", [False, True])
def test_xsph_conserves_momentum_and_smooths(pkg, fp64):
    """XSPH (eq. 17) is a symmetric-weight average of velocity differences: with equal masses it leaves sum m v
    unchanged and lowers the velocity variance."""
    sc = scene("cloud")
    p0 = pkg.default_params(2, 1000.0)
    p1 = p0.copy()
    p1.xsph = 1
    a = pkg.Solver(h=0.1, fp64=fp64)
    a.upload(**sc).step(p0)
    b = pkg.Solver(h=0.1, fp64=fp64)
    b.upload(**sc).step(p1)
    ga, gb = a.download(), b.download()
    assert np.array_equal(ga["id"], gb["id"]) and np.array_equal(ga["pos"], gb["pos"])
    v0, v1 = ga["vel"].astype(np.float64), gb["vel"].astype(np.float64)
    assert not np.array_equal(v0, v1)
    assert np.abs(v1.sum(0) - v0.sum(0)).max() <= (1e-12 if fp64 else 1e-6) * np.abs(v0).sum()
    assert v1.var(0).sum() < v0.var(0).sum()
    ps = ga["pos"].astype(np.float64) / 500.0
    # (the walk only sees the predict-time cells: rebuild them from the state before the step)
    c = pkg.Solver(h=0.1, fp64=fp64)
    c.upload(**sc).stage("predict", p0).stage("sort", p0)
    assert np.array_equal(c.download()["id"], ga["id"])
    cells = NV.predict_cells(c.pstar()[:, :3].astype(np.float64), 0.1, p0.scale, list(p0.min_bound))
    dx = NV.xsph(ps, v0, 0.1, cells) - v0
    assert np.abs((v1 - v0) - dx).max() <= (1e-9 if fp64 else 2e-4) * np.abs(dx).max()
