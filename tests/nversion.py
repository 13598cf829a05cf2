"""A second, independent evaluation of the floating-point stages — TEST INFRASTRUCTURE.

Written from the reference's source text alone (src/omp/ompsph.hpp:67-75, 209-264; src/sph.hpp:251-253;
src/sph_constants.h:5-16), NOT from oracle/pbf_oracle.cpp and not from the kernels: no grid, no sort, no cells —
every particle against every particle, all pairs at once in numpy float64 (O(n^2), n <= a few thousand).

Why this is the same function as the reference's 27-cell walk: cells are h wide, so every b with |a - b| <= h lies in
one of a's 27 cells, and a candidate beyond h contributes exactly 0 to every sum (poly6Kernel and
spikyKernelGradient return 0 there).  Only the ORDER of the additions differs, hence the comparisons are held to a
relative 1e-12 in fp64 (summation noise), not to bit-equality.  It holds for particles inside the grid (key <
Morton(extent)) — a particle outside has no cell and walks nothing in the reference (sph.hpp:206).

One thing the reference's walk does that plain all-pairs does not: a particle's cell is fixed when pStar is PREDICTED
(ompsph.hpp:152) and stays fixed while delta-p moves pStar, so from the second solver iteration on a neighbour can be
within h and yet outside the 27 cells.  `cells` (integer cell coordinates at predict time, from predict_cells())
restricts the pairs to |cell_a - cell_b| <= 1 per axis — still no table, no sort, no Morton code.

The constants are `float` in the reference and promoted at use (sph_constants.h): np.float32 values widened here.
"""
import numpy as np

f32 = np.float32
VD = float(f32(0.49))
RHO = float(f32(6378.0))
RHO_RECIP = float(f32(1.0) / f32(6378.0))          # static float RHO_RECIP = 1.f / RHO
EPSILON = float(f32(0.00000001))
CFM_EPSILON = float(f32(600.0))
CorrDeltaQ = float(f32(0.3))
CorrK = float(f32(0.0001))
CorrN = 4.0
C_XSPH = float(f32(0.00001))                       # sph_constants.h:13 (unused by the reference)
VORTICITY_EPSILON = float(f32(0.0005))             # sph_constants.h:14 (unused by the reference)


def poly6_factor(h):   # sph.hpp:252
    return 315.0 / (64.0 * np.pi * h ** 9)


def spiky_factor(h):   # sph.hpp:253
    return -(45.0 / (np.pi * h ** 6))


def poly6(r, h):       # ompsph.hpp:67-69
    return np.where(r <= h, poly6_factor(h) * ((h * h) - r * r) ** 3, 0.0)


def predict_cells(ps_predicted, h, scale, min_bound):
    """ompsph.hpp:132-135,152 + sph.hpp:198-201: cell = trunc((pStar - minExtent) / h), minExtent = minBound/scale - 2h"""
    lo = np.asarray(min_bound, float) / scale - 2.0 * h
    return np.floor((ps_predicted - lo) / h).astype(np.int64)


def pair_tables(ps, h, cells=None):
    """r[a, b] = |ps[a] - ps[b]| and grad[a, b, :] = spikyKernelGradient(ps[a], ps[b]) (ompsph.hpp:71-75); pairs
    outside each other's 27 predict-time cells (sph.hpp:203-236) are pushed beyond h."""
    d = ps[:, None, :] - ps[None, :, :]                     # x - y
    r = np.sqrt((d * d).sum(-1))
    if cells is not None:
        r = np.where((np.abs(cells[:, None, :] - cells[None, :, :]) <= 1).all(-1), r, np.inf)
    ok = (r >= EPSILON) & (r <= h)
    safe = np.where(ok, r, 1.0)
    g = np.where(ok[..., None], d * (spiky_factor(h) * ((h - safe) ** 2 / safe))[..., None], 0.0)
    return r, g


def lambdas(ps, mass, h, obstacle=None, cells=None):
    """ompsph.hpp:215-232"""
    r, g = pair_tables(ps, h, cells)
    norm2v = (g * RHO_RECIP).sum(1)
    rho = (mass[:, None] * poly6(r, h)).sum(1)
    lam = -(rho / RHO - 1.0) / ((norm2v * norm2v).sum(-1) + CFM_EPSILON)
    if obstacle is not None:
        lam = np.where(obstacle, 0.0, lam)
    return lam, rho


def delta(ps, lam, h, scale, min_bound, max_bound, obstacle=None, cells=None):
    """ompsph.hpp:234-248, every particle from the OLD pStar (Jacobi); returns (new pStar, deltaP)."""
    r, g = pair_tables(ps, h, cells)
    p6dq = float(poly6(np.array(CorrDeltaQ * h), h))        # ompsph.hpp:213
    corr = -CorrK * (poly6(r, h) / p6dq) ** CorrN
    factor = (lam[:, None] + lam[None, :] + corr) / RHO
    dp = (g * factor[..., None]).sum(1)
    pos = np.minimum(np.asarray(max_bound, float), np.maximum(np.asarray(min_bound, float), (ps + dp) * scale))
    out = pos / scale
    if obstacle is not None:
        out = np.where(obstacle[:, None], ps, out)
    return out, dp


def finalise(ps, pos, vel, dt, scale):
    """ompsph.hpp:256-264"""
    dx = ps - pos / scale
    return ps * scale, (dx * (1.0 / dt) + vel) * VD


def predict(pos, vel, mass, dt, scale, force):
    """ompsph.hpp:137-154 without wells"""
    v = (mass[:, None] * np.asarray(force, float)[None, :]) * dt + vel
    return v, v * dt + pos / scale


# ---- the opt-in extras, from Macklin & Mueller 2013 (eq. 15-17) with the reference's kernels and constants ----------

def vorticity(ps, vel, h, cells=None):
    """eq. 15: omega_i = sum_j (v_j - v_i) x grad_{p_j} W(p_i - p_j);  grad_{p_j} W = -grad_{p_i} W"""
    _, g = pair_tables(ps, h, cells)                            # grad_{p_i} W
    vij = vel[None, :, :] - vel[:, None, :]
    return np.cross(vij, -g).sum(1)


def vorticity_force_dv(ps, omega, h, dt, cells=None, return_eta=False):
    """eq. 16: f = eps (N x omega), N = eta / |eta|, eta = grad |omega|; returns the velocity increment f dt"""
    _, g = pair_tables(ps, h, cells)
    mag = np.sqrt((omega * omega).sum(-1))
    eta = (g * mag[None, :, None]).sum(1)
    ln = np.sqrt((eta * eta).sum(-1))
    nn = np.where((ln > EPSILON)[:, None], eta / np.where(ln > EPSILON, ln, 1.0)[:, None], 0.0)
    dv = np.cross(nn, omega) * (VORTICITY_EPSILON * dt)
    return (dv, ln) if return_eta else dv


def xsph(ps, vel, h, cells=None):
    """eq. 17: v_i <- v_i + c sum_j (v_j - v_i) W(p_i - p_j)"""
    r, _ = pair_tables(ps, h, cells)
    w = poly6(r, h)
    return vel + C_XSPH * ((vel[None, :, :] - vel[:, None, :]) * w[..., None]).sum(1)
