"""The list build tests candidates on 8-byte quantised positions (k_quantise / k_build_lists_q,
pbf-sph_amd/csrc/pbf_kernels.hpp).  Its only obligation is to never drop a pair the exact fp test accepts.
This restates the quantisation in numpy float32 — constants parsed from the kernel source — and attacks it
with pairs at and just inside distance h, at small and large grid coordinates and below the grid minimum."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "pbf-sph_amd", "csrc", "pbf_kernels.hpp")).read()
BITS = int(re.search(r"constexpr int QPOS_BITS = (\d+);", SRC).group(1))
T = (1 << BITS) + int(re.search(r"constexpr uint32_t QPOS_T = \(1u << QPOS_BITS\) \+ (\d+);", SRC).group(1))


def quantise(p, gmin, h):
    """quantise_position: low 16 bits of floor((p - gridMin) * (2^BITS / h)), all in float32."""
    k = np.float32(1 << BITS) / np.float32(h)
    f = np.floor((p.astype(np.float32) - gmin.astype(np.float32)) * k)
    return (f.astype(np.int64) & 0xFFFF).astype(np.int64)


def wrapped_d2(qa, qb):
    d = ((qb - qa + 32768) & 0xFFFF) - 32768  # v_pk_sub_i16
    return (d * d).sum(axis=1)                # two v_dot2_i32_i16


def test_threshold_constants():
    assert BITS == 11 and T >= (1 << BITS) + 5


def test_quantised_test_is_a_superset():
    rng = np.random.default_rng(7)
    h = np.float32(0.1)
    n = 400_000
    gmin = np.array([-0.37, 0.011, -2.5], np.float32)
    # cells 0..1023 on every axis, a few below the minimum, many close to the far end (largest fp error)
    cell = np.concatenate([rng.uniform(-3, 1023, (n // 2, 3)), rng.uniform(990, 1023.9, (n // 2, 3))])
    a = (gmin + cell * h).astype(np.float32)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    m = n // 10  # axis-aligned and face-diagonal directions stress single-axis rounding
    u[:m] = np.eye(3)[rng.integers(0, 3, m)] * rng.choice([-1.0, 1.0], (m, 1))
    shrink = rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-3, 0.2], (n, 1))
    b = (a.astype(np.float64) + u * float(h) * (1.0 - shrink)).astype(np.float32)
    # the exact test as the ops evaluate it (fp32 differences and squares): only pairs it accepts oblige us
    d = (b - a).astype(np.float32)
    r2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32)
    accepted = np.sqrt(r2.astype(np.float32)) <= h * np.float32(1 + 1e-5)
    assert accepted.sum() > n // 2
    d2 = wrapped_d2(quantise(a, gmin, h), quantise(b, gmin, h))
    missed = accepted & (d2 > T * T)
    assert not missed.any(), (int(missed.sum()), float(np.sqrt(d2[missed].max())), T)
    # and it is not uselessly loose: pairs beyond 1.01 h are rejected
    far = (a.astype(np.float64) + u * float(h) * 1.01).astype(np.float32)
    assert (wrapped_d2(quantise(a, gmin, h), quantise(far, gmin, h)) > T * T).mean() > 0.99


def test_wraparound_only_adds_candidates():
    """Separations beyond 16 h alias modulo 2^16: that may turn a far pair into a (harmless) candidate, never a
    near pair into a miss — a pair within h is far below the wrap."""
    h = np.float32(0.1)
    gmin = np.zeros(3, np.float32)
    a = np.array([[5.0, 5.0, 5.0]], np.float32)
    near = a + np.array([[0.09, 0.0, 0.0]], np.float32)
    alias = a + np.array([[32.0 * 0.1 + 0.05, 0.0, 0.0]], np.float32)  # 32 h + 0.5 h  ->  0.5 h after the wrap
    assert wrapped_d2(quantise(a, gmin, h), quantise(near, gmin, h))[0] <= T * T
    assert wrapped_d2(quantise(a, gmin, h), quantise(alias, gmin, h))[0] <= T * T  # a false positive, by design
