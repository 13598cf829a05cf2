"""bench.py's host-side helpers (no GPU): the candidates-per-particle figure against a brute-force count."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def spread(v):
    out = 0
    for b in range(10):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def test_candidates_per_particle_matches_brute_force():
    rng = np.random.default_rng(3)
    ext = 6  # cells 0..5 per axis
    cells = rng.integers(0, ext, (500, 3))
    keys = np.sort(np.array([spread(x) | (spread(y) << 1) | (spread(z) << 2) for x, y, z in cells], np.int64))
    tn = spread(ext) | (spread(ext) << 1) | (spread(ext) << 2)  # Morton(extent), sph.hpp:240
    table = np.searchsorted(keys, np.arange(tn))                # exclusive scan of the histogram
    got = bench.candidates_per_particle(keys, table)
    pop = {}
    for k in keys:
        pop[int(k)] = pop.get(int(k), 0) + 1
    pop.pop(tn - 1, None)  # the table's last cell counts as empty
    decode = lambda k: tuple(sum(((k >> (3 * b + a)) & 1) << b for b in range(10)) for a in range(3))
    want = 0
    for k in keys:
        x, y, z = decode(int(k))
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    c = spread((x + dx) & 1023) | (spread((y + dy) & 1023) << 1) | (spread((z + dz) & 1023) << 2)
                    if c < tn:
                        want += pop.get(c, 0)
    assert abs(got - want / len(keys)) < 1e-9
