"""The XCD-aware workgroup -> chunk map of the list kernels (pbf_kernels.hpp xcd_chunk): restated in numpy and checked to
be a bijection of [0, grid) for every grid size — a workgroup id that maps outside, or two that collide, would leave
particles without a neighbour list — and to hand XCD k (workgroup ids congruent k mod 8) one contiguous run of chunks."""
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = (ROOT / "pbf-sph_amd" / "csrc" / "pbf_kernels.hpp").read_text()


def xcd_chunk(block, grid, xcds=8):
    k, q, r = block % xcds, grid // xcds, grid % xcds
    return k * q + np.minimum(k, r) + block // xcds


def test_source_matches_the_restatement():
    body = SRC[SRC.index("__device__ inline uint32_t xcd_chunk()"):]
    body = body[:body.index("}\n") + 1]
    assert re.search(r"k = blockIdx\.x % NUM_XCD, q = g / NUM_XCD, r = g % NUM_XCD", body)
    assert re.search(r"return k \* q \+ min\(k, r\) \+ blockIdx\.x / NUM_XCD;", body)
    assert "constexpr uint32_t NUM_XCD = 8;" in SRC


@pytest.mark.parametrize("grid", list(range(1, 70)) + [255, 256, 257, 1000, 4000, 4001, 4007, 16384, 65535])
def test_bijection_and_contiguous_runs(grid):
    b = np.arange(grid, dtype=np.int64)
    c = xcd_chunk(b, grid)
    assert np.array_equal(np.sort(c), b)  # every chunk exactly once
    for k in range(8):  # one contiguous, ascending run per XCD
        mine = c[b % 8 == k]
        if len(mine):
            assert np.array_equal(mine, np.arange(mine[0], mine[0] + len(mine)))
