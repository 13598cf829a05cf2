"""k_build_lists_q walks a (dy, dz) row of three x cells as TWO runs of the sorted array (pair of adjacent Morton
codes + single cell, chosen by the parity of the own x) fetched with two table loads.  This restates that
row logic in numpy and checks, for every cell of small and edge-case grids, that the candidate sequence equals
the reference's per-cell walk (sph.hpp:206-234: x fastest, a cell outside the table and the table's last cell
are empty) — including x = 0 / x = 1023 wrap-around codes and tables that end inside a row."""
import numpy as np

MX, MY, MZ = 0x09249249, 0x12492492, 0x24924924


def spread(v):
    out = 0
    for b in range(10):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def neigh(key):
    xm, ym, zm = key & MX, key & MY, key & MZ
    xs = [(xm - 1) & MX, xm, ((xm | (~MX & 0x3FFFFFFF)) + 1) & MX]
    ys = [(ym - 2) & MY, ym, ((ym | (~MY & 0x3FFFFFFF)) + 2) & MY]
    zs = [(zm - 4) & MZ, zm, ((zm | (~MZ & 0x3FFFFFFF)) + 4) & MZ]
    return xs, ys, zs


def reference_walk(key, table, tn):
    xs, ys, zs = neigh(key)
    out = []
    for z in zs:
        for y in ys:
            for x in xs:
                code = x | y | z
                if code >= tn:
                    continue
                start = table[code]
                end = table[code + 1] if code + 1 < tn else start
                out.extend(range(start, end))
    return out


def two_run_walk(key, table, tn):
    """load_row + the slot sequence of k_build_lists_q (pairs, run A padded to an even length)."""
    xs, ys, zs = neigh(key)
    odd = key & 1
    x_pair, x_single = (xs[0], xs[2]) if odd else (xs[1], xs[0])
    out = []
    for r in range(9):
        yz = ys[r % 3] | zs[r // 3]
        cp, cs = x_pair | yz, x_single | yz
        t0, t1, t2 = (table[min(cp, tn) + k] for k in range(3))   # the table keeps entries up to tableN + 1 (+ slack)
        s0, s1 = (table[min(cs, tn) + k] for k in range(2))
        lp0 = t1 - t0 if cp + 1 < tn else 0
        lp1 = t2 - t1 if cp + 2 < tn else 0
        sp, lp = (t0 if lp0 else t1), lp0 + lp1
        ls = s1 - s0 if cs + 1 < tn else 0
        sa, la, sb, lb = (sp, lp, s0, ls) if odd else (s0, ls, sp, lp)
        lae = (la + 1) & ~1
        total = lae + lb
        ob = sb - lae
        for sl in range(0, total + (total & 1), 2):   # pairs
            in_a = sl < lae
            b = sl + (sa if in_a else ob)
            lim = la if in_a else total
            if sl < lim:
                out.append(b)
            if sl + 1 < lim:
                out.append(b + 1)
    return out


class LazyTable:
    """table[c] = number of keys < c (the exclusive scan of the cell histogram, incl. the overflow bucket and the
    closing total), evaluated on demand — a 1023-wide grid has 1.5e8 codes."""

    def __init__(self, keys):
        self.keys = keys

    def __getitem__(self, c):
        return int(np.searchsorted(self.keys, c))


def make_table(keys, tn):
    keys = np.sort(np.asarray(keys, np.int64))
    return keys, LazyTable(keys)


def check_grid(ext, cells, rng):
    tn = spread(ext[0]) | (spread(ext[1]) << 1) | (spread(ext[2]) << 2)
    keys = [spread(x) | (spread(y) << 1) | (spread(z) << 2) for x, y, z in cells]
    keys, table = make_table(keys, tn)
    walkers = set(int(k) for k in keys) | {int(k) for k in rng.integers(0, tn + 50, 40)}
    for k in walkers:
        assert two_run_walk(k, table, tn) == reference_walk(k, table, tn), (ext, k)


def test_two_run_rows_equal_the_per_cell_walk():
    rng = np.random.default_rng(11)
    # dense small grid, particles also outside the extent (codes >= tableN land in the overflow bucket)
    check_grid((5, 4, 6), rng.integers(0, 8, (600, 3)), rng)
    # odd extents so that tables end inside rows / pairs
    check_grid((7, 3, 3), rng.integers(0, 8, (400, 3)), rng)
    check_grid((2, 2, 2), rng.integers(0, 3, (60, 3)), rng)
    # sparse: many empty cells, empty pairs, empty singles
    check_grid((9, 9, 9), rng.integers(0, 9, (40, 3)), rng)


def test_two_run_rows_at_the_morton_edge():
    """x = 0 and x = 1023: the -1 / +1 neighbour codes wrap around inside the 10-bit field, like the reference's."""
    rng = np.random.default_rng(5)
    cells = np.concatenate([np.stack([rng.choice([0, 1, 1022, 1023], 300), rng.integers(0, 3, 300), rng.integers(0, 3, 300)], 1)])
    check_grid((1023, 3, 3), cells, rng)
