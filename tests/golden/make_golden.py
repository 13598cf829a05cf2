"""Generate the committed golden fixtures (run in the build container, where /root/reference exists).

  ref_grid.npz        — outputs of the REFERENCE'S OWN code: oracle/_ref/libref_grid.so is
                        src/curves.h + src/sph.hpp compiled where they lie (see oracle/ref_grid.cpp).
                        Pins: Morton encode/decode, zCurveGridIndexAtCoordAt, makeGridTable,
                        foreach_grid order, kernel factors, scene factory, box motion.
  oracle_selfcheck.npz — outputs of OUR oracle (oracle/pbf_oracle.cpp) for the floating-point
                        stages.  SELF-GENERATED: the reference has no goldens for these and its
                        OpenMP backend cannot be built here (glm absent) => parity unpinned; the file
                        guards the oracle against accidental change and gives the GPU tests a
                        machine-independent target.

Fixtures are data only (inputs + expected outputs); no reference source text is stored.
Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ctypes as C  # noqa: E402

import oracle_lib as O  # noqa: E402


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def ref_grid():
    R = O.ref()
    assert R is not None, "oracle/_ref/libref_grid.so missing (needs /root/reference)"
    rng = np.random.default_rng(20261004)
    out = {}
    # Morton KATs: corners, SURVEY §8c values, random coords incl. > 10 bits (low 10 bits are kept)
    coords = np.concatenate([
        np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [24, 24, 24], [23, 23, 23], [128, 128, 128],
                  [1023, 1023, 1023], [1024, 5, 7], [96, 96, 96], [149, 149, 149], [63, 63, 63], [26, 26, 26],
                  [2 ** 64 - 1, 3, 3], [5, 2 ** 64 - 1, 0]], dtype=np.uint64),
        rng.integers(0, 1024, size=(256, 3), dtype=np.uint64),
    ])
    out["morton_coords"] = coords
    out["morton_codes"] = np.array([R.ref_morton_encode(int(x), int(y), int(z)) for x, y, z in coords], np.uint64)
    codes = np.concatenate([np.arange(0, 4096, dtype=np.uint64), rng.integers(0, 2 ** 30, 512, dtype=np.uint64)])
    out["decode_codes"] = codes
    out["decode_xyz"] = np.array([[R.ref_morton_decode(int(c), a) for a in range(3)] for c in codes], np.uint64)
    # zCurveGridIndexAtCoordAt on seeded positions (non-negative, as in the reference's valid domain)
    p32 = (rng.random((512, 3)) * 2.4).astype(np.float32)
    p64 = rng.random((512, 3)) * 2.4
    out["gia_pos_f32"], out["gia_pos_f64"] = p32, p64
    out["gia_f32"] = np.array([R.ref_grid_index_at_f32(*map(float, p), 0.1) for p in p32], np.uint64)
    out["gia_f64"] = np.array([R.ref_grid_index_at_f64(*map(float, p), 0.1) for p in p64], np.uint64)
    # makeGridTable: clustered random keys, some >= table length, extent 24^3 and a non-cubic one
    for tag, ext in (("a", (24, 24, 24)), ("b", (13, 24, 9))):
        tn = R.ref_make_grid_table(*ext, 0, None, None)
        keys = np.sort(np.concatenate([rng.integers(0, tn, 1500, dtype=np.uint64),
                                       rng.integers(tn // 3, tn // 3 + 40, 300, dtype=np.uint64),
                                       np.array([tn - 1, tn - 1, tn, tn + 5, 2 ** 30 - 1], np.uint64)]))
        table = np.empty(tn, np.uint64)
        got = R.ref_make_grid_table(*ext, len(keys), vp(keys), vp(table))
        assert got == tn
        out[f"gt_{tag}_extent"] = np.array(ext, np.uint64)
        out[f"gt_{tag}_keys"] = keys
        out[f"gt_{tag}_table"] = table
        # foreach_grid visitation order for a set of home cells incl. domain corners and the last cell
        homes = np.concatenate([np.array([0, 1, 7, tn - 1, tn - 2, tn // 3 + 3, tn, tn + 5], np.uint64),
                                rng.integers(0, tn, 24, dtype=np.uint64)])
        buf = np.empty(4096, np.uint64)
        visits, counts = [], []
        for hcell in homes:
            k = R.ref_foreach_grid(int(hcell), vp(table), tn, vp(buf), len(buf))
            assert k <= len(buf)
            counts.append(k)
            visits.append(buf[:k].copy())
        out[f"fg_{tag}_homes"] = homes
        out[f"fg_{tag}_counts"] = np.array(counts, np.uint64)
        out[f"fg_{tag}_visits"] = np.concatenate(visits) if visits else np.zeros(0, np.uint64)
    out["factors"] = np.array([R.ref_poly6_factor_f32(0.1), R.ref_spiky_factor_f32(0.1), R.ref_poly6_factor_f64(0.1),
                               R.ref_spiky_factor_f64(0.1), R.ref_poly6_factor_f32(0.05), R.ref_spiky_factor_f64(0.2)])
    # scene factory: simpleConfigWith2Cubes(2048 | 20000, 4, 500)
    for count in (2048, 20000):
        for fp64, nm, dt in ((0, "f32", np.float32), (1, "f64", np.float64)):
            fn = getattr(R, "ref_scene_cubes_" + nm)
            cfg = np.zeros(12)
            n = fn(count, 4, 500.0, None, None, None, None, None, None, vp(cfg))
            ids, ty = np.empty(n, np.uint64), np.empty(n, np.uint8)
            mass, pos, vel, col = np.empty(n, dt), np.empty((n, 3), dt), np.empty((n, 3), dt), np.empty((n, 4), dt)
            fn(count, 4, 500.0, vp(ids), vp(ty), vp(mass), vp(pos), vp(vel), vp(col), vp(cfg))
            out[f"scene_{count}_{nm}_cfg"] = cfg
            if count == 2048:
                out[f"scene_{count}_{nm}_id"], out[f"scene_{count}_{nm}_type"] = ids, ty
                out[f"scene_{count}_{nm}_mass"], out[f"scene_{count}_{nm}_pos"] = mass, pos
                out[f"scene_{count}_{nm}_vel"], out[f"scene_{count}_{nm}_colour"] = vel, col
            else:  # big one: count + checksums only
                out[f"scene_{count}_{nm}_n"] = np.array([n], np.uint64)
                out[f"scene_{count}_{nm}_possum"] = pos.astype(np.float64).sum(0)
                out[f"scene_{count}_{nm}_last"] = pos[-1].astype(np.float64)
    frames = np.arange(0, 64, dtype=np.uint64)
    m32, m64 = np.zeros((len(frames), 6)), np.zeros((len(frames), 6))
    for i, f in enumerate(frames):
        R.ref_motion_f32(int(f), vp(m32[i]))
        R.ref_motion_f64(int(f), vp(m64[i]))
    out["motion_frames"], out["motion_f32"], out["motion_f64"] = frames, m32, m64
    out["sizeof_partially_advected_f32"] = np.array([R.ref_sizeof_partially_advected_f32()], np.uint64)
    np.savez_compressed(os.path.join(HERE, "ref_grid.npz"), **out)
    print("ref_grid.npz:", len(out), "arrays")


def mc_case_digests(tri_rows):
    """Per case: CRC32 of (a) the oriented polygon loops the triangles tile and (b) the oriented triangle set — both
    canonicalised, so the digests do not depend on the order rows are written in.  Digests, not the table."""
    import zlib

    def tris_of(row):
        row = [int(v) for v in row if v != 255]
        return [tuple(row[i:i + 3]) for i in range(0, len(row), 3)]

    def canon(t):
        k = t.index(min(t))
        return (t[k], t[(k + 1) % 3], t[(k + 2) % 3])

    loops_crc, tris_crc = [], []
    for row in tri_rows:
        tris = tris_of(row)
        directed = {e for a, b, c in tris for e in ((a, b), (b, c), (c, a))}
        nxt = {a: b for a, b in directed if (b, a) not in directed}
        loops, seen = [], set()
        for s0 in sorted(nxt):
            if s0 in seen:
                continue
            loop, cur = [], s0
            while cur not in seen:
                seen.add(cur)
                loop.append(cur)
                cur = nxt[cur]
            k = loop.index(min(loop))
            loops.append(tuple(loop[k:] + loop[:k]))
        loops_crc.append(zlib.crc32(repr(sorted(loops)).encode()))
        tris_crc.append(zlib.crc32(repr(sorted(canon(t) for t in tris)).encode()))
    return np.array(loops_crc, np.uint32), np.array(tris_crc, np.uint32)


def ref_mc():
    """Digests of the reference's marching-cubes case tables (src/mc_constants.h compiled where it lies,
    oracle/ref_mc.cpp).  EdgeTable and NumVertsTable follow from first principles (crossed edges; triangle count of
    the case) and are stored as numbers; of TriTable only per-case CRC32 digests are kept."""
    so = os.path.join(O.ORACLE_DIR, "_ref", "libref_mc.so")
    assert os.path.exists(so), "oracle/_ref/libref_mc.so missing (needs /root/reference)"
    R = C.CDLL(so)
    for f in (R.ref_mc_edge, R.ref_mc_numverts, R.ref_mc_tri):
        f.restype = C.c_uint32
    tri = [[R.ref_mc_tri(ci, j) for j in range(16)] for ci in range(256)]
    loops_crc, tris_crc = mc_case_digests(tri)
    np.savez_compressed(os.path.join(HERE, "ref_mc_digest.npz"),
                        edge=np.array([R.ref_mc_edge(ci) for ci in range(256)], np.uint32),
                        numverts=np.array([R.ref_mc_numverts(ci) for ci in range(256)], np.uint32),
                        loops_crc=loops_crc, tris_crc=tris_crc)
    print("ref_mc_digest.npz: 4 arrays")


def by_id(d):
    o = np.argsort(d["id"], kind="stable")
    return {k: v[o] for k, v in d.items()}


def oracle_selfcheck():
    out = {}
    for fp64, nm in ((False, "f32"), (True, "f64")):
        s = O.scene_cubes(1024, fp64)  # 2 x 8^3
        for mode, sort, tag in ((O.JACOBI, O.SORT_STABLE, "jacobi"), (O.GS, O.SORT_STD, "gs")):
            o = O.Oracle(fp64)
            o.set_particles(**s)
            p = O.make_params(mode=mode, sort=sort, threads=1, iteration=4)
            for frame in (1, 2, 3):
                o.step(p)
                if frame in (1, 3):
                    q = by_id(o.get_particles())
                    out[f"cubes1024_{nm}_{tag}_f{frame}_pos"] = q["pos"]
                    out[f"cubes1024_{nm}_{tag}_f{frame}_vel"] = q["vel"]
                    out[f"cubes1024_{nm}_{tag}_f{frame}_colour"] = q["colour"]
        # stage-level vectors, frame 1, jacobi/stable: keys, table, lambda after iteration 1
        o = O.Oracle(fp64)
        o.set_particles(**s)
        p = O.make_params(mode=O.JACOBI, sort=O.SORT_STABLE, threads=1, iteration=4)
        o.predict(p).sort(p).grid_table(p)
        out[f"cubes1024_{nm}_sorted_ids"] = o.get_particles()["id"]
        out[f"cubes1024_{nm}_keys"] = o.keys()
        o.diffuse(p).lambda_(p)
        out[f"cubes1024_{nm}_lambda1"] = o.lambdas()
        o.delta(p)
        out[f"cubes1024_{nm}_pstar1"] = o.pstar()
    np.savez_compressed(os.path.join(HERE, "oracle_selfcheck.npz"), **out)
    print("oracle_selfcheck.npz:", len(out), "arrays")


if __name__ == "__main__":
    O.build()
    ref_grid()
    ref_mc()
    oracle_selfcheck()
