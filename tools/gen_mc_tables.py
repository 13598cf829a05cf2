#!/usr/bin/env python3
"""Generate marching-cubes case tables (EdgeTable / TriTable / NumVertsTable) from first principles.

The reference keeps the classic tables in src/mc_constants.h.  They are data we may not copy and
cannot fetch offline, so this script DERIVES an equivalent set:

  corners  : the reference's CUBE_OFFSETS order (ompsph.hpp:356-358)
             0 (0,0,0) 1 (1,0,0) 2 (1,1,0) 3 (0,1,0) 4 (0,0,1) 5 (1,0,1) 6 (1,1,1) 7 (0,1,1)
  edges    : the reference's lerpAll pairs (ompsph.hpp:439-450)
             0:0-1 1:1-2 2:2-3 3:3-0 4:4-5 5:5-6 6:6-7 7:7-4 8:0-4 9:1-5 10:2-6 11:3-7
  case bit : bit i set <=> value at corner i < isolevel (ompsph.hpp:373,426)

For every case the iso-surface crosses the edges whose end corners differ.  On each cube face the
crossings are joined by segments — two crossings: one segment; four (corners alternate: the
ambiguous face): the two segments that each cut off one SET corner.  That rule depends only on
the face's own four corner states, so the two cubes sharing a face draw the same segments and the
mesh is watertight.  Segments close into loops (every crossed edge lies on exactly two faces); each
loop is wound so that its normal points away from the set corners and triangulated as a fan.

The surface is the same as the classic table's up to how each polygon is split into triangles (and
up to the classic table's choice on ambiguous faces); triangle counts therefore need not equal the
reference's "Final Vertex count" case by case.  Properties are checked in tests/test_mc_tables.py.

Writes pbf-sph_amd/csrc/mc_tables.hpp and oracle/mc_tables.h (identical content).
"""
import itertools
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
EDGE_OF = {frozenset(e): i for i, e in enumerate(EDGES)}


def faces():
    """Each face as its 4 corners in cyclic order, counter-clockwise when seen from OUTSIDE the cube."""
    out = []
    for axis in range(3):
        for side in (0, 1):
            cs = [i for i, c in enumerate(CORNERS) if c[axis] == side]
            cyc = [cs[0]]
            while len(cyc) < 4:  # walk neighbours that differ in exactly one coordinate
                for c in cs:
                    if c not in cyc and sum(a != b for a, b in zip(CORNERS[c], CORNERS[cyc[-1]])) == 1:
                        cyc.append(c)
                        break
            p0, p1, p2 = (CORNERS[c] for c in cyc[:3])
            n = cross(sub(p1, p0), sub(p2, p1))
            outward = tuple((1 if side else -1) if k == axis else 0 for k in range(3))
            if dot(n, outward) < 0:
                cyc = [cyc[0]] + cyc[:0:-1]
            out.append(cyc)
    return out


def sub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def dot(a, b):
    return sum(x * y for x, y in zip(a, b))


FACES = faces()


def case_triangles(ci):
    """Directed segments: on a face seen from outside (corners counter-clockwise), edge k joins corner k and
    k+1; it is an OUT crossing if corner k is set and k+1 is not, an IN crossing the other way round.  Every
    segment runs from an OUT crossing to the IN crossing that closes off the same run of set corners (on an
    ambiguous face: the same single corner), i.e. it keeps the set corners on its left.  A cube edge is OUT on
    one of its two faces and IN on the other, so the segments chain into directed loops, and because the cube
    next door sees the shared face from the other side, the whole surface is consistently oriented."""
    inside = [(ci >> i) & 1 for i in range(8)]
    nxt = {}
    for f in FACES:
        edges = [EDGE_OF[frozenset((f[k], f[(k + 1) % 4]))] for k in range(4)]
        for k in range(4):
            if inside[f[k]] and not inside[f[(k + 1) % 4]]:  # OUT crossing at edge k: walk back to the run's IN crossing
                j = k
                while inside[f[j % 4]]:
                    j -= 1
                assert edges[k] not in nxt
                nxt[edges[k]] = edges[j % 4]
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, cur = [], start
        while cur not in seen:
            seen.add(cur)
            loop.append(cur)
            cur = nxt[cur]
        assert cur == start, (ci, loop)
        loops.append(loop)
    tris = []
    for loop in loops:
        assert len(loop) >= 3, (ci, loop)
        tris.extend(triangulate(loop))
    return tris


def share_face(e1, e2):
    """Do two cube edges lie in a common face?  A polygon diagonal between them would lie IN that face,
    where the neighbouring cube may draw the same diagonal: four triangles on one edge (a pinched,
    non-manifold surface).  Such diagonals are avoided."""
    for f in FACES:
        s = set(f)
        if set(EDGES[e1]) <= s and set(EDGES[e2]) <= s:
            return True
    return False


def all_triangulations(poly):
    """Every triangulation of a convex polygon given as a vertex list (orientation preserved)."""
    if len(poly) == 3:
        yield [tuple(poly)]
        return
    a, b = poly[0], poly[1]
    for k in range(2, len(poly)):  # the triangle on edge (a, b) has apex poly[k]
        left = [[]] if k == 2 else all_triangulations(poly[1:k + 1])
        right = [[]] if k == len(poly) - 1 else all_triangulations([poly[0]] + poly[k:])
        for l in left:
            for r in right:
                yield [(a, b, poly[k])] + list(l) + list(r)


def triangulate(loop):
    """The first triangulation (fans first) none of whose diagonals lies in a cube face."""
    boundary = {frozenset((loop[k], loop[(k + 1) % len(loop)])) for k in range(len(loop))}
    best, best_bad = None, None
    for tri in all_triangulations(loop):
        diagonals = {frozenset(e) for t in tri for e in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0]))} - boundary
        bad = sum(1 for d in diagonals if share_face(*tuple(d)))
        if best is None or bad < best_bad:
            best, best_bad = tri, bad
        if bad == 0:
            break
    return best


def build():
    tri, edge, nverts = [], [], []
    for ci in range(256):
        t = case_triangles(ci) if ci not in (0, 255) else []
        flat = list(itertools.chain.from_iterable(t))
        assert len(flat) <= 15, (ci, len(flat))
        used = 0
        for e in flat:
            used |= 1 << e
        # every crossed edge is used and nothing else
        crossed = 0
        for i, (a, b) in enumerate(EDGES):
            if ((ci >> a) & 1) != ((ci >> b) & 1):
                crossed |= 1 << i
        assert used == crossed, (ci, bin(used), bin(crossed))
        tri.append(flat + [255] * (16 - len(flat)))
        edge.append(crossed)
        nverts.append(len(flat))
    return edge, tri, nverts


def emit(path, edge, tri, nverts, guard):
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_mc_tables.py — do not edit.  Marching-cubes case tables derived from first\n"
                "// principles (see the generator's docstring): same corner / edge numbering as the reference\n"
                "// (src/omp/ompsph.hpp:356-358,439-450), our own polygon triangulation.\n"
                f"#ifndef {guard}\n#define {guard}\n#include <stdint.h>\n"
                "#if defined(__HIPCC__)\n#define PBF_MC_TABLE static __device__ __constant__ const\n#else\n"
                "#define PBF_MC_TABLE static const\n#endif\n\n")
        f.write("// bit e set <=> edge e is crossed in that case\nPBF_MC_TABLE uint16_t kMcEdgeTable[256] = {\n")
        for r in range(0, 256, 16):
            f.write("    " + ", ".join(f"0x{v:03x}" for v in edge[r:r + 16]) + ",\n")
        f.write("};\n\n// number of triangle vertices (3 per triangle) in that case\nPBF_MC_TABLE uint8_t kMcNumVerts[256] = {\n")
        for r in range(0, 256, 16):
            f.write("    " + ", ".join(f"{v:2d}" for v in nverts[r:r + 16]) + ",\n")
        f.write("};\n\n// edge indices, three per triangle, 255-terminated\nPBF_MC_TABLE uint8_t kMcTriTable[256][16] = {\n")
        for row in tri:
            f.write("    {" + ", ".join(f"{v:3d}" for v in row) + "},\n")
        f.write("};\n#endif\n")


if __name__ == "__main__":
    e, t, n = build()
    emit(os.path.join(ROOT, "pbf-sph_amd", "csrc", "mc_tables.hpp"), e, t, n, "PBF_MC_TABLES_HPP")
    emit(os.path.join(ROOT, "oracle", "mc_tables.h"), e, t, n, "PBF_ORACLE_MC_TABLES_H")
    print("cases with triangles:", sum(1 for v in n if v), "max verts:", max(n), "total tris:", sum(n) // 3)
