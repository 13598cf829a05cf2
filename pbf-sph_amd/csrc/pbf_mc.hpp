// pbf_mc.hpp — marching-cubes surface extraction on the device (reference: src/omp/ompsph.hpp:277-477,
// OpenCL kernels mc_lattice / mc_size / mc_eval in src/ocl/oclsph_kernel.h:176-408).
//   k_mc_field : one lane per lattice node — 27-cell gather (cells clamped to the grid, so they may
//                repeat at the domain faces exactly like the reference) of  v = sum size / |l|^infl,
//                the un-normalised gradient and the mean colour of the particles within h*scale;
//   k_mc_count : one lane per cube — case index from the 8 corner values, triangles of that case;
//                an exclusive scan of the counts (the grid-table scan kernels) gives every cube its
//                output offset, so triangles come out in cube order, deterministically (the
//                reference appends through an atomic counter, i.e. in arbitrary order);
//   k_mc_emit  : one lane per cube — edge intersections by linear interpolation, triangles from the
//                case table.
// Case tables: mc_tables.hpp, generated from first principles by tools/gen_mc_tables.py (the
// reference's src/mc_constants.h is data we may not copy): same corner / edge numbering, our own
// polygon triangulation — the surface is the same, the triangle count per case need not be.
// glm's fastDistance / fastLength / fastNormalize (approximations without a bit contract, library
// absent offline) are evaluated with the exact sqrt, like the oracle.
#pragma once

#include "mc_tables.hpp"
#include "pbf_kernels.hpp"

namespace pbf {

template <typename N> struct McConsts {
  N scale, res, isolevel, particleSize, particleInfluence, step, threshold;
  N minExtent[3];
  uint32_t extent[3];
  uint32_t sample[3];
  uint32_t tableN;
  uint32_t hasObstacles;
  // slab mode (pbf_surface after pbf_slab_step; all zero / equal on a single device): this rank's lattice holds the node
  // planes x = nodeX0 .. nodeX0 + sample[0] - 1 of the global lattice (the last one, for every rank but the rightmost,
  // received from the right-hand neighbour), `planes` of them computed here; keys live in the rank's x frame
  uint32_t xoff, nodeX0, planes;
};

// nearMask[code]: bit k = 1 iff slot k of that cell's 27-slot walk (k = (dz * 3 + dy) * 3 + dx, the reference's order,
// coordinates clamped to the grid so that slots may repeat a cell at the domain faces — ompsph.hpp:305-330) holds a
// non-empty cell (round 3).  The lattice of the 1 M dam-break has 7.2 M nodes; the settled fluid is a sheet, so most of a
// node's 27 cells are empty and each used to cost two dependent table loads to find that out: 2.2 ms.  Scatter form: every
// occupied cell c' sets, for each offset d, the bit of the cells c with clamp(c + d) = c' — per axis c' - d when that is
// inside the grid, and c' itself when the clamp folds d back onto it (d = -1 at coordinate 0, d = +1 at the last one).
__global__ __launch_bounds__(BLOCK) void k_mc_mark_near(uint32_t tableN, uint3 extent, uint32_t xoff,
                                                        const uint32_t *__restrict__ table,
                                                        uint32_t *__restrict__ nearMask) {
  const uint32_t code = blockIdx.x * BLOCK + threadIdx.x;
  if (code + 1u >= tableN) return;  // (the table's last cell is empty by definition, sph.hpp:208)
  if (table[code + 1u] == table[code]) return;
  // (x in GLOBAL cell coordinates: the clamp folds happen at the global faces; slab keys carry x - xoff)
  const int c[3] = {int(compact10(code) + xoff), int(compact10(code >> 1)), int(compact10(code >> 2))};
  const int ext[3] = {int(extent.x), int(extent.y), int(extent.z)};
  if (c[0] >= ext[0] || c[1] >= ext[1] || c[2] >= ext[2]) return;  // (a code below tableN outside the box: never a node's cell)
  for (int k = 0; k < 27; ++k) {
    const int d[3] = {k % 3 - 1, (k / 3) % 3 - 1, k / 9 - 1};
    int sol[3][2], ns[3];
    for (int a = 0; a < 3; ++a) {
      ns[a] = 0;
      const int v = c[a] - d[a];
      if (v >= 0 && v < ext[a]) sol[a][ns[a]++] = v;
      if ((d[a] == -1 && c[a] == 0) || (d[a] == 1 && c[a] == ext[a] - 1)) sol[a][ns[a]++] = c[a];
    }
    for (int i = 0; i < ns[0]; ++i)
      for (int j = 0; j < ns[1]; ++j)
        for (int l = 0; l < ns[2]; ++l)
          if (sol[0][i] >= int(xoff)) {
            const uint32_t at = morton_encode(uint32_t(sol[0][i]) - xoff, uint32_t(sol[1][j]), uint32_t(sol[2][l]));
            if (at < tableN) atomicOr(&nearMask[at], 1u << k);  // (a slab's table ends at its right ghost column)
          }
  }
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_mc_field(McConsts<N> m, const uint32_t *__restrict__ table,
                                                    const vec4<N> *__restrict__ pos4, const vec4<N> *__restrict__ pstar,
                                                    const vec4<N> *__restrict__ col4,
                                                    const uint8_t *__restrict__ type, const uint32_t *__restrict__ nearMask,
                                                    vec4<N> *__restrict__ latticePN,
                                                    vec4<N> *__restrict__ latticeC) {
  // lane -> node: 8 consecutive lanes take a 2 x 2 x 2 block of nodes (one grid cell at the stock resolution 2: the same 27
  // cells, so their table and candidate loads coalesce into one request), 64 lanes a 4 x 4 x 4 block
  const uint32_t sx = m.planes, sy = m.sample[1], sz = m.sample[2];  // (slab mode: the received plane is not computed here)
  const uint32_t bx = (sx + 3u) / 4u, by = (sy + 3u) / 4u, bz = (sz + 3u) / 4u;
  const uint32_t t = blockIdx.x * BLOCK + threadIdx.x, blk = t >> 6, l = t & 63u;
  if (blk >= bx * by * bz) return;
  const uint32_t x = (blk / (by * bz)) * 4u + ((l >> 5) & 1u) * 2u + ((l >> 2) & 1u),
                 y = ((blk / bz) % by) * 4u + ((l >> 4) & 1u) * 2u + ((l >> 1) & 1u),
                 z = (blk % bz) * 4u + ((l >> 3) & 1u) * 2u + (l & 1u);
  if (x >= sx || y >= sy || z >= sz) return;
  const uint32_t idx = (x * sy + y) * sz + z;  // index3d (curves.h:17-19)
  const N px = N(x + m.nodeX0), py = N(y), pz = N(z);
  const N ax = (m.minExtent[0] + (px * m.step)) * m.scale, ay = (m.minExtent[1] + (py * m.step)) * m.scale,
          az = (m.minExtent[2] + (pz * m.step)) * m.scale;
  // the node's cell (ompsph.hpp:293-298); coordinates pass through the 10-bit Morton encode / decode
  const uint32_t zX = uint32_t(uint64_t(px / m.res)) & 1023u, zY = uint32_t(uint64_t(py / m.res)) & 1023u,
                 zZ = uint32_t(uint64_t(pz / m.res)) & 1023u;
  const vec4<N> zero = make_vec4<N>(N(0), N(0), N(0), N(0));
  if (zX == m.extent[0] && zY == m.extent[1] && zZ == m.extent[2]) {  // ompsph.hpp:300-303
    latticePN[idx] = zero, latticeC[idx] = zero;
    return;
  }
  auto cl = [](int v, int hi) { return uint32_t(min(max(v, 0), hi)); };
  const uint32_t xs[3] = {cl(int(zX) - 1, int(m.extent[0]) - 1), zX, cl(int(zX) + 1, int(m.extent[0]) - 1)};
  const uint32_t ys[3] = {cl(int(zY) - 1, int(m.extent[1]) - 1), zY, cl(int(zY) + 1, int(m.extent[1]) - 1)};
  const uint32_t zs[3] = {cl(int(zZ) - 1, int(m.extent[2]) - 1), zZ, cl(int(zZ) + 1, int(m.extent[2]) - 1)};
  N v = 0, nx = 0, ny = 0, nz = 0, cr = 0, cg = 0, cb = 0, ca = 0;
  uint32_t nNeighbours = 0;
  const N ninf = (-m.particleInfluence) * m.particleSize;
  // which of the 27 slots hold a non-empty cell (a node whose cell lies outside the grid looks every slot up itself)
  const N t2loose = (m.threshold * m.threshold) * N(1.000001);
  const bool inflHalf = m.particleInfluence == N(0.5);
  const bool inside = zX < m.extent[0] && zY < m.extent[1] && zZ < m.extent[2];
  uint32_t slots = inside ? nearMask[morton_encode(zX - m.xoff, zY, zZ)] : 0x07FFFFFFu;
  // candidates in the reference's order, four loads in flight per trip
  auto fold = [&](const vec4<N> &p, uint32_t b) {
    const N lx = p.x - ax, ly = p.y - ay, lz = p.z - az;
    const N d2 = lx * lx + ly * ly + lz * lz;
    // five of six candidates lie beyond the threshold: a conservative test on the SQUARE keeps the exact sqrt (and
    // the pow and divides behind it) for the rest — sqrt is monotonic, so len < threshold implies d2 < threshold^2
    // (1 + 2^-20) in either precision, and whoever passes here is still tested exactly as the reference writes it
    if (!(d2 < t2loose)) return;
    const N len = sqrt(d2);
    if (!(len < m.threshold)) return;
    // The hits' arithmetic is what the kernel spends its time on (a libm-style pow and four IEEE divides per hit: ~130
    // VALU).  The stock influence 0.5 (sph.hpp:183) makes pow(len, 0.5) a square root — correctly rounded, i.e. at least
    // as close to the reference's glm::pow as the device's pow is —, and the four quotients share ONE refined reciprocal
    // with an exact-residual correction each (div_ranged: the IEEE quotient for operands in the normal range, which a
    // hit's are unless it sits on the node itself: then, wave-wide, the compiler's divide).
    const N denominator = inflHalf ? sqrt(len) : pow(len, m.particleInfluence);
    if (__any(!(denominator > N(1e-18)))) {
      v += (m.particleSize / denominator);
      nx = nx + (lx / denominator) * ninf, ny = ny + (ly / denominator) * ninf, nz = nz + (lz / denominator) * ninf;
    } else {
      v += div_ranged(m.particleSize, denominator);
      nx = nx + div_ranged(lx, denominator) * ninf, ny = ny + div_ranged(ly, denominator) * ninf,
      nz = nz + div_ranged(lz, denominator) * ninf;
    }
    const vec4<N> c = col4[b];
    cr += c.x, cg += c.y, cb += c.z, ca += c.w;
    nNeighbours++;
  };
  while (slots) {  // ascending slot number = the reference's order (x fastest, then y, then z)
    const int k = __builtin_ctz(slots);
    slots &= slots - 1u;
    const uint32_t off = morton_encode(xs[k % 3] - m.xoff, ys[(k / 3) % 3], zs[k / 9]);
    if (off >= m.tableN) continue;
    const uint32_t s0 = table[off], e0 = (off + 1u) < m.tableN ? table[off + 1u] : s0;
    for (uint32_t b = s0; b < e0; b += 4u) {
      vec4<N> p[4];
      bool ok[4];
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t bw = min(b + w, e0 - 1u);  // a tail slot re-reads the last candidate and is masked
        const uint8_t ty = m.hasObstacles ? type[bw] : uint8_t(0);
        ok[w] = b + w < e0 && !(ty & TYPE_OBSTACLE);
        p[w] = pos4[bw];
        if (ty & TYPE_GHOST) {  // a neighbouring slab's particle: its copy carries pStar, and position = pStar * scale
          const vec4<N> q = pstar[bw];  // (ompsph.hpp:260, the very product the owner's finalise stored)
          p[w] = make_vec4<N>(q.x * m.scale, q.y * m.scale, q.z * m.scale, N(0));
        }
      }
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w)
        if (ok[w]) fold(p[w], b + w);
    }
  }
  const N inv = N(1) / sqrt(nx * nx + ny * ny + nz * nz);
  latticePN[idx] = make_vec4<N>(v, nx * inv, ny * inv, nz * inv);
  const N nn = N(nNeighbours);
  latticeC[idx] = make_vec4<N>(cr / nn, cg / nn, cb / nn, ca / nn);
}

__device__ inline void mc_cube_origin(uint32_t i, const uint32_t sample[3], uint32_t &px, uint32_t &py, uint32_t &pz) {
  const uint32_t ry = sample[1] - 1u, rz = sample[2] - 1u;  // utils::to3d over the march range
  px = i / (ry * rz), py = (i / rz) % ry, pz = i % rz;
}

template <typename N>
__device__ inline uint32_t mc_case(const McConsts<N> &m, const vec4<N> *__restrict__ latticePN, uint32_t px, uint32_t py,
                                   uint32_t pz, N values[8], uint32_t node[8]) {
  const uint32_t CX[8] = {0, 1, 1, 0, 0, 1, 1, 0}, CY[8] = {0, 0, 1, 1, 0, 0, 1, 1}, CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  uint32_t ci = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // CUBE_OFFSETS (ompsph.hpp:356-358)
    node[k] = ((px + CX[k]) * m.sample[1] + (py + CY[k])) * m.sample[2] + (pz + CZ[k]);
    values[k] = latticePN[node[k]].x;
    if (values[k] < m.isolevel) ci |= 1u << k;
  }
  return ci;
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_mc_count(McConsts<N> m, uint32_t marchVolume,
                                                    const vec4<N> *__restrict__ latticePN,
                                                    uint32_t *__restrict__ counts) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= marchVolume) return;
  uint32_t px, py, pz, node[8];
  N values[8];
  mc_cube_origin(i, m.sample, px, py, pz);
  const uint32_t ci = mc_case<N>(m, latticePN, px, py, pz, values, node);
  counts[i] = kMcEdgeTable[ci] == 0 ? 0u : uint32_t(kMcNumVerts[ci]) / 3u;  // ompsph.hpp:377
}

template <typename N>
__global__ __launch_bounds__(BLOCK) void k_mc_emit(McConsts<N> m, uint32_t marchVolume,
                                                   const vec4<N> *__restrict__ latticePN,
                                                   const vec4<N> *__restrict__ latticeC,
                                                   const uint32_t *__restrict__ offsets, N *__restrict__ outV,
                                                   N *__restrict__ outN, N *__restrict__ outC) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= marchVolume) return;
  if (offsets[i + 1] == offsets[i]) return;
  uint32_t px, py, pz, node[8];
  N values[8];
  mc_cube_origin(i, m.sample, px, py, pz);
  const uint32_t ci = mc_case<N>(m, latticePN, px, py, pz, values, node);
  const uint32_t CX[8] = {0, 1, 1, 0, 0, 1, 1, 0}, CY[8] = {0, 0, 1, 1, 0, 0, 1, 1}, CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  const int EF[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, ET[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};  // lerpAll pairs
  uint32_t w = offsets[i] * 3u;
  for (int k = 0; kMcTriTable[ci][k] != 255; ++k, ++w) {
    const int e = kMcTriTable[ci][k];
    const int f = EF[e], t = ET[e];
    const N wgt = (m.isolevel - values[f]) / (values[t] - values[f]);  // utils::scale (utils.hpp:85)
    auto mix = [&](N a, N b) { return a * (N(1) - wgt) + b * wgt; };   // glm::mix
    auto coord = [&](uint32_t c, int ax) { return (m.minExtent[ax] + (N(c) * m.step)) * m.scale; };
    const vec4<N> pf = latticePN[node[f]], pt = latticePN[node[t]], cf = latticeC[node[f]], ct = latticeC[node[t]];
    outV[3 * w + 0] = mix(coord(px + m.nodeX0 + CX[f], 0), coord(px + m.nodeX0 + CX[t], 0));
    outV[3 * w + 1] = mix(coord(py + CY[f], 1), coord(py + CY[t], 1));
    outV[3 * w + 2] = mix(coord(pz + CZ[f], 2), coord(pz + CZ[t], 2));
    outN[3 * w + 0] = mix(pf.y, pt.y), outN[3 * w + 1] = mix(pf.z, pt.z), outN[3 * w + 2] = mix(pf.w, pt.w);
    outC[4 * w + 0] = mix(cf.x, ct.x), outC[4 * w + 1] = mix(cf.y, ct.y), outC[4 * w + 2] = mix(cf.z, ct.z),
                 outC[4 * w + 3] = mix(cf.w, ct.w);
  }
}

}  // namespace pbf
