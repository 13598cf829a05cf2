// pbf_scene.cpp — host-side scene factory and parameter helpers of the C ABI (no GPU needed).
//
// Restates the observable behaviour of the reference's scene code (src/sph.hpp:127-186); the
// dam-break scene is ours — the reference has no scene that holds 256 K - 4 M
// particles (its 1000^3 box fits about 2 x 21^3, sph.hpp:165-166,173).
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "pbf_hip.h"

namespace {

template <typename N> struct Writer {
  uint64_t *id;
  uint8_t *type;
  N *mass, *pos, *vel, *colour;
  size_t w = 0;
  void put(uint64_t tag, N x, N y, N z, N r, N g, N b, N a) {
    if (id) {
      id[w] = tag;
      type[w] = PBF_TYPE_FLUID;
      mass[w] = N(1.0);
      pos[3 * w] = x, pos[3 * w + 1] = y, pos[3 * w + 2] = z;
      vel[3 * w] = vel[3 * w + 1] = vel[3 * w + 2] = N(0);
      colour[4 * w] = r, colour[4 * w + 1] = g, colour[4 * w + 2] = b, colour[4 * w + 3] = a;
    }
    ++w;
  }
};

// makeCube (sph.hpp:127-145): len = trunc(cbrt(count)); x outer, z inner; pos = (x,y,z)*spacing + origin
template <typename N>
uint64_t cube(Writer<N> &out, uint64_t tag, N spacing, size_t count, N ox, N oy, N oz, N r, N g, N b, N a) {
  const auto len = static_cast<size_t>(std::cbrt(count));
  for (size_t x = 0; x < len; ++x)
    for (size_t y = 0; y < len; ++y)
      for (size_t z = 0; z < len; ++z)
        out.put(tag++, N(x) * spacing + ox, N(y) * spacing + oy, N(z) * spacing + oz, r, g, b, a);
  return tag;
}

inline size_t icbrt(size_t v) {
  size_t r = static_cast<size_t>(std::cbrt(static_cast<double>(v)));
  while ((r + 1) * (r + 1) * (r + 1) <= v) ++r;
  while (r * r * r > v) --r;
  return r;
}

template <typename N>
size_t cubes(size_t count, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel, void *colour) {
  // simpleConfigWith2Cubes (sph.hpp:165-166)
  Writer<N> w{id, type, (N *)mass, (N *)pos, (N *)vel, (N *)colour};
  uint64_t tag = 0;
  tag = cube<N>(w, tag, N(22.f), count / 2, N(100), N(0), N(100), N(0), N(0.1), N(0.8), N(1));
  tag = cube<N>(w, tag, N(22.f), count / 2, N(600), N(0), N(600), N(0.1), N(0.8), N(0.1), N(1));
  return w.w;
}

template <typename N>
size_t dambreak(size_t nominal, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel, void *colour,
                double *box_side) {
  const size_t nx = icbrt(nominal / 2), ny = 2 * nx, nz = nx;
  const double L = 50.0 * std::ceil(2.5 * double(nx) * 22.0 / 50.0 + 4.0);
  if (box_side) *box_side = L;
  Writer<N> w{id, type, (N *)mass, (N *)pos, (N *)vel, (N *)colour};
  const N spacing = N(22.f);
  const N ox = N(100), oy = N(L - 100.0 - double(ny - 1) * 22.0), oz = N(100);
  uint64_t tag = 0;
  for (size_t x = 0; x < nx; ++x)
    for (size_t y = 0; y < ny; ++y)
      for (size_t z = 0; z < nz; ++z)
        w.put(tag++, N(x) * spacing + ox, N(y) * spacing + oy, N(z) * spacing + oz, N(0), N(0.1), N(0.8), N(1));
  return w.w;
}

}  // namespace

extern "C" {

size_t pbf_scene_cubes(int fp64, size_t count, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel,
                       void *colour) {
  return fp64 ? cubes<double>(count, id, type, mass, pos, vel, colour)
              : cubes<float>(count, id, type, mass, pos, vel, colour);
}

size_t pbf_scene_dambreak(int fp64, size_t nominal, uint64_t *id, uint8_t *type, void *mass, void *pos, void *vel,
                          void *colour, double *box_side) {
  return fp64 ? dambreak<double>(nominal, id, type, mass, pos, vel, colour, box_side)
              : dambreak<float>(nominal, id, type, mass, pos, vel, colour, box_side);
}

void pbf_apply_motion(int fp64, const pbf_params *base, uint64_t frame, pbf_params *out) {
  // applyMotionSinXCosZ (sph.hpp:147-158): offsets are computed in float; the z term is widened
  // to double by `* 0.3` before the conversion to N.
  const float offsetScale = 300.f, offsetRate = 20.f;
  double ox = double(std::sin(float(frame) / offsetRate) * offsetScale);
  double oz = double(std::cos(float(frame) / offsetRate) * offsetScale) * 0.3;
  if (!fp64) ox = double(float(ox)), oz = double(float(oz));
  if (out != base) *out = *base;
  auto add = [&](double v, double o) { return fp64 ? v + o : double(float(v) + float(o)); };
  out->min_bound[0] = add(base->min_bound[0], ox), out->max_bound[0] = add(base->max_bound[0], ox);
  out->min_bound[1] = add(base->min_bound[1], 0.0), out->max_bound[1] = add(base->max_bound[1], 0.0);
  out->min_bound[2] = add(base->min_bound[2], oz), out->max_bound[2] = add(base->max_bound[2], oz);
}

void pbf_default_params(uint64_t iteration, double box_side, pbf_params *out) {
  // simpleConfigWith2Cubes (sph.hpp:168-175): dt = 0.0083 * 1.5f, scale = 500 (benchmark.cpp:25),
  // g = (0, 9.8, 0), bounds [0, side]^3 (1000 in the reference)
  *out = pbf_params{};
  out->dt = 0.0083 * 1.5f;
  out->scale = 500.0;
  out->iteration = iteration;
  out->constant_force[0] = 0, out->constant_force[1] = 9.8, out->constant_force[2] = 0;
  for (int i = 0; i < 3; ++i) out->min_bound[i] = 0.0, out->max_bound[i] = box_side;
  out->n_wells = 0;
  out->wells = nullptr;
}
}
