// hipsph.hpp — sph::hip_impl::Solver<T, N, V>: the MI355X backend behind the reference's
// Solver::advance() surface (reference src/sph.hpp:119-125; sibling of omp_impl::Solver,
// src/omp/ompsph.hpp:77-83).  A thin shim: every number is computed by libpbf_hip.so through the
// C ABI in include/pbf_hip.h.  There is no CPU fallback — construction throws if no gfx950 device
// is usable.
//
//   advance()          the reference's contract: xs is uploaded, stepped once, downloaded in Z-order
//                      (like the reference's OpenCL backend re-uploads every frame,
//                      src/ocl/oclsph.cpp:427-473);
//   upload()/step()/download()   the device-resident fast path the benchmark times as well.
//
// Host-side scene handling restates the observable behaviour of ompsph.hpp:91-126 (sources emit,
// drains erase, empty -> "Particles depleted") and :167-186 (queries).  config.surface runs the
// marching-cubes kernels (pbf_surface) and fills Result::mesh like ompsph.hpp:277-477.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pbf_hip.h"
#include "sph.hpp"

namespace sph::hip_impl {

template <typename T, typename N, template <size_t, typename C = N> typename V = sph::vec>
class Solver final : public sph::Solver<T, N, V> {
  static_assert(std::is_same_v<N, float> || std::is_same_v<N, double>, "N must be float or double");
  static_assert(sizeof(T) == 8, "ids travel as 64-bit (the reference instantiates T = size_t)");
  static_assert(sizeof(V<3>) == 3 * sizeof(N) && sizeof(V<4>) == 4 * sizeof(N), "V must be packed");

  pbf_ctx *ctx_ = nullptr;
  const N h_;
  std::vector<double> wells_;

  void check(int rc, const char *what) const {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + pbf_last_error(ctx_));
  }

  static pbf_aos_layout layout() {
    using P = sph::Particle<T, N, V>;
    const P probe{};
    auto off = [&](const void *m) {
      return uint32_t(reinterpret_cast<const char *>(m) - reinterpret_cast<const char *>(&probe));
    };
    return {uint32_t(sizeof(P)), off(&probe.id),       off(&probe.type),    off(&probe.mass),
            off(&probe.position), off(&probe.velocity), off(&probe.colour)};
  }

  pbf_params params(const sph::SphParams<T, N, V> &c, const sph::Scene<T, N, V> &scene) {
    pbf_params p{};
    p.dt = double(c.dt), p.scale = double(c.scale), p.iteration = c.iteration;
    p.constant_force[0] = c.constantForce.x, p.constant_force[1] = c.constantForce.y,
    p.constant_force[2] = c.constantForce.z;
    p.min_bound[0] = c.minBound.x, p.min_bound[1] = c.minBound.y, p.min_bound[2] = c.minBound.z;
    p.max_bound[0] = c.maxBound.x, p.max_bound[1] = c.maxBound.y, p.max_bound[2] = c.maxBound.z;
    wells_.clear();
    for (const auto &w : scene.wells) {
      wells_.push_back(w.centre.x), wells_.push_back(w.centre.y), wells_.push_back(w.centre.z);
      wells_.push_back(double(w.force));
    }
    p.n_wells = int32_t(scene.wells.size());
    p.wells = wells_.empty() ? nullptr : wells_.data();
    return p;
  }

public:
  explicit Solver(N h, int device = 0, uint32_t flags = 0) : h_(h) {
    pbf_desc d{};
    d.abi_version = PBF_ABI_VERSION;
    d.fp64 = std::is_same_v<N, double> ? 1 : 0;
    d.device = device;
    d.flags = flags;
    d.h = double(h);
    d.stream = nullptr;
    const int rc = pbf_create(&d, &ctx_);
    if (rc != PBF_OK) throw std::runtime_error(std::string("pbf_create: ") + pbf_last_error(nullptr));
  }
  ~Solver() override { pbf_destroy(ctx_); }
  Solver(const Solver &) = delete;
  Solver &operator=(const Solver &) = delete;

  pbf_ctx *context() { return ctx_; }

  // ---- device-resident path -------------------------------------------------------------------
  void upload(const std::vector<sph::Particle<T, N, V>> &xs) {
    const auto l = layout();
    check(pbf_upload_aos(ctx_, xs.size(), xs.data(), &l), "pbf_upload_aos");
  }
  void step(const sph::SphParams<T, N, V> &config, const sph::Scene<T, N, V> &scene = {}, uint32_t count = 1) {
    const pbf_params p = params(config, scene);
    check(pbf_steps(ctx_, &p, count), "pbf_steps");
  }
  void sync() { check(pbf_sync(ctx_), "pbf_sync"); }
  void download(std::vector<sph::Particle<T, N, V>> &xs) {
    xs.resize(pbf_count(ctx_));
    const auto l = layout();
    check(pbf_download_aos(ctx_, xs.data(), &l), "pbf_download_aos");
  }

  // ---- the reference's contract ---------------------------------------------------------------
  sph::Result<T, N, V> advance(const sph::SphParams<T, N, V> &config, const sph::Scene<T, N, V> &scene,
                               std::vector<sph::Particle<T, N, V>> &xs) final {
    // sources: a floor(sqrt(rate)) x ceil(sqrt(rate)) sheet at spacing h*scale/2 (ompsph.hpp:93-105)
    const N spacing = h_ * config.scale / 2;
    for (const auto &src : scene.sources) {
      const N size = std::sqrt(N(src.rate));
      const size_t width = size_t(std::floor(size)), depth = size_t(std::ceil(size));
      const V<3> corner = src.centre - (V<3>(width, 0, depth) * N(0.5)) * spacing;
      for (size_t x = 0; x < width; ++x)
        for (size_t z = 0; z < depth; ++z)
          xs.emplace_back(src.tag, sph::Type::Fluid, N(1), src.colour, corner + V<3>(x, 0, z) * spacing, src.velocity);
    }
    // drains: fluid within `width` of a drain centre disappears (ompsph.hpp:107-118)
    if (!scene.drains.empty())
      xs.erase(std::remove_if(xs.begin(), xs.end(),
                              [&](const sph::Particle<T, N, V> &p) {
                                if (p.type == sph::Type::Obstacle) return false;
                                for (const auto &d : scene.drains) {
                                  const V<3> r = p.position - d.centre;
                                  if (std::sqrt(r.x * r.x + r.y * r.y + r.z * r.z) < d.width) return true;
                                }
                                return false;
                              }),
               xs.end());
    if (xs.empty()) {  // ompsph.hpp:122-126
      std::cout << "Particles depleted" << std::endl;
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
      return {};
    }
    upload(xs);
    step(config, scene, 1);
    sph::Result<T, N, V> result;
    if (!scene.queries.empty()) result.queries = query(config, scene);
    if (config.surface) result.mesh = surface(config, scene);  // ompsph.hpp:277-477
    download(xs);
    return result;
  }

  // Marching-cubes surface of the state the last step left (reference: config.surface, ompsph.hpp:277-477).
  sph::ColouredMesh<N, V> surface(const sph::SphParams<T, N, V> &config, const sph::Scene<T, N, V> &scene = {}) {
    const pbf_params p = params(config, scene);
    const pbf_mc_params mc{double(config.surface->resolution), double(config.surface->isolevel),
                           double(config.surface->particleSize), double(config.surface->particleInfluence)};
    uint64_t triangles = 0;
    check(pbf_surface(ctx_, &p, &mc, &triangles), "pbf_surface");
    sph::ColouredMesh<N, V> mesh(size_t(triangles) * 3);
    check(pbf_download_mesh(ctx_, mesh.vs.data(), mesh.ns.data(), mesh.cs.data()), "pbf_download_mesh");
    return mesh;
  }

private:
  // ids of the fluid particles in the cell that holds each query point (ompsph.hpp:167-186);
  // the cells are those of the predicted positions, as in the reference.
  std::vector<sph::QueryResult<T, N, V>> query(const sph::SphParams<T, N, V> &config,
                                               const sph::Scene<T, N, V> &scene) {
    const size_t n = pbf_count(ctx_);
    std::vector<uint32_t> keys(n);
    std::vector<uint64_t> ids(n);
    std::vector<uint8_t> types(n);
    check(pbf_read_buffer(ctx_, PBF_BUF_KEYS, keys.data(), n * 4), "pbf_read_buffer(keys)");
    check(pbf_download(ctx_, ids.data(), types.data(), nullptr, nullptr, nullptr, nullptr), "pbf_download(ids)");
    uint64_t extent[3];
    double minExtent[3];
    check(pbf_grid_extent(ctx_, extent, minExtent), "pbf_grid_extent");
    const size_t tableN = pbf_table_size(ctx_);
    auto spread = [](uint64_t v) {
      uint32_t x = uint32_t(v);
      x = (x | (x << 16)) & 0x030000FFu, x = (x | (x << 8)) & 0x0300F00Fu;
      x = (x | (x << 4)) & 0x030C30C3u, x = (x | (x << 2)) & 0x09249249u;
      return x;
    };
    std::vector<sph::QueryResult<T, N, V>> out(scene.queries.size());
    for (size_t i = 0; i < scene.queries.size(); ++i) {
      const auto &q = scene.queries[i];
      const N sx = q.point.x / config.scale - N(minExtent[0]), sy = q.point.y / config.scale - N(minExtent[1]),
              sz = q.point.z / config.scale - N(minExtent[2]);
      const uint32_t code = spread(uint64_t(int64_t(sx / h_))) | (spread(uint64_t(int64_t(sy / h_))) << 1) |
                            (spread(uint64_t(int64_t(sz / h_))) << 2);
      std::vector<T> found;
      if (size_t(code) + 1 < tableN) {
        auto lo = std::lower_bound(keys.begin(), keys.end(), code), hi = std::upper_bound(lo, keys.end(), code);
        for (auto it = lo; it != hi; ++it) {
          const size_t a = size_t(it - keys.begin());
          if (types[a] == uint8_t(sph::Type::Fluid)) found.push_back(T(ids[a]));
        }
      }
      out[i] = {q.id, q.point, found};
    }
    return out;
  }
};

}  // namespace sph::hip_impl
