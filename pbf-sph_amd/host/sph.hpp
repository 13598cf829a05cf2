// sph.hpp — the solver API of the MI355X-native PBF-SPH library.
//
// Mirrors the PUBLIC SURFACE of the reference's src/sph.hpp (same namespace, type names, field
// names and free-function names) so that code written against the reference — its benchmark /
// visualise drivers, its backends' callers — compiles against this header unchanged:
//   Type, Query, QueryResult, Particle, Well, Source, Drain, Scene, McParams, SphParams,
//   ColouredMesh, Result, Solver                       (reference src/sph.hpp:15-125)
//   makeCube, applyMotionSinXCosZ, simpleConfigWith2Cubes, save   (src/sph.hpp:127-196)
// Written from the interface, not copied: bodies are ours.  Like the reference it is generic over
// the vector template V<L, T>; the reference instantiates it with glm::vec, this repo ships
// sph::vec (below) so that nothing outside the C++ standard library is needed.
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <limits>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

namespace sph {

// ---- a minimal vector type usable as the V template argument ---------------------------------
#define PBF_SPH_HAS_VEC 1  // (hipsph.hpp: sph::vec exists and is the default V; the reference's sph.hpp has no such type)
template <size_t L, typename T> struct vec;
template <typename T> struct vec<3, T> {
  T x{}, y{}, z{};
  constexpr vec() = default;
  template <typename A, typename B, typename C> constexpr vec(A a, B b, C c) : x(T(a)), y(T(b)), z(T(c)) {}
  template <typename U> constexpr explicit vec(const vec<3, U> &o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}
  constexpr vec &operator+=(const vec &o) {
    x += o.x, y += o.y, z += o.z;
    return *this;
  }
  constexpr bool operator==(const vec &o) const { return x == o.x && y == o.y && z == o.z; }
};
template <typename T> struct vec<4, T> {
  T x{}, y{}, z{}, w{};
  constexpr vec() = default;
  template <typename A, typename B, typename C, typename D>
  constexpr vec(A a, B b, C c, D d) : x(T(a)), y(T(b)), z(T(c)), w(T(d)) {}
  constexpr bool operator==(const vec &o) const { return x == o.x && y == o.y && z == o.z && w == o.w; }
};
template <typename T> constexpr vec<3, T> operator+(const vec<3, T> &a, const vec<3, T> &b) {
  return {a.x + b.x, a.y + b.y, a.z + b.z};
}
template <typename T> constexpr vec<3, T> operator-(const vec<3, T> &a, const vec<3, T> &b) {
  return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename T> constexpr vec<3, T> operator*(const vec<3, T> &a, T s) { return {a.x * s, a.y * s, a.z * s}; }

// ---- model types (reference src/sph.hpp:15-117) ------------------------------------------------
enum class Type : uint8_t { Fluid = 0, Obstacle = 1 };

template <typename T, typename N, template <size_t, typename C = N> typename V> struct Query {
  T id;
  V<3> point;
};

template <typename T, typename N, template <size_t, typename C = N> typename V> struct QueryResult {
  T id{};
  V<3> point{};
  std::vector<T> neighbours{};
};

template <typename T, typename N, template <size_t, typename C = N> typename V> struct Particle {
  T id;
  Type type;
  N mass;
  V<3> position, velocity;
  V<4> colour;
  constexpr Particle() : id{}, type(Type::Fluid), mass{} {}
  constexpr explicit Particle(T id, Type type, N mass, const V<4> &colour, const V<3> &position, const V<3> &velocity)
      : id(id), type(type), mass(mass), position(position), velocity(velocity), colour(colour) {}
  bool operator==(const Particle &o) const {
    return id == o.id && type == o.type && mass == o.mass && colour == o.colour && position == o.position &&
           velocity == o.velocity;
  }
  bool operator!=(const Particle &o) const { return !(*this == o); }  // the reference's recurses forever (sph.hpp:53)
};

template <typename T, typename N, template <size_t, typename C = N> typename V> struct Well {
  T tag;
  V<3> centre;
  N force;
};
template <typename T, typename N, template <size_t, typename C = N> typename V> struct Source {
  T tag;
  V<3> centre, velocity;
  V<4> colour;
  N rate;
};
template <typename T, typename N, template <size_t, typename C = N> typename V> struct Drain {
  T tag;
  V<3> centre;
  N width, depth;
};
template <typename T, typename N, template <size_t S, typename C = N> typename V> struct Scene {
  std::vector<Well<T, N, V>> wells;
  std::vector<Source<T, N, V>> sources;
  std::vector<Drain<T, N, V>> drains;
  std::vector<Query<T, N, V>> queries;
};

template <typename N> struct McParams {
  N resolution, isolevel, particleSize, particleInfluence;
  template <typename O> McParams<O> as() const {
    return {O(resolution), O(isolevel), O(particleSize), O(particleInfluence)};
  }
};

template <typename T, typename N, template <size_t S, typename C = N> typename V> struct SphParams {
  N h, dt, scale;  // h is dead in the reference too (sph.hpp:98): the solver's h is its ctor argument
  size_t iteration;
  V<3> constantForce, minBound, maxBound;
  bool wait;
  std::optional<McParams<N>> surface;
};

template <typename N, template <size_t S, typename C = N> typename V> struct ColouredMesh {
  std::vector<V<3>> vs{}, ns{};
  std::vector<V<4>> cs{};
  ColouredMesh() = default;
  explicit ColouredMesh(size_t size) : vs(size), ns(size), cs(size) {}
  ColouredMesh(const std::vector<V<3>> &vs, const std::vector<V<3>> &ns, const std::vector<V<4>> &cs)
      : vs(vs), ns(ns), cs(cs) {}
};

template <typename T, typename N, template <size_t, typename C = N> typename V> struct Result {
  ColouredMesh<N, V> mesh{};
  std::vector<QueryResult<T, N, V>> queries{};
};

template <typename T, typename N, template <size_t, typename _ = N> typename V> class Solver {
public:
  virtual ~Solver() = default;
  // Mutates xs in place: may grow (sources) / shrink (drains) and comes back in Z-order
  // (reference src/omp/ompsph.hpp:479-481).
  virtual Result<T, N, V> advance(const SphParams<T, N, V> &config, const Scene<T, N, V> &scene,
                                  std::vector<Particle<T, N, V>> &xs) = 0;
};

// ---- scene factory (reference src/sph.hpp:127-186) ---------------------------------------------
template <typename T, typename N, template <size_t, typename C = N> typename V>
T makeCube(T offset, N spacing, const size_t count, V<3> origin, V<4> colour, std::vector<Particle<T, N, V>> &xs) {
  const auto len = static_cast<size_t>(std::cbrt(count));
  for (size_t x = 0; x < len; ++x)
    for (size_t y = 0; y < len; ++y)
      for (size_t z = 0; z < len; ++z)
        xs.emplace_back(offset++, Type::Fluid, N(1.0), colour, (V<3>(x, y, z) * spacing) + origin, V<3>(0, 0, 0));
  return offset;
}

template <typename T, typename N, template <size_t, typename C = N> typename V>
SphParams<T, N, V> applyMotionSinXCosZ(const SphParams<T, N, V> &config, size_t frame) {
  const float scale = 300.f, rate = 20.f;  // offsets are evaluated in float like the reference
  const N ox = N(std::sin(float(frame) / rate) * scale);
  const N oz = N(std::cos(float(frame) / rate) * scale * 0.3);
  SphParams<T, N, V> moved = config;
  const V<3> off(ox, N{}, oz);
  moved.minBound += off;
  moved.maxBound += off;
  return moved;
}

template <typename T, typename N, template <size_t, typename C = N> typename V>
std::tuple<McParams<N>, SphParams<T, N, V>, std::vector<Particle<T, N, V>>>
simpleConfigWith2Cubes(size_t count, size_t solverIter, N scaling) {
  std::vector<Particle<T, N, V>> ps;
  T tag{};
  tag = makeCube<T, N, V>(tag, N(22.f), count / 2, V<3>(100, 0, 100), V<4>(0, 0.1, 0.8, 1), ps);
  tag = makeCube<T, N, V>(tag, N(22.f), count / 2, V<3>(600, 0, 600), V<4>(0.1, 0.8, 0.1, 1), ps);
  SphParams<T, N, V> config{};
  config.dt = N(0.0083 * 1.5f);
  config.scale = scaling;
  config.iteration = solverIter;
  config.constantForce = V<3>(0, 9.8, 0);
  config.minBound = V<3>(0, 0, 0);
  config.maxBound = V<3>(1000, 1000, 1000);
  config.wait = true;
  config.surface = {};
  return {McParams<N>{N(2.0f), N(100), N(25), N(0.5)}, config, ps};
}

// Dam-break column (ours): the reference has no scene for 256 K - 4 M particles.
// Returns {config, particles}; the box is [0, side]^3 with side chosen from the column size.
template <typename T, typename N, template <size_t, typename C = N> typename V>
std::tuple<SphParams<T, N, V>, std::vector<Particle<T, N, V>>> damBreakConfig(size_t nominal, size_t solverIter,
                                                                              N scaling) {
  size_t nx = static_cast<size_t>(std::cbrt(double(nominal / 2)));
  while ((nx + 1) * (nx + 1) * (nx + 1) <= nominal / 2) ++nx;
  while (nx * nx * nx > nominal / 2) --nx;
  const size_t ny = 2 * nx, nz = nx;
  const double side = 50.0 * std::ceil(2.5 * double(nx) * 22.0 / 50.0 + 4.0);
  std::vector<Particle<T, N, V>> ps;
  ps.reserve(nx * ny * nz);
  const V<3> origin(N(100), N(side - 100.0 - double(ny - 1) * 22.0), N(100));
  T tag{};
  for (size_t x = 0; x < nx; ++x)
    for (size_t y = 0; y < ny; ++y)
      for (size_t z = 0; z < nz; ++z)
        ps.emplace_back(tag++, Type::Fluid, N(1.0), V<4>(0, 0.1, 0.8, 1), (V<3>(x, y, z) * N(22.f)) + origin,
                        V<3>(0, 0, 0));
  auto [mc, config, unused] = simpleConfigWith2Cubes<T, N, V>(0, solverIter, scaling);
  (void)mc, (void)unused;
  config.maxBound = V<3>(side, side, side);
  return {config, ps};
}

// The reference's save() creates the directory and stops at "TODO impl" (src/sph.hpp:188-196) while
// its help text promises cloud.ply / mesh.obj (src/args.cpp:40).  This one writes both.
template <typename T, typename N, template <size_t, typename C = N> typename V>
void save(const Result<T, N, V> &result, const std::vector<Particle<T, N, V>> &particles, const std::string &dir) {
  if (dir.empty()) return;
  std::error_code ec;
  std::filesystem::create_directories(dir, ec);
  if (ec) {
    std::cerr << "Can't create directory `" << dir << "`: " << ec.message() << std::endl;
    return;
  }
  {
    std::ofstream ply(std::filesystem::path(dir) / "cloud.ply");
    ply << "ply\nformat ascii 1.0\nelement vertex " << particles.size()
        << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\n"
           "property uchar blue\nend_header\n";
    ply.precision(std::numeric_limits<N>::max_digits10);
    auto u8 = [](N c) { return int(std::min(N(255), std::max(N(0), c * N(255)))); };
    for (const auto &p : particles)
      ply << p.position.x << ' ' << p.position.y << ' ' << p.position.z << ' ' << u8(p.colour.x) << ' '
          << u8(p.colour.y) << ' ' << u8(p.colour.z) << '\n';
  }
  {
    std::ofstream obj(std::filesystem::path(dir) / "mesh.obj");
    for (const auto &v : result.mesh.vs) obj << "v " << v.x << ' ' << v.y << ' ' << v.z << '\n';
    for (const auto &n : result.mesh.ns) obj << "vn " << n.x << ' ' << n.y << ' ' << n.z << '\n';
    for (size_t t = 0; t + 2 < result.mesh.vs.size(); t += 3)
      obj << "f " << t + 1 << "//" << t + 1 << ' ' << t + 2 << "//" << t + 2 << ' ' << t + 3 << "//" << t + 3 << '\n';
  }
}

}  // namespace sph
