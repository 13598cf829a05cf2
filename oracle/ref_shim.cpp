// ref_shim.cpp — builds oracle/_ref/ref_shim_check: the product's C++ shim (pbf-sph_amd/host/hipsph.hpp) compiled
// against the REFERENCE'S OWN src/sph.hpp, where it lies, exactly as INTEGRATION.md tells a maintainer to do it
// ("copy hipsph.hpp + pbf_hip.h next to sph.hpp, add a case to the backend switch").
//
// TEST INFRASTRUCTURE ONLY.  No reference source is in this file: it #includes /root/reference/src/sph.hpp by path
// (-I, glm-free: the header is generic over the vector template V) and a COPY OF OUR OWN hipsph.hpp placed in a
// directory without a sph.hpp of ours, so that its `#include "sph.hpp"` resolves to the reference's.  `Vec` is the
// template argument any caller of the reference's API supplies (the reference uses glm::vec, src/omp/ompsph.hpp:33).
//
// What it proves: sph::hip_impl::Solver<size_t, float|double, V> derives from the reference's abstract
// sph::Solver<T, N, V> (src/sph.hpp:119-125), can be driven through a base-class pointer with the reference's own
// SphParams / Scene / Particle / Result types, scene factory (simpleConfigWith2Cubes, src/sph.hpp:160-186) and box
// motion (applyMotionSinXCosZ, :147-158), the way runN does (src/benchmark.cpp:22-58) — and, on a GPU, produces the
// same bits as the product's own host stack (tests/test_cli_gpu.py compares the dump with the C-ABI path).
//
//   ref_shim_check <fp64:0|1> <frames> <count> <surface:0|1> [dump.bin]
// exit 0 ok; 3 = the solver could not be constructed (no gfx950 device: the message is printed — there is no CPU fallback).
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <tuple>

template <size_t L, typename C> struct Vec;
template <typename C> struct Vec<3, C> {
  C x{}, y{}, z{};
  Vec() = default;
  template <typename A, typename B, typename D> Vec(A a, B b, D d) : x(C(a)), y(C(b)), z(C(d)) {}
  Vec &operator+=(const Vec &o) {
    x += o.x, y += o.y, z += o.z;
    return *this;
  }
  bool operator==(const Vec &o) const { return x == o.x && y == o.y && z == o.z; }
};
template <typename C> struct Vec<4, C> {
  C x{}, y{}, z{}, w{};
  Vec() = default;
  template <typename A, typename B, typename D, typename E> Vec(A a, B b, D d, E e) : x(C(a)), y(C(b)), z(C(d)), w(C(e)) {}
  bool operator==(const Vec &o) const { return x == o.x && y == o.y && z == o.z && w == o.w; }
};
template <typename C> Vec<3, C> operator*(const Vec<3, C> &a, C s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename C> Vec<3, C> operator+(const Vec<3, C> &a, const Vec<3, C> &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename C> Vec<3, C> operator-(const Vec<3, C> &a, const Vec<3, C> &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

#include "sph.hpp"     // the REFERENCE's: -I/root/reference/src
#include "hipsph.hpp"  // ours, from a directory that holds no sph.hpp (its own #include "sph.hpp" is the reference's)

#ifdef PBF_SPH_HAS_VEC
#error "this translation unit must see the reference's sph.hpp, not pbf-sph_amd/host/sph.hpp"
#endif

template <typename N> int run(int frames, size_t count, bool surface, const char *dump) {
  using T = size_t;
  auto [mc, config, particles] = sph::simpleConfigWith2Cubes<T, N, Vec>(count, 4, N(500));
  if (surface) config.surface = mc;  // benchmark.cpp:29
  std::unique_ptr<sph::Solver<T, N, Vec>> solver;  // the reference's abstract interface
  try {
    solver = std::make_unique<sph::hip_impl::Solver<T, N, Vec>>(N(0.1));
  } catch (const std::exception &e) {
    std::printf("construct failed: %s\n", e.what());
    return 3;
  }
  size_t vertices = 0;
  for (int frame = 0; frame < frames; ++frame) {
    const auto result = solver->advance(sph::applyMotionSinXCosZ(config, size_t(frame)), sph::Scene<T, N, Vec>{}, particles);
    vertices = result.mesh.vs.size();
  }
  std::printf("ok n=%zu vertices=%zu sizeof(Particle)=%zu\n", particles.size(), vertices, sizeof(particles[0]));
  if (dump) {
    FILE *f = std::fopen(dump, "wb");
    if (!f) return 2;
    const uint64_t n = particles.size(), v = vertices;
    std::fwrite(&n, 8, 1, f), std::fwrite(&v, 8, 1, f);
    for (const auto &p : particles) {
      const uint64_t id = p.id;
      std::fwrite(&id, 8, 1, f), std::fwrite(&p.position, sizeof(N), 3, f), std::fwrite(&p.velocity, sizeof(N), 3, f);
      std::fwrite(&p.colour, sizeof(N), 4, f);
    }
    std::fclose(f);
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    std::printf("usage: ref_shim_check <fp64> <frames> <count> <surface> [dump.bin]\n");
    return 1;
  }
  const bool fp64 = std::atoi(argv[1]) != 0, surface = std::atoi(argv[4]) != 0;
  const int frames = std::atoi(argv[2]);
  const size_t count = size_t(std::atol(argv[3]));
  const char *dump = argc > 5 ? argv[5] : nullptr;
  return fp64 ? run<double>(frames, count, surface, dump) : run<float>(frames, count, surface, dump);
}
