// san_driver.cpp — runs the CPU checker under the sanitizers (oracle/Makefile `san`; SURVEY.md §5: "TSan run of our
// CPU restatement in Jacobi mode must be clean"; the reference's own Debug flags are -fsanitize=address/undefined,
// CMakeLists.txt:134).  TEST INFRASTRUCTURE: built and run by tests/test_sanitizers_cpu.py, never shipped.
#include <cstdio>
#include <vector>

#include "pbf_oracle.h"

template <typename N> static double run(int fp64, int mode, int sortKind, int threads, int steps, bool surface) {
  double side = 0;
  const size_t n = pbf_oracle_scene_dambreak(fp64, 2048, nullptr, nullptr, nullptr, nullptr, nullptr, &side);
  std::vector<uint64_t> id(n);
  std::vector<uint8_t> type(n, 0);
  std::vector<N> mass(n), pos(3 * n), vel(3 * n), col(4 * n);
  pbf_oracle_scene_dambreak(fp64, 2048, id.data(), mass.data(), pos.data(), vel.data(), col.data(), &side);
  type[5] = 1;  // one obstacle: exercises the special-particle branches
  pbf_oracle *o = pbf_oracle_create(fp64);
  pbf_oracle_set_particles(o, n, id.data(), type.data(), mass.data(), pos.data(), vel.data(), col.data());
  pbf_oracle_params p{};
  p.h = 0.1, p.dt = 0.0083 * 1.5, p.scale = 500.0, p.iteration = 2;
  p.constant_force[1] = 9.8;
  for (int k = 0; k < 3; ++k) p.max_bound[k] = side;
  p.mode = mode, p.sort = sortKind, p.threads = threads;
  const double wells[4] = {300.0, 100.0, 300.0, 500.0};
  p.n_wells = 1, p.wells = wells;
  p.xsph = 1, p.vorticity = 1;
  for (int s = 0; s < steps; ++s) pbf_oracle_step(o, &p);
  uint64_t tris = 0;
  if (surface) {
    pbf_oracle_mc mc{2.0, 100.0, 25.0, 0.5};
    pbf_oracle_surface(o, &p, &mc, &tris);
  }
  pbf_oracle_get_particles(o, id.data(), type.data(), mass.data(), pos.data(), vel.data(), col.data());
  double sum = double(tris);
  for (N v : pos) sum += double(v);
  pbf_oracle_destroy(o);
  return sum;
}

int main() {
  const double a = run<float>(0, PBF_ORACLE_JACOBI, PBF_ORACLE_SORT_STABLE, 4, 3, true);
  const double b = run<double>(1, PBF_ORACLE_JACOBI, PBF_ORACLE_SORT_STABLE, 4, 2, false);
  const double c = run<float>(0, PBF_ORACLE_GS, PBF_ORACLE_SORT_STD, 1, 2, false);
  std::printf("sanitizer run ok: %.6f %.6f %.6f\n", a, b, c);
  return 0;
}
