// benchmark.cpp — drop-in benchmark driver for the MI355X backend.
//
// Observable behaviour follows the reference's src/benchmark.cpp: `warmup` untimed advance() calls,
// then `iter` timed ones, each with applyMotionSinXCosZ(param, frame) and an empty Scene
// (benchmark.cpp:22-58); then the summary block of benchmark.cpp:91-101 and "Results flushed.".
// Stock defaults: 20000 nominal particles (2 x 21^3 = 18522), 6 solver iterations, scale 500, h = 0.1
// (benchmark.cpp:23-25,160-163), marching-cubes surface on (benchmark.cpp:29; --no-surface turns it off).
// Differences, all visible in the output: the only backend is `hip`; extra lines report particle-steps/s;
// --resident times the device-resident loop; the surface's case tables are our own, so the vertex
// count need not equal the reference's triangle for triangle.
#include <hip/hip_runtime_api.h>
#include <malloc.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <iomanip>
#include <numeric>

#include "args.hpp"
#include "hipsph.hpp"

using duration_millis = std::chrono::duration<double, std::milli>;
using hrc = std::chrono::high_resolution_clock;

namespace {

struct Stats {
  double min, max, mean, stdDev;
};
Stats summaryStats(const std::vector<double> &xs) {
  const double sum = std::accumulate(xs.begin(), xs.end(), 0.0);
  const double mean = sum / double(xs.size());
  double var = 0;
  for (double x : xs) var += (x - mean) * (x - mean);
  var /= double(xs.size());
  const auto [mn, mx] = std::minmax_element(xs.begin(), xs.end());
  return {*mn, *mx, mean, std::sqrt(var)};
}

std::vector<std::pair<int, std::string>> listDevices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  std::vector<std::pair<int, std::string>> out;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess) out.emplace_back(i, std::string(p.name) + " (" + p.gcnArchName + ")");
  }
  return out;
}

// devices matching any needle: an index or a substring of the name (src/utils.hpp:87-105,128-159).  The reference
// takes the FIRST match; --all-devices takes every match (one x-slab each), --slabs K repeats the first K times.
std::vector<int> findDevices(const sph::driver::Args &args) {
  const auto devices = listDevices();
  std::vector<int> out;
  if (args.list) {
    for (const auto &[i, name] : devices) std::cout << "[" << i << "] " << name << std::endl;
    return out;
  }
  for (const auto &[i, name] : devices)
    for (const auto &needle : args.devices) {
      bool match = false;
      try {
        size_t pos = 0;
        const int idx = std::stoi(needle, &pos);
        match = pos == needle.size() ? idx == i : name.find(needle) != std::string::npos;
      } catch (...) {
        match = name.find(needle) != std::string::npos;
      }
      if (match && (out.empty() || (args.allDevices && out.back() != i))) {
        std::cout << "Using device: " << name << std::endl;
        out.push_back(i);
      }
    }
  if (out.empty()) std::cerr << "No device matches the --devices list" << std::endl;
  if (!out.empty() && args.slabs > 1) out.assign(args.slabs, out.front());
  return out;
}

template <typename N> int run(sph::driver::Args args, const std::vector<int> &devices) {
  using T = size_t;
  using Particle = sph::Particle<T, N, sph::vec>;
  const auto output = args.renderedOutputName();
  std::cout << "Using " << output << " for output" << std::endl;

  sph::SphParams<T, N, sph::vec> param;
  std::vector<Particle> particles;
  sph::McParams<N> mc{};
  const bool moving = args.scene == "cubes";
  if (moving) {
    std::tie(mc, param, particles) = sph::simpleConfigWith2Cubes<T, N, sph::vec>(args.particles, args.solverIter, N(500));
  } else {
    std::tie(param, particles) = sph::damBreakConfig<T, N, sph::vec>(args.particles, args.solverIter, N(500));
  }
  // the stock driver runs with the surface on: param.surface = initialMcParam (benchmark.cpp:29)
  if (!moving) mc = sph::McParams<N>{N(2.0f), N(100), N(25), N(0.5)};  // the same McParams for the dam-break scene
  if (args.surface) param.surface = mc;

  uint32_t flags = (args.fastMath ? PBF_FLAG_FAST_MATH : 0u) | (args.verbose ? PBF_FLAG_STAGE_TIMING : 0u);
  const bool slabbed = devices.size() > 1;
  if (slabbed) {  // x-slabs: device-resident stepping (the surface: every slab its own node planes, concatenated)
    if (!args.resident) std::cout << "Slab mode (" << devices.size() << " slabs): --resident implied" << std::endl;
    else std::cout << "Slab mode (" << devices.size() << " slabs)" << std::endl;
    args.resident = true;
  }
  sph::hip_impl::Solver<T, N> solver(N(0.1), devices, flags);
  sph::Result<T, N, sph::vec> result;
  auto frameParam = [&](size_t frame) { return moving ? sph::applyMotionSinXCosZ(param, frame) : param; };

  std::vector<double> frameTime;
  hrc::time_point start, end;
  if (args.resident) solver.upload(particles, &param);
  auto one = [&](size_t frame) {
    if (args.resident) {
      solver.step(frameParam(frame));
      if (param.surface) result.mesh = solver.surface(frameParam(frame));
      solver.sync();  // per-frame time like the reference's blocking advance()
    } else {
      result = solver.advance(frameParam(frame), {}, particles);
    }
  };
  for (size_t frame = 0; frame < args.warmup; ++frame) {
    try {
      one(frame);
    } catch (std::exception const &e) {
      std::cout << "Caught asynchronous exception at warmup frame" << frame << ":\n" << e.what() << "\n";
      throw;
    }
  }
  start = hrc::now();
  for (size_t frame = 0; frame < args.iterations; ++frame) {
    const auto f0 = hrc::now();
    try {
      one(frame);
    } catch (std::exception const &e) {
      std::cout << "Caught asynchronous exception at benchmark frame" << frame << ":\n" << e.what() << "\n";
      throw;
    }
    frameTime.push_back(duration_millis(hrc::now() - f0).count());
  }
  end = hrc::now();
  if (args.resident) solver.download(particles);

  const double seconds = duration_millis(end - start).count() / 1000.0;
  const size_t frames = args.iterations;
  const Stats st = frameTime.empty() ? Stats{0, 0, 0, 0} : summaryStats(frameTime);
  std::cout << "Benchmark completed after " << frames << " frames:\n"
            << std::setprecision(4)  //
            << "Runtime              : " << seconds << " s\n"
            << "Framerate            : " << double(frames) / seconds << " fps\n"
            << "Frame-time min       : " << st.min << " ms\n"
            << "Frame-time max       : " << st.max << " ms\n"
            << "Frame-time mean       : " << st.mean << " ms\n"
            << "Frame-time stdDev     : " << st.stdDev << " ms\n"
            << "Final Vertex count   : " << result.mesh.vs.size() << "\n"
            << "Final Particle count : " << particles.size() << " \n"
            << std::endl;
  const double psps = double(particles.size()) * double(frames) / seconds;
  std::cout << std::setprecision(6) << "Particle-steps/s     : " << psps << " (" << (args.resident ? "device-resident" : "advance(): upload+step+download per frame")
            << ", K=" << args.solverIter << ", " << (args.fp64 ? "fp64" : "fp32") << ")\n";
  if (args.verbose) {
    const char *names[16];
    double ms[16];
    uint64_t calls[16];
    const int k = pbf_stage_times(solver.context(), names, ms, calls, 16);
    std::cout << "Stopwatch[ advance]:\n";
    for (int i = 0; i < k; ++i)
      std::cout << "    ->`" << names[i] << "` : " << ms[i] << "ms x " << double(calls[i]) / double(args.warmup + frames) << "/frame\n";
  }
  if (args.json)
    std::cout << "{\"impl\":\"hip\",\"scene\":\"" << args.scene << "\",\"particles\":" << particles.size()
              << ",\"solver_iter\":" << args.solverIter << ",\"fp64\":" << (args.fp64 ? "true" : "false")
              << ",\"resident\":" << (args.resident ? "true" : "false") << ",\"slabs\":" << devices.size() << ",\"frames\":" << frames
              << ",\"seconds\":" << seconds << ",\"particle_steps_per_s\":" << psps << ",\"frame_ms_mean\":" << st.mean
              << "}" << std::endl;
  sph::save(result, particles, output);
  std::cout << "Results flushed." << std::endl;
  return 0;
}

}  // namespace

int main(int argc, char *argv[]) {
  // advance() returns the mesh by value (src/sph.hpp:114-125): three freshly allocated vectors per frame — 54 MB at 1 M
  // particles — which `result = solver.advance(...)` (benchmark.cpp:33,47) frees again one frame later.  Left alone, glibc
  // hands the freed top of the heap back to the system every frame (the trim threshold follows the mmap threshold) and
  // the next frame's vectors fault every page in again: ~2 ms per frame of page faults.  Keep the memory instead.
  mallopt(M_MMAP_THRESHOLD, 32 << 20);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 128 << 20);
  sph::driver::Args args(200, "./out_{impl}_{type}_{iter}");
  if (!args.parse(argc, argv)) return EXIT_SUCCESS;  // the reference exits 0 after help / parse errors too
  if (args.impl != "hip") {
    std::cerr << "Implementation `" << args.impl << "` is not part of this build: this driver ships the `hip` backend only "
              << "(omp/ocl/sycl live in the reference)" << std::endl;
    return EXIT_FAILURE;
  }
  const std::vector<int> devices = findDevices(args);
  if (devices.empty()) return args.list ? EXIT_SUCCESS : EXIT_FAILURE;
  try {
    return args.fp64 ? run<double>(args, devices) : run<float>(args, devices);
  } catch (const std::exception &e) {
    std::cerr << "benchmark failed: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
}
