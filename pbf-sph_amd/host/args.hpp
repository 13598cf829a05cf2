// args.hpp — command line of the benchmark driver.  Same flags, defaults and {iter}/{impl}/{type}
// output templates as the reference's sph::driver::Args (src/args.hpp:38-56, src/args.cpp:7-75),
// parsed by a small hand-written parser instead of the vendored Taywee/args, plus additive flags
// (marked +) that the stock CLI cannot express: its particle count, solver iterations and scene are
// compiled in (src/benchmark.cpp:23-25).
#pragma once

#include <cstddef>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace sph::driver {

struct Args {
  std::string impl = "hip";               // -i/--impl   (reference default "omp": not part of this product)
  bool list = false;                      // -l/--list
  bool verbose = false;                   // -v/--verbose
  std::vector<std::string> devices;       // -d/--devices (index or name substring, first match)
  size_t iterations;                      // -n/--iter
  size_t warmup = 200;                    // -w/--warmup
  bool fp64 = false;                      // --fp64
  std::string output;                     // -o/--output
  // + additive
  size_t particles = 20 * 1000;           // --particles   (stock: benchmark.cpp:23)
  size_t solverIter = 6;                  // --solver-iter (stock: benchmark.cpp:24)
  std::string scene = "cubes";            // --scene cubes|dam-break
  bool resident = false;                  // --resident: keep particles on the GPU between frames
  bool surface = true;                    // --no-surface: skip marching cubes (stock: on, benchmark.cpp:29)
  bool fastMath = false;                  // --fast-math
  bool json = false;                      // --json: one machine-readable line after the summary
  bool allDevices = false;                // --all-devices: EVERY device matching -d becomes one x-slab (RCCL halo)
  size_t slabs = 0;                       // --slabs K: K slabs on the first matching device (in-process exchange: tests)

  Args(size_t defaultIterations, std::string defaultOutput)
      : iterations(defaultIterations), output(std::move(defaultOutput)) {}

  static void usage(std::ostream &os) {
    os << "  benchmark {OPTIONS}\n\n    PBF sph benchmark\n\n  OPTIONS:\n\n"
          "      -h, --help                        Display this help menu\n"
          "      -i[impl], --impl=[impl]           Which implementation to use.\n"
          "                                        One of: hip  (omp, ocl, sycl, sycl2020 live in the reference)\n"
          "                                        Default: hip\n"
          "      -l, --list                        List devices available for [impl] and exit\n"
          "      -v, --verbose                     Show details such as per-stage timings for [impl]\n"
          "      -d[dev...], --devices=[dev...]    Allowed device list (first match only).\n"
          "                                        Entries could be a 0-based index or a substring of the device name\n"
          "                                        Default: 0\n"
          "      -n[iter], --iter=[iter]           How many iterations to run the simulation for.\n"
          "                                        Default: 200\n"
          "      -w[warmup], --warmup=[warmup]     How many iterations to skip for warmup before timing starts.\n"
          "                                        Default: 200\n"
          "      --fp64                            Use FP64 (double) instead of FP32 (float).\n"
          "      -o[out], --output=[out]           Directory to write the final state (cloud.ply, mesh.obj) to.\n"
          "                                        Templates: {iter}, {impl}, {type}; empty string disables output\n"
          "                                        Default: ./out_{impl}_{type}_{iter}\n"
          "    additive (not in the reference CLI):\n"
          "      --particles=[n]                   Nominal particle count. Default: 20000 (stock)\n"
          "      --solver-iter=[k]                 Solver iterations per frame. Default: 6 (stock)\n"
          "      --scene=[cubes|dam-break]         cubes = stock two cubes in a moving box; dam-break = static box\n"
          "      --resident                        Time the device-resident loop (no per-frame host round trip)\n"
          "      --no-surface                      Skip the marching-cubes surface (the stock driver runs with it on)\n"
          "      --fast-math                       v_rsq / fma pair kernels (the reference builds with -Ofast)\n"
          "      --json                            Print one JSON line with the results\n"
          "      --all-devices                     Use EVERY device matching -d: one x-slab per GPU, ghost-layer\n"
          "                                        exchange over RCCL (implies --resident --no-surface)\n"
          "      --slabs=[K]                       K slabs on the first matching device (in-process exchange; tests)\n";
  }

  // returns false if the program should exit (help / parse error), like the reference's parse()
  bool parse(int argc, char *argv[]) {
    auto value = [&](int &i, const std::string &arg, const std::string &shortF, const std::string &longF,
                     std::string &out) -> bool {
      if (arg == shortF || arg == longF) {
        if (i + 1 >= argc) throw std::runtime_error("Flag '" + arg + "' requires an argument");
        out = argv[++i];
        return true;
      }
      if (!shortF.empty() && arg.rfind(shortF, 0) == 0 && arg.size() > shortF.size() && arg[1] != '-') {
        out = arg.substr(shortF.size());
        return true;
      }
      if (arg.rfind(longF + "=", 0) == 0) {
        out = arg.substr(longF.size() + 1);
        return true;
      }
      return false;
    };
    try {
      for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        std::string v;
        if (a == "-h" || a == "--help") {
          usage(std::cout);
          return false;
        } else if (a == "-l" || a == "--list") list = true;
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "--fp64") fp64 = true;
        else if (a == "--resident") resident = true;
        else if (a == "--no-surface") surface = false;
        else if (a == "--fast-math") fastMath = true;
        else if (a == "--json") json = true;
        else if (a == "--all-devices") allDevices = true;
        else if (value(i, a, "", "--slabs", v)) slabs = std::stoull(v);
        else if (value(i, a, "-i", "--impl", v)) impl = v;
        else if (value(i, a, "-d", "--devices", v)) devices.push_back(v);
        else if (value(i, a, "-n", "--iter", v)) iterations = std::stoull(v);
        else if (value(i, a, "-w", "--warmup", v)) warmup = std::stoull(v);
        else if (value(i, a, "-o", "--output", v)) output = v;
        else if (value(i, a, "", "--particles", v)) particles = std::stoull(v);
        else if (value(i, a, "", "--solver-iter", v)) solverIter = std::stoull(v);
        else if (value(i, a, "", "--scene", v)) scene = v;
        else throw std::runtime_error("Flag could not be matched: " + a);
      }
      if (scene != "cubes" && scene != "dam-break") throw std::runtime_error("Unknown scene: " + scene);
    } catch (const std::exception &e) {
      std::cerr << e.what() << std::endl;
      usage(std::cerr);
      return false;
    }
    if (devices.empty()) devices.push_back("0");
    return true;
  }

  std::string renderedOutputName() const {  // src/args.cpp:69-75
    auto replace = [](std::string s, const std::string &from, const std::string &to) {
      for (size_t pos = 0; (pos = s.find(from, pos)) != std::string::npos; pos += to.size()) s.replace(pos, from.size(), to);
      return s;
    };
    std::string name = output;
    name = replace(name, "{iter}", std::to_string(iterations));
    name = replace(name, "{type}", fp64 ? "fp64" : "fp32");
    name = replace(name, "{impl}", impl);
    return name;
  }
};

}  // namespace sph::driver
