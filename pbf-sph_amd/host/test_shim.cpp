// test_shim.cpp — exercises the host side of sph::hip_impl::Solver::advance() that the benchmark
// driver never touches: sources, drains, queries, depletion, and advance() == resident stepping
// (reference behaviour: src/omp/ompsph.hpp:91-126,167-186).  Prints "ok <name>" / "FAIL <name> ..."
// lines; tests/test_cli_gpu.py runs it on a GPU.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <set>
#include <string>

#include "hipsph.hpp"

using T = size_t;
using N = float;
using P = sph::Particle<T, N, sph::vec>;
using V3 = sph::vec<3, N>;
using V4 = sph::vec<4, N>;

static int failures = 0;
#define CHECK(name, cond, ...)                  \
  do {                                          \
    if (cond) std::printf("ok %s\n", name);     \
    else {                                      \
      ++failures;                               \
      std::printf("FAIL %s ", name);            \
      std::printf(__VA_ARGS__);                 \
      std::printf("\n");                        \
    }                                           \
  } while (0)

// --dump <dir>: one fixed scene with sources, a drain, an obstacle and queries run through advance() for two frames;
// the particles and the query answers after each frame go to <dir> as raw arrays.  tests/test_cli_gpu.py replays the
// same scene through the ORACLE's restatement of ompsph.hpp:91-126,167-186 and expects identical ids, order and bits.
static int dump_scene(const char *dir) {
  auto [mc, config, particles] = sph::simpleConfigWith2Cubes<T, N, sph::vec>(2048, 4, N(500));
  (void)mc;
  sph::hip_impl::Solver<T, N> solver(N(0.1));
  auto xs = particles;
  xs[7].type = sph::Type::Obstacle;
  sph::Scene<T, N, sph::vec> scene;
  scene.sources.push_back({T(100777), V3(500, 300, 500), V3(0, 1, 0), V4(1, 0, 0, 1), N(16)});
  scene.sources.push_back({T(100888), V3(200, 700, 800), V3(3, 0, -2), V4(0, 1, 0, 1), N(10)});
  scene.drains.push_back({T(1), xs[7].position, N(60), N(0)});
  scene.queries.push_back({T(5), xs[100].position});
  scene.queries.push_back({T(6), V3(990, 990, 990)});
  scene.queries.push_back({T(7), V3(510, 310, 510)});
  for (int frame = 0; frame < 2; ++frame) {
    const auto res = solver.advance(config, scene, xs);
    const std::string base = std::string(dir) + "/frame" + std::to_string(frame);
    FILE *f = std::fopen((base + "_particles.bin").c_str(), "wb");
    if (!f) return 2;
    const uint64_t n = xs.size();
    std::fwrite(&n, 8, 1, f);
    for (const auto &p : xs) {
      const uint64_t id = p.id;
      const uint8_t ty = uint8_t(p.type);
      std::fwrite(&id, 8, 1, f), std::fwrite(&ty, 1, 1, f), std::fwrite(&p.position, sizeof(N), 3, f);
      std::fwrite(&p.velocity, sizeof(N), 3, f), std::fwrite(&p.colour, sizeof(N), 4, f);
    }
    std::fclose(f);
    f = std::fopen((base + "_queries.bin").c_str(), "wb");
    for (const auto &q : res.queries) {
      const uint64_t qid = q.id, cnt = q.neighbours.size();
      std::fwrite(&qid, 8, 1, f), std::fwrite(&cnt, 8, 1, f);
      for (T v : q.neighbours) {
        const uint64_t id = v;
        std::fwrite(&id, 8, 1, f);
      }
    }
    std::fclose(f);
  }
  std::printf("dumped\n");
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 3 && std::string(argv[1]) == "--dump") return dump_scene(argv[2]);
  auto [mc, config, particles] = sph::simpleConfigWith2Cubes<T, N, sph::vec>(2048, 4, N(500));
  (void)mc;
  sph::hip_impl::Solver<T, N> solver(N(0.1));

  {  // sources: floor(sqrt(rate)) x ceil(sqrt(rate)) particles per frame, spacing h*scale/2 (ompsph.hpp:93-105)
    auto xs = particles;
    sph::Scene<T, N, sph::vec> scene;
    scene.sources.push_back({T(100777), V3(500, 300, 500), V3(0, 1, 0), V4(1, 0, 0, 1), N(16)});
    scene.sources.push_back({T(100888), V3(200, 700, 800), V3(0, 0, 0), V4(0, 1, 0, 1), N(10)});
    const size_t before = xs.size();
    solver.advance(config, scene, xs);
    size_t n777 = 0, n888 = 0;
    for (const auto &p : xs) n777 += p.id == 100777, n888 += p.id == 100888;
    CHECK("sources_count", xs.size() == before + 16 + 12 && n777 == 16 && n888 == 12, "size %zu n777 %zu n888 %zu",
          xs.size(), n777, n888);
    solver.advance(config, scene, xs);
    CHECK("sources_accumulate", xs.size() == before + 2 * 28, "size %zu", xs.size());
  }
  {  // drains: fluid within `width` of the centre disappears, obstacles stay (ompsph.hpp:107-118)
    auto xs = particles;
    xs[0].type = sph::Type::Obstacle;
    sph::Scene<T, N, sph::vec> scene;
    scene.drains.push_back({T(1), xs[0].position, N(60), N(0)});
    size_t expectGone = 0;
    for (size_t i = 1; i < xs.size(); ++i) {
      const V3 r = xs[i].position - xs[0].position;
      expectGone += std::sqrt(r.x * r.x + r.y * r.y + r.z * r.z) < 60;
    }
    const size_t before = xs.size();
    solver.advance(config, scene, xs);
    bool obstacleKept = false;
    for (const auto &p : xs) obstacleKept |= p.id == particles[0].id && p.type == sph::Type::Obstacle;
    CHECK("drains", expectGone > 3 && xs.size() == before - expectGone && obstacleKept, "gone %zu size %zu->%zu", expectGone,
          before, xs.size());
  }
  {  // queries: ids of the fluid particles in the query point's cell (ompsph.hpp:167-186); particles at rest
    auto xs = particles;
    auto still = config;
    still.constantForce = V3(0, 0, 0);
    still.iteration = 0;
    sph::Scene<T, N, sph::vec> scene;
    scene.queries.push_back({T(5), xs[100].position});
    scene.queries.push_back({T(6), V3(990, 990, 990)});  // empty corner
    const V3 q0 = xs[100].position;
    const T id100 = xs[100].id;
    auto cell = [&](const V3 &p, int ax) {
      const N v = ax == 0 ? p.x : ax == 1 ? p.y : p.z;
      return long((v / still.scale - (N(0) / still.scale - N(0.2))) / N(0.1));
    };
    std::set<T> expect;
    for (const auto &p : xs)
      if (cell(p.position, 0) == cell(q0, 0) && cell(p.position, 1) == cell(q0, 1) && cell(p.position, 2) == cell(q0, 2))
        expect.insert(p.id);
    const auto res = solver.advance(still, scene, xs);
    std::set<T> got(res.queries[0].neighbours.begin(), res.queries[0].neighbours.end());
    CHECK("queries", res.queries.size() == 2 && res.queries[0].id == 5 && got == expect && got.count(id100) == 1 &&
                         res.queries[1].neighbours.empty(),
          "got %zu expect %zu", got.size(), expect.size());
  }
  {  // depletion (ompsph.hpp:122-126)
    std::vector<P> none;
    const auto res = solver.advance(config, {}, none);
    CHECK("depleted", none.empty() && res.mesh.vs.empty() && res.queries.empty(), "-");
  }
  {  // advance() per frame == device-resident stepping, bit for bit
    auto a = particles, b = particles;
    for (int f = 0; f < 3; ++f) solver.advance(sph::applyMotionSinXCosZ(config, f), {}, a);
    sph::hip_impl::Solver<T, N> resident(N(0.1));
    resident.upload(b);
    for (int f = 0; f < 3; ++f) resident.step(sph::applyMotionSinXCosZ(config, f));
    resident.download(b);
    bool same = a.size() == b.size();
    for (size_t i = 0; same && i < a.size(); ++i) same = a[i] == b[i];
    CHECK("advance_equals_resident", same, "sizes %zu %zu", a.size(), b.size());
  }
  {  // Solver(h, {devices...}): three x-slabs (sharing this GPU: in-process exchange behind the library's host-callback
     // transport; distinct GPUs take RCCL) == ONE solver up to summation order; nothing lost; cuts re-balanced
    auto [dparam, dparticles] = sph::damBreakConfig<T, N, sph::vec>(8192, 2, N(500));
    auto a = dparticles, b = dparticles;
    sph::hip_impl::Solver<T, N> slabs(N(0.1), std::vector<int>{0, 0, 0});
    slabs.setRebalanceEvery(2);
    slabs.upload(a, &dparam);
    const auto cuts0 = slabs.cuts();
    slabs.step(dparam, {}, 12);
    slabs.download(a);
    sph::hip_impl::Solver<T, N> one(N(0.1));
    one.upload(b);
    one.step(dparam, {}, 12);
    one.download(b);
    auto byId = [](const P &x, const P &y) { return x.id < y.id; };
    std::sort(a.begin(), a.end(), byId);
    std::sort(b.begin(), b.end(), byId);
    bool ids = a.size() == b.size();
    double worst = 0, sum = 0;
    for (size_t i = 0; ids && i < a.size(); ++i) {
      ids = a[i].id == b[i].id;
      const V3 d = a[i].position - b[i].position;
      const double e = std::sqrt(double(d.x) * d.x + double(d.y) * d.y + double(d.z) * d.z);
      worst = std::max(worst, e), sum += e;
    }
    CHECK("multi_device_slabs", ids && slabs.deviceCount() == 3 && sum / double(a.size()) <= 0.05 && slabs.cuts() != cuts0,
          "sizes %zu %zu worst %g mean %g", a.size(), b.size(), worst, sum / double(a.size()));
    // advance() on several devices keeps the reference's contract too (upload -> step -> download)
    auto c = dparticles;
    sph::hip_impl::Solver<T, N> two(N(0.1), std::vector<int>{0, 0});
    two.advance(dparam, {}, c);
    CHECK("multi_device_advance", c.size() == dparticles.size(), "size %zu", c.size());
  }
  {  // the surface across slabs (round 3): every slab extracts the cubes of its own node planes — ghost layer for the
     // 27-cell neighbourhoods, the owners' diffused colours refreshed on the copies, one node plane handed to the left —
     // and the parts concatenated in slab order are the single-device mesh.  K = 0 (no delta-p: positions cannot differ by a
     // summation order): vertices and normals bit for bit, colours to rounding (a cell's particles may sit in another
     // order on a slab); K = 4: the same surface up to the solver's summation noise.
    auto [dparam, dparticles] = sph::damBreakConfig<T, N, sph::vec>(8192, 0, N(500));
    dparam.surface = sph::McParams<N>{N(2.0f), N(100), N(25), N(0.5)};
    auto colour = dparticles;
    for (size_t i = 0; i < colour.size(); ++i) colour[i].colour = V4(N(i % 7) / 7, N(i % 5) / 5, N(i % 3) / 3, 1);
    for (int iteration : {0, 4}) {
      dparam.iteration = size_t(iteration);
      auto a = colour, b = colour;
      sph::hip_impl::Solver<T, N> slabs(N(0.1), std::vector<int>{0, 0, 0});
      sph::hip_impl::Solver<T, N> one(N(0.1));
      sph::Result<T, N, sph::vec> ra, rb;
      for (int f = 0; f < 3; ++f) ra = slabs.advance(dparam, {}, a), rb = one.advance(dparam, {}, b);
      const auto &ma = ra.mesh, &mb = rb.mesh;
      if (iteration == 0) {
        bool same = ma.vs.size() == mb.vs.size() && ma.vs.size() > 3000 && ma.ns.size() == mb.ns.size() && ma.cs.size() == mb.cs.size();
        double worstC = 0;
        size_t badN = 0;
        for (size_t i = 0; same && i < ma.vs.size(); ++i) {
          same = ma.vs[i] == mb.vs[i];
          // (normals: NaN == NaN counts as equal — nodes without a neighbour carry 0 / 0)
          auto eq = [](N x, N y) { return x == y || (x != x && y != y); };
          badN += !(eq(ma.ns[i].x, mb.ns[i].x) && eq(ma.ns[i].y, mb.ns[i].y) && eq(ma.ns[i].z, mb.ns[i].z));
          worstC = std::max({worstC, double(std::fabs(ma.cs[i].x - mb.cs[i].x)), double(std::fabs(ma.cs[i].y - mb.cs[i].y)),
                             double(std::fabs(ma.cs[i].z - mb.cs[i].z)), double(std::fabs(ma.cs[i].w - mb.cs[i].w))});
        }
        CHECK("multi_device_surface_exact", same && badN <= ma.vs.size() / 1000 && worstC <= 1e-5, "vertices %zu vs %zu, normals off %zu, colours off by %g",
              ma.vs.size(), mb.vs.size(), badN, worstC);
      } else {
        const double ra_n = double(ma.vs.size()), rb_n = double(mb.vs.size());
        CHECK("multi_device_surface_solved", rb_n > 3000 && std::fabs(ra_n - rb_n) <= 0.02 * rb_n + 60, "vertices %zu vs %zu", ma.vs.size(),
              mb.vs.size());
      }
    }
  }
  std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
