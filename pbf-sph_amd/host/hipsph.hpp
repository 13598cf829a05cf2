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
//   Solver(h, {d0, d1, ...})     several GPUs of one node (reference: the -d/--devices list, src/args.cpp:20-24):
//                      the box is cut into x-slabs with equal particle counts, one ctx + one host thread per
//                      device, every step runs pbf_slab_step on all of them concurrently with the neighbour
//                      exchange over RCCL (ncclSend/ncclRecv, xGMI) inside the library; cuts are re-balanced
//                      every few steps from the all-device column histogram.  Devices listed more than once share
//                      a GPU and exchange through an in-process transport (RCCL refuses two ranks on one GPU):
//                      the same code path, used by the tests on a one-GPU box.
//
// Host-side scene handling restates the observable behaviour of ompsph.hpp:91-126 (sources emit,
// drains erase, empty -> "Particles depleted") and :167-186 (queries).  config.surface runs the
// marching-cubes kernels (pbf_surface) and fills Result::mesh like ompsph.hpp:277-477.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pbf_hip.h"
#include "sph.hpp"

namespace sph::hip_impl {

namespace detail {

// One direction of a link between two slabs that share a GPU: the sender posts its (pinned host) buffer, the receiver
// copies it out.  Used through pbf_comm_create_host_callback; never on the RCCL path.
struct Mailbox {
  std::mutex m;
  std::condition_variable cv;
  const void *data = nullptr;
  size_t bytes = 0;
  bool full = false;
  void post(const void *d, size_t n) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return !full; });
    data = d, bytes = n, full = true;
    cv.notify_all();
  }
  void take(void *dst, size_t n) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return full; });
    std::memcpy(dst, data, std::min(n, bytes));
    full = false;
    cv.notify_all();
  }
  void drained() {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return !full; });
  }
};
struct InProcessLink {  // rank's view: mailboxes to / from both neighbours
  Mailbox *toLeft = nullptr, *toRight = nullptr, *fromLeft = nullptr, *fromRight = nullptr;
  static int exchange(void *user, const void *sL, size_t nSL, const void *sR, size_t nSR, void *rL, size_t nRL, void *rR,
                      size_t nRR) {
    auto *k = static_cast<InProcessLink *>(user);
    if (nSL && k->toLeft) k->toLeft->post(sL, nSL);
    if (nSR && k->toRight) k->toRight->post(sR, nSR);
    if (nRL && k->fromLeft) k->fromLeft->take(rL, nRL);
    if (nRR && k->fromRight) k->fromRight->take(rR, nRR);
    if (nSL && k->toLeft) k->toLeft->drained();  // our staging buffers are free again once the neighbours copied
    if (nSR && k->toRight) k->toRight->drained();
    return 0;
  }
};

// A fixed crew of host threads, one per slab, that runs f(0) .. f(n - 1) concurrently on every run() call.
class Workers {
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable go_, done_;
  std::function<void(size_t)> task_;
  std::vector<std::string> errs_;
  uint64_t epoch_ = 0;
  size_t pending_ = 0;
  bool stop_ = false;

public:
  size_t size() const { return threads_.size(); }
  void start(size_t n) {
    shutdown();
    stop_ = false;
    errs_.assign(n, std::string());
    for (size_t g = 0; g < n; ++g)
      threads_.emplace_back([this, g] {
        uint64_t seen = 0;
        for (;;) {
          std::function<void(size_t)> job;
          {
            std::unique_lock<std::mutex> l(m_);
            go_.wait(l, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
            job = task_;
          }
          std::string err;
          try {
            job(g);
          } catch (const std::exception &e) {
            err = e.what();
          }
          std::unique_lock<std::mutex> l(m_);
          errs_[g] = err;
          if (--pending_ == 0) done_.notify_all();
        }
      });
  }
  void run(std::function<void(size_t)> f) {
    std::unique_lock<std::mutex> l(m_);
    task_ = std::move(f);
    pending_ = threads_.size();
    ++epoch_;
    go_.notify_all();
    done_.wait(l, [&] { return pending_ == 0; });
    for (const auto &e : errs_)
      if (!e.empty()) throw std::runtime_error(e);
  }
  void shutdown() {
    {
      std::unique_lock<std::mutex> l(m_);
      stop_ = true;
      go_.notify_all();
    }
    for (auto &t : threads_) t.join();
    threads_.clear();
  }
  ~Workers() { shutdown(); }
};

// slab.py recut(): particle-count quantiles of the column histogram, each cut moved by at most `maxMove` columns
// (a transferred column always goes to the ADJACENT slab) and slabs kept >= minWidth columns wide.
inline std::vector<uint32_t> recut(const std::vector<uint32_t> &old, const std::vector<uint64_t> &hist, int maxMove = 2,
                                   int minWidth = 4) {
  const int n = int(old.size()) - 1;
  std::vector<uint64_t> cum(hist.size() + 1, 0);
  for (size_t c = 0; c < hist.size(); ++c) cum[c + 1] = cum[c] + hist[c];
  const double total = double(cum.back());
  std::vector<long> nw(old.begin(), old.end());
  auto clampMove = [&](int g) { nw[g] = std::min<long>(std::max<long>(nw[g], long(old[g]) - maxMove), long(old[g]) + maxMove); };
  for (int g = 1; g < n; ++g) {
    const double target = total * g / n;
    nw[g] = long(std::lower_bound(cum.begin(), cum.end(), target, [](uint64_t a, double t) { return double(a) < t; }) - cum.begin());
    clampMove(g);
  }
  for (int g = 1; g < n; ++g) nw[g] = std::max(nw[g], nw[g - 1] + (g > 1 ? minWidth : 1));
  for (int g = n - 1; g > 0; --g) nw[g] = std::min(nw[g], nw[g + 1] - (g < n - 1 ? minWidth : 1));
  for (int g = 1; g < n; ++g) clampMove(g);
  return std::vector<uint32_t>(nw.begin(), nw.end());
}

}  // namespace detail

// V: the vector template of the hosting code base — the reference instantiates its API with glm::vec (src/omp/ompsph.hpp:33),
// this repo's host/sph.hpp ships sph::vec and announces it with PBF_SPH_HAS_VEC, which only supplies a default here.  Built
// against the REFERENCE's own src/sph.hpp (no sph::vec there) the argument is mandatory: oracle/ref_shim.cpp does exactly
// that, so the "copy two headers, add a case" recipe of INTEGRATION.md is compiled, linked and run by the test-suite.
// What V must offer (glm::vec does): x / y / z[/ w] members, V<3>(a, b, c) from mixed arithmetic types, V<3> + V<3>,
// V<3> - V<3>, V<3> * N, and a packed layout.
#ifdef PBF_SPH_HAS_VEC
template <typename T, typename N, template <size_t, typename C = N> typename V = sph::vec>
#else
template <typename T, typename N, template <size_t, typename C = N> typename V>
#endif
class Solver final : public sph::Solver<T, N, V> {
  static_assert(std::is_same_v<N, float> || std::is_same_v<N, double>, "N must be float or double");
  static_assert(sizeof(T) == 8, "ids travel as 64-bit (the reference instantiates T = size_t)");
  static_assert(sizeof(V<3>) == 3 * sizeof(N) && sizeof(V<4>) == 4 * sizeof(N), "V must be packed");

  pbf_ctx *ctx_ = nullptr;
  const N h_;
  std::vector<double> wells_;
  // ---- several devices (slabs): ctx_ aliases slabs_[0] ----------------------------------------
  std::vector<int> devices_;
  std::vector<pbf_ctx *> slabs_;
  std::vector<pbf_comm *> comms_;
  std::vector<std::unique_ptr<detail::Mailbox>> boxes_;
  std::vector<detail::InProcessLink> links_;
  std::vector<uint32_t> cuts_;
  uint32_t flags_ = 0;
  uint64_t frame_ = 0;
  unsigned rebalanceEvery_ = 8;
  bool attached_ = false;
  bool multi() const { return slabs_.size() > 1; }

  // PBF_SHIM_TIMING=1: mean host time of advance()'s phases, printed by the destructor (diagnostic)
  struct Phases {
    bool on = std::getenv("PBF_SHIM_TIMING") != nullptr;
    double ms[5] = {0, 0, 0, 0, 0};
    uint64_t frames = 0;
  } phase_;
  struct PhaseClock {
    Phases &p;
    std::chrono::steady_clock::time_point t;
    explicit PhaseClock(Phases &ph) : p(ph), t(std::chrono::steady_clock::now()) { p.frames += p.on; }
    void lap(int k) {
      if (!p.on) return;
      const auto n = std::chrono::steady_clock::now();
      p.ms[k] += std::chrono::duration<double, std::milli>(n - t).count();
      t = n;
    }
  };
  // One PERSISTENT host thread per device (round 3: fresh std::threads every step cost ~50 us of creation and join per
  // step): the workers sleep on a condition variable between tasks; errors are rethrown on the caller's thread.
  detail::Workers workers_;
  template <typename F> void parallel(F &&f) {
    if (workers_.size() != slabs_.size()) workers_.start(slabs_.size());
    workers_.run(std::function<void(size_t)>(std::forward<F>(f)));
  }
  void checkOn(size_t g, int rc, const char *what) const {
    if (rc < 0) throw std::runtime_error(std::string(what) + " (device " + std::to_string(devices_[g]) + "): " + pbf_last_error(slabs_[g]));
  }
  uint32_t columnOf(const sph::SphParams<T, N, V> &c, N x) const {  // ompsph.hpp:132-135,152 in N
    const N minExtent = c.minBound.x / c.scale - h_ * 2;
    return uint32_t(int64_t((x / c.scale - minExtent) / h_));
  }

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
  // Several GPUs of one node: x-slabs, one ctx per entry of `devices` (an entry may repeat: see the file header).
  Solver(N h, std::vector<int> devices, uint32_t flags = 0) : h_(h), devices_(std::move(devices)), flags_(flags) {
    if (devices_.empty()) devices_.push_back(0);
    for (int dev : devices_) {
      pbf_desc d{};
      d.abi_version = PBF_ABI_VERSION;
      d.fp64 = std::is_same_v<N, double> ? 1 : 0;
      d.device = dev, d.flags = flags, d.h = double(h), d.stream = nullptr;
      pbf_ctx *c = nullptr;
      if (pbf_create(&d, &c) != PBF_OK) {
        const std::string msg = pbf_last_error(nullptr);
        for (auto *s : slabs_) pbf_destroy(s);
        throw std::runtime_error("pbf_create: " + msg);
      }
      slabs_.push_back(c);
    }
    ctx_ = slabs_[0];
    if (!multi()) return;
    const int n = int(slabs_.size());
    comms_.assign(n, nullptr);
    const bool distinct = std::set<int>(devices_.begin(), devices_.end()).size() == devices_.size();
    if (distinct) {  // RCCL over xGMI: ncclCommInitRank must be entered by all ranks concurrently
      unsigned char id[PBF_COMM_ID_BYTES];
      if (pbf_comm_unique_id(id) != PBF_OK) throw std::runtime_error(std::string("pbf_comm_unique_id: ") + pbf_comm_last_error(nullptr));
      parallel([&](size_t g) {
        if (pbf_comm_create_rccl(id, n, int(g), devices_[g], &comms_[g]) != PBF_OK)
          throw std::runtime_error(std::string("pbf_comm_create_rccl: ") + pbf_comm_last_error(nullptr));
      });
    } else {  // slabs sharing a GPU: in-process mailboxes behind the library's host-callback transport
      for (int k = 0; k < 2 * (n - 1); ++k) boxes_.emplace_back(new detail::Mailbox());
      links_.resize(n);
      for (int g = 0; g < n; ++g) {
        if (g > 0) links_[g].toLeft = boxes_[2 * (g - 1) + 1].get(), links_[g].fromLeft = boxes_[2 * (g - 1)].get();
        if (g + 1 < n) links_[g].toRight = boxes_[2 * g].get(), links_[g].fromRight = boxes_[2 * g + 1].get();
      }
      for (int g = 0; g < n; ++g)
        if (pbf_comm_create_host_callback(&detail::InProcessLink::exchange, &links_[g], n, g, &comms_[g]) != PBF_OK)
          throw std::runtime_error("pbf_comm_create_host_callback failed");
    }
  }
  ~Solver() override {
    if (phase_.on && phase_.frames)
      std::fprintf(stderr, "advance() phases, mean ms over %llu frames: upload %.3f | step %.3f | surface kernels + mesh DMA %.3f | "
                           "particle download %.3f | mesh vectors (rest) %.3f\n", (unsigned long long)phase_.frames,
                   phase_.ms[0] / phase_.frames, phase_.ms[1] / phase_.frames, phase_.ms[2] / phase_.frames,
                   phase_.ms[3] / phase_.frames, phase_.ms[4] / phase_.frames);
    workers_.shutdown();
    if (slabs_.empty()) pbf_destroy(ctx_);
    for (auto *s : slabs_) pbf_destroy(s);
    for (auto *c : comms_) pbf_comm_destroy(c);
  }
  size_t deviceCount() const { return slabs_.empty() ? 1 : slabs_.size(); }
  const std::vector<uint32_t> &cuts() const { return cuts_; }
  void setRebalanceEvery(unsigned steps) { rebalanceEvery_ = steps; }
  Solver(const Solver &) = delete;
  Solver &operator=(const Solver &) = delete;

  pbf_ctx *context() { return ctx_; }

  // ---- device-resident path -------------------------------------------------------------------
  // Several devices: `config` places the cuts (its bounds and scale define the grid columns); the slabs start with
  // equal particle counts.
  void upload(const std::vector<sph::Particle<T, N, V>> &xs, const sph::SphParams<T, N, V> *config = nullptr) {
    const auto l = layout();
    if (!multi()) {
      check(pbf_upload_aos(ctx_, xs.size(), xs.data(), &l), "pbf_upload_aos");
      return;
    }
    if (!config) throw std::runtime_error("upload() on several devices needs the SphParams (grid columns for the cuts)");
    const int n = int(slabs_.size());
    std::vector<uint32_t> col(xs.size());
    for (size_t i = 0; i < xs.size(); ++i) col[i] = std::min<uint32_t>(columnOf(*config, xs[i].position.x), 1023u);
    if (!attached_) {  // first upload: cuts at the particle-count quantiles of the columns
      std::vector<uint32_t> sorted(col);
      std::sort(sorted.begin(), sorted.end());
      cuts_.assign(n + 1, 0);
      for (int g = 1; g < n; ++g) cuts_[g] = sorted.empty() ? uint32_t(g) : sorted[std::min(sorted.size() - 1, sorted.size() * g / n)];
      cuts_[n] = 1024;
      for (int g = 1; g <= n; ++g) cuts_[g] = std::max(cuts_[g], cuts_[g - 1] + 1);
    }
    std::vector<std::vector<sph::Particle<T, N, V>>> part(n);
    for (size_t i = 0; i < xs.size(); ++i) {
      const int g = int(std::upper_bound(cuts_.begin() + 1, cuts_.end() - 1, col[i]) - (cuts_.begin() + 1));
      part[g].push_back(xs[i]);
    }
    const size_t per = std::max<size_t>(xs.size() / n, 1);
    // records in the first message of the two assembly rounds (the remainder follows when needed)
    const uint32_t capGhost = uint32_t(std::max<size_t>(per / 8, 1u << 14)), capMig = uint32_t(std::max<size_t>(per / 64, 1u << 12));
    for (int g = 0; g < n; ++g) {
      if (!attached_) checkOn(g, pbf_reserve(slabs_[g], 3 * per + (1u << 16)), "pbf_reserve");
      checkOn(g, pbf_upload_aos(slabs_[g], part[g].size(), part[g].data(), &l), "pbf_upload_aos");
      if (!attached_) checkOn(g, pbf_slab_attach(slabs_[g], comms_[g], cuts_.data(), capMig, capGhost), "pbf_slab_attach");
    }
    attached_ = true;
  }
  void step(const sph::SphParams<T, N, V> &config, const sph::Scene<T, N, V> &scene = {}, uint32_t count = 1) {
    const pbf_params p = params(config, scene);
    if (!multi()) {
      check(pbf_steps(ctx_, &p, count), "pbf_steps");
      return;
    }
    if (!attached_) throw std::runtime_error("step() before upload()");
    for (uint32_t k = 0; k < count; ++k) {
      if (rebalanceEvery_ && frame_ && frame_ % rebalanceEvery_ == 0) rebalance();
      ++frame_;
      parallel([&](size_t g) { checkOn(g, pbf_slab_step(slabs_[g], &p), "pbf_slab_step"); });
    }
  }
  // Load balance: all-device column histogram -> new cuts (detail::recut) -> every ctx.
  bool rebalance() {
    std::vector<uint64_t> hist(1024, 0);
    for (size_t g = 0; g < slabs_.size(); ++g) {
      uint32_t h[1024];
      checkOn(g, pbf_slab_column_histogram(slabs_[g], h), "pbf_slab_column_histogram");
      for (int c = 0; c < 1024; ++c) hist[c] += h[c];
    }
    const auto nw = detail::recut(cuts_, hist);
    if (nw == cuts_) return false;
    cuts_ = nw;
    for (size_t g = 0; g < slabs_.size(); ++g) checkOn(g, pbf_slab_set_cuts(slabs_[g], cuts_.data()), "pbf_slab_set_cuts");
    return true;
  }
  void sync() {
    if (!multi()) check(pbf_sync(ctx_), "pbf_sync");
    else
      for (size_t g = 0; g < slabs_.size(); ++g) checkOn(g, pbf_sync(slabs_[g]), "pbf_sync");
  }
  size_t count() const {
    if (!multi()) return pbf_count(ctx_);
    size_t n = 0;
    for (auto *s : slabs_) n += pbf_owned_count(s);
    return n;
  }
  // Several devices: slab after slab (each slab in its own Z-order).
  void download(std::vector<sph::Particle<T, N, V>> &xs) {
    const auto l = layout();
    if (!multi()) {
      xs.resize(pbf_count(ctx_));
      check(pbf_download_aos(ctx_, xs.data(), &l), "pbf_download_aos");
      return;
    }
    xs.resize(count());
    size_t at = 0;
    for (size_t g = 0; g < slabs_.size(); ++g) {
      checkOn(g, pbf_download_aos(slabs_[g], xs.data() + at, &l), "pbf_download_aos");
      at += pbf_owned_count(slabs_[g]);
    }
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
    if (multi()) {
      if (!scene.queries.empty()) throw std::runtime_error("queries are a single-device feature");
      upload(xs, &config);
      step(config, scene, 1);
      sph::Result<T, N, V> result;
      if (config.surface) result.mesh = surface(config, scene);  // every slab its part of the lattice, concatenated
      download(xs);
      return result;
    }
    PhaseClock clk(phase_);
    upload(xs);
    clk.lap(0);
    step(config, scene, 1);
    if (phase_.on) sync();
    clk.lap(1);
    sph::Result<T, N, V> result;
    if (!scene.queries.empty()) result.queries = query(config, scene);
    if (config.surface) {  // ompsph.hpp:277-477
      // the mesh lands in page-locked staging (one DMA); its three vectors are then built on host threads WHILE the
      // particles travel back over PCIe
      // — and the particles set off BEFORE the surface kernels start (pbf_download_aos_begin: their DMA runs on a copy
      // stream while the field / count / emit kernels, which only read the state, execute)
      static const bool early = std::getenv("PBF_SHIM_LATE_DOWNLOAD") == nullptr;  // (A/B switch: the round-2 order)
      const auto l = layout();
      xs.resize(pbf_count(ctx_));
      if (early) check(pbf_download_aos_begin(ctx_, xs.data(), &l), "pbf_download_aos_begin");
      MeshCopy copy(*this, config, scene, result.mesh);
      clk.lap(2);
      if (early) check(pbf_download_aos_end(ctx_), "pbf_download_aos_end");
      else check(pbf_download_aos(ctx_, xs.data(), &l), "pbf_download_aos");
      clk.lap(3);
      copy.join();
      clk.lap(4);
    } else {
      download(xs);
      clk.lap(3);
    }
    return result;
  }

  // Marching-cubes surface of the state the last step left (reference: config.surface, ompsph.hpp:277-477).
  // Several devices: every slab extracts the cubes of its own node planes (pbf_surface is collective there: the slabs
  // refresh their copies' colours and hand one node plane to the left), the parts are concatenated in slab order — the
  // single-device mesh's cube order.
  sph::ColouredMesh<N, V> surface(const sph::SphParams<T, N, V> &config, const sph::Scene<T, N, V> &scene = {}) {
    sph::ColouredMesh<N, V> mesh;
    if (!multi()) {
      MeshCopy(*this, config, scene, mesh).join();
      return mesh;
    }
    const pbf_params p = params(config, scene);
    const pbf_mc_params mc{double(config.surface->resolution), double(config.surface->isolevel),
                           double(config.surface->particleSize), double(config.surface->particleInfluence)};
    std::vector<uint64_t> tri(slabs_.size(), 0);
    parallel([&](size_t g) { checkOn(g, pbf_surface(slabs_[g], &p, &mc, &tri[g]), "pbf_surface"); });
    size_t total = 0;
    for (uint64_t t : tri) total += size_t(t) * 3;
    mesh.vs.reserve(total), mesh.ns.reserve(total), mesh.cs.reserve(total);
    for (size_t g = 0; g < slabs_.size(); ++g) {
      const void *pv = nullptr, *pn = nullptr, *pc = nullptr;
      checkOn(g, pbf_map_mesh(slabs_[g], &pv, &pn, &pc), "pbf_map_mesh");
      const size_t nv = size_t(tri[g]) * 3;
      if (!nv) continue;
      const V<3> *v3 = static_cast<const V<3> *>(pv), *n3 = static_cast<const V<3> *>(pn);
      const V<4> *c4 = static_cast<const V<4> *>(pc);
      mesh.vs.insert(mesh.vs.end(), v3, v3 + nv), mesh.ns.insert(mesh.ns.end(), n3, n3 + nv);
      mesh.cs.insert(mesh.cs.end(), c4, c4 + nv);
    }
    return mesh;
  }

private:
  // pbf_surface + the mesh hand-over.  ColouredMesh(size) would zero-fill 54 MB (1 M particles) that a pageable
  // device-to-host copy then overwrites at a few GB/s: 11 ms per frame in round 2.  Instead: ONE DMA into the library's
  // page-locked staging (pbf_map_mesh), then the three vectors are range-assigned from it (no fill), each on a thread of its
  // own — so the caller can do something useful (the particle download) until join().
  struct MeshCopy {
    sph::ColouredMesh<N, V> &mesh;
    const V<3> *v3 = nullptr, *n3 = nullptr;
    const V<4> *c4 = nullptr;
    size_t nv = 0;
    std::thread tv, tn, tc;
    MeshCopy(Solver &s, const sph::SphParams<T, N, V> &config, const sph::Scene<T, N, V> &scene, sph::ColouredMesh<N, V> &out)
        : mesh(out) {
      const pbf_params p = s.params(config, scene);
      const pbf_mc_params mc{double(config.surface->resolution), double(config.surface->isolevel),
                             double(config.surface->particleSize), double(config.surface->particleInfluence)};
      uint64_t triangles = 0;
      s.check(pbf_surface(s.ctx_, &p, &mc, &triangles), "pbf_surface");
      const void *pv = nullptr, *pn = nullptr, *pc = nullptr;
      s.check(pbf_map_mesh(s.ctx_, &pv, &pn, &pc), "pbf_map_mesh");
      nv = size_t(triangles) * 3;
      v3 = static_cast<const V<3> *>(pv), n3 = static_cast<const V<3> *>(pn), c4 = static_cast<const V<4> *>(pc);
      if (nv >= (size_t(1) << 16)) {  // (small meshes — the stock 18 K-particle run: a thread costs more than the copy)
        tv = std::thread([this] { mesh.vs.assign(v3, v3 + nv); });
        tn = std::thread([this] { mesh.ns.assign(n3, n3 + nv); });
        tc = std::thread([this] { mesh.cs.assign(c4, c4 + nv); });
      }
    }
    void join() {
      if (nv == 0) return;
      if (tv.joinable()) {
        tv.join(), tn.join(), tc.join();
      } else {
        mesh.vs.assign(v3, v3 + nv), mesh.ns.assign(n3, n3 + nv), mesh.cs.assign(c4, c4 + nv);
      }
      nv = 0;
    }
    ~MeshCopy() {
      if (tv.joinable()) tv.join();
      if (tn.joinable()) tn.join();
      if (tc.joinable()) tc.join();
    }
  };

public:

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
