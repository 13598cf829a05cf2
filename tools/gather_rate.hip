// (GPU box) What does one wave-wide GATHER cost on gfx950 at the occupancy the list readers run at (8 waves per SIMD)?
// Every lane loads W bytes (8 / 12 / 16) from its own index; the indices mimic the list readers' pattern: the 64 lanes of
// a wave are neighbours in a Morton-sorted particle array, so their candidates fall into a window of a few thousand
// particles around the wave's own position.  Compared: global gathers from a 16 MB array (L2-resident), the same with
// every lane in its OWN 128-byte line inside a small window, a fully coalesced load, and the same gathers out of an LDS
// tile.  Prints ns per wave-gather per CU.   hipcc -O3 --offload-arch=gfx950 -o tools/gather_rate tools/gather_rate.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

constexpr int TRIPS = 64;   // gathers per lane per launch (4 per loop trip, like the readers)
constexpr int PATTERN_BLOCKS = 64;  // distinct index patterns (relative offsets, L2-resident), reused round robin
constexpr uint32_t NMASK = (1u << 20) - 1;

template <int W> struct Vec;
template <> struct Vec<8> { typedef float2 T; };
template <> struct Vec<12> { typedef float3 T; };
template <> struct Vec<16> { typedef float4 T; };

template <int W> __device__ inline float sum(const typename Vec<W>::T &v);
template <> __device__ inline float sum<8>(const float2 &v) { return v.x + v.y; }
template <> __device__ inline float sum<12>(const float3 &v) { return v.x + v.y + v.z; }
template <> __device__ inline float sum<16>(const float4 &v) { return v.x + v.y + v.z + v.w; }

// idx: [block][trip][thread] like the neighbour lists; src: 16-byte slots
template <int W> __global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ src, const uint32_t *__restrict__ idx,
                                                                 float *__restrict__ out) {
  const uint32_t *mine = idx + size_t(blockIdx.x % PATTERN_BLOCKS) * TRIPS * 256 + threadIdx.x;
  const uint32_t base = blockIdx.x * 256;
  float acc = 0;
  for (int t = 0; t < TRIPS; t += 4) {
    uint32_t b[4];
    typename Vec<W>::T v[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) b[w] = (base + mine[(t + w) * 256]) & NMASK;
#pragma unroll
    for (int w = 0; w < 4; ++w) v[w] = *reinterpret_cast<const typename Vec<W>::T *>(src + b[w]);
#pragma unroll
    for (int w = 0; w < 4; ++w) acc += sum<W>(v[w]);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// the same gathers out of an LDS tile of TILE 16-byte slots staged from src (indices taken modulo TILE)
template <int W, int TILE> __global__ __launch_bounds__(256) void k_gather_lds(const float4 *__restrict__ src,
                                                                               const uint32_t *__restrict__ idx,
                                                                               float *__restrict__ out) {
  __shared__ float4 tile[TILE];
  for (int s = threadIdx.x; s < TILE; s += 256) tile[s] = src[(size_t(blockIdx.x) * 224 + s) % (1u << 20)];
  __syncthreads();
  const uint32_t *mine = idx + size_t(blockIdx.x % PATTERN_BLOCKS) * TRIPS * 256 + threadIdx.x;
  float acc = 0;
  for (int t = 0; t < TRIPS; t += 4) {
    uint32_t b[4];
    typename Vec<W>::T v[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) b[w] = mine[(t + w) * 256] % TILE;
#pragma unroll
    for (int w = 0; w < 4; ++w) v[w] = *reinterpret_cast<const typename Vec<W>::T *>(&tile[b[w]]);
#pragma unroll
    for (int w = 0; w < 4; ++w) acc += sum<W>(v[w]);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K> static double run(K kernel, int blocks, const float4 *src, const uint32_t *idx, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, src, idx, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, src, idx, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return double(ms) / 5;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const uint32_t n = 1u << 20;
  const int blocks = int(n / 256);  // 4096 blocks = 16 per CU, two rounds of 8 resident per CU
  std::vector<float4> h(n);
  for (uint32_t i = 0; i < n; ++i) h[i] = make_float4(float(i), 1.f, 2.f, 3.f);
  float4 *src;
  uint32_t *idx;
  float *out;
  (void)hipMalloc(&src, size_t(n) * 16);
  (void)hipMalloc(&idx, size_t(PATTERN_BLOCKS) * TRIPS * 256 * 4);
  (void)hipMalloc(&out, size_t(n) * 4);
  (void)hipMemcpy(src, h.data(), size_t(n) * 16, hipMemcpyHostToDevice);
  std::mt19937 rng(7);
  std::vector<uint32_t> hi(size_t(PATTERN_BLOCKS) * TRIPS * 256);
  struct Pattern {
    const char *name;
    int kind;
    uint32_t window;
  };
  const Pattern patterns[] = {{"coalesced (lane i reads slot base + i)", 0, 0},
                              {"window 512 around the lane's own slot", 1, 512},
                              {"window 4096 around the lane's own slot", 1, 4096},
                              {"window 65536 around the lane's own slot", 1, 65536},
                              {"uniform over the whole 16 MB array", 1, n},
                              {"cell-like: lanes in groups of 7 share a candidate inside a window of 4096", 2, 4096}};
  std::printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.name, cus);
  for (const Pattern &p : patterns) {
    for (int b = 0; b < PATTERN_BLOCKS; ++b)  // offsets relative to the block's first slot (added modulo n on the device)
      for (int t = 0; t < TRIPS; ++t) {
        uint32_t shared = 0;
        for (int l = 0; l < 256; ++l) {
          uint32_t v;
          if (p.kind == 0) {
            v = uint32_t(l) + uint32_t(t) * 256;
          } else {
            if (p.kind == 1 || l % 7 == 0) shared = (uint32_t(l) + n - p.window / 2 + rng() % p.window) % n;
            v = p.kind == 1 ? shared : (shared + uint32_t(l % 7)) % n;
          }
          hi[(size_t(b) * TRIPS + t) * 256 + l] = v;
        }
      }
    (void)hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    const double wavesGathers = double(blocks) * 4 * TRIPS;  // wave-wide gather instructions per launch
    auto report = [&](const char *what, int w, double ms) {
      std::printf("{\"pattern\": \"%s\", \"from\": \"%s\", \"bytes\": %d, \"ms\": %.4f, \"ns_per_wave_gather_per_cu\": %.2f}\n", p.name,
                  what, w, ms, ms * 1e6 / (wavesGathers / cus));
      std::fflush(stdout);
    };
    report("global", 8, run(k_gather<8>, blocks, src, idx, out));
    report("global", 12, run(k_gather<12>, blocks, src, idx, out));
    report("global", 16, run(k_gather<16>, blocks, src, idx, out));
    if (p.kind != 0) {
      report("lds tile 1024", 8, run(k_gather_lds<8, 1024>, blocks, src, idx, out));
      report("lds tile 1024", 12, run(k_gather_lds<12, 1024>, blocks, src, idx, out));
      report("lds tile 1024", 16, run(k_gather_lds<16, 1024>, blocks, src, idx, out));
    }
  }
  return 0;
}
