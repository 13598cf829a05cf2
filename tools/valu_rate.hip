// (GPU box) VALU issue rates on gfx950 with the occupancy the solver's kernels run at (8 waves per SIMD): how many SIMD
// cycles does one wave64 instruction of each class cost when enough waves are resident to fill the pipe?  Decides what
// "VALU-bound" means for the list readers (profiles/r02_pmc_sq_final.md: 2 951 VALU per wave in 115 us) and whether packed
// fp32 forms could pay.     hipcc -O3 --offload-arch=gfx950 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 4096, UNROLL = 8;
__device__ inline f2 mk2(float x, float y) {
  f2 r;
  r.x = x, r.y = y;
  return r;
}

#define BODY(NAME, DECL, STEP, SINK)                                                            \
  __global__ __launch_bounds__(256) void NAME(float *out, long long *cyc, float seed) {         \
    DECL;                                                                                       \
    const long long t0 = clock64();                                                             \
    for (int i = 0; i < ITER; ++i) {                                                            \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { STEP; }                              \
    }                                                                                           \
    const long long t1 = clock64();                                                             \
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * 256 + threadIdx.x) / 64] = t1 - t0;            \
    out[blockIdx.x * 256 + threadIdx.x] = SINK;                                                 \
  }

// four independent chains per lane: a wave alone can issue back to back
BODY(k_fma, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a + b + c + d)
BODY(k_add, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a + b + c + d)
BODY(k_pk_fma, f2 a = mk2(seed + threadIdx.x, seed); f2 b = a + 1.f; f2 c = a + 2.f; f2 d = a + 3.f; const f2 m = mk2(seed * 0.5f, seed),
     asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a.x + b.y + c.x + d.y)
BODY(k_pk_mul, f2 a = mk2(seed + threadIdx.x, seed); f2 b = a + 1.f; f2 c = a + 2.f; f2 d = a + 3.f; const f2 m = mk2(seed * 0.5f, seed),
     asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a.x + b.y + c.x + d.y)
BODY(k_sqrt, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3,
     asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)),
     a + b + c + d)
BODY(k_rcp, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3,
     asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)),
     a + b + c + d)
BODY(k_cndmask, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)
                  : "vcc"),
     a + b + c + d)
BODY(k_cmp, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4"
                  :
                  : "v"(a), "v"(b), "v"(c), "v"(d), "v"(m)
                  : "vcc"),
     a + b + c + d)
BODY(k_cmp_sgpr, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_cmp_le_f32 s[20:21], %0, %4\n v_cmp_le_f32 s[22:23], %1, %4\n v_cmp_le_f32 s[24:25], %2, %4\n v_cmp_le_f32 s[26:27], %3, %4"
                  :
                  : "v"(a), "v"(b), "v"(c), "v"(d), "v"(m)
                  : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"),
     a + b + c + d)
BODY(k_dot2, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_dot2_i32_i16 %0, %4, %4, %0\n v_dot2_i32_i16 %1, %4, %4, %1\n v_dot2_i32_i16 %2, %4, %4, %2\n v_dot2_i32_i16 %3, %4, %4, %3"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     float(a + b + c + d))
BODY(k_pk_sub_i16, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     float(a + b + c + d))
BODY(k_add_u32, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     float(a + b + c + d))
BODY(k_lshl_or, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_lshl_or_b32 %0, %0, 10, %4\n v_lshl_or_b32 %1, %1, 10, %4\n v_lshl_or_b32 %2, %2, 10, %4\n v_lshl_or_b32 %3, %3, 10, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     float(a + b + c + d))

// --- second batch: the select / carry / shift forms the list build and the readers lean on
BODY(k_cndmask_init, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f;
     asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a), "v"(m) : "vcc"),
     asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)
                  : "vcc"),
     a + b + c + d)
BODY(k_cndmask_sgpr, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f;
     asm volatile("v_cmp_le_f32 s[20:21], %0, %1" : : "v"(a), "v"(m) : "s20", "s21"),
     asm volatile("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)
                  : "s20", "s21"),
     a + b + c + d)
BODY(k_cmp_cndmask, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_le_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)
                  : "vcc"),
     a + b + c + d)
BODY(k_cndmask_distinct, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f; float e; float f; float g; float h,
     asm volatile("v_cndmask_b32 %4, %0, %8, vcc\n v_cndmask_b32 %5, %1, %8, vcc\n v_cndmask_b32 %6, %2, %8, vcc\n v_cndmask_b32 %7, %3, %8, vcc"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h)
                  : "v"(m)
                  : "vcc"),
     a + b + c + d + e + f + g + h)
BODY(k_addc, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)
                  : "vcc"),
     float(a + b + c + d))
BODY(k_mul, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a + b + c + d)
BODY(k_max, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a + b + c + d)
BODY(k_fma_neg, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_fma_f32 %0, -%0, %4, %0\n v_fma_f32 %1, -%1, %4, %1\n v_fma_f32 %2, -%2, %4, %2\n v_fma_f32 %3, -%3, %4, %3"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a + b + c + d)
BODY(k_and, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     float(a + b + c + d))
BODY(k_lshl, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     float(a + b + c + d))
BODY(k_add3, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     float(a + b + c + d))
BODY(k_mul_e64, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_mul_f32_e64 %0, %0, %4\n v_mul_f32_e64 %1, %1, %4\n v_mul_f32_e64 %2, %2, %4\n v_mul_f32_e64 %3, %3, %4"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)),
     a + b + c + d)
BODY(k_cmp_u32_sgpr, int a = int(seed) + threadIdx.x; int b = a + 1; int c = a + 2; int d = a + 3; const int m = int(seed) * 3,
     asm volatile("v_cmp_le_u32 s[20:21], %0, %4\n v_cmp_le_u32 s[22:23], %1, %4\n v_cmp_le_u32 s[24:25], %2, %4\n v_cmp_le_u32 s[26:27], %3, %4"
                  :
                  : "v"(a), "v"(b), "v"(c), "v"(d), "v"(m)
                  : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"),
     float(a + b + c + d))
BODY(k_mix_fma_cnd, float a = seed + threadIdx.x; float b = a + 1; float c = a + 2; float d = a + 3; const float m = seed * 0.5f,
     asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_cndmask_b32 %3, %3, %4, vcc"
                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                  : "v"(m)
                  : "vcc"),
     a + b + c + d)

// one dependent chain per lane: what a single wave's latency-bound stream costs
BODY(k_fma_chain, float a = seed + threadIdx.x; const float m = seed * 0.5f,
     asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(m)), a)

template <typename K> static void run(const char *name, K kernel, int blocksPerCu, float *out, long long *cyc, int cus) {
  const int blocks = cus * blocksPerCu;  // 256 threads = 4 waves = one per SIMD; blocksPerCu waves per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(size_t(blocks) * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (long long v : h) mean += double(v);
  mean /= double(h.size());
  const double instr = double(ITER) * UNROLL * 4;  // per wave
  // clock64 = s_memtime = shader cycles: ticks per instruction of one wave / waves per SIMD = SIMD cycles per instruction
  const double nsPerInstrPerSimd = double(ms) * 1e6 / (instr * blocksPerCu);
  std::printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_wave_instr_per_simd\": %.4f, \"cycles_per_instr_per_wave\": %.3f, \"simd_cycles_per_instr\": %.3f}\n",
              name, blocksPerCu, ms, nsPerInstrPerSimd, mean / instr, mean / instr / blocksPerCu);
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, cus, prop.clockRate);
  float *out;
  long long *cyc;
  (void)hipMalloc(&out, size_t(cus) * 8 * 256 * sizeof(float));
  (void)hipMalloc(&cyc, size_t(cus) * 8 * 4 * sizeof(long long));
  for (int w : {1, 8}) {
    run("v_fma_f32", k_fma, w, out, cyc, cus);
    run("v_add_f32", k_add, w, out, cyc, cus);
    run("v_pk_fma_f32", k_pk_fma, w, out, cyc, cus);
    run("v_pk_mul_f32", k_pk_mul, w, out, cyc, cus);
    run("v_sqrt_f32", k_sqrt, w, out, cyc, cus);
    run("v_rcp_f32", k_rcp, w, out, cyc, cus);
    run("v_cndmask_b32", k_cndmask, w, out, cyc, cus);
    run("v_cmp_le_f32 vcc", k_cmp, w, out, cyc, cus);
    run("v_cmp_le_f32 sgpr", k_cmp_sgpr, w, out, cyc, cus);
    run("v_dot2_i32_i16", k_dot2, w, out, cyc, cus);
    run("v_pk_sub_i16", k_pk_sub_i16, w, out, cyc, cus);
    run("v_add_u32", k_add_u32, w, out, cyc, cus);
    run("v_lshl_or_b32", k_lshl_or, w, out, cyc, cus);
    run("v_cndmask_b32 vcc set by v_cmp", k_cndmask_init, w, out, cyc, cus);
    run("v_cndmask_b32 sgpr pair", k_cndmask_sgpr, w, out, cyc, cus);
    run("v_cmp + v_cndmask pairs", k_cmp_cndmask, w, out, cyc, cus);
    run("v_cndmask_b32 dst != src", k_cndmask_distinct, w, out, cyc, cus);
    run("v_addc_co_u32", k_addc, w, out, cyc, cus);
    run("v_mul_f32", k_mul, w, out, cyc, cus);
    run("v_max_f32", k_max, w, out, cyc, cus);
    run("v_fma_f32 neg", k_fma_neg, w, out, cyc, cus);
    run("v_and_b32", k_and, w, out, cyc, cus);
    run("v_lshlrev_b32", k_lshl, w, out, cyc, cus);
    run("v_add3_u32", k_add3, w, out, cyc, cus);
    run("v_mul_f32_e64", k_mul_e64, w, out, cyc, cus);
    run("v_cmp_le_u32 sgpr", k_cmp_u32_sgpr, w, out, cyc, cus);
    run("3 v_fma + 1 v_cndmask", k_mix_fma_cnd, w, out, cyc, cus);
    run("v_fma_f32 dependent chain", k_fma_chain, w, out, cyc, cus);
  }
  return 0;
}
