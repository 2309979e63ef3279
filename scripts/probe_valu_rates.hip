// Probe: issue rates of packed-f32 and scalar-f32 vector instructions on gfx950 (profiles/r03_experiments.md §9).
// One wave per SIMD (256 threads per block, one block per CU... one block total), 8 independent chains per instruction
// kind, 4096 rounds; cycles per instruction = s_memtime delta / (rounds * 8).
// build: hipcc --offload-arch=gfx950 -O3 -o build_probe/probe_valu_rates scripts/probe_valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef CHAINS
#define CHAINS 8
#endif
#define ROUNDS 4096
#define LOOP(NAME, INSTR)                                                                       \
  __global__ void NAME(float* out, long long* cyc, float seed) {                                \
    f32x2 r[CHAINS];                                                                            \
    for (int i = 0; i < CHAINS; ++i) r[i] = f32x2{seed + i + threadIdx.x, seed - i};           \
    f32x2 c = {seed * 0.5f, 1.0f};                                                              \
    const long long t0 = __builtin_amdgcn_s_memtime();                                          \
    for (int k = 0; k < ROUNDS; ++k) {                                                          \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(INSTR : "+v"(r[i]) : "v"(c)); \
    }                                                                                           \
    const long long t1 = __builtin_amdgcn_s_memtime();                                          \
    float s = 0;                                                                                \
    for (int i = 0; i < CHAINS; ++i) s += r[i].x + r[i].y;                                      \
    out[threadIdx.x] = s;                                                                       \
    if (threadIdx.x == 0) *cyc = t1 - t0;                                                       \
  }
LOOP(k_pk_add, "v_pk_add_f32 %0, %0, %1")
LOOP(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
LOOP(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %1")
#define LOOP1(NAME, INSTR)                                                                      \
  __global__ void NAME(float* out, long long* cyc, float seed) {                                \
    float r[CHAINS];                                                                            \
    for (int i = 0; i < CHAINS; ++i) r[i] = seed + i + threadIdx.x;                             \
    float c = seed * 0.5f;                                                                      \
    const long long t0 = __builtin_amdgcn_s_memtime();                                          \
    for (int k = 0; k < ROUNDS; ++k) {                                                          \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(INSTR : "+v"(r[i]) : "v"(c)); \
    }                                                                                           \
    const long long t1 = __builtin_amdgcn_s_memtime();                                          \
    float s = 0;                                                                                \
    for (int i = 0; i < CHAINS; ++i) s += r[i];                                                 \
    out[threadIdx.x] = s;                                                                       \
    if (threadIdx.x == 0) *cyc = t1 - t0;                                                       \
  }
LOOP1(k_add, "v_add_f32 %0, %0, %1")
LOOP1(k_fma, "v_fma_f32 %0, %0, %1, %1")
LOOP1(k_med3, "v_med3_f32 %0, %0, %1, %1")
LOOP1(k_cvt, "v_cvt_pk_f16_f32 %0, %0, %1")

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  struct { const char* n; void (*f)(float*, long long*, float); } ks[] = {{"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul},
    {"v_pk_fma_f32", k_pk_fma}, {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_med3_f32", k_med3}, {"v_cvt_pk_f16_f32", k_cvt}};
  for (int waves : {1, 2}) {
    for (auto& k : ks) {
      long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k.f, dim3(1), dim3(256 * waves), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
      }
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%d wave(s) per SIMD  %-26s %.2f cycles per instruction (per wave)\n", waves, k.n, (double)h / (ROUNDS * CHAINS));
    }
  }
  return 0;
}
