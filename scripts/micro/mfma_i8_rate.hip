// Issue rate of v_mfma_i32_16x16x64_i8 (and v_mfma_f32_16x16x32_f16 beside it): back-to-back independent MFMAs, no memory.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_i8_rate.hip -o gpurun_out/mfma_i8_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, int* out) {
  i32x4 a = {int(threadIdx.x), 1, 2, 3}, b = {3, 2, 1, int(threadIdx.x)};
  i32x4 acc[16];
  f32x4 facc[16];
  for (int i = 0; i < 16; ++i) { acc[i] = i32x4{0, 0, 0, 0}; facc[i] = f32x4{0, 0, 0, 0}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
      else facc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<f16x8*>(&a), *reinterpret_cast<f16x8*>(&b), facc[i], 0, 0, 0);
    }
  }
  int s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + int(facc[i][0]);
  if (s == 0x7fffffff) out[0] = s;
}

int main() {
  int* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int threads : {256, 512}) {
      const int iters = 20000, blocks = 256;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, iters, out);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = double(blocks) * (threads / 64) * iters * 16;
        const double ops = n * 2.0 * 16 * 16 * (kind == 0 ? 64 : 32);
        if (rep) printf("%s, %d waves per CU: %.3f ms, %.1f T%s/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", kind == 0 ? "i32_16x16x64_i8" : "f32_16x16x32_f16",
                        threads / 64, ms, ops / ms / 1e9, kind == 0 ? "OP" : "FLOP", ms * 1e-3 * 2.4e9 / (n / (256.0 * 4)));
      }
    }
  return 0;
}
