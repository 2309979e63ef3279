// Probe: does MODE.FP16_OVFL (bit 23 of HW_REG_MODE) make f32 -> f16 conversions saturate at +-65504 on gfx950, and is the
// result for every finite input the same as clamp-then-convert? (profiles/r03_experiments.md)
// build: hipcc --offload-arch=gfx950 -O3 -o build_probe/probe_fp16_ovfl scripts/probe_fp16_ovfl.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 half_t;

__global__ void k(const float* in, unsigned short* clamped, unsigned short* ovfl, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = in[i];
  const half_t a = static_cast<half_t>(fminf(fmaxf(v, -65504.0f), 65504.0f));
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  float w = v;
  asm volatile("" : "+v"(w));
  const half_t b = static_cast<half_t>(w);
  unsigned short ub = __builtin_bit_cast(unsigned short, b);
  asm volatile("" : "+v"(ub));
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
  clamped[i] = __builtin_bit_cast(unsigned short, a);
  ovfl[i] = ub;
}

int main() {
  std::vector<float> h = {70000.f, -70000.f, 65504.f, 65519.9f, 65520.f, 65536.f, 1e30f, -1e30f, INFINITY, -INFINITY, NAN, 1e-8f, 0.f, -0.f, 1.0f, 3.14159f};
  for (int i = 0; i < 100000; ++i) h.push_back((float)(((i * 2654435761u) >> 8) % 200000) / 1.37f - 70000.f);
  const int n = (int)h.size();
  float* d; unsigned short *a, *b;
  hipMalloc(&d, n * 4); hipMalloc(&a, n * 2); hipMalloc(&b, n * 2);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(d, a, b, n);
  std::vector<unsigned short> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 2, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 2, hipMemcpyDeviceToHost);
  int diff = 0;
  for (int i = 0; i < n; ++i) {
    if (i < 16) printf("%14g clamp %04x ovfl %04x\n", h[i], ha[i], hb[i]);
    if (ha[i] != hb[i] && std::isfinite(h[i])) ++diff;
  }
  printf("finite inputs that differ: %d of %d\n", diff, n);
  return 0;
}
