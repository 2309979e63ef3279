// Probe: what bounds the GEMM epilogue's write burst? (profiles/r03_experiments.md)
// Every block (8 waves) repeatedly "computes" for `gap` cycles (s_sleep) and then writes a 256 x 256 f16 output tile
// (128 KiB) from registers with non-temporal 16-byte stores, in one of three lane->address patterns, waits for
// vmcnt(0), and stamps the drain with s_memtime. Blocks: 1, 8 (one per XCD), 32, 256. Phase: all blocks together or
// spread uniformly over the period.
//   pattern 0: the shipped epilogue's — an instruction covers 16 token rows x 64 bytes (half lines), the other half of
//              each line follows 8 instructions later
//   pattern 1: an instruction covers 8 token rows x 128 bytes (whole lines)
//   pattern 2: an instruction covers 1 KiB contiguous (a different output layout; the upper bound)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_store_burst scripts/probe_store_burst.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef half_t f16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <int PATTERN>
__global__ __launch_bounds__(512) void burst_kernel(half_t* __restrict__ out, int N, int tiles_n, int rounds, int gap_sleeps,
                                                    int spread_sleeps, long long* __restrict__ stamps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = gridDim.x;
  const int local = (G % 8 == 0) ? (static_cast<int>(blockIdx.x) % 8) * (G / 8) + static_cast<int>(blockIdx.x) / 8
                                 : static_cast<int>(blockIdx.x);
  for (int d = static_cast<int>((static_cast<long long>(spread_sleeps) * local) / G); d > 0; d -= 16) __builtin_amdgcn_s_sleep(16);
  const int wm = wave >> 2, wn = wave & 3;
  f16x8 v;
  for (int i = 0; i < 8; ++i) v[i] = static_cast<half_t>(lane + i);
  long long drain = 0, whole = 0;
  const long long t_begin = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    const int tile = local + r * G;
    const int bm = (tile / tiles_n) * 256, bn = (tile % tiles_n) * 256;
    for (int d = gap_sleeps; d > 0; d -= 16) __builtin_amdgcn_s_sleep(16);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    half_t* base = out + static_cast<long long>(bm + wm * 128) * N + bn + wn * 64;
    if (PATTERN == 0) {
      const int tok = lane & 15, fg = lane >> 4;
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc)
          __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(base + static_cast<long long>(16 * pc + tok) * N + 32 * p2 + 8 * fg));
    } else if (PATTERN == 1) {
      const int tok = lane >> 3, c = lane & 7;
#pragma unroll
      for (int pc = 0; pc < 16; ++pc)
        __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(base + static_cast<long long>(8 * pc + tok) * N + 8 * c));
    } else if (PATTERN == 3) {  // pattern 1's lines with the lane order an exchange between lanes t and t ^ 8 would give
      const int t = lane & 15, g = lane >> 4;
#pragma unroll
      for (int pc = 0; pc < 16; ++pc)
        __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(base + static_cast<long long>(8 * pc + (t & 7)) * N + 8 * (g + 4 * (t >> 3))));
    } else if (PATTERN == 4) {  // 16 token rows x 64 bytes as shipped, but the two halves of a line back to back
      const int tok = lane & 15, fg = lane >> 4;
#pragma unroll
      for (int pc = 0; pc < 8; ++pc)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
          __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(base + static_cast<long long>(16 * pc + tok) * N + 32 * p2 + 8 * fg));
    } else if (PATTERN == 5) {  // pattern 3 with plain (temporal) stores
      const int t = lane & 15, g = lane >> 4;
#pragma unroll
      for (int pc = 0; pc < 16; ++pc)
        *reinterpret_cast<f16x8*>(base + static_cast<long long>(8 * pc + (t & 7)) * N + 8 * (g + 4 * (t >> 3))) = v;
    } else if (PATTERN == 6) {  // 8 bytes per lane, 16 lanes = one 128-byte line, 4 token rows per instruction, 32 instructions
      typedef half_t f16x4 __attribute__((ext_vector_type(4)));
      const f16x4 v4 = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int pc = 0; pc < 32; ++pc)
        __builtin_nontemporal_store(v4, reinterpret_cast<f16x4*>(base + static_cast<long long>(4 * pc + (lane >> 4)) * N + 4 * (lane & 15)));
    } else if (PATTERN == 7) {  // 16 bytes per lane, even lanes one row, odd lanes the next (what a swap of lane pairs gives)
#pragma unroll
      for (int pc = 0; pc < 16; ++pc)
        __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(base + static_cast<long long>(8 * pc + 2 * (lane >> 4) + (lane & 1)) * N + 8 * ((lane & 15) >> 1)));
    } else if (PATTERN == 8) {  // 16 bytes per lane, 4 adjacent lanes = 64 contiguous bytes, 16 rows per instruction
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc)
          __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(base + static_cast<long long>(16 * pc + (lane >> 2)) * N + 32 * p2 + 8 * (lane & 3)));
    } else {
      half_t* flat = out + (static_cast<long long>(tile) * 8 + wave) * 8192;
#pragma unroll
      for (int pc = 0; pc < 16; ++pc) __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(flat + pc * 512 + lane * 8));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    drain += t1 - t0;
  }
  whole = __builtin_amdgcn_s_memtime() - t_begin;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = drain;
    stamps[2 * blockIdx.x + 1] = whole;
  }
}

int main() {
  const int N = 3072, M = 262144;  // the FFN-up output of a 262144-token chunk: 1.5 GiB
  const int tiles_n = N / 256;
  half_t* out;
  CK(hipMalloc(&out, static_cast<size_t>(M) * N * 2));
  long long* stamps;
  CK(hipMalloc(&stamps, 2 * 256 * sizeof(long long)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("pattern blocks gap_us spread | drain per burst, 100 s_memtime ticks | (ignore) | wall ms | chip GB/s over wall\n");
  for (int pattern : {0, 1, 6, 7, 8})
    for (int blocks : {1, 256})
      for (int gap_us : {0, 20})
        for (int spread : {0, 1}) {
          if (spread && (gap_us == 0 || blocks == 1)) continue;
          const int rounds = 40;
          // s_sleep(n) sleeps ~64 n cycles; at ~2.1 GHz 1 us ~ 33 sleeps-units of 64 cycles
          const int gap_sleeps = gap_us * 33;
          const int spread_sleeps = spread ? (gap_us + 6) * 33 : 0;
          auto launch = [&]() {
            if (pattern == 0) burst_kernel<0><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 1) burst_kernel<1><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 2) burst_kernel<2><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 3) burst_kernel<3><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 4) burst_kernel<4><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 6) burst_kernel<6><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 7) burst_kernel<7><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 8) burst_kernel<8><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
            if (pattern == 5) burst_kernel<5><<<blocks, 512>>>(out, N, tiles_n, rounds, gap_sleeps, spread_sleeps, stamps);
          };
          launch();
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0));
          launch();
          CK(hipEventRecord(e1));
          CK(hipDeviceSynchronize());
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          std::vector<long long> h(2 * blocks);
          CK(hipMemcpy(h.data(), stamps, 2 * blocks * sizeof(long long), hipMemcpyDeviceToHost));
          double drain = 0;
          for (int b = 0; b < blocks; ++b) drain += static_cast<double>(h[2 * b]);
          const double us = drain / blocks / rounds / 100.0;  // in units of 100 s_memtime ticks (the counter runs at the shader clock here: compare with wall)
          printf("%d %4d %3d %d | %8.2f | %8.1f | %8.3f | %8.1f\n", pattern, blocks, gap_us, spread, us, 131072.0 / us / 1e3, ms,
                 131072.0 * blocks * rounds / ms / 1e6);
        }
  return 0;
}
