"""Yardstick only (not used by the product): what the vendor GEMM (hipBLASLt through torch.mm) sustains on
the encoder's four projection shapes, f16 in / f32 accumulate / f16 out, no epilogue. Sets the practical
ceiling against which gemm_f16x3_256_kernel<EPI, 1> (bias/GELU/residual epilogues fused) is read."""
import torch, time
M = 225280
dev = torch.device("cuda:0")
for name, K, N in (("qkv", 768, 2304), ("out", 768, 768), ("ffn1", 768, 3072), ("ffn2", 3072, 768)):
    a = torch.randn(M, K, device=dev, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.1
    for _ in range(3): torch.mm(a, w.t())
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): torch.mm(a, w.t())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"{name:5s} M={M} N={N} K={K}: {dt * 1e3:.3f} ms  {2 * M * N * K / dt / 1e12:.0f} TFLOP/s")
