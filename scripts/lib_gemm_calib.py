"""Calibration only (not product): what the vendor GEMM library reaches on the DiT's GEMM shapes, to judge the
headroom of the hand-written kernels.  fp16 in, fp32 accumulate, torch.matmul (hipBLASLt / rocBLAS)."""
import torch, time
dev = torch.device("cuda")
shapes = {"qkv": (2112, 3072, 1024), "out": (2112, 1024, 1024), "ff_in": (2112, 8192, 1024), "ff_out": (2112, 1024, 4096),
          "big": (8192, 8192, 8192)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(5): torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): torch.matmul(a, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:7s} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
