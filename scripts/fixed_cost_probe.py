"""Development: launch + prologue + epilogue cost of the row-panel GEMM kernels at the DiT shapes (M = 2112), from a
linear fit of the launch time over K (the k loop is the only part that grows with K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native
def PV(rows, bn): return 0x20 | (rows << 8) | (bn << 20) | 0x40
eng = native.Engine(precision=3, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
M = 2112
for name, N, rows, bn, ks in (("ff1 264x256", 8192, 264, 256, 1), ("out 66x128", 1024, 66, 128, 1), ("ff2 132x256/k4 (K = 4x)", 1024, 132, 256, 4),
                              ("qkv 104x256", 3072, 104, 256, 1)):
    pts = []
    for K in (64, 128, 256, 512, 1024, 2048):
        ms = eng.bench_igemm(1, M, K * ks, N, 1, 1, 0, ks, PV(rows, bn), 50)
        pts.append((K, ms * 1e3))
    (k0, t0), (k1, t1) = pts[2], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print(name, " ".join(f"K={k}:{t:.1f}us" for k, t in pts), f"| slope {slope*64:.2f} us per 64-wide k-tile, intercept {t1 - slope*k1:.1f} us", flush=True)
eng.close()
