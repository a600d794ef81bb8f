"""Tile / split-K sweep for the NCSN++ 3x3 conv GEMM shapes (run as 9-tap 1-D convs: same K structure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native

def V(bm, bn, nst, bk):
    return nst | (0x10 if bk == 64 else 0) | (bm << 8) | (bn << 20)

shapes = [  # (name, B, L, Cin, N, taps, dil, pad, ksplits)
    ("L0 M131072 N128 C128", 64, 2048, 128, 128, 9, 1, 4, (1,)),
    ("L0 M131072 N128 C256cat", 64, 2048, 256, 128, 9, 1, 4, (1,)),
    ("L1 M32768 N256 C128", 64, 512, 128, 256, 9, 1, 4, (1, 2)),
    ("L1 M32768 N256 C256", 64, 512, 256, 256, 9, 1, 4, (1, 2)),
    ("L1 M32768 N256 C512cat", 64, 512, 512, 256, 9, 1, 4, (1, 2)),
    ("L2 M8192 N256 C256", 64, 128, 256, 256, 9, 1, 4, (1, 2, 4)),
    ("L2 M8192 N256 C512cat", 64, 128, 512, 256, 9, 1, 4, (1, 2, 4)),
]
cfgs = {2: [(128, 128, 2, 32), (256, 128, 2, 32), (256, 256, 2, 32), (256, 128, 3, 32)],
        1: [(128, 128, 3, 32), (256, 128, 3, 32), (128, 128, 3, 64), (256, 128, 2, 64), (256, 128, 3, 64),
            (256, 256, 2, 64)]}
precs = [int(x) for x in os.environ.get("PRECS", "3,2").split(",")]
for prec in precs:
    eng = native.Engine(precision=prec, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
    P = 2 if prec in (2, 4) else 1
    for name, B, L, Cin, N, taps, dil, pad, ks in shapes:
        flops = 2.0 * B * L * N * taps * Cin
        res = []
        for (bm, bn, nst, bk) in cfgs[P]:
            for k in ks:
                ms = eng.bench_igemm(B, L, Cin, N, taps, dil, pad, k, V(bm, bn, nst, bk), 10)
                res.append((f"{bm}x{bn}s{nst}k{bk}", k, ms))
        best = min(res, key=lambda r: r[2])
        line = " ".join(f"{c}/{k}:{ms*1e3:.0f}" for c, k, ms in res)
        print(f"P={prec} {name:26s} best {best[0]}/{best[1]} {best[2]*1e3:.1f}us {flops/best[2]/1e9:.0f}TF | us: {line}", flush=True)
    eng.close()
