"""Micro-benchmark of the implicit-GEMM kernel on the path's shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native

def V(bm, bn, nst, bk):
    return nst | (0x10 if bk == 64 else 0) | (bm << 8) | (bn << 20)

shapes = [  # (name, B, L, Cin, N, taps, dil, pad, ksplits)
    ("dit qkv  M2112 N3072 K1024", 1, 2112, 1024, 3072, 1, 1, 0, (1,)),
    ("dit out  M2112 N1024 K1024", 1, 2112, 1024, 1024, 1, 1, 0, (1, 2, 3)),
    ("dit ff1  M2112 N8192 K1024", 1, 2112, 1024, 8192, 1, 1, 0, (1,)),
    ("dit ff2  M2112 N1024 K4096", 1, 2112, 4096, 1024, 1, 1, 0, (1, 2, 3, 4)),
    ("dec cT 2048>1024 s8 L32 x128", 128, 33, 4096, 8192, 1, 1, 0, (1,)),
    ("dec c7 C1024 L256 x128", 128, 256, 1024, 1024, 7, 1, 3, (1,)),
    ("dec c7 C512 L2048 x128", 128, 2048, 512, 512, 7, 3, 9, (1,)),
    ("dec c7 C256 L8192 x128", 128, 8192, 256, 256, 7, 9, 27, (1,)),
    ("dec c7 C128 L32768 x64", 64, 32768, 128, 128, 7, 1, 3, (1,)),
    ("dec c1 C128 L32768 x64", 64, 32768, 128, 128, 1, 1, 0, (1,)),
]
cfgs = {2: [(128, 128, 2, 32), (256, 128, 2, 32), (128, 256, 2, 32), (256, 256, 2, 32), (256, 128, 3, 32)],
        1: [(128, 128, 3, 32), (256, 128, 3, 32), (256, 256, 3, 32), (128, 128, 2, 64), (128, 128, 3, 64),
            (256, 128, 2, 64), (128, 256, 2, 64), (256, 256, 2, 64), (256, 128, 3, 64), (128, 256, 3, 64)]}
precs = [int(x) for x in os.environ.get("PRECS", "3").split(",")]
for prec in precs:
    eng = native.Engine(precision=prec, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
    P = 2 if prec in (2, 4) else 1
    for name, B, L, Cin, N, taps, dil, pad, ks in shapes:
        flops = 2.0 * B * L * N * taps * Cin
        res = []
        ms = eng.bench_igemm(B, L, Cin, N, taps, dil, pad, 1, 1, 10)
        res.append(("v1", 1, ms))
        for (bm, bn, nst, bk) in cfgs[P]:
            for k in ks:
                ms = eng.bench_igemm(B, L, Cin, N, taps, dil, pad, k, V(bm, bn, nst, bk), 10)
                res.append((f"{bm}x{bn}s{nst}k{bk}", k, ms))
        best = min(res, key=lambda r: r[2])
        line = " ".join(f"{c}/{k}:{flops/ms/1e9:.0f}" for c, k, ms in res)
        print(f"P={prec} {name:30s} best {best[0]}/{best[1]} {best[2]*1e3:.1f}us {flops/best[2]/1e9:.0f}TF | {line}", flush=True)
    eng.close()
