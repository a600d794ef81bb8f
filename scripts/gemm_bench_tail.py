"""Tile sweep for the HBM-bound decoder-tail GEMMs (ConvTranspose phase GEMMs and the 256-channel 1x1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native
def V(bm, bn, nst, bk): return nst | (0x10 if bk == 64 else 0) | (bm << 8) | (bn << 20)
shapes = [  # (name, B, L, Cin, N, taps, dil, pad)
    ("cT 512>256 s4 L2048", 128, 2049, 512, 1024, 2, 1, 0),
    ("cT 256>128 s4 L8192", 128, 8193, 256, 512, 2, 1, 0),
    ("cT 128>128 s2 L32768", 128, 32769, 128, 256, 2, 1, 0),
    ("c1 C256 L8192", 128, 8192, 256, 256, 1, 1, 0),
    ("c7 C256 L8192", 128, 8192, 256, 256, 7, 9, 27),
    ("c1 C512 L2048", 128, 2048, 512, 512, 1, 1, 0),
]
cfgs = [(128, 128, 3, 32), (256, 128, 3, 32), (256, 256, 3, 32), (128, 128, 2, 64), (256, 128, 2, 64), (256, 256, 2, 64),
        (128, 256, 2, 64)]
eng = native.Engine(precision=3, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
for name, B, L, Cin, N, taps, dil, pad in shapes:
    flops = 2.0 * B * L * N * taps * Cin
    res = []
    for (bm, bn, nst, bk) in cfgs:
        ms = eng.bench_igemm(B, L, Cin, N, taps, dil, pad, 1, V(bm, bn, nst, bk), 5)
        res.append((f"{bm}x{bn}s{nst}k{bk}", ms))
    best = min(res, key=lambda r: r[1])
    print(f"{name:24s} best {best[0]} {best[1]*1e3:.0f}us | " + " ".join(f"{c}:{ms*1e3:.0f}" for c, ms in res), flush=True)
