import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native
def V(bm, bn, nst, bk): return nst | (0x10 if bk == 64 else 0) | (bm << 8) | (bn << 20)
def PV(rows, bn): return 0x20 | (rows << 8) | (bn << 20)
shapes = [("qkv", 2112, 1024, 3072), ("out", 2112, 1024, 1024), ("ff1", 2112, 1024, 8192), ("ff2", 2112, 4096, 1024)]
for prec in (3,):
    eng = native.Engine(precision=prec, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
    for name, M, K, N in shapes:
        fl = 2.0 * M * N * K
        res = []
        for rows in (264, 132):
            for bn in (128, 256):
                for ks in ((1,) if N > 1024 else (1, 2, 4, 8)):
                    try:
                        ms = eng.bench_igemm(1, M, K, N, 1, 1, 0, ks, PV(rows, bn), 20)
                        res.append((f"p{rows}x{bn}/k{ks}", ms))
                    except RuntimeError as ex:
                        res.append((f"p{rows}x{bn}/k{ks}:ERR", 1e9))
        keep = [r for r in res if r[1] < 1e8]
        res = keep
        for mf in (0, 0x40):
            for (bm, bn, nst, bk) in ((128, 128, 3 if prec == 3 else 2, 32), (256, 128, 3 if prec == 3 else 2, 32), (256, 256, 3 if prec == 3 else 2, 32)) + (((256, 128, 3, 64), (256, 128, 2, 64), (256, 256, 2, 64)) if prec == 3 else ()):
                for ks in ((1,) if N > 1024 else (1, 3, 4)):
                    ms = eng.bench_igemm(1, M, K, N, 1, 1, 0, ks, V(bm, bn, nst, bk) | mf, 20)
                    res.append((f"{'mf' if mf else 'nf'}{bm}x{bn}s{nst}k{bk}/k{ks}", ms))
        for rows, bn in ((264, 256),):
            for mf in (0, 0x40):
                ms = eng.bench_igemm(1, M, K, N, 1, 1, 0, 1 if N > 1024 else 8, PV(rows, bn) | mf, 20)
                res.append((f"{'mf' if mf else 'nf'}p{rows}x{bn}", ms))
        res.sort(key=lambda r: r[1])
        res = res[:12]
        print(f"P={prec} {name}: " + " ".join(f"{c}:{ms*1e3:.1f}us/{fl/ms/1e9:.0f}TF" for c, ms in res), flush=True)
    eng.close()
