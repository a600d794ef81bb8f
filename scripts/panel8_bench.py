"""Development: 8-wave (2 x 4) row-panel variants against the 16-wave ones on the DiT layer shapes at M = 2112."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native
def PV(rows, bn, wm2=False, nst=0, bk64=True): return 0x20 | (rows << 8) | (bn << 20) | ((0x80 | nst | (0x10 if bk64 else 0)) if wm2 else 0)
eng = native.Engine(precision=3, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
M = 2112
for rnd in range(2):
    for name, K, N, rows, ks, cfgs in (("ff1", 1024, 8192, 264, 1, [(False, 0, True), (True, 2, True), (True, 4, False)]),
                                       ("ff2", 4096, 1024, 132, 4, [(False, 0, True), (True, 3, True)])):
        fl = 2.0 * M * N * K
        out = []
        for wm2, nst, bk64 in cfgs:
            ms = eng.bench_igemm(1, M, K, N, 1, 1, 0, ks, PV(rows, 256, wm2, nst, bk64) | 0x40, 50)
            out.append(f"{'8w' if wm2 else '16w'}/s{nst}/bk{64 if bk64 else 32}: {ms*1e3:.1f} us {fl/ms/1e9:.0f} TF")
        print(name, " | ".join(out), flush=True)
eng.close()
