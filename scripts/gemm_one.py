import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native
def V(bm, bn, nst, bk): return nst | (0x10 if bk == 64 else 0) | (bm << 8) | (bn << 20)
def PV(rows, bn): return 0x20 | (rows << 8) | (bn << 20)
eng = native.Engine(precision=3, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
which = os.environ.get("WHICH", "t128")
var = {"t128": V(128, 128, 3, 32), "t256": V(256, 256, 2, 64), "panel": PV(264, 256)}[which]
if os.environ.get("SHAPE", "ff1") == "ff1":
    ms = eng.bench_igemm(1, 2112, 1024, 8192, 1, 1, 0, 1, var, 3)
else:
    ms = eng.bench_igemm(128, 2048, 512, 512, 7, 3, 9, 1, var, 3)
print(which, ms * 1e3, "us", flush=True)
