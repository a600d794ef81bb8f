import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditsep_amd import native
def V(bm, bn, nst, bk): return nst | (0x10 if bk == 64 else 0) | (bm << 8) | (bn << 20)
eng = native.Engine(precision=3, score_kind=0, vae_has_encoder=False, vae_has_decoder=False)
for name, B, L, K, N, taps, dil, pad in (("ff1", 1, 2112, 1024, 8192, 1, 1, 0), ("c7 C512", 128, 2048, 512, 512, 7, 3, 9)):
    for cfg in ((128, 128, 3, 32), (256, 256, 2, 64)):
        ms = eng.bench_igemm(B, L, K, N, taps, dil, pad, 1, V(*cfg), 10)
        print(os.environ.get("DSN_GEMM_DBG", "0"), name, cfg, f"{ms*1e3:.1f} us {2.0*B*L*N*taps*K/ms/1e9:.0f} TF", flush=True)
