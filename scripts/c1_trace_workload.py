"""Eager single-mixture DiT score calls at the C1 shape (B=1, T=16) for rocprofv3 --kernel-trace (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import synthetic
import bench
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=0.002, skip_gain=0.02)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=0.08)
eng = bench.build_engine(0, bench.precisions()["fp16"][0], dcfg, vcfg, dsd, vsd)
B, T = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 16
xt = torch.randn(B, 2, 64, T, device="cuda"); mix = torch.randn(B, 1, 64, T, device="cuda"); t = torch.full((B,), 0.5, device="cuda")
for _ in range(4): eng.score(xt, t, mix)
torch.cuda.synchronize(); print("done", flush=True)
