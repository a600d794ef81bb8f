"""Workload for rocprofv3 --kernel-trace: the C2 sampler (B = 64, hipGraph replay) for a few steps, to read the
kernel-to-kernel gaps inside a captured score call (scripts/graph_gap_parse.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ditsep_amd import native, synthetic
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=bench.DIT_OUT_GAIN, skip_gain=bench.DIT_SKIP_GAIN)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=bench.DEC_IN_GAIN)
prec = bench.precisions()[os.environ.get("PRECISION", "fp16")][0]
eng = bench.build_engine(0, prec, dcfg, vcfg, dsd, vsd)
eng.enable_graphs(True)
y = torch.randn(64, 1, 64, 32, device="cuda")
for i in range(3):
    x, _ = eng.pc_sample(y, None, N=4, corrector_steps=1, snr=0.5, t_eps=0.03, seed=42 + i)
torch.cuda.synchronize()
print("done", flush=True)
