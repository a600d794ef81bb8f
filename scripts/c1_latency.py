"""Config C1 (WSJ0-2mix shape: 8 kHz x 4 s -> T=16 latent frames, N=10, 1 corrector step, batch 1): single-mixture
latency of the native path (sampler + decode), graphs on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ditsep_amd import native, synthetic
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=bench.DIT_OUT_GAIN, skip_gain=bench.DIT_SKIP_GAIN)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=bench.DEC_IN_GAIN)
eng = bench.build_engine(0, native.PREC_FP16, dcfg, vcfg, dsd, vsd)
eng.enable_graphs(True)
L = 32000
src = synthetic.synthetic_sources(1, 2, L, 8000, seed=1)
y = eng.encode(src.sum(1, keepdim=True), seed=3)
for B in (1, 4):
    yy = y.repeat(B, 1, 1, 1)
    def step(i):
        x, _ = eng.pc_sample(yy, None, N=10, corrector_steps=1, snr=0.5, t_eps=0.03, seed=i)
        return eng.decode(x, L)
    for i in range(3): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for i in range(n): step(10 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"C1 shape, batch {B}: {dt*1e3:.1f} ms per batch -> {B/dt:.1f} utt/s", flush=True)
