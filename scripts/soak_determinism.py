"""Soak: the full-size C2 step (sampler + decode, B=64, graphs on) repeated with identical inputs must be
bit-identical every time on the DiT path (no atomics anywhere in it) -- a cheap detector for intra-kernel races."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ditsep_amd import native, synthetic
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=bench.DIT_OUT_GAIN, skip_gain=bench.DIT_SKIP_GAIN)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=bench.DEC_IN_GAIN)
reps = int(os.environ.get("REPS", "12"))
ncfg = synthetic.NCSNppConfig()
nsd = synthetic.random_ncsnpp_weights(ncfg, 1, out_gain=bench.NCSN_OUT_GAIN)
# (score config, weights, precision): the DiT path in the headline and the strict mode, the fp8 mode, and the NCSN++ path
# (round 3: its GroupNorm epilogue partials were run-to-run different at this batch before the rewrite)
for cfg, sd, prec in ((dcfg, dsd, native.PREC_FP16), (dcfg, dsd, native.PREC_BF16X3), (dcfg, dsd, native.PREC_FP8),
                      (ncfg, nsd, native.PREC_FP16)):
    eng = bench.build_engine(0, prec, cfg, vcfg, sd, vsd)
    eng.enable_graphs(True)
    B, L = 64, bench.FS * bench.SECONDS
    src = synthetic.synthetic_sources(B, 2, L, bench.FS, seed=1234)
    y = eng.encode(src.sum(1, keepdim=True), seed=7)
    ref = None
    bad = 0
    for i in range(reps):
        x, _ = eng.pc_sample(y, None, N=30, corrector_steps=1, snr=0.5, t_eps=0.03, seed=42)
        w = eng.decode(x, L)
        if ref is None:
            ref = w.clone()
        elif not torch.equal(w, ref):
            bad += 1
            print(f"  prec {prec} rep {i}: differs, rel {float((w - ref).norm() / ref.norm()):.3e}", flush=True)
    print(f"{type(cfg).__name__} precision {prec}: {reps} repetitions, {bad} differing", flush=True)
    eng.close()
