import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from ditsep_amd import native, synthetic
vcfg = synthetic.OobleckConfig(); vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=bench.DEC_IN_GAIN)
ncfg = synthetic.NCSNppConfig(); nsd = synthetic.random_ncsnpp_weights(ncfg, 1, out_gain=bench.NCSN_OUT_GAIN)
eng = bench.build_engine(0, native.PREC_FP16, ncfg, vcfg, nsd, vsd)
eng.enable_graphs(True)
for B in (64, 48, 8):
    L = bench.FS * bench.SECONDS
    src = synthetic.synthetic_sources(B, 2, L, bench.FS, seed=1234)
    y = eng.encode(src.sum(1, keepdim=True), seed=7)
    ref = None; bad = 0
    for i in range(int(os.environ.get("REPS", "25"))):
        x, _ = eng.pc_sample(y, None, N=30, corrector_steps=1, snr=0.5, t_eps=0.03, seed=42)
        w = eng.decode(x, L)
        if ref is None: ref = w.clone()
        elif not torch.equal(w, ref): bad += 1
    print(f"NCSN++ fp16 B={B}: {os.environ.get('REPS','25')} repetitions of the N=30 chain + decode, {bad} differing", flush=True)
