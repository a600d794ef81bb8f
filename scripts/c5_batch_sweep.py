import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from ditsep_amd import native, synthetic
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=bench.DIT_OUT_GAIN, skip_gain=bench.DIT_SKIP_GAIN)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=bench.DEC_IN_GAIN)
for pname in ("fp16", "fp8"):
    eng = bench.build_engine(0, bench.precisions()[pname][0], dcfg, vcfg, dsd, vsd)
    eng.enable_graphs(True)
    for b in (8, 16, 32):
        r = bench.measure_c5(eng, dcfg, torch.device("cuda", 0), steps=2, batch=b)
        print(pname, b, r["value"], r["ms_per_step"], flush=True)
    eng.close()
