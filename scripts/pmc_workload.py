"""Small eager workload for rocprofv3 passes (--pmc FETCH_SIZE / WRITE_SIZE, --kernel-trace --stats):
PART=score -> two score calls at the C2 batch (DiT: 64 x 33 tokens; SCORE=ncsnpp: the NCSN++ U-Net on 64 x 64 x 32
latents), PART=decode -> one Oobleck decode of 128 sequences, PART=all -> both.  Every kernel shape of a bench step appears; 60 identical score calls are not needed
for bytes per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import synthetic
import bench

prec = os.environ.get("PRECISION", "fp16")
part = os.environ.get("PART", "all")
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=0.002, skip_gain=0.02)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=0.08)
if os.environ.get("SCORE", "dit") == "ncsnpp":
    from tests.util import make_engine
    ncfg = synthetic.NCSNppConfig()
    eng = make_engine(ncfg=ncfg, nsd=synthetic.random_ncsnpp_weights(ncfg, 1), precision=bench.precisions()[prec][0])
else:
    eng = bench.build_engine(0, bench.precisions()[prec][0], dcfg, vcfg, dsd, vsd)
dev = torch.device("cuda")
B = 64
xt = torch.randn(B, 2, 64, 32, device=dev); mix = torch.randn(B, 1, 64, 32, device=dev); t = torch.full((B,), 0.5, device=dev)
if part in ("score", "all"):
    for _ in range(2):
        eng.score(xt, t, mix)
if part in ("decode", "all"):
    eng.decode(xt, 64000)
torch.cuda.synchronize()
print("done", flush=True)
