"""Small eager NCSN++ workload for rocprofv3 --kernel-trace: 3 score calls at the C2 batch (B=64, T=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import native, synthetic
from tests.util import make_engine
ncfg = synthetic.NCSNppConfig()
nsd = synthetic.random_ncsnpp_weights(ncfg, 1)
eng = make_engine(ncfg=ncfg, nsd=nsd, precision=int(os.environ.get("PREC", "3")))
dev = torch.device("cuda"); B = int(os.environ.get("B", "64")); T = int(os.environ.get("T", "32"))
xt = torch.randn(B, 2, 64, T, device=dev); mix = torch.randn(B, 1, 64, T, device=dev); t = torch.full((B,), 0.5, device=dev)
for _ in range(3):
    eng.score(xt, t, mix)
torch.cuda.synchronize()
print("done", flush=True)
