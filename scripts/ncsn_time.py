"""Time one NCSN++ score call at the C2 batch (B=64, T=32) (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import synthetic
from tests.util import make_engine
ncfg = synthetic.NCSNppConfig()
nsd = synthetic.random_ncsnpp_weights(ncfg, 1)
eng = make_engine(ncfg=ncfg, nsd=nsd, precision=int(os.environ.get("PREC", "3")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
xt = torch.randn(B, 2, 64, T, device="cuda"); mix = torch.randn(B, 1, 64, T, device="cuda"); t = torch.full((B,), 0.5, device="cuda")
for _ in range(3): out = eng.score(xt, t, mix)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.score(xt, t, mix)
torch.cuda.synchronize()
print(f"B={B}: ncsnpp score {1e3*(time.perf_counter()-t0)/20:.3f} ms (eager)  checksum {float(out.double().abs().sum()):.6f}")
eng.profile_begin(); eng.score(xt, t, mix); p = eng.profile_end()
for r in sorted(p["rows"], key=lambda r: -r["ms"]):
    print(f"  {r['site']:22s} {r['launches']:3d} launches {r['ms']:8.3f} ms")
