"""Time one DiT score call at a given (B, T) per call site (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import synthetic
import bench
B, T = int(sys.argv[1]), int(sys.argv[2])
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=0.002, skip_gain=0.02)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=0.08)
eng = bench.build_engine(0, bench.precisions()[os.environ.get("PRECISION", "fp16")][0], dcfg, vcfg, dsd, vsd)
xt = torch.randn(B, 2, 64, T, device="cuda"); mix = torch.randn(B, 1, 64, T, device="cuda"); t = torch.full((B,), 0.5, device="cuda")
for _ in range(3): eng.score(xt, t, mix)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.score(xt, t, mix)
torch.cuda.synchronize()
print(f"B={B} T={T}: score {1e3*(time.perf_counter()-t0)/20:.3f} ms (eager)")
eng.profile_begin(); eng.score(xt, t, mix); p = eng.profile_end()
for r in sorted(p["rows"], key=lambda r: -r["ms"]):
    print(f"  {r['site']:22s} {r['launches']:3d} launches {1e3*r['ms']/r['launches']:8.2f} us avg")
