"""Time the Oobleck decode (C2 batch: 128 sequences) per call site (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import synthetic
import bench
dcfg, vcfg = synthetic.DiTConfig(depth=1), synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=0.002, skip_gain=0.02)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=0.08)
eng = bench.build_engine(0, bench.precisions()[os.environ.get("PRECISION", "fp16")][0], dcfg, vcfg, dsd, vsd)
x = torch.randn(64, 2, 64, 32, device="cuda")
for _ in range(2): eng.decode(x, 64000)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): eng.decode(x, 64000)
torch.cuda.synchronize()
print(f"decode {1e3*(time.perf_counter()-t0)/5:.2f} ms (eager)")
eng.profile_begin(); eng.decode(x, 64000); p = eng.profile_end()
for r in sorted(p["rows"], key=lambda r: -r["ms"]):
    print(f"  {r['site']:28s} {r['launches']:3d} launches {r['ms']:8.3f} ms")
