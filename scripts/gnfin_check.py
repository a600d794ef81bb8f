"""Development: NCSN++ score call with the producer-finished GroupNorm against the separate gn_apply pass
(DSN_NO_GN_FIN=1): the outputs must be bit-identical; prints both timings."""
import os, sys, subprocess
import torch
code = r'''
import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
from ditsep_amd import synthetic
from tests.util import make_engine
ncfg = synthetic.NCSNppConfig()
nsd = synthetic.random_ncsnpp_weights(ncfg, 1)
eng = make_engine(ncfg=ncfg, nsd=nsd, precision=3)
B = int(sys.argv[2])
g = torch.Generator().manual_seed(5)
xt = torch.randn(B, 2, 64, 32, generator=g).cuda(); mix = torch.randn(B, 1, 64, 32, generator=g).cuda(); t = torch.linspace(0.9, 0.1, B).cuda()
o = eng.score(xt, t, mix); o2 = eng.score(xt, t, mix)
torch.cuda.synchronize()
print("repeat equal", torch.equal(o, o2), "finite", bool(torch.isfinite(o).all()))
torch.save(o.cpu(), sys.argv[1])
for _ in range(3): eng.score(xt, t, mix)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.score(xt, t, mix)
torch.cuda.synchronize(); print("score ms %.3f" % (1e3 * (time.perf_counter() - t0) / 20))
'''
for B in (64, 8):
    outs = []
    for nofin in (0, 1):
        env = dict(os.environ)
        if nofin: env["DSN_NO_GN_FIN"] = "1"
        out = "/tmp/gnfin_%d.pt" % nofin
        r = subprocess.run([sys.executable, "-c", code, out, str(B)], env=env, capture_output=True, text=True, timeout=300)
        print(f"B={B}", "separate pass" if nofin else "fused", r.stdout.strip().replace("\n", " | "), r.stderr[-400:] if r.returncode else "", flush=True)
        outs.append(torch.load(out))
    print(f"B={B} bitwise equal:", torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max()), flush=True)
