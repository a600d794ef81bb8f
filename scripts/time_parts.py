"""Coarse timing of the path's stages on one GPU (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import native, synthetic

def log(*a):
    print(*a, flush=True)

prec = int(os.environ.get("PREC", "2"))
B = int(os.environ.get("B", "64"))
t0 = time.time()
dcfg = synthetic.DiTConfig(); vcfg = synthetic.OobleckConfig()
dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=0.002, skip_gain=0.02)
vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=0.08)
log("weights generated", time.time() - t0)
eng = native.Engine(precision=prec)
eng.load_state_dict(dsd, prefix="score_model."); eng.load_state_dict(vsd, prefix="vae."); eng.finalize()
torch.cuda.synchronize(); log("engine ready", time.time() - t0)
dev = torch.device("cuda")
def timeit(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    log(f"{name}: {(time.time()-t)/n*1e3:.2f} ms")
xt = torch.randn(B, 2, 64, 32, device=dev); mix = torch.randn(B, 1, 64, 32, device=dev); t = torch.full((B,), 0.5, device=dev)
timeit("score B=%d" % B, lambda: eng.score(xt, t, mix))
timeit("decode B=%d" % B, lambda: eng.decode(xt, 64000), n=2)
wav = torch.randn(B, 1, 64000, device=dev) * 0.1
timeit("encode B=%d" % B, lambda: eng.encode(wav), n=2)
timeit("pc_sample N=30 eager", lambda: eng.pc_sample(mix, None, N=30), n=1)
eng.enable_graphs(True)
eng.pc_sample(mix, None, N=30); eng.pc_sample(mix, None, N=30)
timeit("pc_sample N=30 graph", lambda: eng.pc_sample(mix, None, N=30), n=2)
log("workspace GB", eng.workspace_bytes() / 1e9)
