"""Development check: fp16 decode with the fused ResidualUnit v2 / v1 / unfused path on identical input
(each variant in its own process: the switches are read once), compared against the bf16x3 strict mode."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = "/tmp/ru_variant_%s.pt"
if len(sys.argv) > 1:
    import torch
    from ditsep_amd import native, synthetic
    tag, prec = sys.argv[1], int(sys.argv[2])
    vcfg = synthetic.OobleckConfig()
    vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=0.08)
    eng = native.Engine(precision=prec, n_src=2, score_kind=native.SCORE_NONE)
    eng.load_state_dict(vsd, prefix="vae."); eng.finalize()
    g = torch.Generator().manual_seed(1)
    est = torch.randn((2, 2, 64, 32), generator=g)
    torch.save(eng.decode(est).cpu(), OUT % tag)
    sys.exit(0)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = {"x3": ({}, 2), "v2": ({}, 3), "v1": ({"DSN_RU_V1": "1"}, 3), "unfused": ({"DSN_NO_FUSED_RU": "1"}, 3)}
if os.path.exists(os.path.join(ROOT, "ditsep_amd", "libditsep_dbg.so")):
    runs["v2safe"] = ({"DSN_LIB": os.path.join(ROOT, "ditsep_amd", "libditsep_dbg.so")}, 3)
for tag, (env, prec) in runs.items():
    subprocess.run([sys.executable, __file__, tag, str(prec)], env={**os.environ, **env}, check=True)
ref = torch.load(OUT % "x3").double()
for tag in [t for t in runs if t != "x3"]:
    w = torch.load(OUT % tag).double()
    print(tag, "rel_l2 vs bf16x3:", float((w - ref).norm() / ref.norm()), flush=True)
a, b = torch.load(OUT % "v2").double(), torch.load(OUT % "v1").double()
print("v2 vs v1:", float((a - b).norm() / b.norm()))
d = (a - b).abs()
rms = b.pow(2).mean().sqrt()
print("rms", float(rms), "max abs diff", float(d.max()), "frac > 1e-2 rms", float((d > 1e-2 * rms).double().mean()))
flat = d.reshape(-1, d.shape[-1])
big = (flat > 1e-2 * rms)
pos = big.any(0).nonzero().flatten()
print("n positions with big diff:", pos.numel(), "first:", pos[:20].tolist(), "last:", pos[-10:].tolist())
import collections
print("pos mod 512 histogram (top):", collections.Counter((pos % 512).tolist()).most_common(12))
print("per-sequence rel:", [(float((a.reshape(-1, a.shape[-1])[i] - b.reshape(-1, b.shape[-1])[i]).norm() / b.reshape(-1, b.shape[-1])[i].norm())) for i in range(4)])
