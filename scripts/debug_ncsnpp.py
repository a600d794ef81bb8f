import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ncsnpp as oncs
from tests.util import make_engine, rel_l2
cfg = oncs.NCSNppConfig(nf=32)
sd = oncs.random_ncsnpp_weights(cfg, 41)
g = torch.Generator().manual_seed(42)
B, T = 2, 8
xt = 3.0 * torch.randn((B, 2, 64, T), generator=g); mix = torch.randn((B, 1, 64, T), generator=g); t = torch.tensor([0.8, 0.1])
rec = {}
x = torch.cat((xt, mix), 1)
ref = oncs.ncsnpp_forward(sd, cfg, x, t, rec=rec)
eng = make_engine(ncfg=cfg, nsd=sd, precision=2)
out = eng.score(xt, t, mix)
print("final", rel_l2(out, ref))
hs = rec["hs"]
nhs = len(hs)
# hs[i] lives in CB[u = nhs-1-i] at channel offset cb_ch[u]
lvC = [32, 64, 64]
in_ch = 64; cb = []
u = 0
for l in (2, 1, 0):
    for k in range(3):
        cs = hs[nhs - 1 - u].shape[1]
        cb.append((in_ch, cs, l)); in_ch = lvC[l]; u += 1
for i in range(nhs):
    u = nhs - 1 - i
    ich, cs, l = cb[u]
    H, W = 64 >> l, T >> l
    buf = eng.debug_read(f"ncs_cb{u}_f", (B, H * W, ich + cs))
    got = buf[:, :, ich:].reshape(B, H, W, cs).permute(0, 3, 1, 2)
    print("hs", i, "level", l, tuple(hs[i].shape), rel_l2(got, hs[i]))
p = eng.debug_read("ncs_pyo0", (B, 64 * T, 4)).reshape(B, 64, T, 4).permute(0, 3, 1, 2)[:, :3]
print("pyramid", rel_l2(p, rec["pyramid"]))
