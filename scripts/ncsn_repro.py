"""Development: localise run-to-run differences of the NCSN++ score call.  Runs the same call REPS times and compares
the persistent workspace tensors (skip / concat buffers, middle, up-path outputs, pyramid heads, score) between
runs, in forward order; prints the first tensors that differ.  Environment: B, T, PREC, REPS + the engine's DSN_*
switches (DSN_NO_HALO, DSN_NO_GN_FUSE, ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditsep_amd import synthetic
from tests.util import make_engine

B, T = int(os.environ.get("B", "64")), int(os.environ.get("T", "32"))
prec, reps = int(os.environ.get("PREC", "3")), int(os.environ.get("REPS", "3"))
ncfg = synthetic.NCSNppConfig()
nsd = synthetic.random_ncsnpp_weights(ncfg, 1, out_gain=0.01)
eng = make_engine(ncfg=ncfg, nsd=nsd, precision=prec)
g = torch.Generator().manual_seed(144)
xt = 3.0 * torch.randn((B, 2, 64, T), generator=g)
mix = torch.randn((B, 1, 64, T), generator=g)
t = torch.linspace(0.97, 0.03, B)

nf, mult, nrb, H0 = ncfg.nf, ncfg.ch_mult, ncfg.num_res_blocks, ncfg.image_size
levels = len(mult)
lv = [(H0 >> l, T >> l, nf * mult[l]) for l in range(levels)]
hs = [(0, nf)]
for l in range(levels):
    hs += [(l, lv[l][2])] * nrb
    if l != levels - 1:
        hs.append((l + 1, lv[l][2]))
names = [("ncs_xin_f", B * H0 * T * 32)]
u, in_ch, cbs = 0, lv[-1][2], []
for l in range(levels - 1, -1, -1):
    for k in range(nrb + 1):
        cs = hs[len(hs) - 1 - u][1]
        cbs.append((f"ncs_cb{u}_f", B * lv[l][0] * lv[l][1] * (in_ch + cs), l))
        in_ch = lv[l][2]
        u += 1
# forward order: down path writes the skip halves of cb8 .. cb0, then hm, then the up path cb0 .. cb8
order = [(n, c) for n, c, _ in reversed(cbs)] + [("ncs_hm_f", B * lv[-1][0] * lv[-1][1] * lv[-1][2])]
order += [(f"ncs_hup{l}_f", B * lv[l][0] * lv[l][1] * lv[l][2]) for l in range(levels - 1, -1, -1)]
order += [("sc", B * T * 2 * 64)]
names += order
slot_floats = B * ((H0 * T + 63) // 64) * (2 * nf * max(mult) // 4) * 2
names.append(("ncs_stats", 96 * slot_floats))

snaps = []
for r in range(reps):
    out = eng.score(xt, t, mix)
    torch.cuda.synchronize()
    snap = {"out": out.cpu()}
    for n, c in names:
        try:
            snap[n] = eng.debug_read(n, (c,))
        except RuntimeError as e:
            snap[n] = None
            if r == 0:
                print("skip", n, str(e)[:80])
    snaps.append(snap)
for r in range(1, reps):
    print(f"--- run {r} vs run 0")
    for n in ["out"] + [n for n, _ in names]:
        a, b = snaps[0][n], snaps[r][n]
        if a is None:
            continue
        ne = (a != b)
        if ne.any():
            idx = ne.flatten().nonzero().flatten()
            d = (a.double() - b.double()).abs().max().item()
            print(f"  DIFF {n}: {int(ne.sum())} of {a.numel()} elements, max abs {d:.3e}, first flat idx {int(idx[0])}, last {int(idx[-1])}")
        else:
            print(f"  same {n}")
a, b = snaps[0]["ncs_stats"], snaps[1]["ncs_stats"]
if a is not None:
    a, b = a.view(96, -1), b.view(96, -1)
    for sl in range(96):
        ne = (a[sl] != b[sl]) & ~(torch.isnan(a[sl]) & torch.isnan(b[sl]))
        if ne.any():
            idx = ne.nonzero().flatten()
            print(f"slot {sl}: {int(ne.sum())} floats differ; first {idx[:12].tolist()} last {int(idx[-1])}; "
                  f"vals run0 {a[sl][idx[:4]].tolist()} run1 {b[sl][idx[:4]].tolist()}")
# slot 0 = partials of conv_in's output (the skip half, channels [128, 256) of cb8 at level 0): recompute from the tensor
for r in range(min(reps, 2)):
    cb = snaps[r]["ncs_cb8_f"].view(B, H0 * T, 2 * nf)[:, :, nf:].double()
    if r == 1:
        o0 = snaps[0]["ncs_cb8_f"].view(B, H0 * T, 2 * nf)[:, :, nf:]
        o1 = snaps[1]["ncs_cb8_f"].view(B, H0 * T, 2 * nf)[:, :, nf:]
        print("conv_in output run0 vs run1: differing elements", int((o0 != o1).sum()))
    x = cb.view(B, H0 * T // 64, 64, nf // 4, 4)
    mean = x.mean(dim=(2, 4))
    m2 = ((x - mean[:, :, None, :, None]) ** 2).sum(dim=(2, 4))
    st = snaps[r]["ncs_stats"].view(96, -1)[0][: B * (H0 * T // 64) * (nf // 4) * 2].view(B, H0 * T // 64, nf // 4, 2).double()
    bad = ((st[..., 0] - mean).abs() > 1e-4 + 1e-3 * mean.abs()) | ((st[..., 1] - m2).abs() > 1e-3 * m2)
    idx = bad.nonzero()
    print(f"run {r}: slot-0 partials that do not match the tensor they describe: {int(bad.sum())} of {bad.numel()}; "
          f"first (b, slice, quad): {idx[:16].tolist()}")
    if len(idx):
        import collections
        print("   by quad:", sorted(collections.Counter(idx[:, 2].tolist()).items()))
        print("   by slice%4:", sorted(collections.Counter((idx[:, 1] % 4).tolist()).items()))
        def chan(a, b2):
            na, ma, m2a = a
            nb, mb, m2b = b2
            n = na + nb
            d_ = mb - ma
            return (n, ma + d_ * nb / n, m2a + m2b + d_ * d_ * na * nb / n)
        for b_, s_, q_ in idx[:6].tolist():
            print("   (b, slice, quad)", (b_, s_, q_), "stored", st[b_, s_, q_].tolist(), "true", [float(mean[b_, s_, q_]), float(m2[b_, s_, q_])])
            # per row-lane values: lane r covers rows r, r+16, r+32, r+48 (sub-tiles tm) x 4 channels
            xx = x[b_, s_, :, q_, :].view(4, 16, 4)                      # [tm][lane][ch]
            lane = [(16, float(xx[:, r].mean()), float(((xx[:, r] - xx[:, r].mean()) ** 2).sum())) for r in range(16)]
            lv_ = lane
            for o in (8, 4, 2, 1):
                lv_ = [chan(lv_[r], lv_[r ^ o]) for r in range(16)]
                print(f"      after xor {o}: lane 0 ->", [round(v, 5) for v in lv_[0][1:]])
            # does the stored pair equal the TRUE pair of some other (item, slice, quad)?  (a misdirected store)
            dist = (mean - st[b_, s_, q_, 0]).abs() / (mean.abs() + 1e-3) + (m2 - st[b_, s_, q_, 1]).abs() / m2
            am = int(dist.argmin())
            bb, ss, qq = am // (dist.shape[1] * dist.shape[2]), (am // dist.shape[2]) % dist.shape[1], am % dist.shape[2]
            print(f"      closest true pair anywhere: (b, slice, quad) = {(bb, ss, qq)} dist {float(dist.flatten()[am]):.3e} value", [float(mean[bb, ss, qq]), float(m2[bb, ss, qq])])
            # the neighbouring quads' true values (a value stored to the wrong place / taken from the wrong column sub-tile)
            for dq in (-4, -1, 1, 4):
                if 0 <= q_ + dq < nf // 4:
                    print(f"      true of quad {q_ + dq}:", [float(mean[b_, s_, q_ + dq]), float(m2[b_, s_, q_ + dq])])
print("done", flush=True)
