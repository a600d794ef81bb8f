"""Generate tests/golden/*.npz from the reference's own classes (build container only).

TEST INFRASTRUCTURE ONLY.  Run:  python -m oracle.make_golden
Requires /root/reference (imported through oracle/reference_loader.py stubs).
Every fixture holds inputs, seeds and the *reference's* outputs on small
configurations with every parameter re-randomised; weights are
regenerated from their seed by the oracle's `random_*` helpers and guarded by a
stored checksum.  The reference never travels to the GPU box; these vectors do.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch

from . import dit as odit
from . import oobleck as ovae
from . import reference_loader as rl
from . import sampler as osmp

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def checksum(sd: dict) -> np.ndarray:
    tot = sum(float(v.double().sum()) for v in sd.values())
    atot = sum(float(v.double().abs().sum()) for v in sd.values())
    return np.array([tot, atot])


def save(name: str, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def toy_score(x, t, y):
    """Closed-form stand-in score used by the sampler-only fixtures."""
    tt = t.reshape(-1, 1, 1, 1)
    return -(x - y) * 0.05 / (1 + tt) + 0.01 * torch.tanh(x)


def gen_sde_tables(ns):
    out = {}
    for N in (10, 30):
        ref = ns.OUVESDE(theta=1.5, sigma_min=0.96, sigma_max=10.0, N=N)
        ts = torch.linspace(ref.T, 0.03, N)
        std = ref._std(ts)
        _, g = ref.sde(torch.zeros(N), ts, torch.zeros(N))
        _, G = ref.discretize(torch.zeros(N), ts, torch.zeros(N))
        out[f"t_{N}"] = ts
        out[f"std_{N}"] = std
        out[f"g_{N}"] = g
        out[f"G_{N}"] = G
        out[f"stdT_{N}"] = ref._std(torch.ones(1))
    save("sde_tables", **out)


def gen_sampler_toy(ns):
    B, n, D, T, N = 2, 2, 64, 8, 10
    g = torch.Generator().manual_seed(100)
    y = torch.randn((B, 1, D, T), generator=g)
    out = {"y": y, "N": N, "eps": 0.03, "snr": 0.5, "seed": 5}
    ref_sde = ns.OUVESDE(theta=1.5, sigma_min=0.96, sigma_max=10.0, N=N)
    for c in (0, 1, 2):
        for dn in (True, False):
            torch.manual_seed(5)
            smp = ns.sdes.get_pc_sampler("reverse_diffusion", "ald", sde=ref_sde, score_fn=toy_score,
                                         y=y, eps=0.03, snr=0.5, corrector_steps=c, denoise=dn,
                                         n_spkrs=n)
            x, nfe = smp()
            out[f"x_c{c}_dn{int(dn)}"] = x
            out[f"nfe_c{c}_dn{int(dn)}"] = nfe
    # 3-speaker variant (n_spkrs must be passed explicitly)
    torch.manual_seed(6)
    smp = ns.sdes.get_pc_sampler("reverse_diffusion", "ald", sde=ref_sde, score_fn=toy_score, y=y,
                                 eps=0.03, snr=0.5, corrector_steps=1, denoise=True, n_spkrs=3)
    out["x_3spk"], _ = smp()
    save("sampler_toy", **out)


SAMPLER_VARIANTS = (  # (predictor, corrector, probability_flow, corrector_steps)
    ("euler_maruyama", "ald", False, 1), ("reverse_diffusion", "langevin", False, 2),
    ("euler_maruyama", "langevin", False, 1), ("none", "ald", False, 1), ("none", "langevin", False, 2),
    ("reverse_diffusion", "ald", True, 1), ("euler_maruyama", "ald", True, 0),
)


def gen_sampler_variants(ns):
    """The other registered predictors / correctors reachable through get_pc_sampler's names
    (predictors.py:39-77, correctors.py:35-55) and the (inert) probability_flow flag."""
    B, n, D, T, N = 2, 2, 64, 8, 10
    g = torch.Generator().manual_seed(100)
    y = torch.randn((B, 1, D, T), generator=g)
    out = {"y": y, "N": N, "eps": 0.03, "snr": 0.5, "seed": 5}
    ref_sde = ns.OUVESDE(theta=1.5, sigma_min=0.96, sigma_max=10.0, N=N)
    for pred, corr, pf, c in SAMPLER_VARIANTS:
        for dn in (True, False):
            torch.manual_seed(5)
            smp = ns.sdes.get_pc_sampler(pred, corr, sde=ref_sde, score_fn=toy_score, y=y, eps=0.03, snr=0.5,
                                         corrector_steps=c, denoise=dn, n_spkrs=n, probability_flow=pf)
            x, nfe = smp()
            key = f"{pred}_{corr}_pf{int(pf)}_c{c}_dn{int(dn)}"
            out["x_" + key] = x
            out["nfe_" + key] = nfe
    save("sampler_variants", **out)


def gen_dit(ns):
    for tag, n_src, T in (("2spk", 2, 8), ("3spk", 3, 5)):
        cfg = odit.DiTConfig(n_src=n_src, embed_dim=128, depth=2, num_heads=2)
        sd = odit.random_dit_weights(cfg, 11)
        ref = ns.DiffusionTransformer(**cfg.reference_kwargs()).eval()
        ref.load_state_dict(sd, strict=False)
        g = torch.Generator().manual_seed(12)
        B = 2
        xt = 3.0 * torch.randn((B, n_src, 64, T), generator=g)
        mix = torch.randn((B, 1, 64, T), generator=g)
        t = torch.tensor([0.9, 0.13])
        with torch.no_grad():
            out = ref(xt.flatten(1, 2), t, input_concat_cond=mix.squeeze(1)).unflatten(1, (n_src, 64))
        save(f"dit_tiny_{tag}", xt=xt, mix=mix, t=t, out=out, wsum=checksum(sd), seed=11,
             embed_dim=128, depth=2, num_heads=2, n_src=n_src)


def _ref_vae(ns, cfg):
    dec = ns.OobleckDecoder(out_channels=1, channels=cfg.channels, latent_dim=cfg.latent_dim,
                            c_mults=list(cfg.c_mults), strides=list(cfg.strides),
                            use_snake=cfg.use_snake).eval()
    enc = ns.OobleckEncoder(in_channels=1, channels=cfg.channels, latent_dim=cfg.enc_latent_dim,
                            c_mults=list(cfg.c_mults), strides=list(cfg.strides),
                            use_snake=cfg.use_snake).eval()
    return enc, dec


def tiny_vae_weights(cfg, seed):
    from ditsep_amd.synthetic import vae_weights
    return vae_weights(cfg, seed)


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def gen_vae(ns, channels=8, tag="tiny"):
    """`channels=32` ("c32") is the smallest width the HIP decoder / encoder kernels take (one 32-channel K chunk),
    so the GPU tests can compare the native VAE with reference outputs directly, not only through the oracle."""
    for snake in (False, True):
        cfg = ovae.OobleckConfig(channels=channels, use_snake=snake)
        sd = tiny_vae_weights(cfg, 21)
        enc, dec = _ref_vae(ns, cfg)
        enc.load_state_dict(_sub(sd, "encoder."))
        dec.load_state_dict(_sub(sd, "decoder."))
        g = torch.Generator().manual_seed(22)
        z = torch.randn((3, 64, 2), generator=g)
        wav_in = 0.3 * torch.randn((2, 1, 4096), generator=g)
        vn = torch.randn((2, 64, 2), generator=g)
        with torch.no_grad():
            wav = dec(z)
            e = enc(wav_in)
            torch.manual_seed(23)
            lat = ns.vae_sample(*e.chunk(2, dim=1))[0]
        torch.manual_seed(23)
        vn = torch.randn((2, 64, 2))
        save(f"vae_{tag}_{'snake' if snake else 'elu'}", z=z, wav=wav, wav_in=wav_in, enc_out=e,
             vae_noise=vn, latent=lat, wsum=checksum(sd), seed=21, channels=channels)


def gen_vae_chunked(ns):
    """AudioAutoencoder.decode_audio / encode_audio with chunked=True (autoencoders.py:596-731) on the tiny
    ELU autoencoder, no bottleneck (the stitch rule is what is pinned; sampling is tested elsewhere)."""
    cfg = ovae.OobleckConfig(channels=8, c_mults=(1, 2), strides=(2, 4))
    sd = tiny_vae_weights(cfg, 24)
    enc, dec = _ref_vae(ns, cfg)
    enc.load_state_dict(_sub(sd, "encoder."))
    dec.load_state_dict(_sub(sd, "decoder."))
    ae = ns.AudioAutoencoder(enc, dec, latent_dim=cfg.latent_dim, downsampling_ratio=cfg.hop, sample_rate=16000,
                             io_channels=1).eval()
    # encode side: no bottleneck, so the stitched tensor is the raw encoder output (mean ++ scale, 2 x latent_dim)
    ae_enc = ns.AudioAutoencoder(enc, dec, latent_dim=cfg.enc_latent_dim, downsampling_ratio=cfg.hop,
                                 sample_rate=16000, io_channels=1).eval()
    g = torch.Generator().manual_seed(25)
    z = torch.randn((2, cfg.latent_dim, 45), generator=g)
    wav = 0.3 * torch.randn((2, 1, 45 * cfg.hop), generator=g)
    out = {"z": z, "wav_in": wav, "wsum": checksum(sd), "seed": 24}
    with torch.no_grad():
        for cs, ov in ((16, 4), (16, 5), (15, 0), (45, 6), (20, 10)):
            out[f"dec_{cs}_{ov}"] = ae.decode_audio(z, chunked=True, overlap=ov, chunk_size=cs)
            out[f"enc_{cs}_{ov}"] = ae_enc.encode_audio(wav, chunked=True, overlap=ov, chunk_size=cs)
    save("vae_chunked", **out)


CHUNK_CASES = ((16, 4), (16, 5), (15, 0), (45, 6), (20, 10))


def gen_e2e(ns, channels=8, tag="tiny"):
    """encode -> PC sampler (DiT score) -> decode, composed from the reference's
    own pieces exactly as LatentDiffSep.separate does (diffsep_latent.py:471-487)."""
    vcfg = ovae.OobleckConfig(channels=channels)
    vsd = tiny_vae_weights(vcfg, 31)
    enc, dec = _ref_vae(ns, vcfg)
    enc.load_state_dict(_sub(vsd, "encoder."))
    dec.load_state_dict(_sub(vsd, "decoder."))
    dcfg = odit.DiTConfig(n_src=2, embed_dim=128, depth=2, num_heads=2)
    dsd = odit.random_dit_weights(dcfg, 32, out_gain=0.005)
    ref_dit = ns.DiffusionTransformer(**dcfg.reference_kwargs()).eval()
    ref_dit.load_state_dict(dsd, strict=False)

    def score(xt, t, mix):
        return ref_dit(xt.flatten(1, 2), t, input_concat_cond=mix.squeeze(1)).unflatten(1, (2, 64))

    B, L, N = 2, 4000, 4
    g = torch.Generator().manual_seed(33)
    mix = 0.3 * torch.randn((B, 1, L), generator=g)
    sde = ns.OUVESDE(theta=1.5, sigma_min=0.96, sigma_max=10.0, N=N)
    torch.manual_seed(34)
    with torch.no_grad():
        xin = ns.torch_utils.pad(mix, vcfg.hop)
        e = enc(xin)
        lat, _ = ns.vae_sample(*e.chunk(2, dim=1))
        y = lat.unsqueeze(1)
        smp = ns.sdes.get_pc_sampler("reverse_diffusion", "ald", sde=sde, score_fn=score, y=y,
                                     eps=0.03, snr=0.5, corrector_steps=1, denoise=True, n_spkrs=2)
        x, nfe = smp()
        wav = dec(x.reshape(B * 2, 64, -1)).reshape(B, 2, -1)[..., :L]
    save(f"e2e_{tag}", mix=mix, y=y, x=x, wav=wav, nfe=nfe, seed=34, N=N,
         wsum_vae=checksum(vsd), wsum_dit=checksum(dsd), channels=channels)


def toy_score3(x, t, y):
    """Closed-form stand-in score for the flattened-latent sampler fixtures (x [B, n, L], y [B, 1, L])."""
    tt = t.reshape(-1, 1, 1)
    return -(x - 0.5 * y) * 0.05 / (1 + tt) + 0.01 * torch.tanh(x)


MIX_VARIANTS = (  # (sde kind, predictor, corrector, corrector_steps)
    ("mix", "reverse_diffusion", "ald2", 1), ("mix", "euler_maruyama", "ald2", 2), ("mix", "none", "ald2", 1),
    ("priormix", "reverse_diffusion", "ald2", 1), ("priormix", "euler_maruyama", "ald2", 1),
)


def gen_sampler_mix(ns):
    """MixSDE / PriorMixSDE with the ald2 corrector through the reference's get_pc_sampler, on latents flattened to
    [B, 1, D*T] (the 3-D layout these classes are written for): sdes.py:182-593, correctors.py:87-121."""
    import sdes.sdes as S
    B, D, T, N = 2, 64, 8, 6
    g = torch.Generator().manual_seed(200)
    y4 = torch.randn((B, 1, D, T), generator=g)
    yf = y4.reshape(B, 1, D * T)
    out = {"y": y4, "N": N, "eps": 0.03, "snr": 0.5, "seed": 7, "d_lambda": 2.0, "sigma_min": 0.05, "sigma_max": 0.5,
           "avg_len": 50}
    for kind, pred, corr, c in MIX_VARIANTS:
        sde = (S.MixSDE(2, 2.0, 0.05, 0.5, N=N) if kind == "mix"
               else S.PriorMixSDE(2, 2.0, 0.05, 0.5, N=N, avg_len=50))
        for dn in (True, False):
            torch.manual_seed(7)
            smp = ns.sdes.get_pc_sampler(pred, corr, sde=sde, score_fn=toy_score3, y=yf, eps=0.03, snr=0.5,
                                         corrector_steps=c, denoise=dn, n_spkrs=2)
            x, nfe = smp()
            key = f"{kind}_{pred}_{corr}_c{c}_dn{int(dn)}"
            out["x_" + key] = x.reshape(B, 2, D, T)
            out["nfe_" + key] = nfe
    # PriorMixSDE with 3 sources and an even / odd averaging window
    for avg_len in (7, 8):
        sde = S.PriorMixSDE(3, 1.5, 0.05, 0.5, N=N, avg_len=avg_len)
        torch.manual_seed(8)
        smp = ns.sdes.get_pc_sampler("reverse_diffusion", "ald2", sde=sde, score_fn=toy_score3, y=yf, eps=0.03,
                                     snr=0.5, corrector_steps=1, denoise=True, n_spkrs=3)
        x, _ = smp()
        out[f"x_priormix3_avg{avg_len}"] = x.reshape(B, 3, D, T)
    save("sampler_mix", **out)


def gen_sampler_sb(ns):
    """get_sb_sampler (sde and ode types) with SBVESDE on flattened latents (sdes.py:701-779, __init__.py:284-389);
    `model` = a closed-form data estimate."""
    import sdes.sdes as S
    B, D, T, N = 2, 64, 8, 5
    g = torch.Generator().manual_seed(210)
    y4 = torch.randn((B, 1, D, T), generator=g)
    yf = y4.reshape(B, 1, D * T)

    def model(x, t, y):
        return 0.7 * x + 0.2 * y + 0.05 * torch.tanh(x) * t.reshape(-1, 1, 1)

    out = {"y": y4, "N": N, "eps": 1e-4, "seed": 9, "k": 2.6, "c": 0.4}
    for st in ("sde", "ode"):
        sde = S.SBVESDE(2.6, 0.4, N=N)
        torch.manual_seed(9)
        x, ns_ = ns.sdes.get_sb_sampler(sde, model, yf, eps=1e-4, n_steps=17, sampler_type=st,
                                        pad_dim=(slice(None), None, None))()
        out["x_" + st] = x.reshape(B, 2, D, T)
        out["n_steps_" + st] = ns_
    save("sampler_sb", **out)


def gen_state_keys(ns):
    """state_dict key order and parameters() order of the reference modules a checkpoint of this path holds
    (DiffusionTransformer incl. its LayerNorm `beta` / rotary `inv_freq` BUFFERS, Oobleck encoder / decoder with
    SnakeBeta parameters, LatentScoreModelNCSNpp) -- what ditsep_amd.checkpoint.parameter_names must reproduce to
    match torch_ema's shadow_params (diffsep_latent.py:341-392).  Names only, written as JSON."""
    import json

    class Holder(torch.nn.Module):      # LatentDiffSep's attribute order: score_model, then vae (diffsep_latent.py:39-45)
        def __init__(self, score_model, vae):
            super().__init__()
            self.score_model = score_model
            self.vae = vae

    class Vae(torch.nn.Module):         # AudioAutoencoder registers encoder before decoder (autoencoders.py)
        def __init__(self, enc, dec):
            super().__init__()
            self.encoder, self.decoder = enc, dec

    out = {}
    dcfg = odit.DiTConfig(n_src=2, embed_dim=128, depth=2, num_heads=2)
    for snake in (False, True):
        enc, dec = _ref_vae(ns, ovae.OobleckConfig(channels=8, use_snake=snake))
        m = Holder(ns.DiffusionTransformer(**dcfg.reference_kwargs()), Vae(enc, dec))
        out[f"dit_{'snake' if snake else 'elu'}"] = {
            "state_dict": list(m.state_dict().keys()),
            "parameters": [k for k, _ in m.named_parameters()],
            "score_model_parameters": ["score_model." + k for k, _ in m.score_model.named_parameters()]}
    try:
        from . import make_golden_ncsnpp
        nm = make_golden_ncsnpp.reference_model(2)
        enc, dec = _ref_vae(ns, ovae.OobleckConfig(channels=8))
        m = Holder(nm, Vae(enc, dec))
        out["ncsnpp_elu"] = {"state_dict": list(m.state_dict().keys()),
                             "parameters": [k for k, _ in m.named_parameters()],
                             "score_model_parameters": ["score_model." + k for k, _ in m.score_model.named_parameters()]}
    except (ImportError, AttributeError) as e:
        print("  (ncsnpp state keys skipped:", e, ")")
    path = os.path.join(OUT, "state_keys.json")
    with open(path, "w") as fh:
        json.dump(out, fh)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    ns = rl.load()
    print("generating golden vectors from", rl.REF_ROOT)
    gen_sde_tables(ns)
    gen_sampler_toy(ns)
    gen_sampler_variants(ns)
    gen_sampler_mix(ns)
    gen_sampler_sb(ns)
    gen_dit(ns)
    gen_vae(ns)
    gen_vae(ns, channels=32, tag="c32")
    gen_vae_chunked(ns)
    gen_e2e(ns)
    gen_e2e(ns, channels=32, tag="c32")
    gen_state_keys(ns)
    try:
        from . import make_golden_ncsnpp
        make_golden_ncsnpp.main(save, checksum)
    except ImportError:
        pass


if __name__ == "__main__":
    main()
