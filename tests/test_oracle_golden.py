"""Oracle (CPU restatement) vs the committed golden vectors that
oracle/make_golden.py captured from the reference's own classes.
Tolerances: fp32 round-off only (max-abs <= 2e-5 relative to O(1)-O(10) data)."""
import numpy as np
import pytest
import torch

from oracle import dit as odit
from oracle import oobleck as ovae
from oracle import pipeline, sampler
from oracle.make_golden import CHUNK_CASES, SAMPLER_VARIANTS, checksum, tiny_vae_weights, toy_score, _sub

T = torch.from_numpy


def close(a, b, tol=2e-5):
    a = a if isinstance(a, torch.Tensor) else T(np.asarray(a))
    b = b if isinstance(b, torch.Tensor) else T(np.asarray(b))
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max-abs {err} > {tol}*{scale}"


def test_sde_tables(golden):
    g = golden("sde_tables")
    for N in (10, 30):
        sde = sampler.OUVE(N=N)
        ts = torch.linspace(1, 0.03, N)
        close(ts, g[f"t_{N}"], 1e-7)
        close(sde.std(ts), g[f"std_{N}"], 1e-6)
        close(sde.diffusion(ts), g[f"g_{N}"], 1e-6)
        co = sampler.step_coefficients(sde, ts, 0.5)
        close(co["G"], g[f"G_{N}"], 1e-6)
        close(sde.std(torch.ones(1)), g[f"stdT_{N}"], 1e-6)


@pytest.mark.parametrize("c", [0, 1, 2])
@pytest.mark.parametrize("dn", [True, False])
def test_sampler_toy_bit_exact(golden, c, dn):
    g = golden("sampler_toy")
    y = T(g["y"])
    N = int(g["N"])
    noise = sampler.draw_noise(int(g["seed"]), 1 + N * (c + 1), (2, 2, 64, 8))
    x, nfe = sampler.pc_sample(toy_score, y, noise, sampler.OUVE(N=N), eps=0.03, snr=0.5,
                               corrector_steps=c, denoise=dn, n_spkrs=2)
    assert nfe == int(g[f"nfe_c{c}_dn{int(dn)}"]) == N * (c + 1)
    assert torch.equal(x, T(g[f"x_c{c}_dn{int(dn)}"]))


@pytest.mark.parametrize("pred,corr,pf,c", SAMPLER_VARIANTS)
@pytest.mark.parametrize("dn", [True, False])
def test_sampler_variants_bit_exact(golden, pred, corr, pf, c, dn):
    """Other registered predictors / correctors (euler_maruyama, none, langevin) and the reference's inert
    probability_flow flag, against vectors produced by the reference's get_pc_sampler."""
    g = golden("sampler_variants")
    N = int(g["N"])
    noise = sampler.draw_noise(int(g["seed"]), sampler.noise_draws(N, c, pred), (2, 2, 64, 8))
    x, nfe = sampler.pc_sample(toy_score, T(g["y"]), noise, sampler.OUVE(N=N), eps=0.03, snr=0.5, corrector_steps=c,
                               denoise=dn, n_spkrs=2, predictor=pred, corrector=corr, probability_flow=pf)
    key = f"{pred}_{corr}_pf{int(pf)}_c{c}_dn{int(dn)}"
    assert nfe == int(g["nfe_" + key]) == N * (c + 1)
    assert torch.equal(x, T(g["x_" + key]))


def test_sampler_three_speakers(golden):
    g = golden("sampler_toy")
    y = T(g["y"])
    noise = sampler.draw_noise(6, 1 + 10 * 2, (2, 3, 64, 8))
    x, _ = sampler.pc_sample(toy_score, y, noise, sampler.OUVE(N=10), eps=0.03, snr=0.5,
                             corrector_steps=1, denoise=True, n_spkrs=3)
    assert torch.equal(x, T(g["x_3spk"]))


@pytest.mark.parametrize("tag", ["2spk", "3spk"])
def test_dit_tiny(golden, tag):
    g = golden(f"dit_tiny_{tag}")
    cfg = odit.DiTConfig(n_src=int(g["n_src"]), embed_dim=int(g["embed_dim"]),
                         depth=int(g["depth"]), num_heads=int(g["num_heads"]))
    sd = odit.random_dit_weights(cfg, int(g["seed"]))
    np.testing.assert_allclose(checksum(sd), g["wsum"], rtol=1e-9)
    out = odit.DiTScore(sd, cfg)(T(g["xt"]), T(g["t"]), T(g["mix"]))
    close(out, g["out"])


@pytest.mark.parametrize("tag", ["tiny", "c32"])
@pytest.mark.parametrize("act", ["elu", "snake"])
def test_vae_tiny(golden, act, tag):
    g = golden(f"vae_{tag}_{act}")
    cfg = ovae.OobleckConfig(channels=int(g["channels"]), use_snake=(act == "snake"))
    sd = tiny_vae_weights(cfg, int(g["seed"]))
    np.testing.assert_allclose(checksum(sd), g["wsum"], rtol=1e-9)
    close(ovae.decoder_forward(sd, cfg, T(g["z"]), "decoder."), g["wav"])
    e = ovae.encoder_forward(sd, cfg, T(g["wav_in"]), "encoder.")
    close(e, g["enc_out"])
    close(ovae.vae_sample(e, T(g["vae_noise"])), g["latent"])


def test_pad_full_extra_hop():
    x = torch.randn(1, 1, 4096)
    assert sampler.pad_to_hop(x, 2048).shape[-1] == 6144      # reference quirk F7
    assert sampler.pad_to_hop(x[..., :4000], 2048).shape[-1] == 4096
    assert sampler.pad_to_hop(x[..., :0], 2048).shape[-1] == 2048


def test_weight_norm_fold_transposed_axis():
    g = torch.Generator().manual_seed(0)
    v = torch.randn((6, 4, 8), generator=g)
    gg = torch.rand((6, 1, 1), generator=g) + 0.5
    w = ovae.fold_weight_norm({"c.weight_g": gg, "c.weight_v": v}, "c.")
    torch.testing.assert_close(w.flatten(1).norm(dim=1), gg.flatten())


@pytest.mark.parametrize("tag", ["tiny", "c32"])
def test_e2e_tiny(golden, tag):
    g = golden(f"e2e_{tag}")
    vcfg = ovae.OobleckConfig(channels=int(g["channels"]))
    vsd = tiny_vae_weights(vcfg, 31)
    dcfg = odit.DiTConfig(n_src=2, embed_dim=128, depth=2, num_heads=2)
    dsd = odit.random_dit_weights(dcfg, 32, out_gain=0.005)
    np.testing.assert_allclose(checksum(vsd), g["wsum_vae"], rtol=1e-9)
    np.testing.assert_allclose(checksum(dsd), g["wsum_dit"], rtol=1e-9)
    mix = T(g["mix"])
    r = pipeline.separate(odit.DiTScore(dsd, dcfg), vsd, vcfg, mix, sampler.OUVE(N=int(g["N"])),
                          int(g["seed"]), n_spkrs=2, eps=0.03, snr=0.5, corrector_steps=1,
                          target_dim=mix.shape[-1])
    assert r["nfe"] == int(g["nfe"])
    close(r["y"], g["y"])
    close(r["x"], g["x"], 5e-5)
    close(r["wav"], g["wav"], 5e-5)


@pytest.mark.parametrize("tag", ["2spk", "3spk"])
def test_ncsnpp_tiny(golden, tag):
    from oracle import ncsnpp as oncs

    g = golden(f"ncsnpp_tiny_{tag}")
    cfg = oncs.NCSNppConfig(n_src=int(g["n_src"]), nf=int(g["nf"]))
    sd = oncs.random_ncsnpp_weights(cfg, int(g["seed"]))
    np.testing.assert_allclose(checksum(sd), g["wsum"], rtol=1e-9)
    out = oncs.NCSNppScore(sd, cfg)(T(g["xt"]), T(g["t"]), T(g["mix"]))     # 3spk case has W=6 -> padded to 8
    close(out, g["out"])


def test_fir_resamplers_match_upfirdn_definition():
    """Closed-form separable FIR == zero-insert / pad / correlate / decimate with [1,3,3,1]."""
    from oracle.ncsnpp import fir_down, fir_up

    x = torch.randn(2, 3, 8, 6, generator=torch.Generator().manual_seed(0))
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k2 = torch.outer(k1, k1)
    k2 = k2 / k2.sum()
    # up: zero-insert x2, pad (2,1), true convolution with 4*k2
    up = torch.zeros(2, 3, 16, 12)
    up[:, :, ::2, ::2] = x
    up = torch.nn.functional.pad(up, (2, 1, 2, 1))
    w = (4 * k2).flip(0, 1)[None, None].repeat(3, 1, 1, 1)
    ref_up = torch.nn.functional.conv2d(up, w, groups=3)
    torch.testing.assert_close(fir_up(x), ref_up, atol=1e-6, rtol=1e-6)
    dn = torch.nn.functional.pad(x, (1, 1, 1, 1))
    w = k2.flip(0, 1)[None, None].repeat(3, 1, 1, 1)
    ref_dn = torch.nn.functional.conv2d(dn, w, groups=3)[:, :, ::2, ::2]
    torch.testing.assert_close(fir_down(x), ref_dn, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("cs,ov", CHUNK_CASES)
def test_vae_chunked_stitch(golden, cs, ov):
    """AudioAutoencoder.decode_audio / encode_audio(chunked=True) of the reference (autoencoders.py:596-731)."""
    g = golden("vae_chunked")
    cfg = ovae.OobleckConfig(channels=8, c_mults=(1, 2), strides=(2, 4))
    sd = tiny_vae_weights(cfg, int(g["seed"]))
    np.testing.assert_allclose(checksum(sd), g["wsum"], rtol=1e-9)
    close(ovae.decode_chunked(sd, cfg, T(g["z"]), cs, ov), g[f"dec_{cs}_{ov}"])
    close(ovae.encode_chunked(sd, cfg, T(g["wav_in"]), cs, ov), g[f"enc_{cs}_{ov}"])


def test_vae_chunk_plan_edges():
    assert ovae.chunk_plan(45, 45, 6) == [(0, 0, 45, 0, 45)]                   # a single chunk: nothing trimmed
    assert ovae.chunk_plan(32, 16, 0) == [(0, 0, 16, 0, 16), (16, 16, 32, 0, 16)]
    plan = ovae.chunk_plan(45, 16, 4)                                          # grid 0,12,24 + a final flush chunk
    assert [p[0] for p in plan] == [0, 12, 24, 29] and plan[-1][1:] == (31, 45, 2, 16)
    covered = sorted(t for p in plan for t in range(p[1], p[2]))
    assert covered[0] == 0 and covered[-1] == 44 and set(covered) == set(range(45))
    with pytest.raises(ValueError):
        ovae.chunk_plan(10, 16, 4)                                             # shorter than one chunk


# ------------------------------------------------------------------ secondary sampler family on flattened latents
def _toy_score4(x, t, y):
    tt = t.reshape(-1, 1, 1, 1)
    return -(x - 0.5 * y) * 0.05 / (1 + tt) + 0.01 * torch.tanh(x)


def _priormix_noise(seed, draws, B, n, D, Tn):
    """The seeded draws as the REFERENCE's PriorMixSDE loop consumes them.  Its state comes out of
    einsum("bcdt,bdt->bct") laid out source-fastest in memory ([B, L, n] strides), and every later `randn_like(x)`
    inherits those strides: torch then fills the tensor through its non-contiguous normal_ path (memory order, scalar
    sampler) instead of the vectorised contiguous one -- a different value sequence from the same generator state.
    (The prior's own draw is randn_like of an expanded tensor: plain contiguous.)  The oracle and the native path
    take noise in the logical [B, n, D, T] layout; only this fixture comparison needs the emulation."""
    g = torch.Generator().manual_seed(seed)
    L = D * Tn
    out = [torch.randn((B, n, L), generator=g)]
    for _ in range(draws - 1):
        out.append(torch.empty_strided((B, n, L), (n * L, 1, n)).normal_(generator=g).contiguous())
    return torch.stack(out).reshape(draws, B, n, D, Tn)


def test_mix_sde_samplers_vs_reference(golden):
    """MixSDE / PriorMixSDE + ald2 through the reference's get_pc_sampler (on latents flattened to [B,1,D*T]) vs the
    oracle's source-mean restatement, same seeded noise stream."""
    from oracle import sampler_variants as sv
    from oracle.make_golden import MIX_VARIANTS

    g = golden("sampler_mix")
    y = T(g["y"])
    B, _, D, Tn = y.shape
    N = int(g["N"])
    for kind, pred, corr, c in MIX_VARIANTS:
        sde = sv.MixSDE(2, float(g["d_lambda"]), float(g["sigma_min"]), float(g["sigma_max"]), N=N,
                        prior_mix=(kind == "priormix"), avg_len=int(g["avg_len"]))
        for dn in (True, False):
            draws = sv.mix_noise_draws(N, c, pred)
            noise = (_priormix_noise(int(g["seed"]), draws, B, 2, D, Tn) if kind == "priormix"
                     else sampler.draw_noise(int(g["seed"]), draws, (B, 2, D, Tn)))
            x, nfe = sv.pc_sample_mix(_toy_score4, y, noise, sde, predictor=pred, corrector=corr, eps=float(g["eps"]),
                                      snr=float(g["snr"]), corrector_steps=c, denoise=dn)
            key = f"{kind}_{pred}_{corr}_c{c}_dn{int(dn)}"
            assert nfe == int(g["nfe_" + key])
            close(x, g["x_" + key], 2e-5)
    for avg_len in (7, 8):
        sde = sv.MixSDE(3, 1.5, 0.05, 0.5, N=N, prior_mix=True, avg_len=avg_len)
        noise = _priormix_noise(8, sv.mix_noise_draws(N, 1, "reverse_diffusion"), B, 3, D, Tn)
        x, _ = sv.pc_sample_mix(_toy_score4, y, noise, sde, eps=0.03, snr=0.5, corrector_steps=1)
        close(x, g[f"x_priormix3_avg{avg_len}"], 2e-5)


def test_sb_sampler_vs_reference(golden):
    from oracle import sampler_variants as sv

    g = golden("sampler_sb")
    y = T(g["y"])
    B, _, D, Tn = y.shape
    N = int(g["N"])

    def model(x, t, yy):
        return 0.7 * x + 0.2 * yy + 0.05 * torch.tanh(x) * t.reshape(-1, 1, 1, 1)

    for st in ("sde", "ode"):
        noise = sampler.draw_noise(int(g["seed"]), N, (B, 2, D, Tn))
        x, ns = sv.sb_sample(model, y, noise, sv.SBVE(float(g["k"]), float(g["c"]), N=N), eps=float(g["eps"]),
                             sampler_type=st, n_steps=17)
        assert ns == int(g["n_steps_" + st]) == 17
        close(x, g["x_" + st], 2e-5)
