"""Shared helpers for the GPU parity tests: build a native Engine from the
oracle's seeded weights (reference key names)."""
import torch

from oracle import dit as odit
from oracle import oobleck as ovae
from oracle.make_golden import tiny_vae_weights


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


def make_engine(dcfg=None, dsd=None, vcfg=None, vsd=None, precision=2, ncfg=None, nsd=None, **kw):
    from ditsep_amd import native

    args = dict(precision=precision, score_kind=native.SCORE_NONE, vae_has_encoder=False, vae_has_decoder=False)
    if dcfg is not None:
        args.update(score_kind=native.SCORE_DIT, n_src=dcfg.n_src, latent_dim=dcfg.latent_dim,
                    dit_embed_dim=dcfg.embed_dim, dit_depth=dcfg.depth, dit_heads=dcfg.num_heads)
    if ncfg is not None:
        args.update(score_kind=native.SCORE_NCSNPP, n_src=ncfg.n_src, latent_dim=ncfg.image_size,
                    ncsn_nf=ncfg.nf, ncsn_ch_mult=ncfg.ch_mult, ncsn_num_res_blocks=ncfg.num_res_blocks,
                    ncsn_attn_resolution=ncfg.attn_resolutions[0], ncsn_image_size=ncfg.image_size,
                    ncsn_max_latent_length=ncfg.max_latent_length)
    if vcfg is not None:
        args.update(vae_channels=vcfg.channels, vae_c_mults=vcfg.c_mults, vae_strides=vcfg.strides,
                    vae_enc_latent_dim=vcfg.enc_latent_dim, vae_use_snake=vcfg.use_snake,
                    vae_final_tanh=vcfg.final_tanh, latent_dim=vcfg.latent_dim,
                    vae_has_encoder=any(k.startswith("encoder.") for k in vsd),
                    vae_has_decoder=any(k.startswith("decoder.") for k in vsd))
    args.update(kw)
    eng = native.Engine(**args)
    if dsd is not None:
        eng.load_state_dict(dsd, prefix="score_model.")
    if nsd is not None:
        eng.load_state_dict(nsd, prefix="score_model.")
    if vsd is not None:
        eng.load_state_dict(vsd, prefix="vae.")
    eng.finalize()
    return eng
