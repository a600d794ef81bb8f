"""Oracle: the whole `LatentDiffSep.separate()` path on CPU (fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  encode  (pad -> Oobleck encoder -> VAE sample)   reference src/diffsep_latent.py:107-118
  sampler (OUVE PC, reverse_diffusion + ald)        reference src/diffsep_latent.py:471-483
  decode  (Oobleck decoder, crop to target_dim)     reference src/diffsep_latent.py:120-128

Also the synthetic workload of SURVEY.md section 8d (band-limited, amplitude-
modulated noise bursts) used by tests and bench.
"""
from __future__ import annotations

import math

import torch

from ditsep_amd.synthetic import synthetic_sources  # noqa: F401

from . import oobleck, sampler
from .sampler import OUVE


def separate(score_fn, vae_sd, vae_cfg: oobleck.OobleckConfig, mix: torch.Tensor, sde: OUVE,
             seed: int, *, n_spkrs: int = 2, eps: float = 0.03, snr: float = 0.5,
             corrector_steps: int = 1, denoise: bool = True, target_dim=None):
    """mix [B,1,L] -> dict(latent y, noise, x, wav).  One CPU generator seeded
    with `seed` supplies, in reference order, the VAE noise then the sampler
    draws."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        x_in = sampler.pad_to_hop(mix, vae_cfg.hop)
        T = x_in.shape[-1] // vae_cfg.hop
        B = mix.shape[0]
        vae_noise = torch.randn((B, vae_cfg.latent_dim, T), generator=g)
        enc = oobleck.encoder_forward(vae_sd, vae_cfg, x_in, "encoder.")
        y = oobleck.vae_sample(enc, vae_noise).unsqueeze(1)
        n_draws = 1 + sde.N * (corrector_steps + 1)
        noise = sampler.draw_noise(g, n_draws, (B, n_spkrs, vae_cfg.latent_dim, T))
        x, nfe = sampler.pc_sample(score_fn, y, noise, sde, eps=eps, snr=snr,
                                   corrector_steps=corrector_steps, denoise=denoise,
                                   n_spkrs=n_spkrs)
        wav = oobleck.decode_sources(vae_sd, vae_cfg, x, target_dim, "decoder.")
    return {"y": y, "vae_noise": vae_noise, "noise": noise, "x": x, "wav": wav, "nfe": nfe}
