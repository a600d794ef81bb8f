"""Oracle: OUVE SDE + reverse-diffusion / annealed-Langevin PC sampler (CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, as closed-form tensor algebra with *injected* noise:
  * OUVESDE.sde / _std / prior_sampling      reference src/sdes/sdes.py:595-698
  * SDE.discretize (dt is always 1/N)         reference src/sdes/sdes.py:94-108
  * RSDE.discretize                           reference src/sdes/sdes.py:165-173
  * ReverseDiffusionPredictor.update_fn       reference src/sdes/predictors.py:55-66
  * EulerMaruyamaPredictor / NonePredictor    reference src/sdes/predictors.py:39-52,69-77
  * AnnealedLangevinDynamics.update_fn        reference src/sdes/correctors.py:58-84
  * LangevinCorrector.update_fn               reference src/sdes/correctors.py:35-55
  * RSDE.sde / rsde_parts (probability flow)  reference src/sdes/sdes.py:134-163
  * get_pc_sampler -> pc_sampler()            reference src/sdes/__init__.py:133-193

Noise draw order: prior, then per step
[corrector_steps x corrector noise], predictor noise.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class OUVE:
    """Scalar parameters of the Ornstein-Uhlenbeck variance-exploding SDE."""

    theta: float = 1.5
    sigma_min: float = 0.96
    sigma_max: float = 10.0
    N: int = 30

    @property
    def logsig(self) -> float:
        # np.log of a python float ratio, as the reference does (sdes.py:641)
        return float(np.log(self.sigma_max / self.sigma_min))

    def std(self, t: torch.Tensor) -> torch.Tensor:
        """Perturbation-kernel std at time t (fp32 tensor in, fp32 out)."""
        th, ls, smin = self.theta, self.logsig, self.sigma_min
        num = smin**2 * torch.exp(-2 * th * t) * (torch.exp(2 * (th + ls) * t) - 1) * ls
        return torch.sqrt(num / (th + ls))

    def diffusion(self, t: torch.Tensor) -> torch.Tensor:
        """g(t) = sigma_min (sigma_max/sigma_min)^t sqrt(2 logsig)."""
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return sigma * np.sqrt(2 * self.logsig)


def _bcast(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return v.reshape(v.shape + (1,) * (like.ndim - v.ndim))


def step_coefficients(sde: OUVE, timesteps: torch.Tensor, snr: float):
    """Per-step scalars of the PC loop, computed exactly as the torch reference
    would on a [B]-vector of identical times (all fp32 tensor arithmetic).

    Returns dict of fp32 1-D tensors of length N:
      eps_c  = 2 (snr * std(t))^2         (corrector step size)
      cn     = sqrt(2 eps_c)              (corrector noise gain)
      fdt    = theta * dt                 (predictor drift factor, dt = 1/N)
      G      = g(t) sqrt(dt)
      G2     = G^2
    """
    t = timesteps.to(torch.float32)
    dt = 1 / sde.N
    std = sde.std(t)
    eps_c = (snr * std) ** 2 * 2
    cn = torch.sqrt(eps_c * 2)
    G = sde.diffusion(t) * torch.sqrt(torch.tensor(dt))
    return {
        "std": std,
        "eps_c": eps_c,
        "cn": cn,
        "fdt": torch.full_like(t, sde.theta * dt),
        "G": G,
        "G2": G**2,
    }


PREDICTORS = ("reverse_diffusion", "euler_maruyama", "none")
CORRECTORS = ("ald", "langevin")


def noise_draws(N: int, corrector_steps: int, predictor: str = "reverse_diffusion") -> int:
    """Standard-normal tensors the sampler consumes: prior + per step [corrector draws] + predictor draw
    (NonePredictor draws nothing, predictors.py:69-77)."""
    return 1 + N * (corrector_steps + (0 if predictor == "none" else 1))


def pc_sample(
    score_fn,
    y: torch.Tensor,
    noise: torch.Tensor,
    sde: OUVE,
    *,
    eps: float = 3e-2,
    snr: float = 0.1,
    corrector_steps: int = 1,
    denoise: bool = True,
    n_spkrs: int = 2,
    intermediate: bool = False,
    timesteps=None,
    predictor: str = "reverse_diffusion",
    corrector: str = "ald",
    probability_flow: bool = False,
):
    """Run the PC sampler with injected noise.  `timesteps` (>= N entries) replaces linspace(1, eps, N):
    the scheduled sampler of src/sdes/__init__.py:49-130 (dt stays 1/N there as well).

    predictor  'reverse_diffusion' (predictors.py:55-66) | 'euler_maruyama' (:39-52) | 'none' (:69-77)
    corrector  'ald' (correctors.py:58-84) | 'langevin' (:35-55, batch-mean norms)
    probability_flow  accepted and WITHOUT EFFECT, as in the reference: Predictor.__init__ stores the flag
               but builds its reverse SDE with `sde.reverse(score_fn)` (predictors.py:13-18), so RSDE's own
               probability_flow stays False and the ODE branch of sdes.py:140-173 is never taken.

    y      [B, 1, D, T]   mixture latent (the OU steady-state mean)
    noise  [noise_draws(N, corrector_steps, predictor), B, n_spkrs, D, T] standard normal draws
    returns (x [B, n_spkrs, D, T], nfe[, intermediates])
    """
    if predictor not in PREDICTORS or corrector not in CORRECTORS:
        raise ValueError(f"unknown predictor/corrector {predictor!r}/{corrector!r}")
    N = sde.N
    B = y.shape[0]
    shape = (B, n_spkrs) + tuple(y.shape[2:])
    assert noise.shape[0] == noise_draws(N, corrector_steps, predictor), noise.shape
    assert tuple(noise.shape[1:]) == shape
    it = iter(noise)
    dt = 1 / N
    with torch.no_grad():
        ones = torch.ones(B, dtype=y.dtype)
        std_T = sde.std(ones)
        x = y + next(it) * _bcast(std_T, y)
        x_mean = x
        timesteps = torch.linspace(1, eps, N) if timesteps is None else torch.as_tensor(timesteps, dtype=torch.float32)
        im = []
        for i in range(N):
            t = ones * timesteps[i]
            if corrector == "ald":          # annealed Langevin dynamics: step from the kernel std
                std = sde.std(t)
                for _ in range(corrector_steps):
                    grad = score_fn(x, t, y)
                    z = next(it)
                    step = (snr * std) ** 2 * 2
                    x_mean = x + _bcast(step, x) * grad
                    x = x_mean + z * _bcast(torch.sqrt(step * 2), x)
            else:                           # langevin: one step size from batch-mean norms
                for _ in range(corrector_steps):
                    grad = score_fn(x, t, y)
                    z = next(it)
                    grad_norm = torch.norm(grad.reshape(B, -1), dim=-1).mean()
                    noise_norm = torch.norm(z.reshape(B, -1), dim=-1).mean()
                    step = (snr * noise_norm / grad_norm) ** 2 * 2
                    x_mean = x + step * grad
                    x = x_mean + z * torch.sqrt(step * 2)
            if intermediate:
                im.append((x, x_mean))
            if predictor == "reverse_diffusion":
                f = sde.theta * (y - x) * dt
                G = sde.diffusion(t) * torch.sqrt(torch.tensor(dt))
                rev_f = f - _bcast(G, x) ** 2 * score_fn(x, t, y)
                z = next(it)
                x_mean = x - rev_f
                x = x_mean + _bcast(G, x) * z
            elif predictor == "euler_maruyama":
                z = next(it)
                g = sde.diffusion(t)
                total = sde.theta * (y - x) + (-_bcast(g, x) ** 2 * score_fn(x, t, y))
                x_mean = x + total * (-dt)
                x = x_mean + _bcast(g, x) * np.sqrt(dt) * z
            else:
                x_mean = x
        out = x_mean if denoise else x
    nfe = N * (corrector_steps + 1)        # the reference reports this whatever the predictor (__init__.py:186)
    if intermediate:
        return out, nfe, im
    return out, nfe


def draw_noise(seed, n_draws: int, shape) -> torch.Tensor:
    """Seeded CPU standard-normal draws in sampler order.  Drawing tensor by
    tensor keeps the stream identical to the reference's successive
    torch.randn / randn_like calls under the same torch.manual_seed.
    `seed` may be an int or a torch.Generator to continue from."""
    g = seed if isinstance(seed, torch.Generator) else torch.Generator().manual_seed(seed)
    return torch.stack([torch.randn(shape, generator=g) for _ in range(n_draws)])


def pad_to_hop(x: torch.Tensor, hop: int) -> torch.Tensor:
    """reference src/utils/torch_utils.py:11-18 -- pads a *full* extra hop when
    the length is already a multiple of hop."""
    pad_len = hop - (x.shape[-1] % hop)
    return torch.nn.functional.pad(x, (0, pad_len))
