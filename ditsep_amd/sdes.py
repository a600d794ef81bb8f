"""Host mirror of the reference sampler library's interface (reference src/sdes/).

`get_pc_sampler(predictor_name, corrector_name, sde, score_fn, y, ...)` keeps the
reference signature and closure return (reference src/sdes/__init__.py:133-193) but
the whole predictor-corrector loop -- prior draw, N x (corrector, predictor), every
score-network call -- runs inside the HIP engine as one captured launch sequence.
Native kernels exist for the predictors reverse_diffusion / euler_maruyama / none and
the correctors ald / langevin (/ none).  There is no Python loop and no PyTorch
fallback: combinations the native path does not implement raise NotImplementedError.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .registry import Registry

PredictorRegistry = Registry("Predictor")
CorrectorRegistry = Registry("Corrector")
SDERegistry = Registry("SDE")


@SDERegistry.register("ouve")
class OUVESDE:
    """Parameter carrier + closed forms of the OU variance-exploding SDE
    (reference src/sdes/sdes.py:595-698).  The sampling arithmetic lives in the engine
    (dsn_ouve_schedule / dsn_pc_sample); these torch forms serve callers that query the SDE."""

    def __init__(self, theta, sigma_min, sigma_max, N=1000, **ignored_kwargs):
        self.theta, self.sigma_min, self.sigma_max, self.N = float(theta), float(sigma_min), float(sigma_max), int(N)
        self.logsig = float(np.log(self.sigma_max / self.sigma_min))

    @property
    def T(self):
        return 1

    def copy(self):
        return OUVESDE(self.theta, self.sigma_min, self.sigma_max, N=self.N)

    def sde(self, x, t, y):
        drift = self.theta * (y - x)
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return drift, sigma * np.sqrt(2 * self.logsig)

    def _mean(self, x0, t, y):
        e = torch.exp(-self.theta * t)
        e = e.reshape(e.shape + (1,) * (x0.ndim - e.ndim))
        return e * x0 + (1 - e) * y

    def _std(self, t):
        smin, th, ls = self.sigma_min, self.theta, self.logsig
        return torch.sqrt(smin**2 * torch.exp(-2 * th * t) * (torch.exp(2 * (th + ls) * t) - 1) * ls / (th + ls))

    def marginal_prob(self, x0, t, y):
        return self._mean(x0, t, y), self._std(t)


@PredictorRegistry.register("reverse_diffusion")
class ReverseDiffusionPredictor:
    """Marker: executed natively (reference src/sdes/predictors.py:55-66)."""


@PredictorRegistry.register("euler_maruyama")
class EulerMaruyamaPredictor:
    """Marker: executed natively (reference src/sdes/predictors.py:39-52)."""


@PredictorRegistry.register("none")
class NonePredictor:
    """Marker: executed natively as 'no predictor step' (reference src/sdes/predictors.py:69-77)."""


@CorrectorRegistry.register("ald")
class AnnealedLangevinDynamics:
    """Marker: executed natively (reference src/sdes/correctors.py:58-84)."""


@CorrectorRegistry.register("langevin")
class LangevinCorrector:
    """Marker: executed natively (reference src/sdes/correctors.py:35-55).  Its step size is a batch mean:
    results depend on which mixtures share a batch (and a rank), exactly as in the reference."""


@CorrectorRegistry.register("ald2")
class AnnealedLangevinDynamics2:
    """Registered name only: needs MixSDE / PriorMixSDE (correctors.py:93-96), which are outside this path."""


@CorrectorRegistry.register("none")
class NoneCorrector:
    """Runs as zero corrector steps.  (In the reference this name cannot be used with get_pc_sampler at all:
    NoneCorrector.update_fn returns a 1-tuple that the loop's `xt, xt_mean = ...` cannot unpack,
    correctors.py:132-133 vs __init__.py:181.)"""


def get_pc_sampler(predictor_name, corrector_name, sde, score_fn, y, true_mean=None, denoise=True, eps=3e-2,
                   snr=0.1, corrector_steps=1, probability_flow=False, intermediate=False, n_spkrs=2,
                   noise=None, seed=None, **kwargs):
    """Same arguments as the reference; `score_fn` must be a native-backed model (an object exposing
    `.engine`, e.g. ditsep_amd.LatentDiffSep).  Extra keywords: `noise` (the injected standard-normal
    draws in reference order, for bit-comparable parity runs) and `seed` (on-device Philox)."""
    PredictorRegistry.get_by_name(predictor_name)      # ValueError for unknown names, as the reference
    CorrectorRegistry.get_by_name(corrector_name)
    engine = getattr(score_fn, "engine", None)
    if engine is None:
        raise NotImplementedError("get_pc_sampler needs a native score model (object with `.engine`); "
                                  "arbitrary Python score functions are not supported (no PyTorch fallback)")
    if corrector_name == "ald2":
        raise NotImplementedError("ald2 needs MixSDE / PriorMixSDE (reference correctors.py:93-96); the native "
                                  "sampler implements the OUVE SDE")
    if not isinstance(sde, OUVESDE):
        raise NotImplementedError("native sampler implements the OUVE SDE")
    # probability_flow: accepted and without effect, as in the reference -- Predictor.__init__ keeps the flag but
    # builds its reverse SDE with sde.reverse(score_fn) (predictors.py:13-18), so the ODE branch is never reached.
    if n_spkrs != engine.n_src:
        raise ValueError(f"n_spkrs={n_spkrs} but the engine was built for {engine.n_src} sources")
    if (abs(sde.theta - engine.cfg.sde_theta) > 1e-6 or abs(sde.sigma_min - engine.cfg.sde_sigma_min) > 1e-6
            or abs(sde.sigma_max - engine.cfg.sde_sigma_max) > 1e-6):
        raise ValueError("sde parameters differ from the ones the engine was built with")
    c_steps = 0 if corrector_name == "none" else int(corrector_steps)
    corr = "ald" if corrector_name == "none" else corrector_name
    counter = {"calls": 0}

    timesteps = kwargs.pop("timesteps", None)

    def pc_sampler():
        s = seed if seed is not None else int(torch.randint(0, 2**31 - 1, (1,)).item()) + counter["calls"]
        counter["calls"] += 1
        return engine.pc_sample(y, noise, N=sde.N, corrector_steps=c_steps, snr=float(snr), t_eps=float(eps),
                                denoise=bool(denoise), seed=s, timesteps=timesteps, predictor=predictor_name,
                                corrector=corr, prior_mean=true_mean, intermediate=bool(intermediate))

    return pc_sampler


def schedule_timesteps(schedule: str, T: float, eps: float, N: int) -> torch.Tensor:
    """The N+1-point time grids of the reference's scheduled sampler (src/sdes/__init__.py:95-116)."""
    base = 10
    if schedule == "linear":
        return torch.linspace(T, eps, N + 1)
    if schedule == "log":
        return torch.logspace(math.log(T) / math.log(base), math.log(eps) / math.log(base), N + 1, base=base)
    if schedule == "revlog":
        return torch.logspace(math.log(eps) / math.log(base), math.log(T) / math.log(base), N + 1,
                              base=base).flip(dims=(0,))
    raise NotImplementedError(f"Schedule '{schedule}' does not exist")


def get_pc_scheduled_sampler(predictor_name, corrector_name, sde, score_fn, y, denoise=True, true_mean=None,
                             eps=3e-2, snr=0.1, corrector_steps=1, probability_flow=False, intermediate=False,
                             schedule="linear", **kwargs):
    """Reference signature (src/sdes/__init__.py:49-64).  Runs the same native loop on the schedule's time
    grid; the step size stays 1/N exactly as in the reference (its `dt` keyword is never picked up).
    Deviation, stated: the state has n_spkrs sources (the reference samples `y.shape`, i.e. one source,
    in this sampler, which cannot separate)."""
    ts = schedule_timesteps(schedule, sde.T, eps, sde.N)
    return get_pc_sampler(predictor_name, corrector_name, sde, score_fn, y, true_mean=true_mean, denoise=denoise,
                          eps=float(ts[sde.N - 1]), snr=snr, corrector_steps=corrector_steps,
                          probability_flow=probability_flow, intermediate=intermediate, timesteps=ts[: sde.N],
                          **kwargs)
