"""Oracle: SI-SDR with brute-force permutation-invariant assignment (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference scores with the third-party `fast_bss_eval.si_bss_eval_sources(
..., compute_permutation=True)` (reference src/evaluate_latent.py:118-136,
src/models/diffsep/losses.py:6-35); that package is not vendored nor installed
here, so this restates the published scale-invariant SDR definition
(Le Roux et al., "SDR - half-baked or well done?", 2019) and is *parity
unpinned* against fast_bss_eval.
"""
from __future__ import annotations

import itertools

import torch


def si_sdr(ref: torch.Tensor, est: torch.Tensor, eps: float = 1e-10) -> torch.Tensor:
    """ref, est [..., L] -> SI-SDR in dB [...] (no mean removal, as the
    reference calls with zero_mean=False)."""
    ref = ref.double()
    est = est.double()
    alpha = (ref * est).sum(-1, keepdim=True) / (ref.pow(2).sum(-1, keepdim=True) + eps)
    target = alpha * ref
    noise = est - target
    return 10 * torch.log10((target.pow(2).sum(-1) + eps) / (noise.pow(2).sum(-1) + eps))


def si_sdr_pit(ref: torch.Tensor, est: torch.Tensor):
    """ref, est [B, n, L] -> (best mean SI-SDR [B], best permutation [B, n])."""
    B, n, _ = ref.shape
    perms = list(itertools.permutations(range(n)))
    scores = torch.stack([si_sdr(ref, est[:, list(p)]).mean(-1) for p in perms], dim=1)
    best = scores.argmax(dim=1)
    return scores.gather(1, best[:, None])[:, 0], torch.tensor(perms)[best]


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a - b|| / ||b|| over the whole tensor (b is the reference side)."""
    return float((a.double() - b.double()).norm() / b.double().norm())
