"""Oracle: SI-SDR / SI-SIR / SI-SAR with brute-force permutation-invariant assignment (CPU).

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


def si_bss_eval(ref: torch.Tensor, est: torch.Tensor, perm_by: str = "sir", clamp_db: float = 100.0):
    """SI-SDR / SI-SIR / SI-SAR with the permutation solved, from the bss_eval decomposition with a one-tap
    (scale-invariant) distortion filter -- what the reference obtains from
    `fast_bss_eval.si_bss_eval_sources(ref, est, zero_mean=False, compute_permutation=True, clamp_db=100)`
    (reference src/evaluate_latent.py:118-136).  *Parity unpinned* against fast_bss_eval (not installed, not
    vendored): restated from the definitions (Vincent et al. 2006; Scheibler 2022), with the signals' explicit
    projections (least squares on the waveforms, fp64) rather than the Gram shortcut the product path takes.

    ref, est [B, n, L] -> (si_sdr, si_sir, si_sar [B, n], perm [B, n]); est[:, perm[b, i]] is matched to ref i."""
    ref, est = ref.double(), est.double()
    B, n, L = ref.shape
    tiny = 1e-300

    def db(num, den):
        v = 10 * torch.log10(num.clamp_min(tiny) / den.clamp_min(tiny))
        return v.clamp(-clamp_db, clamp_db) if clamp_db and clamp_db > 0 else v

    sdr = torch.zeros((B, n, n), dtype=torch.float64)
    sir = torch.zeros_like(sdr)
    sar = torch.zeros((B, n), dtype=torch.float64)
    for b in range(B):
        R = ref[b].T                                            # [L, n]
        coef = torch.linalg.lstsq(R, est[b].T).solution         # [n, n]: column j = coefficients of est_j
        proj = (R @ coef).T                                     # P est_j   [n, L]
        art = est[b] - proj
        sar[b] = db(proj.pow(2).sum(-1), art.pow(2).sum(-1))
        for i in range(n):
            alpha = (est[b] @ ref[b, i]) / ref[b, i].pow(2).sum().clamp_min(tiny)      # [n] over est j
            tgt = alpha[:, None] * ref[b, i][None]                                      # e_target of est_j on ref_i
            sdr[b, i] = db(tgt.pow(2).sum(-1), (est[b] - tgt).pow(2).sum(-1))
            sir[b, i] = db(tgt.pow(2).sum(-1), (proj - tgt).pow(2).sum(-1))
    perms = list(itertools.permutations(range(n)))
    key = sir if perm_by == "sir" else sdr
    idx = torch.arange(n)
    scores = torch.stack([key[:, idx, list(p)].sum(-1) for p in perms], dim=1)
    best = torch.tensor(perms)[scores.argmax(dim=1)]            # [B, n]
    bi = torch.arange(B)[:, None]
    return sdr[bi, idx[None], best], sir[bi, idx[None], best], sar[bi, best], best


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a - b|| / ||b|| over the whole tensor (b is the reference side)."""
    return float((a.double() - b.double()).norm() / b.double().norm())
