"""Reading the reference's Lightning checkpoints (reference src/diffsep_latent.py:341-392) without the reference.

A checkpoint holds `state_dict` (parameters AND persistent buffers, `score_model.*` / `vae.*`), `ema`
(a torch_ema state: `shadow_params` = one tensor per entry of `module.parameters()`, in that order, buffers
excluded) and `trainable_vae` (whether the EMA tracks the whole module or `score_model` only).  Matching the
EMA tensors to names therefore needs to know which state_dict entries are buffers; that is decided here from
the key alone, for the module types on this path:

  * `*.inv_freq`                       RotaryEmbedding buffer          (transformer.py:109)
  * `*.num_batches_tracked`            BatchNorm counter
  * `score_model.*norm.beta`           the DiT's bias-less LayerNorm keeps `beta` as a zero BUFFER
                                        (transformer.py:188-191: pre_norm, ff_norm, cross_attend_norm)
  * `score_model.*norm.gamma` when fixed (fix_scale=True, transformer.py:183-184) cannot be told from the key;
    the DiT on this path never sets it (dit.py builds ContinuousTransformer with the defaults).
SnakeBeta's `alpha` / `beta` under `vae.*` are parameters (blocks.py:291-315).
`nn.Module.state_dict()` and `nn.Module.parameters()` walk the module tree in the same order, so the
parameters are the state_dict keys in order with the buffers removed.
"""
from __future__ import annotations

from typing import Iterable, List

BUFFER_SUFFIXES = ("inv_freq", "num_batches_tracked")


def is_buffer(key: str) -> bool:
    if key.endswith(BUFFER_SUFFIXES):
        return True
    if key.startswith("score_model.") and key.endswith("norm.beta"):
        return True
    return False


def parameter_names(state_dict_keys: Iterable[str], scope: str = "") -> List[str]:
    """Names, in `parameters()` order, of the parameters among `state_dict_keys` under `scope`
    ("" = the whole LatentDiffSep module when `trainable_vae`, else "score_model.")."""
    return [k for k in state_dict_keys if k.startswith(scope) and not is_buffer(k)]


def match_ema(shadow_params, state_dict, scope: str = ""):
    """{name: EMA tensor} for a torch_ema `shadow_params` list against the checkpoint's own state_dict.
    Matching is by position in `parameters()` order; it is checked by COUNT and by SHAPE of every pair, so a
    buffer taken for a parameter (or the reverse) is an error naming the first mismatch instead of EMA weights
    silently loaded under the wrong names."""
    names = parameter_names(state_dict.keys(), scope)
    if len(names) != len(shadow_params):
        raise ValueError(f"EMA holds {len(shadow_params)} tensors but {len(names)} parameters are named under "
                         f"'{scope or '<root>'}': buffer classification (checkpoint.is_buffer) does not fit this checkpoint")
    for i, (name, t) in enumerate(zip(names, shadow_params)):
        if tuple(t.shape) != tuple(state_dict[name].shape):
            raise ValueError(f"EMA tensor {i} has shape {tuple(t.shape)} but parameter '{name}' is "
                             f"{tuple(state_dict[name].shape)}: the EMA list does not line up with the parameter names")
    return dict(zip(names, shadow_params))
