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
