"""Oracle: DiT score network over Oobleck latent tokens (CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional restatement (explicit state_dict with the reference's key names) of
  DiffusionTransformer._forward          reference src/stable_audio_tools/models/dit.py:149-244
  ContinuousTransformer.forward          reference src/stable_audio_tools/models/transformer.py:838-899
  TransformerBlock.forward (no adaLN)    reference transformer.py:717-764
  Attention.forward (einsum branch)      reference transformer.py:435-598
  FeedForward / GLU (SwiGLU)             reference transformer.py:214-288
  LayerNorm (bias-less, eps 1e-5)        reference transformer.py:176-202
  RotaryEmbedding / apply_rotary_pos_emb reference transformer.py:92-173
  FourierFeatures                        reference src/stable_audio_tools/models/blocks.py:85-94
for the configuration the separation adapter uses: continuous_transformer,
timestep as a prepended global token, `input_concat_cond` = mixture latent,
no cross-attention, no adaLN, patch_size 1.

Adapter (SURVEY.md F2 -- the reference has no (xt, t, mix) wiring for its DiT):
  score(xt[B,n,D,T], t[B], mix[B,1,D,T]) =
      DiT(x = xt.flatten(1,2), t, input_concat_cond = mix.squeeze(1)).unflatten(1,(n,D))
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F

from ditsep_amd.synthetic import DiTConfig, dit_param_shapes, random_dit_weights  # noqa: F401


def rope_tables(seq_len: int, rot_dim: int, base: float = 10000.0):
    """cos/sin tables [seq_len, rot_dim] with the (freqs, freqs) duplication."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    pos = torch.arange(seq_len).float()
    freqs = torch.outer(pos, inv_freq)
    freqs = torch.cat((freqs, freqs), dim=-1)
    return freqs.cos(), freqs.sin()


def apply_rope(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Partial rotary on the first rot_dim features; t is [B, H, S, dh]."""
    rot = cos.shape[-1]
    tr, keep = t[..., :rot], t[..., rot:]
    half = rot // 2
    rotated = torch.cat((-tr[..., half:], tr[..., :half]), dim=-1)
    tr = tr * cos + rotated * sin
    return torch.cat((tr, keep), dim=-1)


def dit_forward(sd: dict, cfg: DiTConfig, x: torch.Tensor, t: torch.Tensor,
                concat_cond: torch.Tensor, matmul=None, ff_in=None) -> torch.Tensor:
    """x [B, io, T], t [B], concat_cond [B, latent_dim, T] -> [B, io, T].

    `matmul(a, w)` (a [..., K], w [N, K] -> [..., N]) may be overridden to
    emulate reduced-precision GEMM operands; default is exact fp32.
    `ff_in(h, layer_index)` (h [B, S, D] the residual stream -> [B, S, 2 * 4 D] = ff.0.proj(ff_norm(h)) with bias) may
    be overridden to emulate a device mode that evaluates that pair in another algebraic form (tests only).
    """
    mm = matmul or (lambda a, w: a @ w.t())
    D, H, dh = cfg.embed_dim, cfg.num_heads, cfg.dim_heads

    h = torch.cat([x, concat_cond], dim=1)                      # [B, dim_in, T]
    # timestep token
    f = 2 * math.pi * t[:, None] @ sd["timestep_features.weight"].t()
    ff = torch.cat([f.cos(), f.sin()], dim=-1)                  # [B, 256]
    te = ff @ sd["to_timestep_embed.0.weight"].t() + sd["to_timestep_embed.0.bias"]
    te = F.silu(te)
    te = te @ sd["to_timestep_embed.2.weight"].t() + sd["to_timestep_embed.2.bias"]

    h = h.transpose(1, 2)                                       # [B, T, dim_in]
    h = mm(h, sd["preprocess_conv.weight"][:, :, 0]) + h
    h = mm(h, sd["transformer.project_in.weight"])             # [B, T, D]
    h = torch.cat([te[:, None, :], h], dim=1)                   # [B, T+1, D]
    B, S, _ = h.shape
    cos, sin = rope_tables(S, cfg.rot_dim)
    scale = 1.0 / math.sqrt(dh)

    for i in range(cfg.depth):
        p = f"transformer.layers.{i}."
        a = F.layer_norm(h, (D,), sd[p + "pre_norm.gamma"], sd.get(p + "pre_norm.beta"), 1e-5)
        qkv = mm(a, sd[p + "self_attn.to_qkv.weight"])
        q, k, v = qkv.chunk(3, dim=-1)
        q, k, v = (u.reshape(B, S, H, dh).transpose(1, 2) for u in (q, k, v))
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        dots = torch.einsum("bhid,bhjd->bhij", q, k) * scale
        attn = dots.softmax(dim=-1)
        o = torch.einsum("bhij,bhjd->bhid", attn, v)
        o = o.transpose(1, 2).reshape(B, S, D)
        h = h + mm(o, sd[p + "self_attn.to_out.weight"])

        if ff_in is not None:
            u = ff_in(h, i)
        else:
            a = F.layer_norm(h, (D,), sd[p + "ff_norm.gamma"], sd.get(p + "ff_norm.beta"), 1e-5)
            u = mm(a, sd[p + "ff.ff.0.proj.weight"]) + sd[p + "ff.ff.0.proj.bias"]
        val, gate = u.chunk(2, dim=-1)
        u = val * F.silu(gate)
        h = h + mm(u, sd[p + "ff.ff.2.weight"]) + sd[p + "ff.ff.2.bias"]

    out = mm(h, sd["transformer.project_out.weight"])          # [B, S, io]
    out = out[:, 1:, :]                                          # drop timestep token
    out = mm(out, sd["postprocess_conv.weight"][:, :, 0]) + out
    return out.transpose(1, 2)


class DiTScore:
    """score_fn(xt, t, mix) adapter around dit_forward."""

    def __init__(self, sd: dict, cfg: DiTConfig, matmul=None, ff_in=None):
        self.sd, self.cfg, self.matmul, self.ff_in = sd, cfg, matmul, ff_in

    def __call__(self, xt, t, mix):
        B, n, Dl, T = xt.shape
        out = dit_forward(self.sd, self.cfg, xt.reshape(B, n * Dl, T), t,
                          mix.reshape(B, Dl, T), self.matmul, self.ff_in)
        return out.reshape(B, n, Dl, T)
