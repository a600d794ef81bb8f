"""Oracle: Oobleck VAE decoder / encoder / bottleneck (CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional restatement over an explicit state_dict (reference key names,
old-style weight-norm `weight_g` / `weight_v`) of
  ResidualUnit                reference src/stable_audio_tools/models/autoencoders.py:59-82
  EncoderBlock / DecoderBlock reference autoencoders.py:229-279
  OobleckEncoder / Decoder    reference autoencoders.py:281-356
  get_activation / SnakeBeta  reference autoencoders.py:33-46, blocks.py:291-329
  VAEBottleneck / vae_sample  reference src/stable_audio_tools/models/bottleneck.py:57-86
  LatentDiffSep.encode/decode reference src/diffsep_latent.py:107-128
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from .sampler import pad_to_hop


@dataclass
class OobleckConfig:
    """Defaults = src/stable_audio_tools/configs/model_configs/autoencoders/oobleck_finetune.json"""

    io_channels: int = 1
    channels: int = 128
    c_mults: tuple = (1, 2, 4, 8, 16)
    strides: tuple = (2, 4, 4, 8, 8)
    latent_dim: int = 64          # decoder input / bottleneck output
    enc_latent_dim: int = 128     # encoder output (mean ++ scale)
    use_snake: bool = False
    final_tanh: bool = True

    @property
    def hop(self) -> int:
        return int(math.prod(self.strides))

    @property
    def mults(self):
        return (1,) + tuple(self.c_mults)


def fold_weight_norm(sd: dict, prefix: str) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (old torch weight_norm, dim=0;
    for ConvTranspose1d dim 0 is the IN-channel axis)."""
    if prefix + "weight" in sd:
        return sd[prefix + "weight"]
    g, v = sd[prefix + "weight_g"], sd[prefix + "weight_v"]
    norm = v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.ndim - 1)))
    return v * (g / norm)


def _act(sd: dict, prefix: str, x: torch.Tensor, snake: bool) -> torch.Tensor:
    if not snake:
        return F.elu(x)
    alpha = torch.exp(sd[prefix + "alpha"])[None, :, None]
    beta = torch.exp(sd[prefix + "beta"])[None, :, None]
    return x + (1.0 / (beta + 1e-9)) * torch.sin(x * alpha) ** 2


def _conv(sd, prefix, x, **kw):
    return F.conv1d(x, fold_weight_norm(sd, prefix), sd.get(prefix + "bias"), **kw)


def _res_unit(sd, prefix, x, dilation, snake):
    h = _act(sd, prefix + "layers.0.", x, snake)
    h = _conv(sd, prefix + "layers.1.", h, dilation=dilation, padding=3 * dilation)
    h = _act(sd, prefix + "layers.2.", h, snake)
    h = _conv(sd, prefix + "layers.3.", h)
    return x + h


def decoder_forward(sd: dict, cfg: OobleckConfig, z: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """z [S, latent_dim, T] -> wav [S, io_channels, hop*T]."""
    snake = cfg.use_snake
    m = cfg.mults
    x = _conv(sd, prefix + "layers.0.", z, padding=3)
    li = 1
    for i in range(len(m) - 1, 0, -1):
        p = f"{prefix}layers.{li}."
        s = cfg.strides[i - 1]
        x = _act(sd, p + "layers.0.", x, snake)
        w = fold_weight_norm(sd, p + "layers.1.")
        x = F.conv_transpose1d(x, w, sd.get(p + "layers.1.bias"), stride=s, padding=math.ceil(s / 2))
        for j, d in enumerate((1, 3, 9)):
            x = _res_unit(sd, f"{p}layers.{2 + j}.", x, d, snake)
        li += 1
    x = _act(sd, f"{prefix}layers.{li}.", x, snake)
    x = _conv(sd, f"{prefix}layers.{li + 1}.", x, padding=3)
    return torch.tanh(x) if cfg.final_tanh else x


def encoder_forward(sd: dict, cfg: OobleckConfig, wav: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """wav [S, io_channels, L] (L multiple of hop) -> [S, enc_latent_dim, L/hop]."""
    snake = cfg.use_snake
    m = cfg.mults
    x = _conv(sd, prefix + "layers.0.", wav, padding=3)
    li = 1
    for i in range(len(m) - 1):
        p = f"{prefix}layers.{li}."
        s = cfg.strides[i]
        for j, d in enumerate((1, 3, 9)):
            x = _res_unit(sd, f"{p}layers.{j}.", x, d, snake)
        x = _act(sd, p + "layers.3.", x, snake)
        x = _conv(sd, p + "layers.4.", x, stride=s, padding=math.ceil(s / 2))
        li += 1
    x = _act(sd, f"{prefix}layers.{li}.", x, snake)
    return _conv(sd, f"{prefix}layers.{li + 1}.", x, padding=1)


def vae_sample(enc_out: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """mean + noise * (softplus(scale) + 1e-4); bottleneck.py:57-65 (the
    reference samples even in eval mode)."""
    mean, scale = enc_out.chunk(2, dim=1)
    stdev = F.softplus(scale) + 1e-4
    return noise * stdev + mean


def encode_mix(sd, cfg: OobleckConfig, mix: torch.Tensor, noise: torch.Tensor, prefix="encoder."):
    """LatentDiffSep.encode for the mixture: pad -> encoder -> vae sample -> [B,1,D,T]."""
    x = pad_to_hop(mix, cfg.hop)
    lat = vae_sample(encoder_forward(sd, cfg, x, prefix), noise)
    return lat.unsqueeze(1)


def decode_sources(sd, cfg: OobleckConfig, est: torch.Tensor, target_dim=None, prefix="decoder."):
    """LatentDiffSep.decode: [B,n,D,T] -> [B,n,L] cropped to target_dim."""
    B, n, D, T = est.shape
    wav = decoder_forward(sd, cfg, est.reshape(B * n, D, T), prefix).reshape(B, n, -1)
    return wav if target_dim is None else wav[..., :target_dim]


# ---------------------------------------------------------------------------
# parameter inventory + seeded re-randomisation (reference key names)
# ---------------------------------------------------------------------------

def _wn(shapes, prefix, w_shape, bias=True, transposed=False):
    shapes[prefix + "weight_g"] = (w_shape[0],) + (1,) * (len(w_shape) - 1)
    shapes[prefix + "weight_v"] = tuple(w_shape)
    if bias:
        shapes[prefix + "bias"] = (w_shape[1] if transposed else w_shape[0],)


def _act_shapes(shapes, prefix, ch, snake):
    if snake:
        shapes[prefix + "alpha"] = (ch,)
        shapes[prefix + "beta"] = (ch,)


def _ru_shapes(shapes, prefix, ch, snake):
    _act_shapes(shapes, prefix + "layers.0.", ch, snake)
    _wn(shapes, prefix + "layers.1.", (ch, ch, 7))
    _act_shapes(shapes, prefix + "layers.2.", ch, snake)
    _wn(shapes, prefix + "layers.3.", (ch, ch, 1))


def decoder_param_shapes(cfg: OobleckConfig, prefix: str = "") -> dict:
    m, ch, snake = cfg.mults, cfg.channels, cfg.use_snake
    s = {}
    _wn(s, prefix + "layers.0.", (m[-1] * ch, cfg.latent_dim, 7))
    li = 1
    for i in range(len(m) - 1, 0, -1):
        p = f"{prefix}layers.{li}."
        cin, cout, st = m[i] * ch, m[i - 1] * ch, cfg.strides[i - 1]
        _act_shapes(s, p + "layers.0.", cin, snake)
        _wn(s, p + "layers.1.", (cin, cout, 2 * st), transposed=True)
        for j in range(3):
            _ru_shapes(s, f"{p}layers.{2 + j}.", cout, snake)
        li += 1
    _act_shapes(s, f"{prefix}layers.{li}.", m[0] * ch, snake)
    _wn(s, f"{prefix}layers.{li + 1}.", (cfg.io_channels, m[0] * ch, 7), bias=False)
    return s


def encoder_param_shapes(cfg: OobleckConfig, prefix: str = "") -> dict:
    m, ch, snake = cfg.mults, cfg.channels, cfg.use_snake
    s = {}
    _wn(s, prefix + "layers.0.", (m[0] * ch, cfg.io_channels, 7))
    li = 1
    for i in range(len(m) - 1):
        p = f"{prefix}layers.{li}."
        cin, cout, st = m[i] * ch, m[i + 1] * ch, cfg.strides[i]
        for j in range(3):
            _ru_shapes(s, f"{p}layers.{j}.", cin, snake)
        _act_shapes(s, p + "layers.3.", cin, snake)
        _wn(s, p + "layers.4.", (cout, cin, 2 * st))
        li += 1
    _act_shapes(s, f"{prefix}layers.{li}.", m[-1] * ch, snake)
    _wn(s, f"{prefix}layers.{li + 1}.", (cfg.enc_latent_dim, m[-1] * ch, 3))
    return s


def random_weights(shapes: dict, seed: int, res_gain: float = 0.3) -> dict:
    """Seeded fill for weight-normed conv stacks: v ~ N(0,1); g chosen so the
    folded weight has per-output-row norm ~ sqrt(2*fan_out_ratio) keeping
    activations O(1); biases ~ 0.1 N; snake alpha/beta ~ 0.3 N (log scale)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        if name.endswith("weight_v"):
            sd[name] = torch.randn(shape, generator=g)
        elif name.endswith("weight_g"):
            sd[name] = 0.9 + 0.2 * torch.rand(shape, generator=g)
        elif name.endswith("bias"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        else:  # alpha / beta
            sd[name] = 0.3 * torch.randn(shape, generator=g)
    # folded row norm == g, so a conv maps unit-variance input to ~g^2 variance:
    # keep g ~ 0.9..1.1, and damp the residual-branch output convs (k=1) by
    # `res_gain` so 15 stacked residual units keep activations O(1) (Snake is
    # identity + bounded, it does not shrink variance the way ELU does).
    for name in list(sd):
        if name.endswith("layers.3.weight_g") and sd[name].ndim == 3:
            sd[name] = sd[name] * res_gain
    return sd
