"""Synthetic workload for tests and benchmarks: architecture configs, parameter
inventories (reference state_dict key names), seeded re-randomised weights and
seeded synthetic mixtures.

No trained checkpoints or datasets exist for this path and the
reference's default init zeroes most output projections (F5), so every parameter
is re-randomised from a seed with scales chosen to keep activations O(1).
Pure PyTorch-CPU data generation: nothing here computes the separation path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class DiTConfig:
    n_src: int = 2
    latent_dim: int = 64
    embed_dim: int = 1024
    depth: int = 24
    num_heads: int = 16

    @property
    def io_channels(self) -> int:
        return self.n_src * self.latent_dim

    @property
    def dim_in(self) -> int:
        return self.io_channels + self.latent_dim

    @property
    def dim_heads(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def rot_dim(self) -> int:
        # RotaryEmbedding(max(dim_heads // 2, 32))   transformer.py:800
        return max(self.dim_heads // 2, 32)

    def reference_kwargs(self) -> dict:
        """kwargs for the reference DiffusionTransformer matching this adapter."""
        return dict(
            io_channels=self.io_channels,
            input_concat_dim=self.latent_dim,
            embed_dim=self.embed_dim,
            depth=self.depth,
            num_heads=self.num_heads,
            transformer_type="continuous_transformer",
            global_cond_type="prepend",
        )



def dit_param_shapes(cfg: DiTConfig) -> dict:
    """name -> shape of every tensor dit_forward reads (reference key names)."""
    D, di, io = cfg.embed_dim, cfg.dim_in, cfg.io_channels
    s = {
        "timestep_features.weight": (128, 1),
        "to_timestep_embed.0.weight": (D, 256), "to_timestep_embed.0.bias": (D,),
        "to_timestep_embed.2.weight": (D, D), "to_timestep_embed.2.bias": (D,),
        "preprocess_conv.weight": (di, di, 1),
        "postprocess_conv.weight": (io, io, 1),
        "transformer.project_in.weight": (D, di),
        "transformer.project_out.weight": (io, D),
    }
    for i in range(cfg.depth):
        p = f"transformer.layers.{i}."
        s[p + "pre_norm.gamma"] = (D,)
        s[p + "self_attn.to_qkv.weight"] = (3 * D, D)
        s[p + "self_attn.to_out.weight"] = (D, D)
        s[p + "ff_norm.gamma"] = (D,)
        s[p + "ff.ff.0.proj.weight"] = (8 * D, D)
        s[p + "ff.ff.0.proj.bias"] = (8 * D,)
        s[p + "ff.ff.2.weight"] = (D, 4 * D)
        s[p + "ff.ff.2.bias"] = (D,)
    return s


def random_dit_weights(cfg: DiTConfig, seed: int, out_gain: float = 1.0,
                       skip_gain: float = 0.0) -> dict:
    """Seeded re-randomisation of EVERY parameter (the reference's default init
    zeroes to_out / ff-out / pre/post convs, SURVEY.md F5, which would make
    parity vacuous).  Linear weights ~ N(0, 1/fan_in) so activations stay O(1);
    norm gains ~ 1 + 0.1 N(0,1); biases ~ 0.1 N(0,1).  `out_gain` scales
    project_out so the score magnitude suits the sampler dynamics.

    `skip_gain` = kappa > 0 adds, through the network's own linear skip path
    (project_in -> residual stream -> project_out), the term -kappa (x_s - y):
    the shape of a trained OU score, so that the synthetic sampler contracts
    towards the mixture like a trained model does instead of random-walking to
    |x| ~ 100 (no trained weights exist)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in dit_param_shapes(cfg).items():
        if name == "timestep_features.weight":
            w = torch.randn(shape, generator=g)
        elif name.endswith("gamma"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            w = 0.1 * torch.randn(shape, generator=g)
        elif name in ("preprocess_conv.weight", "postprocess_conv.weight"):
            w = torch.randn(shape, generator=g) * (0.5 / math.sqrt(shape[1]))
        else:
            w = torch.randn(shape, generator=g) / math.sqrt(shape[1])
            if name.endswith("to_out.weight") or name.endswith("ff.ff.2.weight"):
                w = w * 0.5
        if name == "transformer.project_out.weight":
            w = w * out_gain
        sd[name] = w
    if skip_gain:
        n, Dl = cfg.n_src, cfg.latent_dim
        M = torch.zeros(cfg.io_channels, cfg.dim_in)
        M[:, : cfg.io_channels] = torch.eye(cfg.io_channels)
        for s_ in range(n):
            M[s_ * Dl:(s_ + 1) * Dl, cfg.io_channels:] = -torch.eye(Dl)
        pinv = torch.linalg.pinv(sd["transformer.project_in.weight"].double()).float()
        sd["transformer.project_out.weight"] = (
            sd["transformer.project_out.weight"] - skip_gain * (M @ pinv))
    return sd


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


def vae_weights(cfg: OobleckConfig, seed: int, dec_in_gain: float = 1.0) -> dict:
    """`encoder.*` + `decoder.*` weights (state_dict of the reference AudioAutoencoder
    minus the `vae.` prefix).  `dec_in_gain` scales the decoder's first conv: the
    untrained sampler hands the decoder latents of std ~10-20 instead of ~1, and this
    keeps the decoder's activations O(1) (away from tanh saturation)."""
    sd = {}
    sd.update(random_weights(encoder_param_shapes(cfg, "encoder."), seed))
    sd.update(random_weights(decoder_param_shapes(cfg, "decoder."), seed + 1))
    if dec_in_gain != 1.0:
        sd["decoder.layers.0.weight_g"] = sd["decoder.layers.0.weight_g"] * dec_in_gain
    return sd


def synthetic_sources(B: int, n: int, L: int, fs: int = 16000, seed: int = 1234) -> torch.Tensor:
    """[B, n, L] seeded band-limited noise bursts, peak 0.3.
    Low-pass: 4 cascaded one-pole sections at 0.4*fs/2-ish, then a 3-8 Hz
    raised-cosine envelope."""
    out = torch.empty(B, n, L)
    tt = torch.arange(L, dtype=torch.float64) / fs
    for b in range(B):
        for k in range(n):
            g = torch.Generator().manual_seed(seed + 1000 * b + k)
            w = torch.randn(L, generator=g, dtype=torch.float64)
            # cheap zero-phase low-pass via FFT brick wall at 0.4 * fs
            spec = torch.fft.rfft(w)
            freqs = torch.fft.rfftfreq(L, 1.0 / fs)
            spec = spec * (1.0 / (1.0 + (freqs / (0.4 * fs)) ** 8))
            w = torch.fft.irfft(spec, n=L)
            rate = 3.0 + 5.0 * torch.rand(1, generator=g, dtype=torch.float64)
            phase = 2 * math.pi * torch.rand(1, generator=g, dtype=torch.float64)
            env = 0.5 * (1 - torch.cos(2 * math.pi * rate * tt + phase))
            s = w * env
            out[b, k] = (0.3 * s / s.abs().max()).float()
    return out


# ---------------------------------------------------------------------------
# NCSN++ latent score network (the reference's wired-in score model)
# ---------------------------------------------------------------------------
@dataclass
class NCSNppConfig:
    """Defaults = src/config/latent_diffsep_ouve/model/default.yaml:16-28."""

    n_src: int = 2
    nf: int = 128
    ch_mult: tuple = (1, 2, 2)
    num_res_blocks: int = 2
    attn_resolutions: tuple = (16,)
    image_size: int = 64           # = latent dim (the "height" of the latent image)
    max_latent_length: int = 4     # W is padded to a multiple of this
    fourier_scale: float = 16.0

    @property
    def ch_in(self) -> int:
        return self.n_src + 1

    def layout(self):
        """Module list in the reference's construction order (ncsnpp.py:107-309):
        entries (index, kind, info)."""
        nf, levels = self.nf, len(self.ch_mult)
        res = [self.image_size // (2**i) for i in range(levels)]
        mods = [(0, "fourier", {}), (1, "linear", dict(cin=2 * nf, cout=4 * nf)),
                (2, "linear", dict(cin=4 * nf, cout=4 * nf)), (3, "conv3", dict(cin=self.ch_in, cout=nf))]
        i = 4
        hs_c = [nf]
        in_ch = nf
        for lv in range(levels):
            for _ in range(self.num_res_blocks):
                out_ch = nf * self.ch_mult[lv]
                mods.append((i, "res", dict(cin=in_ch, cout=out_ch))); i += 1
                in_ch = out_ch
                if res[lv] in self.attn_resolutions:
                    mods.append((i, "attn", dict(c=in_ch))); i += 1
                hs_c.append(in_ch)
            if lv != levels - 1:
                mods.append((i, "res", dict(cin=in_ch, cout=in_ch, down=True))); i += 1
                mods.append((i, "combine", dict(cin=self.ch_in, cout=in_ch))); i += 1
                hs_c.append(in_ch)
        mods.append((i, "res", dict(cin=in_ch, cout=in_ch))); i += 1
        mods.append((i, "attn", dict(c=in_ch))); i += 1
        mods.append((i, "res", dict(cin=in_ch, cout=in_ch))); i += 1
        for lv in reversed(range(levels)):
            for _ in range(self.num_res_blocks + 1):
                out_ch = nf * self.ch_mult[lv]
                mods.append((i, "res", dict(cin=in_ch + hs_c.pop(), cout=out_ch))); i += 1
                in_ch = out_ch
            if res[lv] in self.attn_resolutions:
                mods.append((i, "attn", dict(c=in_ch))); i += 1
            mods.append((i, "gn", dict(c=in_ch))); i += 1
            mods.append((i, "conv3", dict(cin=in_ch, cout=self.ch_in))); i += 1
            if lv != 0:
                mods.append((i, "res", dict(cin=in_ch, cout=in_ch, up=True))); i += 1
        assert not hs_c
        return mods

    @property
    def n_modules(self) -> int:
        return len(self.layout())

    def reference_backbone_args(self) -> dict:
        return dict(_target_="models.diffsep.ncsnpp.NCSNpp", nf=self.nf, ch_mult=list(self.ch_mult),
                    num_res_blocks=self.num_res_blocks, attn_resolutions=list(self.attn_resolutions),
                    resamp_with_conv=True, image_size=self.image_size, centered=True)


def ncsnpp_param_shapes(cfg: NCSNppConfig, prefix: str = "backbone.") -> dict:
    s = {}
    temb = 4 * cfg.nf
    for i, kind, a in cfg.layout():
        p = f"{prefix}all_modules.{i}."
        if kind == "fourier":
            s[p + "W"] = (cfg.nf,)
        elif kind == "linear":
            s[p + "weight"], s[p + "bias"] = (a["cout"], a["cin"]), (a["cout"],)
        elif kind == "conv3":
            s[p + "weight"], s[p + "bias"] = (a["cout"], a["cin"], 3, 3), (a["cout"],)
        elif kind == "gn":
            s[p + "weight"], s[p + "bias"] = (a["c"],), (a["c"],)
        elif kind == "combine":
            s[p + "Conv_0.weight"], s[p + "Conv_0.bias"] = (a["cout"], a["cin"], 1, 1), (a["cout"],)
        elif kind == "attn":
            c = a["c"]
            s[p + "GroupNorm_0.weight"], s[p + "GroupNorm_0.bias"] = (c,), (c,)
            for k in range(4):
                s[p + f"NIN_{k}.W"], s[p + f"NIN_{k}.b"] = (c, c), (c,)
        elif kind == "res":
            ci, co = a["cin"], a["cout"]
            s[p + "GroupNorm_0.weight"], s[p + "GroupNorm_0.bias"] = (ci,), (ci,)
            s[p + "Conv_0.weight"], s[p + "Conv_0.bias"] = (co, ci, 3, 3), (co,)
            s[p + "Dense_0.weight"], s[p + "Dense_0.bias"] = (co, temb), (co,)
            s[p + "GroupNorm_1.weight"], s[p + "GroupNorm_1.bias"] = (co,), (co,)
            s[p + "Conv_1.weight"], s[p + "Conv_1.bias"] = (co, co, 3, 3), (co,)
            if ci != co or a.get("up") or a.get("down"):
                s[p + "Conv_2.weight"], s[p + "Conv_2.bias"] = (co, ci, 1, 1), (co,)
    s[prefix + "output_layer.weight"] = (cfg.n_src, cfg.ch_in, 1, 1)
    s[prefix + "output_layer.bias"] = (cfg.n_src,)
    return s


def random_ncsnpp_weights(cfg: NCSNppConfig, seed: int, out_gain: float = 1.0) -> dict:
    """Seeded re-randomisation of every parameter (the reference zero-initialises Conv_1 / NIN_3 /
    pyramid convs through init_scale=0 -> 1e-10): conv / linear weights ~ N(0, 1/fan_in),
    norm gains 1 + 0.1 N, biases 0.1 N, Fourier W ~ N(0, scale^2); `out_gain` scales the 1x1 output layer."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in ncsnpp_param_shapes(cfg).items():
        if name.endswith("all_modules.0.W"):
            w = torch.randn(shape, generator=g) * cfg.fourier_scale
        elif "GroupNorm" in name and name.endswith("weight") or (len(shape) == 1 and name.endswith("weight")):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias") or name.endswith(".b"):
            w = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".W"):           # NIN [in, out]
            w = torch.randn(shape, generator=g) / math.sqrt(shape[0])
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            w = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        sd[name] = w
    sd["backbone.output_layer.weight"] = sd["backbone.output_layer.weight"] * out_gain
    return sd
