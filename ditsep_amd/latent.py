"""`LatentDiffSep`-shaped facade over the HIP engine (the drop-in boundary).

Mirrors the inference surface of the reference Lightning module
(reference src/diffsep_latent.py): `encode` :107-118, `decode` :120-128, `forward` :147-148,
`get_pc_sampler` :406-469, `separate` :471-487, with the same config keys
(`model.score_model`, `model.vae`, `model.sde`, `model.t_eps`, `model.sampler`,
`model.n_speakers`) and the same state_dict naming (`score_model.*`, `vae.*`).
Training, EMA and logging are out of scope.
"""
from __future__ import annotations

import json
import math
from typing import Any, Mapping, Optional

import torch

from . import checkpoint, native, sdes


def _get(cfg: Any, path: str, default=None):
    cur = cfg
    for key in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, Mapping):
            cur = cur.get(key, None)
        else:
            cur = getattr(cur, key, None)
    return default if cur is None else cur


def _vae_arch(vae_cfg) -> dict:
    """Read the Oobleck architecture from a stable-audio-tools model config (dict or JSON path;
    reference src/utils/load_stable_model.py:9-35, autoencoders.py:866-909)."""
    cfg = vae_cfg
    path = _get(vae_cfg, "config_path")
    if path is not None and _get(vae_cfg, "model") is None:
        with open(path) as fh:
            cfg = json.load(fh)
    enc = _get(cfg, "model.encoder.config")
    dec = _get(cfg, "model.decoder.config")
    if _get(cfg, "model.encoder.type", "oobleck") != "oobleck" or _get(cfg, "model.decoder.type", "oobleck") != "oobleck":
        raise NotImplementedError("only Oobleck encoders/decoders are implemented natively")
    if _get(cfg, "model.bottleneck.type", "vae") != "vae":
        raise NotImplementedError("only the VAE bottleneck is implemented natively")
    src = dec if dec is not None else enc
    return dict(vae_channels=int(_get(src, "channels", 128)), vae_c_mults=tuple(_get(src, "c_mults", (1, 2, 4, 8))),
                vae_strides=tuple(_get(src, "strides", (2, 4, 8, 8))),
                latent_dim=int(_get(cfg, "model.latent_dim", _get(dec, "latent_dim", 64))),
                vae_enc_latent_dim=int(_get(enc, "latent_dim", 128)) if enc is not None else 128,
                vae_use_snake=bool(_get(src, "use_snake", False)), vae_final_tanh=bool(_get(dec, "final_tanh", True)),
                vae_has_encoder=enc is not None, vae_has_decoder=dec is not None)


_PRECISIONS = {"bf16": native.PREC_BF16, "bf16x3": native.PREC_BF16X3, "fp16": native.PREC_FP16,
               "fp16x3": native.PREC_FP16X3, "fp8": native.PREC_FP8}


class LatentDiffSep:
    """Native latent-diffusion separator.  `config` is the reference's Hydra tree (a nested dict or an
    OmegaConf object); `config.model.score_model._target_` selects the score network:
      * `...DiTScoreModel` / `...DiffusionTransformer`  DiT over latent tokens (kwargs embed_dim, depth,
        num_heads; io = n_speakers*latent, input_concat = latent)
      * `...LatentScoreModelNCSNpp`                      the NCSN++ latent U-Net the reference wires in
    """

    def __init__(self, config, device: int = 0, precision: str = "fp16"):
        self.config = config
        self.device_index = device
        n_src = int(_get(config, "model.n_speakers", 2))
        sm = _get(config, "model.score_model")
        target = str(_get(sm, "_target_", ""))
        args = dict(device=device, precision=_PRECISIONS[precision], n_src=n_src)
        if target.endswith("DiTScoreModel") or target.endswith("DiffusionTransformer"):
            args.update(score_kind=native.SCORE_DIT, dit_embed_dim=int(_get(sm, "embed_dim", 1024)),
                        dit_depth=int(_get(sm, "depth", 24)), dit_heads=int(_get(sm, "num_heads", 16)))
        elif target.endswith("LatentScoreModelNCSNpp"):
            ba = _get(sm, "backbone_args")
            attn = tuple(_get(ba, "attn_resolutions", (16,)))
            if len(attn) != 1:
                raise NotImplementedError("NCSN++: exactly one attention resolution is implemented natively")
            args.update(score_kind=native.SCORE_NCSNPP, ncsn_nf=int(_get(ba, "nf", 128)),
                        ncsn_ch_mult=tuple(_get(ba, "ch_mult", (1, 2, 2))),
                        ncsn_num_res_blocks=int(_get(ba, "num_res_blocks", 2)), ncsn_attn_resolution=int(attn[0]),
                        ncsn_image_size=int(_get(ba, "image_size", 64)),
                        ncsn_max_latent_length=int(_get(sm, "max_latent_length", 16)))
        elif target == "":
            args.update(score_kind=native.SCORE_NONE)
        else:
            raise ValueError(f"unknown score_model _target_ '{target}'")
        args.update(_vae_arch(_get(config, "model.vae")))
        self.sde = sdes.OUVESDE(theta=_get(config, "model.sde.theta", 1.5),
                                sigma_min=_get(config, "model.sde.sigma_min", 0.96),
                                sigma_max=_get(config, "model.sde.sigma_max", 10.0),
                                N=_get(config, "model.sde.N", 30))
        args.update(sde_theta=self.sde.theta, sde_sigma_min=self.sde.sigma_min, sde_sigma_max=self.sde.sigma_max)
        self.t_eps = float(_get(config, "model.t_eps", 0.03))
        self.t_max = self.sde.T
        self.n_src = n_src
        self.engine = native.Engine(**args)
        # the object `config.model.score_model._target_` names (reference diffsep_latent.py:39), bound to the engine
        from . import score_models
        if args["score_kind"] == native.SCORE_DIT:
            self.score_model = score_models.DiTScoreModel(embed_dim=args["dit_embed_dim"], depth=args["dit_depth"],
                                                          num_heads=args["dit_heads"]).bind(self.engine)
        elif args["score_kind"] == native.SCORE_NCSNPP:
            self.score_model = score_models.LatentScoreModelNCSNpp(
                num_sources=n_src, backbone_args=dict(_get(sm, "backbone_args", {}) or {}),
                max_latent_length=args["ncsn_max_latent_length"]).bind(self.engine)
        else:
            self.score_model = None
        self.max_len_lat = 0
        self._finalized = False

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict: bool = True):
        """Reference checkpoint naming: `score_model.*`, `vae.encoder.*`, `vae.decoder.*`
        (non-EMA parameters, as evaluate_latent.py:203-208 uses them)."""
        self.engine.load_state_dict(state_dict)
        # a missing tensor always raises; with strict (nn.Module's default, and what the reference's loaders use) so
        # does a tensor the configured network does not consume -- e.g. cross-attention / global-conditioning /
        # qk-norm weights of a DiT trained with other options, which would otherwise be silently ignored
        self.engine.finalize(strict=strict)
        self._finalized = True
        return self

    def load_checkpoint(self, ckpt, use_ema: bool = False):
        """Load a checkpoint written by the reference's training loop (reference src/diffsep_latent.py:341-392):
        `ckpt["state_dict"]` (`score_model.*`, `vae.*`, weight-norm `weight_g/weight_v`), `ckpt["ema"]`
        (torch_ema state: `shadow_params` in `parameters()` order -- of the whole module when
        `ckpt["trainable_vae"]`, else of `score_model` only) and `ckpt["trainable_vae"]`.
        `ckpt` is that dict or a path; a path is read with `torch.load(..., weights_only=True)` only (nothing
        in the file is executed; a checkpoint that loader refuses must be re-saved as plain tensors).
        `use_ema` selects the EMA weights now; `eval(no_ema=...)` switches later (reference :356-388)."""
        if not isinstance(ckpt, dict):
            ckpt = torch.load(str(ckpt), map_location="cpu", weights_only=True)
        if "state_dict" not in ckpt:
            raise KeyError("checkpoint has no 'state_dict'")
        raw = dict(ckpt["state_dict"])
        self._raw_state, self._ema_state = raw, None
        ema = ckpt.get("ema")
        if ema is not None:
            scope = "" if ckpt.get("trainable_vae", False) else "score_model."
            # by position in parameters() order, checked by count AND by the shape of every pair
            self._ema_state = {**raw, **checkpoint.match_ema(list(ema["shadow_params"]), raw, scope)}
        if use_ema and self._ema_state is None:
            raise ValueError("use_ema=True but the checkpoint has no 'ema' entry")   # reference: _error_loading_ema
        self._using_ema = bool(use_ema)
        return self.load_state_dict(self._ema_state if use_ema else raw)

    def to(self, device=None, *args, **kwargs):
        """nn.Module.to for the one thing it can mean here: the engine lives on the GPU it was created on.  The
        same device (or a dtype-only / no-op call) returns self; another device is refused loudly rather than
        silently ignored -- build a new LatentDiffSep(config, device=i) there instead."""
        if device is None or isinstance(device, torch.dtype):
            return self
        dev = torch.device(device)
        if dev.type == "cuda" and (dev.index is None or dev.index == self.device_index):
            return self
        raise RuntimeError(f"LatentDiffSep lives on cuda:{self.device_index}; cannot move it to {dev} "
                           "(create a new instance with device=<index>)")

    def train(self, mode: bool = True, no_ema: bool = False):
        """Reference semantics (src/diffsep_latent.py:356-388): eval mode swaps the EMA weights in unless
        `no_ema`; train mode restores the raw parameters.  Only meaningful after `load_checkpoint` with an
        'ema' entry; otherwise a no-op (as the reference behaves when the EMA failed to load)."""
        ema_state = getattr(self, "_ema_state", None)
        if ema_state is None:
            return self
        want = (not mode) and (not no_ema)
        if want != self._using_ema:
            self._using_ema = want
            self.load_state_dict(ema_state if want else self._raw_state)
        return self

    def eval(self, no_ema: bool = False):
        return self.train(False, no_ema=no_ema)

    @property
    def hop_length(self) -> int:
        return self.engine.hop_length

    # ------------------------------------------------------------------ reference surface
    @torch.no_grad()
    def encode(self, mix, target=None, vae_noise=None, seed=None, chunked=False, overlap=32, chunk_size=128):
        """reference src/diffsep_latent.py:107-118.  `seed=None` draws fresh posterior noise per call (as the
        reference's randn does); an int makes the draw reproducible."""
        if seed is None:
            seed = int(torch.randint(0, 2**31 - 2, (1,)).item())
        y = self.engine.encode(mix, vae_noise, seed=seed, chunked=chunked, overlap=overlap, chunk_size=chunk_size)
        self.max_len_lat = max(self.max_len_lat, y.shape[-1])
        if target is None:
            return y, None
        B, n, L = target.shape
        t = self.engine.encode(target.reshape(B * n, 1, L), None, seed=seed + 1)
        return y, t.reshape(B, n, *t.shape[2:])

    @torch.no_grad()
    def decode(self, est, target_dim=None, chunked=False, overlap=32, chunk_size=128):
        """reference src/diffsep_latent.py:120-128.  `chunked` / `overlap` / `chunk_size` (latent frames) select
        the VAE's long-form mode, AudioAutoencoder.decode_audio(..., chunked=True) (autoencoders.py:665-731)."""
        return self.engine.decode(est, target_dim, chunked=chunked, overlap=overlap, chunk_size=chunk_size)

    def forward(self, xt, time, mix):
        return self.engine.score(xt, time, mix)

    __call__ = forward

    def get_pc_sampler(self, predictor_name, corrector_name, y, N=None, minibatch=None, schedule=None, **kwargs):
        N = self.sde.N if N is None else N
        sde = self.sde.copy()
        sde.N = N
        kwargs = {"eps": self.t_eps, "n_spkrs": self.n_src, **kwargs}
        if schedule is not None:
            if minibatch is not None:
                raise NotImplementedError("minibatch + schedule")
            return sdes.get_pc_scheduled_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=y,
                                                 schedule=schedule, **kwargs)
        if minibatch is None:
            return sdes.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=y, **kwargs)
        M = y.shape[0]
        noise = kwargs.pop("noise", None)
        seed = kwargs.pop("seed", None)

        def batched_sampling_fn():
            samples, ns = [], []
            for i in range(int(math.ceil(M / minibatch))):
                sl = slice(i * minibatch, (i + 1) * minibatch)
                nz = None if noise is None else noise[:, sl]
                # independent draws per minibatch, as the reference's successive randn calls give: an explicit seed
                # is offset by the minibatch index (the same seed would repeat one Philox stream in every minibatch)
                sampler = sdes.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=y[sl],
                                              noise=nz, seed=None if seed is None else int(seed) + i, **kwargs)
                s, n = sampler()
                samples.append(s)
                ns.append(n)
            return torch.cat(samples, dim=0), ns

        return batched_sampling_fn

    @torch.no_grad()
    def separate(self, mix, target_dim=None, latent=False, **kwargs):
        if not latent:
            # the reference draws fresh VAE posterior noise on every call (bottleneck.py:57-83): without an explicit
            # seed, so does this (an explicit seed makes the whole call reproducible)
            seed = kwargs.get("seed")
            enc_seed = int(torch.randint(0, 2**31 - 1, (1,)).item()) if seed is None else int(seed)
            mix, _ = self.encode(mix, None, vae_noise=kwargs.pop("vae_noise", None), seed=enc_seed)
        sampler_kwargs = dict(_get(self.config, "model.sampler", {}) or {})
        sampler_kwargs.update(kwargs)
        sampler = self.get_pc_sampler("reverse_diffusion", "ald", mix, **sampler_kwargs)
        est, *others = sampler()
        est = self.decode(est, target_dim)
        return (est, *others)

    def close(self):
        self.engine.close()
