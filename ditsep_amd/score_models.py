"""Config targets for `config.model.score_model._target_` (reference src/diffsep_latent.py:39: the score network is
`hydra.utils.instantiate(config.model.score_model)`, any `forward(xt, time_cond, mix)` module).

The networks themselves run inside the HIP engine; these classes are what a Hydra `_target_` string resolves to:
they carry the architecture keywords (so `instantiate` succeeds and `LatentDiffSep` can read them back) and, once
bound to an engine, evaluate the score natively.  There is no PyTorch implementation behind them: calling an
unbound instance raises.

  ditsep_amd.score_models.DiTScoreModel            DiT over latent tokens (reference class DiffusionTransformer,
                                                   src/stable_audio_tools/models/dit.py:12-244) through the
                                                   (xt, t, mix) adapter: io = n_speakers * latent_dim channels,
                                                   input_concat_cond = the mixture latent
  ditsep_amd.score_models.LatentScoreModelNCSNpp   same keywords as the reference's
                                                   models.diffsep.score_models.LatentScoreModelNCSNpp (:140-186)
"""
from __future__ import annotations


class _NativeScoreModel:
    kind = "none"

    def __init__(self, **kwargs):
        self.kwargs = dict(kwargs)
        self.engine = None

    def bind(self, engine):
        self.engine = engine
        return self

    def forward(self, xt, time_cond, mix):
        if self.engine is None:
            raise RuntimeError(f"{type(self).__name__} is not bound to a native engine (no PyTorch fallback): "
                               "construct ditsep_amd.LatentDiffSep(config) and call it")
        return self.engine.score(xt, time_cond, mix)

    __call__ = forward


class DiTScoreModel(_NativeScoreModel):
    kind = "dit"

    def __init__(self, embed_dim: int = 1024, depth: int = 24, num_heads: int = 16, **kwargs):
        if embed_dim % num_heads != 0 or embed_dim // num_heads != 64:
            raise ValueError(f"DiTScoreModel: only 64-wide attention heads are implemented natively "
                             f"(embed_dim {embed_dim} / num_heads {num_heads})")
        super().__init__(embed_dim=embed_dim, depth=depth, num_heads=num_heads, **kwargs)


class LatentScoreModelNCSNpp(_NativeScoreModel):
    kind = "ncsnpp"

    def __init__(self, num_sources: int = 2, backbone_args=None, max_latent_length: int = 16, **kwargs):
        super().__init__(num_sources=num_sources, backbone_args=dict(backbone_args or {}),
                         max_latent_length=max_latent_length, **kwargs)
