"""MI355X-native latent-diffusion separation inference (drop-in for the hot path of
eduardburlacu/DiTSep: `LatentDiffSep.separate()` = encode -> PC sampler -> decode).
The compute lives in libditsep_hip.so (ditsep_amd/csrc, C-ABI include/ditsep_hip.h);
importing this package does not load it -- the first Engine / LatentDiffSep does, and
fails loudly if it is missing."""

__all__ = ["LatentDiffSep", "Engine", "sdes", "synthetic", "distributed"]


def __getattr__(name):
    if name == "LatentDiffSep":
        from .latent import LatentDiffSep
        return LatentDiffSep
    if name == "Engine":
        from .native import Engine
        return Engine
    if name in ("sdes", "synthetic", "distributed", "native", "latent", "score_models", "checkpoint", "evaluate"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
