"""Import the reference's Python classes for oracle pinning (build container only).

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py and by the optional
`tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent,
i.e. always on the GPU box).  Nothing here is copied from the reference: it
only arranges `sys.modules` stubs for third-party packages that are not
installed in this image so that the reference's *own* source files import
(recipe: SURVEY.md section 8c).

F6: `models.diffsep.ncsnpp_utils.op` JIT-builds CUDA extensions and
hipifies in place at import time -- it is never imported; a stub module exposing
`upfirdn2d` backed by the reference's pure-PyTorch `upfirdn2d_native` (extracted
from the file text by `ast`) takes its place.
"""
from __future__ import annotations

import ast
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("DITSEP_REFERENCE", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")


def available() -> bool:
    return os.path.isdir(REF_SRC)


def _ns(name: str, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference classes on the hot path."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)

    import torch

    # 1. `utils` as a bare namespace (skip utils/__init__.py -> lightning)
    _ns("utils", os.path.join(REF_SRC, "utils"))
    # 2. hydra.utils.instantiate
    def instantiate(cfg, *a, **kw):
        cfg = dict(cfg)
        cfg.pop("_recursive_", None)
        kw.pop("_recursive_", None)
        target = cfg.pop("_target_")
        mod, _, attr = target.rpartition(".")
        return getattr(importlib.import_module(mod), attr)(*a, **cfg, **kw)

    hyd = _ns("hydra")
    hyd.utils = _ns("hydra.utils", instantiate=instantiate, to_absolute_path=lambda p: p)
    # 3. torchaudio / alias_free_torch dummies (import-time only)
    class _Dummy(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    ta = _ns("torchaudio")
    ta.transforms = _ns("torchaudio.transforms", Resample=_Dummy, Spectrogram=_Dummy,
                        InverseSpectrogram=_Dummy)
    _ns("alias_free_torch", Activation1d=_Dummy)

    import stable_audio_tools  # noqa: F401  (imports factory only)

    _ns("stable_audio_tools.inference", os.path.join(REF_SRC, "stable_audio_tools", "inference"))
    _ns("stable_audio_tools.inference.sampling", sample=None)
    _ns("stable_audio_tools.inference.utils", prepare_audio=None)
    _ns("stable_audio_tools.models.diffusion", ConditionedDiffusionModel=_Dummy,
        DAU1DCondWrapper=_Dummy, UNet1DCondWrapper=_Dummy, DiTWrapper=_Dummy)

    from stable_audio_tools.models import autoencoders, bottleneck, dit
    import sdes
    import sdes.sdes as sdes_sdes
    from utils import torch_utils

    ns = types.SimpleNamespace(
        torch_utils=torch_utils, sdes=sdes, OUVESDE=sdes_sdes.OUVESDE,
        DiffusionTransformer=dit.DiffusionTransformer,
        OobleckDecoder=autoencoders.OobleckDecoder,
        OobleckEncoder=autoencoders.OobleckEncoder,
        AudioAutoencoder=autoencoders.AudioAutoencoder,
        VAEBottleneck=bottleneck.VAEBottleneck,
        vae_sample=bottleneck.vae_sample,
    )
    _loaded = ns
    return ns


def load_ncsnpp():
    """Reference LatentScoreModelNCSNpp with the native-op package stubbed (F6)."""
    ns = load()
    if hasattr(ns, "LatentScoreModelNCSNpp"):
        return ns
    import torch  # noqa: F401
    import torch.nn.functional as F  # noqa: F401

    base = os.path.join(REF_SRC, "models")
    _ns("models", base)
    _ns("models.diffsep", os.path.join(base, "diffsep"))
    _ns("models.diffsep.ncsnpp_utils", os.path.join(base, "diffsep", "ncsnpp_utils"))
    # extract ONLY the pure-PyTorch `upfirdn2d_native` def from the file text
    op_file = os.path.join(base, "diffsep", "ncsnpp_utils", "op", "upfirdn2d.py")
    with open(op_file) as fh:
        src = fh.read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "upfirdn2d_native"]
    assert len(fn) == 1
    env = {"torch": __import__("torch"), "F": __import__("torch").nn.functional}
    exec(compile(ast.Module(body=fn, type_ignores=[]), op_file, "exec"), env)
    native = env["upfirdn2d_native"]

    def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
        return native(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])

    _ns("models.diffsep.ncsnpp_utils.op", upfirdn2d=upfirdn2d, FusedLeakyReLU=None,
        fused_leaky_relu=None)
    ncsnpp = importlib.import_module("models.diffsep.ncsnpp")
    # score_models.py imports torchaudio/hydra at top; both are stubbed above
    score_models = importlib.import_module("models.diffsep.score_models")
    ns.NCSNpp = ncsnpp.NCSNpp
    ns.LatentScoreModelNCSNpp = score_models.LatentScoreModelNCSNpp
    return ns
