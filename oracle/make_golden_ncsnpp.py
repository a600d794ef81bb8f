"""NCSN++ golden vectors (called from oracle/make_golden.py::main).  TEST INFRASTRUCTURE ONLY."""
import torch

from . import reference_loader as rl
from .ncsnpp import NCSNppConfig, random_ncsnpp_weights


def main(save, checksum):
    ns = rl.load_ncsnpp()
    for tag, n_src, T in (("2spk", 2, 8), ("3spk", 3, 6)):
        cfg = NCSNppConfig(n_src=n_src, nf=32)
        sd = random_ncsnpp_weights(cfg, 41)
        ref = ns.LatentScoreModelNCSNpp(num_sources=n_src, backbone_args=cfg.reference_backbone_args(),
                                        max_latent_length=cfg.max_latent_length).eval()
        ref.load_state_dict(sd)
        g = torch.Generator().manual_seed(42)
        xt = 3.0 * torch.randn((2, n_src, 64, T), generator=g)
        mix = torch.randn((2, 1, 64, T), generator=g)
        t = torch.tensor([0.8, 0.1])
        with torch.no_grad():
            out = ref(xt, t, mix)
        save(f"ncsnpp_tiny_{tag}", xt=xt, mix=mix, t=t, out=out, wsum=checksum(sd), seed=41, nf=32, n_src=n_src)


def reference_model(n_src: int = 2, nf: int = 32):
    """A reference LatentScoreModelNCSNpp instance (tiny), for fixtures that only need its structure."""
    ns = rl.load_ncsnpp()
    cfg = NCSNppConfig(n_src=n_src, nf=nf)
    return ns.LatentScoreModelNCSNpp(num_sources=n_src, backbone_args=cfg.reference_backbone_args(),
                                     max_latent_length=cfg.max_latent_length).eval()
