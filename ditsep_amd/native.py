"""ctypes binding of libditsep_hip.so (C-ABI: include/ditsep_hip.h).

The HIP library is the product; this module only marshals torch device tensors
(raw pointers + the current HIP stream) across the C boundary.  There is no
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Mapping, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSN_LIB", os.path.join(_HERE, "libditsep_hip.so"))

PREC_BF16 = 1
PREC_BF16X3 = 2
PREC_FP16 = 3
PREC_FP16X3 = 4
PREC_FP8 = 5
SCORE_NONE, SCORE_DIT, SCORE_NCSNPP = 0, 1, 2
MAX_VAE_BLOCKS = 8

EXPORTS = [
    "dsn_create", "dsn_destroy", "dsn_last_error", "dsn_load_tensor", "dsn_finalize_weights", "dsn_finalize_weights_ex",
    "dsn_score", "dsn_ouve_schedule", "dsn_pc_sample", "dsn_pc_sample_sched", "dsn_pc_sample_ex", "dsn_pc_sample_mix", "dsn_sb_sample", "dsn_decode",
    "dsn_encode", "dsn_decode_chunked", "dsn_encode_chunked",
    "dsn_latent_frames", "dsn_hop_length", "dsn_separate", "dsn_enable_graphs",
    "dsn_workspace_bytes", "dsn_profile_begin", "dsn_profile_end", "dsn_profile_hbm", "dsn_profile_rows", "dsn_test_igemm",
    "dsn_bench_igemm", "dsn_debug_read", "dsn_si_sdr_pit", "dsn_si_bss_eval",
]


PREDICTORS = {"reverse_diffusion": 0, "euler_maruyama": 1, "none": 2}
CORRECTORS = {"ald": 0, "langevin": 1}


class DsnSamplerOpts(C.Structure):
    _fields_ = [
        ("predictor", C.c_int), ("corrector", C.c_int), ("corrector_steps", C.c_int),
        ("snr", C.c_float), ("t_eps", C.c_float), ("denoise", C.c_int),
        ("timesteps", C.POINTER(C.c_float)), ("prior_mean", C.c_void_p), ("intermediates", C.c_void_p),
    ]


class DsnMixOpts(C.Structure):
    _fields_ = [
        ("prior_mix", C.c_int), ("d_lambda", C.c_float), ("sigma_min", C.c_float), ("sigma_max", C.c_float),
        ("avg_len", C.c_int), ("predictor", C.c_int), ("corrector", C.c_int), ("corrector_steps", C.c_int),
        ("snr", C.c_float), ("t_eps", C.c_float), ("denoise", C.c_int),
    ]


MIX_CORRECTORS = {"ald2": 0, "none": 1}
SB_TYPES = {"sde": 0, "ode": 1}


class DsnConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("precision", C.c_int32), ("n_src", C.c_int32), ("latent_dim", C.c_int32),
        ("score_kind", C.c_int32), ("dit_embed_dim", C.c_int32), ("dit_depth", C.c_int32),
        ("dit_heads", C.c_int32),
        ("ncsn_nf", C.c_int32), ("ncsn_n_levels", C.c_int32), ("ncsn_ch_mult", C.c_int32 * 4),
        ("ncsn_num_res_blocks", C.c_int32), ("ncsn_attn_resolution", C.c_int32), ("ncsn_image_size", C.c_int32),
        ("ncsn_max_latent_length", C.c_int32),
        ("vae_channels", C.c_int32), ("vae_n_blocks", C.c_int32),
        ("vae_c_mults", C.c_int32 * MAX_VAE_BLOCKS), ("vae_strides", C.c_int32 * MAX_VAE_BLOCKS),
        ("vae_enc_latent_dim", C.c_int32), ("vae_use_snake", C.c_int32), ("vae_final_tanh", C.c_int32),
        ("vae_has_encoder", C.c_int32), ("vae_has_decoder", C.c_int32),
        ("sde_theta", C.c_float), ("sde_sigma_min", C.c_float), ("sde_sigma_max", C.c_float),
    ]


_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library; loud failure when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU/PyTorch fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C ditsep_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    vp, ci, cf, fp = C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_float)
    lib.dsn_create.restype = vp
    lib.dsn_create.argtypes = [C.POINTER(DsnConfig)]
    lib.dsn_destroy.restype = None
    lib.dsn_destroy.argtypes = [vp]
    lib.dsn_last_error.restype = C.c_char_p
    lib.dsn_last_error.argtypes = [vp]
    lib.dsn_load_tensor.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), ci, ci]
    lib.dsn_finalize_weights.argtypes = [vp]
    lib.dsn_finalize_weights_ex.argtypes = [vp, ci]
    lib.dsn_score.argtypes = [vp, vp, vp, vp, vp, ci, ci, vp]
    lib.dsn_ouve_schedule.argtypes = [vp, ci, cf, cf, fp, fp, fp, fp, fp, fp]
    lib.dsn_pc_sample.argtypes = [vp, vp, vp, C.c_uint64, vp, ci, ci, ci, ci, cf, cf, ci, C.POINTER(ci), vp]
    lib.dsn_pc_sample_sched.argtypes = [vp, vp, vp, C.c_uint64, vp, ci, ci, ci, fp, ci, cf, ci, C.POINTER(ci), vp]
    lib.dsn_pc_sample_ex.argtypes = [vp, vp, vp, C.c_uint64, vp, ci, ci, ci, C.POINTER(DsnSamplerOpts),
                                     C.POINTER(ci), vp]
    lib.dsn_pc_sample_mix.argtypes = [vp, vp, vp, C.c_uint64, vp, ci, ci, ci, C.POINTER(DsnMixOpts), C.POINTER(ci), vp]
    lib.dsn_sb_sample.argtypes = [vp, vp, vp, C.c_uint64, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp]
    lib.dsn_decode.argtypes = [vp, vp, vp, ci, ci, ci, vp]
    lib.dsn_encode.argtypes = [vp, vp, vp, C.c_uint64, vp, ci, ci, vp]
    lib.dsn_decode_chunked.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.dsn_encode_chunked.argtypes = [vp, vp, vp, C.c_uint64, vp, ci, ci, ci, ci, vp]
    lib.dsn_latent_frames.argtypes = [vp, ci]
    lib.dsn_hop_length.argtypes = [vp]
    lib.dsn_separate.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, ci, ci, ci, ci, ci, cf, cf, ci,
                                 C.POINTER(ci), vp]
    lib.dsn_enable_graphs.argtypes = [vp, ci]
    lib.dsn_workspace_bytes.restype = C.c_int64
    lib.dsn_workspace_bytes.argtypes = [vp]
    lib.dsn_profile_begin.argtypes = [vp]
    lib.dsn_profile_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.dsn_profile_hbm.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.dsn_profile_rows.argtypes = [vp, ci, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.dsn_test_igemm.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.dsn_si_sdr_pit.argtypes = [vp, vp, vp, ci, ci, ci, fp, C.POINTER(ci), vp]
    lib.dsn_si_bss_eval.argtypes = [vp, vp, vp, ci, ci, ci, ci, cf, fp, fp, fp, C.POINTER(ci), vp]
    lib.dsn_debug_read.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.dsn_bench_igemm.argtypes = [vp] + [ci] * 10 + [C.POINTER(C.c_double)]
    for name in EXPORTS:
        getattr(lib, name)      # every symbol include/ditsep_hip.h declares must resolve
    _lib = lib
    return lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


class Engine:
    """One native context on one GPU."""

    def __init__(self, *, device: int = 0, precision: int = PREC_BF16X3, n_src: int = 2, latent_dim: int = 64,
                 score_kind: int = SCORE_DIT, dit_embed_dim: int = 1024, dit_depth: int = 24,
                 dit_heads: int = 16, ncsn_nf: int = 128, ncsn_ch_mult=(1, 2, 2), ncsn_num_res_blocks: int = 2,
                 ncsn_attn_resolution: int = 16, ncsn_image_size: int = 64, ncsn_max_latent_length: int = 4,
                 vae_channels: int = 128, vae_c_mults=(1, 2, 4, 8, 16),
                 vae_strides=(2, 4, 4, 8, 8), vae_enc_latent_dim: int = 128, vae_use_snake: bool = False,
                 vae_final_tanh: bool = True, vae_has_encoder: bool = True, vae_has_decoder: bool = True,
                 sde_theta: float = 1.5, sde_sigma_min: float = 0.96, sde_sigma_max: float = 10.0):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("ditsep_amd needs a ROCm GPU (torch.cuda.is_available() is False)")
        cfg = DsnConfig()
        cfg.device, cfg.precision, cfg.n_src, cfg.latent_dim = device, precision, n_src, latent_dim
        cfg.score_kind = score_kind
        cfg.dit_embed_dim, cfg.dit_depth, cfg.dit_heads = dit_embed_dim, dit_depth, dit_heads
        cfg.ncsn_nf, cfg.ncsn_n_levels = ncsn_nf, len(ncsn_ch_mult)
        for i, m in enumerate(ncsn_ch_mult):
            cfg.ncsn_ch_mult[i] = int(m)
        cfg.ncsn_num_res_blocks, cfg.ncsn_attn_resolution = ncsn_num_res_blocks, ncsn_attn_resolution
        cfg.ncsn_image_size, cfg.ncsn_max_latent_length = ncsn_image_size, ncsn_max_latent_length
        cfg.vae_channels, cfg.vae_n_blocks = vae_channels, len(vae_c_mults)
        assert len(vae_c_mults) == len(vae_strides) <= MAX_VAE_BLOCKS
        for i, (m, s) in enumerate(zip(vae_c_mults, vae_strides)):
            cfg.vae_c_mults[i], cfg.vae_strides[i] = int(m), int(s)
        cfg.vae_enc_latent_dim = vae_enc_latent_dim
        cfg.vae_use_snake, cfg.vae_final_tanh = int(vae_use_snake), int(vae_final_tanh)
        cfg.vae_has_encoder, cfg.vae_has_decoder = int(vae_has_encoder), int(vae_has_decoder)
        cfg.sde_theta, cfg.sde_sigma_min, cfg.sde_sigma_max = sde_theta, sde_sigma_min, sde_sigma_max
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.n_src, self.latent_dim = n_src, latent_dim
        self.ctx = self.lib.dsn_create(C.byref(cfg))
        if not self.ctx:
            raise RuntimeError("dsn_create failed: " + self.lib.dsn_last_error(None).decode())

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "ctx", None):
            self.lib.dsn_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.dsn_last_error(self.ctx).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Mapping[str, torch.Tensor], prefix: str = ""):
        """Feed reference-named tensors (e.g. `score_model.*`, `vae.decoder.*`)."""
        for k, v in sd.items():
            if not torch.is_floating_point(v):
                continue
            t = v.detach().to(torch.float32).contiguous()
            shape = (C.c_int64 * max(t.ndim, 1))(*t.shape)
            self._check(self.lib.dsn_load_tensor(self.ctx, (prefix + k).encode(), C.c_void_p(t.data_ptr()),
                                                 shape, t.ndim, int(t.is_cuda)), f"dsn_load_tensor({prefix + k})")

    def finalize(self, strict: bool = True):
        """strict: tensors the configured network does not consume are an error (load_state_dict(strict=True))."""
        self._check(self.lib.dsn_finalize_weights_ex(self.ctx, int(strict)), "dsn_finalize_weights")

    # ------------------------------------------------------------------ path
    def score(self, xt, t, mix):
        xt, t, mix = (_dev32(a, self.device) for a in (xt, t, mix))
        B, n, D, T = xt.shape
        out = torch.empty_like(xt)
        self._check(self.lib.dsn_score(self.ctx, _ptr(xt), _ptr(t), _ptr(mix), _ptr(out), B, T, self._stream()),
                    "dsn_score")
        return out

    def ouve_schedule(self, N: int, t_eps: float, snr: float):
        arr = [(C.c_float * N)() for _ in range(5)]
        stdT = C.c_float()
        self._check(self.lib.dsn_ouve_schedule(self.ctx, N, t_eps, snr, *arr, C.byref(stdT)), "dsn_ouve_schedule")
        names = ("t", "std", "step", "gain", "G")
        out = {k: torch.tensor(list(a), dtype=torch.float32) for k, a in zip(names, arr)}
        out["std_T"] = stdT.value
        return out

    def pc_sample(self, y, noise=None, *, N=30, corrector_steps=1, snr=0.5, t_eps=0.03, denoise=True, seed=0,
                  timesteps=None, predictor="reverse_diffusion", corrector="ald", prior_mean=None,
                  intermediate=False):
        """timesteps: optional explicit schedule (>= N floats, host) -> the scheduled sampler.
        predictor / corrector: the reference's registered names (the ones with a native kernel).
        prior_mean: `true_mean` [B,n,D,T].  intermediate: also return the per-step (x, x_mean) list."""
        if predictor not in PREDICTORS or corrector not in CORRECTORS:
            raise NotImplementedError(f"no native kernel for predictor {predictor!r} / corrector {corrector!r}")
        y = _dev32(y, self.device)
        B, _, D, T = y.shape
        draws = 1 + N * (corrector_steps + (0 if predictor == "none" else 1))
        if noise is not None:
            noise = _dev32(noise, self.device)
            assert tuple(noise.shape) == (draws, B, self.n_src, D, T), noise.shape
        x = torch.empty((B, self.n_src, D, T), device=self.device, dtype=torch.float32)
        nfe = C.c_int()
        plain = (predictor == "reverse_diffusion" and corrector == "ald" and prior_mean is None and not intermediate)
        if plain and timesteps is None:
            self._check(self.lib.dsn_pc_sample(self.ctx, _ptr(y), _ptr(noise), seed, _ptr(x), B, T, N,
                                               corrector_steps, snr, t_eps, int(denoise), C.byref(nfe),
                                               self._stream()), "dsn_pc_sample")
            return x, nfe.value
        ts = None if timesteps is None else (C.c_float * N)(*[float(v) for v in list(timesteps)[:N]])
        if plain:
            self._check(self.lib.dsn_pc_sample_sched(self.ctx, _ptr(y), _ptr(noise), seed, _ptr(x), B, T, N, ts,
                                                     corrector_steps, snr, int(denoise), C.byref(nfe),
                                                     self._stream()), "dsn_pc_sample_sched")
            return x, nfe.value
        pm = None if prior_mean is None else _dev32(prior_mean, self.device)
        if pm is not None and tuple(pm.shape) != tuple(x.shape):
            raise ValueError(f"prior_mean must be {tuple(x.shape)}, got {tuple(pm.shape)}")
        im = (torch.empty((N, 2, B, self.n_src, D, T), device=self.device, dtype=torch.float32)
              if intermediate else None)
        o = DsnSamplerOpts(PREDICTORS[predictor], CORRECTORS[corrector], int(corrector_steps), float(snr),
                           float(t_eps), int(denoise), ts if ts is not None else None,
                           None if pm is None else pm.data_ptr(), None if im is None else im.data_ptr())
        self._check(self.lib.dsn_pc_sample_ex(self.ctx, _ptr(y), _ptr(noise), seed, _ptr(x), B, T, N, C.byref(o),
                                              C.byref(nfe), self._stream()), "dsn_pc_sample_ex")
        if intermediate:
            return x, nfe.value, [(im[i, 0], im[i, 1]) for i in range(N)]
        return x, nfe.value

    def pc_sample_mix(self, y, noise=None, *, N=30, prior_mix=False, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5,
                      avg_len=510, predictor="reverse_diffusion", corrector="ald2", corrector_steps=1, snr=0.5,
                      t_eps=0.03, denoise=True, seed=0):
        """MixSDE / PriorMixSDE predictor-corrector sampler (ald2 corrector) on the latent state."""
        if predictor not in PREDICTORS or corrector not in MIX_CORRECTORS:
            raise NotImplementedError(f"no native kernel for predictor {predictor!r} / corrector {corrector!r} with MixSDE")
        y = _dev32(y, self.device)
        B, _, D, T = y.shape
        c = 0 if corrector == "none" else int(corrector_steps)
        draws = 1 + N * (c + (0 if predictor == "none" else 1))
        if noise is not None:
            noise = _dev32(noise, self.device)
            assert tuple(noise.shape) == (draws, B, self.n_src, D, T), noise.shape
        x = torch.empty((B, self.n_src, D, T), device=self.device, dtype=torch.float32)
        o = DsnMixOpts(int(prior_mix), float(d_lambda), float(sigma_min), float(sigma_max), int(avg_len),
                       PREDICTORS[predictor], MIX_CORRECTORS[corrector], c, float(snr), float(t_eps), int(denoise))
        nfe = C.c_int()
        self._check(self.lib.dsn_pc_sample_mix(self.ctx, _ptr(y), _ptr(noise), seed, _ptr(x), B, T, N, C.byref(o),
                                               C.byref(nfe), self._stream()), "dsn_pc_sample_mix")
        return x, nfe.value

    def sb_sample(self, y, noise=None, *, N=50, k=2.6, c=0.4, sb_eps=1e-8, t_eps=1e-4, sampler_type="ode", seed=0):
        """Schroedinger-bridge sampler (reference get_sb_sampler + SBVESDE) on the latent state."""
        if sampler_type not in SB_TYPES:
            raise ValueError("Invalid type. Choose 'ode' or 'sde'.")
        y = _dev32(y, self.device)
        B, _, D, T = y.shape
        if noise is not None:
            noise = _dev32(noise, self.device)
            assert tuple(noise.shape) == (N, B, self.n_src, D, T), noise.shape
        x = torch.empty((B, self.n_src, D, T), device=self.device, dtype=torch.float32)
        self._check(self.lib.dsn_sb_sample(self.ctx, _ptr(y), _ptr(noise), seed, _ptr(x), B, T, N, float(k), float(c),
                                           float(sb_eps), float(t_eps), SB_TYPES[sampler_type], self._stream()),
                    "dsn_sb_sample")
        return x

    def decode(self, est, target_len: Optional[int] = None, chunked: bool = False, overlap: int = 32,
               chunk_size: int = 128):
        """chunked / overlap / chunk_size: AudioAutoencoder.decode_audio's long-form mode (latent frames)."""
        est = _dev32(est, self.device)
        B, n, D, T = est.shape
        if n != self.n_src or D != self.latent_dim:
            raise ValueError(f"est must be [B,{self.n_src},{self.latent_dim},T], got {tuple(est.shape)}")
        L = target_len if target_len else self.hop_length * T
        wav = torch.empty((B, n, L), device=self.device, dtype=torch.float32)
        if chunked:
            self._check(self.lib.dsn_decode_chunked(self.ctx, _ptr(est), _ptr(wav), B, T, L, int(chunk_size),
                                                    int(overlap), self._stream()), "dsn_decode_chunked")
        else:
            self._check(self.lib.dsn_decode(self.ctx, _ptr(est), _ptr(wav), B, T, L, self._stream()), "dsn_decode")
        return wav

    def encode(self, mix, vae_noise=None, seed=0, chunked: bool = False, overlap: int = 32, chunk_size: int = 128):
        mix = _dev32(mix, self.device)
        B, _, L = mix.shape
        T = self.latent_frames(L)
        if vae_noise is not None:
            vae_noise = _dev32(vae_noise, self.device)
            assert tuple(vae_noise.shape) == (B, self.latent_dim, T)
        y = torch.empty((B, 1, self.latent_dim, T), device=self.device, dtype=torch.float32)
        if chunked:
            self._check(self.lib.dsn_encode_chunked(self.ctx, _ptr(mix), _ptr(vae_noise), seed, _ptr(y), B, L,
                                                    int(chunk_size), int(overlap), self._stream()),
                        "dsn_encode_chunked")
        else:
            self._check(self.lib.dsn_encode(self.ctx, _ptr(mix), _ptr(vae_noise), seed, _ptr(y), B, L,
                                            self._stream()), "dsn_encode")
        return y

    def separate(self, mix, *, vae_noise=None, noise=None, seed=0, target_len=None, N=30, corrector_steps=1,
                 snr=0.5, t_eps=0.03, denoise=True):
        mix = _dev32(mix, self.device)
        B, _, L = mix.shape
        Lt = target_len if target_len else L
        wav = torch.empty((B, self.n_src, Lt), device=self.device, dtype=torch.float32)
        nfe = C.c_int()
        vn = None if vae_noise is None else _dev32(vae_noise, self.device)
        nz = None if noise is None else _dev32(noise, self.device)
        self._check(self.lib.dsn_separate(self.ctx, _ptr(mix), _ptr(vn), _ptr(nz), seed, _ptr(wav), B, L, Lt, N,
                                          corrector_steps, snr, t_eps, int(denoise), C.byref(nfe),
                                          self._stream()), "dsn_separate")
        return wav, nfe.value

    @property
    def hop_length(self) -> int:
        return self.lib.dsn_hop_length(self.ctx)

    def latent_frames(self, L: int) -> int:
        return self.lib.dsn_latent_frames(self.ctx, L)

    def enable_graphs(self, on: bool = True):
        self._check(self.lib.dsn_enable_graphs(self.ctx, int(on)), "dsn_enable_graphs")

    def workspace_bytes(self) -> int:
        return self.lib.dsn_workspace_bytes(self.ctx)

    def profile_begin(self):
        self._check(self.lib.dsn_profile_begin(self.ctx), "dsn_profile_begin")

    def profile_end(self):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.dsn_profile_end(self.ctx, C.byref(ms), C.byref(fl), C.byref(n)), "dsn_profile_end")
        hm, hb, hn = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.dsn_profile_hbm(self.ctx, C.byref(hm), C.byref(hb), C.byref(hn)), "dsn_profile_hbm")
        nrows = self.lib.dsn_profile_rows(self.ctx, 0, None, None, None, None, None)
        rows = []
        if nrows > 0:
            NL = 48                                     # DSN_PROFILE_NAME_LEN
            names = C.create_string_buffer(nrows * NL)
            rms, rfl, rby = ((C.c_double * nrows)() for _ in range(3))
            rn = (C.c_int64 * nrows)()
            self.lib.dsn_profile_rows(self.ctx, nrows, names, rms, rfl, rby, rn)
            for i in range(nrows):
                nm = names.raw[i * NL:(i + 1) * NL].split(b"\0", 1)[0].decode()
                rows.append({"site": nm, "ms": rms[i], "flops": rfl[i], "bytes": rby[i], "launches": int(rn[i])})
        return {"gemm_ms": ms.value, "gemm_flops": fl.value, "gemm_launches": n.value,
                "hbm_ms": hm.value, "hbm_bytes": hb.value, "hbm_launches": hn.value, "rows": rows}

    def si_sdr_pit(self, ref, est):
        """ref, est [B,n,L] -> (si_sdr [B,n] dB, perm [B,n]): est[:, perm[b,i]] matches ref[:, i]."""
        ref, est = _dev32(ref, self.device), _dev32(est, self.device)
        B, n, L = ref.shape
        sdr = (C.c_float * (B * n))()
        perm = (C.c_int * (B * n))()
        self._check(self.lib.dsn_si_sdr_pit(self.ctx, _ptr(ref), _ptr(est), B, n, L, sdr, perm, self._stream()),
                    "dsn_si_sdr_pit")
        return (torch.tensor(list(sdr), dtype=torch.float32).reshape(B, n),
                torch.tensor(list(perm), dtype=torch.long).reshape(B, n))

    def si_bss_eval(self, ref, est, perm_by: str = "sir", clamp_db: float = 100.0):
        """ref, est [B,n,L] -> (si_sdr, si_sir, si_sar [B,n] dB, perm [B,n]) with the permutation solved on
        `perm_by` ("sir" as bss_eval / the reference's evaluate_latent.py:118-124, or "sdr")."""
        ref, est = _dev32(ref, self.device), _dev32(est, self.device)
        B, n, L = ref.shape
        bufs = [(C.c_float * (B * n))() for _ in range(3)]
        perm = (C.c_int * (B * n))()
        self._check(self.lib.dsn_si_bss_eval(self.ctx, _ptr(ref), _ptr(est), B, n, L, {"sdr": 0, "sir": 1}[perm_by],
                                             float(clamp_db), *bufs, perm, self._stream()), "dsn_si_bss_eval")
        out = [torch.tensor(list(b), dtype=torch.float32).reshape(B, n) for b in bufs]
        return (*out, torch.tensor(list(perm), dtype=torch.long).reshape(B, n))

    def debug_read(self, name: str, shape):
        out = torch.empty(shape, dtype=torch.float32)
        self._check(self.lib.dsn_debug_read(self.ctx, name.encode(), C.c_void_p(out.data_ptr()), out.numel()),
                    f"dsn_debug_read({name})")
        return out

    def bench_igemm(self, B, Lin, Cin, N, taps=1, tap_dil=1, in_pad=0, ksplit=1, variant=2, iters=10):
        ms = C.c_double()
        self._check(self.lib.dsn_bench_igemm(self.ctx, B, Lin, Cin, N, taps, tap_dil, in_pad, ksplit, variant,
                                             iters, C.byref(ms)), "dsn_bench_igemm")
        return ms.value

    def test_igemm(self, a, w, *, taps=1, in_stride=1, tap_dil=1, in_pad=0, rows_per_b=None, panel_rows=0,
                   panel_bn=256):
        """a [B,Lin,Cin] channels-last, w [N, taps*Cin] -> [B, rows_per_b, N] (kernel test hook)."""
        a, w = _dev32(a, self.device), _dev32(w, self.device)
        B, Lin, Cin = a.shape
        N = w.shape[0]
        rpb = rows_per_b or Lin
        out = torch.empty((B, rpb, N), device=self.device, dtype=torch.float32)
        self._check(self.lib.dsn_test_igemm(self.ctx, _ptr(a), _ptr(w), _ptr(out), B, Lin, Cin, N, taps, in_stride,
                                            tap_dil, in_pad, rpb, panel_rows, panel_bn, self._stream()), "dsn_test_igemm")
        return out
