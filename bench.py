#!/usr/bin/env python3
"""Headline benchmark: separated utterances/sec on the BASELINE.json C2 workload.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under `python -m torch.distributed.run ... bench.py --gpus N` (how the
round driver starts it) the ranks read RANK / LOCAL_RANK / WORLD_SIZE; a plain `python bench.py --gpus N`
starts the same N ranks itself as child processes (ditsep_amd.distributed.launch_ranks) before this process
has touched the GPU.  It never silently runs fewer ranks than asked for.

A "step" = one pass of the hot path over one batch of synthetic Libri2Mix-shape mixtures already resident in
HBM as latents: N=30 predictor-corrector sampler (1 corrector step, 60 score-network calls, on-device Philox
noise) followed by the Oobleck decode to waveforms -- the region the reference times
(src/evaluate_latent.py:273-277).  Mixtures are independent: each rank owns a 64-mixture shard (weak
scaling, BASELINE C3 = 8 x 64) and the only data-path collective is one RCCL gather of the separated
waveforms to rank 0 per step (ditsep_amd.distributed.separate_sharded -- the function the gloo test drives).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL on this driver stack shares buffers between ranks through dmabuf IPC only; must be in the environment
# before the HIP runtime starts, whoever launched the ranks (torch.distributed.run or launch_ranks)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


PEAK_MFMA_DENSE_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 / fp16 MFMA
PEAK_FP8_DENSE_TFLOPS = 5000.0       # ~5 PF dense fp8 MFMA
PEAK_HBM_GBS = 8000.0                # HBM3E 8 TB/s spec (6.3 TB/s achievable)
FS, SECONDS, N_STEPS, CORR, SNR, T_EPS = 16000, 4, 30, 1, 0.5, 0.03
DIT_OUT_GAIN, DIT_SKIP_GAIN, DEC_IN_GAIN, NCSN_OUT_GAIN = 0.002, 0.02, 0.08, 0.01

# call site (dsn_profile_rows) -> what it is, and the kernel name rocprofv3 shows for it at the C2 shape in the
# headline mode (for cross-checking the table against profiles/*_kernel_stats.csv)
SITES = {
    "dit.ff_in": "DiT FF-in GEMM 1024->8192 + SwiGLU (transformer.py:214-288)",
    "dit.ff_out": "DiT FF-out GEMM 4096->1024 (transformer.py:214-288)",
    "dit.qkv": "DiT to_qkv GEMM 1024->3072 + RoPE (transformer.py:290-598)",
    "dit.attn_out": "DiT to_out GEMM 1024->1024 (transformer.py:290-598)",
    "dit.attention": "DiT softmax(QK^T)V, 16 heads x 64",
    "dit.qkv_attention": "DiT to_qkv GEMM 1024->3072 + RoPE + softmax(QK^T)V in one launch (transformer.py:290-598)",
    "dit.residual_norm": "residual add (+ split-K reduce) + LayerNorm -> operand planes",
    "dit.project_in": "DiT project_in (+ folded preprocess conv)",
    "dit.project_out": "DiT project_out (+ folded postprocess conv)",
    "score.time_embed": "timestep features + to_timestep_embed MLP, all steps at once",
    "vae.residual_unit_fused": "Oobleck ResidualUnit fused (k7 dilated conv + act + 1x1 + residual)",
    "vae.residual_unit_2gemm": "Oobleck ResidualUnit as two implicit GEMMs (wide layers)",
    "vae.dec_convT": "Oobleck ConvTranspose1d as 2-tap phase GEMM",
    "vae.dec_conv_in": "Oobleck decoder first conv",
    "vae.dec_conv_out": "Oobleck decoder last conv (Cout = 1) + tanh",
    "ncsnpp.conv": "NCSN++ 3x3 / 1x1 convs and dense layers (implicit GEMM)",
}


def precisions():
    from ditsep_amd import native
    return {"bf16": (native.PREC_BF16, "bf16 MFMA operands, fp32 accumulate"),
            "bf16x3": (native.PREC_BF16X3, "split-bf16 (hi,lo) MFMA operands x3, fp32 accumulate"),
            "fp16": (native.PREC_FP16, "fp16 MFMA operands, fp32 accumulate"),
            "fp16x3": (native.PREC_FP16X3, "split-fp16 (hi,lo) MFMA operands x3, fp32 accumulate"),
            **({"fp8": (native.PREC_FP8, "fp8 e4m3 MFMA operands for the DiT layer GEMMs (per-row scales), "
                                         "fp16 elsewhere, fp32 accumulate")} if hasattr(native, "PREC_FP8") else {})}


def build_engine(device, precision, dcfg, vcfg, dsd, vsd):
    from ditsep_amd import native, synthetic
    score = dict(score_kind=native.SCORE_DIT, dit_embed_dim=dcfg.embed_dim, dit_depth=dcfg.depth,
                 dit_heads=dcfg.num_heads, latent_dim=dcfg.latent_dim) if isinstance(dcfg, synthetic.DiTConfig) else \
        dict(score_kind=native.SCORE_NCSNPP, ncsn_nf=dcfg.nf, ncsn_ch_mult=dcfg.ch_mult,
             ncsn_num_res_blocks=dcfg.num_res_blocks, ncsn_attn_resolution=dcfg.attn_resolutions[0],
             ncsn_image_size=dcfg.image_size, ncsn_max_latent_length=dcfg.max_latent_length,
             latent_dim=dcfg.image_size)
    eng = native.Engine(device=device, precision=precision, n_src=dcfg.n_src, vae_channels=vcfg.channels,
                        vae_c_mults=vcfg.c_mults, vae_strides=vcfg.strides,
                        vae_enc_latent_dim=vcfg.enc_latent_dim, vae_use_snake=vcfg.use_snake,
                        vae_final_tanh=vcfg.final_tanh, **score)
    eng.load_state_dict(dsd, prefix="score_model.")
    eng.load_state_dict(vsd, prefix="vae.")
    eng.finalize()
    return eng


def _cpu_run(score, vsd, vcfg, dcfg, y, noise, L, N):
    """One pass of the CPU restatement (oracle/, test infrastructure -- here only as the timed baseline)."""
    import torch
    from oracle import oobleck as ovae
    from oracle import sampler as osmp
    t0 = time.perf_counter()
    with torch.no_grad():
        x, nfe = osmp.pc_sample(score, y, noise, osmp.OUVE(N=N), eps=T_EPS, snr=SNR, corrector_steps=CORR,
                                denoise=True, n_spkrs=dcfg.n_src)
        wav = ovae.decode_sources(vsd, vcfg, x, L, "decoder.")
    return time.perf_counter() - t0, wav


def cpu_baseline(dcfg, vcfg, dsd, vsd, y_c2, y_c1):
    """SURVEY 8(d) / BASELINE.md 3: the CPU restatement of the reference path (kind = "port") timed on this
    host, sampler + decode, 1 warm-up then best of 3: the C2 shape at batch 1 (the headline workload's own
    mixtures) and config C1 exactly (8 kHz x 4 s, N = 10, batch 1)."""
    import torch
    from ditsep_amd import synthetic
    from oracle import dit as odit
    from oracle import ncsnpp as oncs
    from oracle import sampler as osmp

    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("DITSEP_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    score = odit.DiTScore(dsd, dcfg) if isinstance(dcfg, synthetic.DiTConfig) else oncs.NCSNppScore(dsd, dcfg)
    g = torch.Generator().manual_seed(99)
    out = {}
    for name, y, L, N, reps in (("c2_b1", y_c2[:1], FS * SECONDS, N_STEPS, 4), ("c2_b4", y_c2, FS * SECONDS, N_STEPS, 2),
                                ("c1", y_c1, 8000 * SECONDS, 10, 4)):
        y = y.detach().cpu().float()
        noise = osmp.draw_noise(g, 1 + N * (CORR + 1), (y.shape[0], dcfg.n_src, vcfg.latent_dim, y.shape[-1]))
        times, wav = [], None
        for rep in range(reps):                   # warm-up + the timed runs (the batch-4 leg is already warm)
            dt, wav = _cpu_run(score, vsd, vcfg, dcfg, y, noise, L, N)
            if rep or name == "c2_b4":
                times.append(dt)
            log(f"cpu {name} run {rep}: {dt:.2f} s")
        out[name] = dict(times=times, best=min(times), noise=noise, wav=wav, y=y, L=L, N=N)
    c2, c4, c1 = out["c2_b1"], out["c2_b4"], out["c1"]
    nb = c4["y"].shape[0]
    rec = {"value": round(1.0 / c2["best"], 4), "unit": "utt/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"1 mixture of the C2 workload (16 kHz x 4 s, N=30, 60 NFE, sampler+decode; batch 1), "
                     f"PyTorch-CPU fp32 oracle, 1 warm-up + best of 3: {[round(t, 2) for t in c2['times']]} s",
           "c2_b4": {"value": round(nb / c4["best"], 4), "unit": "utt/s",
                     "sample": f"{nb} mixtures of the C2 workload as one batch, sampler+decode, best of 2 (warm): "
                               f"{[round(t, 2) for t in c4['times']]} s"},
           "c1": {"value": round(1.0 / c1["best"], 4), "unit": "utt/s",
                  "sample": f"config C1 exactly: 8 kHz x 4 s (T=16), N=10 (20 NFE), batch 1, sampler+decode, "
                            f"1 warm-up + best of 3: {[round(t, 2) for t in c1['times']]} s"}}
    return rec, c4, c1


def roofline_rows(prof, fp8_sites=()):
    """Per call-site roofline rows from the engine's per-launch HIP events (one un-timed step)."""
    rows = []
    for r in prof["rows"]:
        ms = r["ms"]
        if ms <= 0:
            continue
        row = {"site": r["site"], "what": SITES.get(r["site"], ""), "launches": r["launches"],
               "ms_per_step": round(ms, 3), "avg_launch_us": round(1e3 * ms / r["launches"], 2)}
        if r["flops"] > 0:
            peak = PEAK_FP8_DENSE_TFLOPS if r["site"] in fp8_sites else PEAK_MFMA_DENSE_TFLOPS
            tf = r["flops"] / (ms * 1e-3) / 1e12
            row.update(bound="mfma", achieved=round(tf, 1), peak=peak, unit="TFLOP/s", frac=round(tf / peak, 4),
                       gflop_per_launch=round(r["flops"] / r["launches"] / 1e9, 3))
        elif r["bytes"] > 0:
            gbs = r["bytes"] / (ms * 1e-3) / 1e9
            row.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                       frac=round(gbs / PEAK_HBM_GBS, 4), mb_per_launch=round(r["bytes"] / r["launches"] / 1e6, 3))
        if r["flops"] > 0 and r["bytes"] > 0:       # HBM-bound member of the GEMM family (fused ResidualUnit)
            gbs = r["bytes"] / (ms * 1e-3) / 1e9
            row["hbm"] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": round(gbs / PEAK_HBM_GBS, 4)}
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def measure_c5(engine, dcfg, dev, steps=2, batch=16):
    """BASELINE config 5 shape on one GPU: 30 s mixtures at 16 kHz (T = 235 latent frames, 236 tokens), N = 30 +
    1 corrector, hipGraph-captured sampler loop + plain decode; `batch` mixtures per step (BASELINE names no batch for
    this config; 16 x 236 = 3776 token rows is about the row count of the C2 step -- scripts/c5_batch_sweep.py:
    fp16 31.7 / 39.3 / 42.2 utt/s and fp8 36.3 / 46.3 / 50.6 at batch 8 / 16 / 32)."""
    import torch
    from ditsep_amd import synthetic
    L5 = 30 * FS
    mix5 = synthetic.synthetic_sources(batch, dcfg.n_src, L5, FS, seed=4242).sum(1, keepdim=True).to(dev)
    y5 = engine.encode(mix5, seed=11)

    def step5(i):
        x, _ = engine.pc_sample(y5, None, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS, seed=700 + i)
        return engine.decode(x, L5)
    for i in range(2):
        step5(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step5(2 + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": round(batch * steps / el, 3), "unit": "utt/s (30 s mixtures)", "ms_per_step": round(1e3 * el / steps, 1),
            "batch": batch, "latent_frames": int(y5.shape[-1]),
            "workload": "C5 shape: 2-spk 16 kHz 30 s mixtures, N=30 PC sampler (60 NFE, hipGraph) + Oobleck decode"}


def measure_ncsnpp(local, dev, prec, vcfg, vsd, mix, src, L, graphs, steps=3, parity=True):
    """The same C2 workload with the score network the reference actually wires in (LatentScoreModelNCSNpp, nf = 128):
    its own engine, 3 timed steps (2 set-up calls + 1 warm-up first), its own parity leg -- the CPU oracle's latents and
    injected noise ride as items 0..1 of the timed batch -- and the roofline row of its conv family."""
    import torch
    from ditsep_amd import synthetic
    ncfg = synthetic.NCSNppConfig()
    nsd = synthetic.random_ncsnpp_weights(ncfg, 1, out_gain=NCSN_OUT_GAIN)
    eng = build_engine(local, prec, ncfg, vcfg, nsd, vsd)
    eng.enable_graphs(graphs)
    B = mix.shape[0]
    y = eng.encode(mix, seed=7)

    def step(i):
        x, _ = eng.pc_sample(y, None, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS, denoise=True, seed=900 + i)
        return eng.decode(x, L)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(3 + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rec = {"value": round(B * steps / el, 3), "unit": "utt/s", "ms_per_step": round(1e3 * el / steps, 2), "steps": steps,
           "workload": "C2 Libri2Mix-shape, batch %d, N=30 + 1 corrector (60 NFE) + Oobleck decode; score function: NCSN++ "
                       "latent U-Net nf=128 ch_mult (1,2,2) (LatentScoreModelNCSNpp, as wired in the reference)" % B}
    eng.profile_begin()
    step(50)
    prof = eng.profile_end()
    rows = [r for r in roofline_rows(prof) if r["site"].startswith("ncsnpp.")]
    if rows:
        rec["roofline"] = rows[0]
    if parity:
        from oracle import metrics as omet
        from oracle import ncsnpp as oncs
        from oracle import sampler as osmp
        nb = min(2, B)
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), int(os.environ.get("DITSEP_CPU_THREADS", "16"))))
        yc = y[:nb].cpu()
        noise = osmp.draw_noise(4711, 1 + N_STEPS * (CORR + 1), (nb, ncfg.n_src, vcfg.latent_dim, yc.shape[-1]))
        dt, wav_ref = _cpu_run(oncs.NCSNppScore(nsd, ncfg), vsd, vcfg, ncfg, yc, noise, L, N_STEPS)
        nz = torch.randn((noise.shape[0], B) + tuple(noise.shape[2:]), device=dev)
        nz[:, :nb] = noise.to(dev)
        xg, _ = eng.pc_sample(y, nz, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS)
        wg = eng.decode(xg, L)[:nb].cpu()
        rec["parity"] = {"rel_l2_waveform_vs_cpu_fp32": float((wg.double() - wav_ref.double()).norm() / wav_ref.double().norm()),
                         "tolerance": 1e-3, "mixtures": nb, "cpu_seconds": round(dt, 2),
                         "how": f"items 0..{nb - 1} of the timed batch carry the CPU oracle's latents and noise",
                         "si_sdr": "tests/test_gpu_headline.py (random-init NCSN++ estimates are uncorrelated with the "
                                   "synthetic sources; the dB criterion is evaluated at a working separator's operating point)"}
    eng.close()
    return rec


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="mixtures per GPU")
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--alt", default="bf16x3,bf16,fp8", help="comma list of secondary precisions to also measure")
    ap.add_argument("--score", choices=["dit", "ncsnpp"], default="dit",
                    help="dit: the north-star DiT score network (ditsep.json dims); ncsnpp: the NCSN++ the reference wires in")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary-precision measurement")
    ap.add_argument("--no-extra", action="store_true", help="skip the C1 latency / long-form side measurements")
    args = ap.parse_args()

    force_dist = bool(os.environ.get("DITSEP_FORCE_DIST"))     # rehearse the RCCL path with world size 1
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as fresh children (this process has not touched
        # the GPU: torch is not even imported yet) and relay their exit code
        from ditsep_amd import distributed
        import torch
        have = torch.cuda.device_count()                        # counting devices does not initialise the GPU
        if have < args.gpus:
            log(f"--gpus {args.gpus} but only {have} GPU(s) visible: refusing to run fewer ranks than asked for")
            sys.exit(2)
        sys.exit(distributed.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    from ditsep_amd import distributed, synthetic
    PRECISIONS = precisions()
    if args.precision not in PRECISIONS:
        log(f"unknown precision {args.precision}")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
            f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
            f"bench.py --gpus {args.gpus} ...` or plain `python bench.py --gpus {args.gpus}`")
        sys.exit(2)
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    vcfg = synthetic.OobleckConfig()                   # oobleck_finetune.json
    if args.score == "dit":
        dcfg = synthetic.DiTConfig()                   # ditsep.json dims: 1024 x 24 layers x 16 heads
        dsd = synthetic.random_dit_weights(dcfg, 1, out_gain=DIT_OUT_GAIN, skip_gain=DIT_SKIP_GAIN)
        score_desc = "DiT 1024x24x16 (ditsep.json dims) via (xt,t,mix) adapter"
    else:
        dcfg = synthetic.NCSNppConfig()                # latent_diffsep_ouve/model/default.yaml:16-28
        dsd = synthetic.random_ncsnpp_weights(dcfg, 1, out_gain=NCSN_OUT_GAIN)
        score_desc = "NCSN++ latent U-Net nf=128 ch_mult (1,2,2) (LatentScoreModelNCSNpp, as wired in the reference)"
    vsd = synthetic.vae_weights(vcfg, 2, dec_in_gain=DEC_IN_GAIN)
    prec = PRECISIONS[args.precision][0]
    log("weights generated")
    eng = build_engine(local, prec, dcfg, vcfg, dsd, vsd)
    eng.enable_graphs(not args.no_graphs)
    log("engine ready")

    B, L = args.batch, FS * SECONDS
    src = synthetic.synthetic_sources(B, dcfg.n_src, L, FS, seed=1234 + 100000 * rank)
    mix = src.sum(1, keepdim=True).to(dev)
    y = eng.encode(mix, seed=7 + rank)                 # latents resident in HBM before the timed region
    torch.cuda.synchronize()
    log("latents encoded", tuple(y.shape))
    # the one data-path collective: buffers and shard sizes fixed before the timed region
    plan = distributed.ShardPlan(B, dcfg.n_src, L, dev) if dist is not None else None
    n_ranks = dist.get_world_size() if dist is not None else 1

    def step(i, engine=eng):
        def separate_shard(y_shard):                   # this rank's mixtures: sampler + decode
            x, _ = engine.pc_sample(y_shard, None, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS,
                                    denoise=True, seed=(1000 * rank + i) & 0x7FFFFFFF)
            return engine.decode(x, L)
        return distributed.separate_sharded(separate_shard, y, presharded=True, plan=plan)

    def timed(k, w, engine=eng):
        # engine setup, not warm-up: the first call per shape sizes the workspace (eager), the second
        # captures the hipGraphs; W warm-up replays follow, then exactly K timed steps
        for i in range(2):
            step(-1 - i, engine)
        for i in range(w):
            step(i, engine)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            step(w + i, engine)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            # the contract's number is the MAX over ranks; min / max per rank are reported beside it
            every = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(dist.get_world_size())]
            dist.all_gather(every, torch.tensor([el], device=dev, dtype=torch.float64))
            rank_times[:] = [float(t.item()) for t in every]
            el = max(rank_times)
        return el

    rank_times = []

    elapsed = timed(args.steps, args.warmup)
    value = n_ranks * B * args.steps / elapsed
    log(f"timed region: {elapsed:.3f} s for {args.steps} steps -> {value:.2f} utt/s")
    dist_info = None
    if dist is not None:
        # the one data-path collective on its own (waveforms already computed): an upper bound on what a step can expose
        x0, _ = eng.pc_sample(y, None, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS, denoise=True, seed=1)
        w0 = eng.decode(x0, L)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.gather(w0)
        torch.cuda.synchronize()
        g_ms = 1e3 * (time.perf_counter() - t0) / 5
        gt = torch.tensor([g_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        dist_info = {"per_rank_ms_per_step": {"min": round(1e3 * min(rank_times) / args.steps, 2),
                                              "max": round(1e3 * max(rank_times) / args.steps, 2)},
                     "gather_ms_standalone": round(float(gt.item()), 3),
                     "gather_bytes_per_rank": int(w0.numel() * 4),
                     "note": "gather timed alone on finished waveforms (max over ranks): an upper bound on its exposed time"}

    # roofline: per-launch HIP events (on the launch stream) around every profiled launch of one extra step
    eng.profile_begin()
    step(10_000)
    prof = eng.profile_end()
    table = roofline_rows(prof, fp8_sites=("dit.qkv", "dit.attn_out", "dit.ff_in", "dit.ff_out")
                          if args.precision == "fp8" else ())
    fam = prof["gemm_flops"] / (prof["gemm_ms"] * 1e-3) / 1e12
    dom = next((r for r in table if r.get("bound") == "mfma"), None)
    # HBM traffic per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate
    # runs over scripts/pmc_workload.py, gfx950 corrections by scripts/pmc_summary.py), taken at the commit
    # recorded in the file -- not re-measured by this run
    traffic, traffic_source = None, None
    pmc_file = os.path.join(ROOT, "profiles", f"r03_pmc_traffic_{args.precision}_{args.score}.json")
    if os.path.exists(pmc_file) and B == 64 and dom is not None:
        pm = json.load(open(pmc_file))
        site = pm.get("sites", {}).get(dom["site"])
        if site:
            traffic = round(site["hbm_bytes_per_launch"])
            traffic_source = {"file": os.path.relpath(pmc_file, ROOT), "commit": pm.get("commit"),
                              "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over scripts/pmc_workload.py (scripts/pmc_collect.sh), separate passes, gfx950 corrections by scripts/pmc_summary.py; replayed from the file (the bench itself under PMC exceeds the box's silence limit)",
                              "score_call_hbm_bytes": pm.get("score_call_hbm_bytes"),
                              "decode_hbm_bytes": pm.get("decode_hbm_bytes")}
    roofline = {"bound": "mfma", "achieved": None, "peak": PEAK_MFMA_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "traffic": traffic, "traffic_source": traffic_source}
    if dom is not None:
        roofline.update(achieved=dom["achieved"], frac=dom["frac"], peak=dom["peak"],
                        kernel=f"{dom['site']}: {dom['what']}", launches_per_step=dom["launches"],
                        avg_launch_us=dom["avg_launch_us"], gflop_per_launch=dom["gflop_per_launch"])
    roofline["family"] = {"kernel": "implicit-GEMM MFMA family, all launches (igemm_panel / igemm2 / fused ResidualUnit)",
                          "achieved": round(fam, 2), "frac": round(fam / PEAK_MFMA_DENSE_TFLOPS, 4),
                          "launches_per_step": prof["gemm_launches"],
                          "algorithmic_tflop_per_step": round(prof["gemm_flops"] / 1e12, 3),
                          "gemm_ms_per_step": round(prof["gemm_ms"], 2)}
    roofline["table"] = table[:10]
    roofline["note"] = ("per-launch HIP events on the launch stream, graphs bypassed, one extra un-timed step; events add "
                        "~2 us per launch, so `achieved` reads slightly low vs profiles/*_kernel_stats.csv")
    if prof["hbm_launches"]:
        gbs = prof["hbm_bytes"] / (prof["hbm_ms"] * 1e-3) / 1e9
        roofline["secondary"] = {"kernel": "ru_fused kernel (fused Oobleck ResidualUnit, 128 ch)", "bound": "hbm",
                                 "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": prof["hbm_launches"],
                                 "avg_launch_us": round(1e3 * prof["hbm_ms"] / prof["hbm_launches"], 1),
                                 "algorithmic_gb_per_step": round(prof["hbm_bytes"] / 1e9, 2)}

    out = {
        "metric": "separated utterances/sec @ N=30, 2-spk 4 s mixtures",
        "value": round(value, 3), "unit": "utt/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": f"{args.precision} ({PRECISIONS[args.precision][1]})",
        "data": "synthetic",
        "config": {"workload": "C2 Libri2Mix-shape: 2-spk 16 kHz 4 s mixtures, N=30 PC sampler "
                               "(reverse_diffusion + ald, 1 corrector step, 60 NFE) + Oobleck decode, "
                               f"batch={B} per GPU; score function: {score_desc}; "
                               "decoder: Oobleck 128ch x(1,2,4,8,16), strides (2,4,4,8,8), ELU",
                   "global_batch": n_ranks * B, "latent_frames": int(y.shape[-1]), "graphs": not args.no_graphs,
                   "parallelism": f"dp{n_ranks}: batch sharded, one RCCL gather of waveforms per step",
                   "precision_note": "BASELINE C2 names bf16; fp16 operands (same MFMA rate, 11-bit mantissa) are the "
                                     "headline because single-plane bf16 misses the 1e-3 waveform bound (see alt_precision)"},
        "roofline": roofline,
    }
    if dist_info is not None:
        out["distributed"] = dist_info

    if rank == 0 and n_ranks == 1:
        # the same step with the encoder included (the reference times sampler + decode only, so this
        # is a secondary figure, never `value`)
        def full_step(i):
            yy = eng.encode(mix, seed=7 + i)
            xx, _ = eng.pc_sample(yy, None, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS, denoise=True,
                                  seed=5000 + i)
            return eng.decode(xx, L)
        for i in range(2):
            full_step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            full_step(2 + i)
        torch.cuda.synchronize()
        el_full = time.perf_counter() - t0
        out["with_encode"] = {"value": round(B * args.steps / el_full, 3), "unit": "utt/s",
                              "ms_per_step": round(1e3 * el_full / args.steps, 2)}
        # config C1 on the same engine: 8 kHz x 4 s, N = 10, batch 1 -- latency per mixture
        L1 = 8000 * SECONDS
        mix1 = synthetic.synthetic_sources(1, dcfg.n_src, L1, 8000, seed=77).sum(1, keepdim=True).to(dev)
        y1 = eng.encode(mix1, seed=3)
        if not args.no_extra:
            def c1_step(i):
                xx, _ = eng.pc_sample(y1, None, N=10, corrector_steps=CORR, snr=SNR, t_eps=T_EPS, seed=100 + i)
                return eng.decode(xx, L1)
            for i in range(4):
                c1_step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(10):
                c1_step(10 + i)
            torch.cuda.synchronize()
            c1_ms = 1e2 * (time.perf_counter() - t0)
            out["extra"] = {"c1_latency_ms_per_mixture": round(c1_ms, 2),
                            "c1": "config C1: 8 kHz x 4 s (T=16), N=10 + 1 corrector (20 NFE), batch 1, sampler + "
                                  "decode, hipGraph replay, mean of 10"}
            if args.score == "dit":
                out["extra"]["c5_long_form"] = measure_c5(eng, dcfg, dev)
                out["extra"]["ncsnpp"] = measure_ncsnpp(local, dev, prec, vcfg, vsd, mix, src, L, not args.no_graphs,
                                                         parity=not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            log("cpu baseline (oracle) ...")
            cb, c2, c1 = cpu_baseline(dcfg, vcfg, dsd, vsd, y[:min(4, B)], y1)
            log("cpu baseline done:", cb["value"], "utt/s on", cb["cores"], "threads")
            out["cpu_baseline"] = cb
            # live parity of the native path on the very samples the CPU just computed
            from oracle import metrics as omet
            par = {}
            nb = c2["y"].shape[0]
            for name, c, s_ref in (("c2", c2, src[:nb]), ("c1", c1, None)):
                if name == "c2":
                    # the kernels the headline times: the oracle's noise rides as items 0..nb-1 of the FULL timed
                    # batch (mixtures are independent), the other items draw their own; the replayed graph runs
                    nz = torch.randn((c["noise"].shape[0], B) + tuple(c["noise"].shape[2:]), device=dev)
                    nz[:, :nb] = c["noise"].to(dev)
                    xg, _ = eng.pc_sample(y, nz, N=c["N"], corrector_steps=CORR, snr=SNR, t_eps=T_EPS)
                    wg = eng.decode(xg, c["L"])[:nb].cpu()
                    del nz
                else:
                    xg, _ = eng.pc_sample(c["y"], c["noise"], N=c["N"], corrector_steps=CORR, snr=SNR, t_eps=T_EPS)
                    wg = eng.decode(xg, c["L"]).cpu()
                par[name] = float((wg.double() - c["wav"].double()).norm() / c["wav"].double().norm())
                if s_ref is not None:
                    # BASELINE's second metric: |SI-SDR(build, s) - SI-SDR(CPU path, s)| under PIT against the
                    # synthetic sources (random-init weights: only the delta is meaningful)
                    sdr_g, _ = eng.si_sdr_pit(s_ref, wg)
                    sdr_c, _ = omet.si_sdr_pit(s_ref, c["wav"])
                    par["si_sdr_delta_db_vs_cpu_fp32"] = float((sdr_g.mean(-1) - sdr_c).abs().max())
            out["parity"] = {"rel_l2_waveform_vs_cpu_fp32": par["c2"], "rel_l2_waveform_vs_cpu_fp32_c1": par["c1"],
                             "tolerance": 1e-3, "si_sdr_delta_db_vs_cpu_fp32": par["si_sdr_delta_db_vs_cpu_fp32"],
                             "si_sdr_tolerance_db": 0.05, "mixtures": nb,
                             "how": f"items 0..{nb - 1} of the timed batch of {B} (same kernels, same graph as the "
                                    "timed steps) carry the CPU oracle's latents and injected noise",
                             "note": "B=64 score call / N=30 chains (DiT and NCSN++), C4, C5 parity: "
                                     "tests/test_gpu_headline.py, tests/test_gpu_configs.py"}
        if not args.no_alt:
            noise = torch.randn((1 + N_STEPS * (CORR + 1), 4, dcfg.n_src, 64, int(y.shape[-1])), device=dev)
            wa = eng.decode(eng.pc_sample(y[:4], noise, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS)[0], L)
            out["alt_precision"] = []
            for name in [a for a in args.alt.split(",") if a and a != args.precision and a in PRECISIONS]:
                log("secondary precision", name, "...")
                eng2 = build_engine(local, PRECISIONS[name][0], dcfg, vcfg, dsd, vsd)
                eng2.enable_graphs(not args.no_graphs)
                el2 = timed(max(1, args.steps), 3, eng2)
                wb = eng2.decode(eng2.pc_sample(y[:4], noise, N=N_STEPS, corrector_steps=CORR, snr=SNR,
                                                t_eps=T_EPS)[0], L)
                rec = {"dtype": name, "value": round(B * max(1, args.steps) / el2, 3), "unit": "utt/s",
                       "rel_l2_waveform_vs_headline_mode": float((wb.double() - wa.double()).norm()
                                                                 / wa.double().norm())}
                if name == "fp8":
                    if not args.no_extra:
                        rec["c5_long_form"] = measure_c5(eng2, dcfg, dev)
                        # BASELINE config 5 in its named precision: the same 30 s mixture, latent and injected noise
                        # through the headline engine and through this one (N = 30, hipGraph-captured loop)
                        L5 = 30 * FS
                        mix5 = synthetic.synthetic_sources(1, dcfg.n_src, L5, FS, seed=4242).sum(1, keepdim=True).to(dev)
                        y5 = eng.encode(mix5, seed=11)
                        nz5 = torch.randn((1 + N_STEPS * (CORR + 1), 1, dcfg.n_src, 64, int(y5.shape[-1])), device=dev)
                        w16 = eng.decode(eng.pc_sample(y5, nz5, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS)[0], L5)
                        w8 = eng2.decode(eng2.pc_sample(y5, nz5, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS)[0], L5)
                        rec["c5_long_form"]["rel_l2_waveform_vs_headline_mode"] = float(
                            (w8.double() - w16.double()).norm() / w16.double().norm())
                        rec["c5_long_form"]["parity_note"] = ("vs the fp32 CPU oracle (T = 235, N = 3): "
                                                              "tests/test_gpu_headline.py::test_dit_c5_fp8_long_form_chain_parity_figure")
                    eng2.profile_begin()
                    step(20_000, eng2)
                    p2 = eng2.profile_end()
                    rec["roofline_table"] = [r for r in roofline_rows(
                        p2, fp8_sites=("dit.qkv", "dit.attn_out", "dit.ff_in", "dit.ff_out")) if r["site"].startswith("dit.")][:6]
                out["alt_precision"].append(rec)
                eng2.close()
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        out["commit"] = git_head()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
