#!/usr/bin/env python3
"""Headline benchmark: separated utterances/sec on the BASELINE.json C2 workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" = one pass of the hot path over one batch of synthetic Libri2Mix-shape
mixtures already resident in HBM as latents: N=30 predictor-corrector sampler
(1 corrector step, 60 score-network calls, on-device Philox noise) followed by the
Oobleck decode to waveforms -- the region the reference times
(src/evaluate_latent.py:273-277).  Mixtures are independent: each rank owns a
64-mixture shard (weak scaling, BASELINE C3 = 8 x 64) and the only collective is
one RCCL gather of the separated waveforms to rank 0 per step.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ditsep_amd import native, synthetic  # noqa: E402

def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


PRECISIONS = {"bf16": (native.PREC_BF16, "bf16 MFMA operands, fp32 accumulate"),
              "bf16x3": (native.PREC_BF16X3, "split-bf16 (hi,lo) MFMA operands x3, fp32 accumulate"),
              "fp16": (native.PREC_FP16, "fp16 MFMA operands, fp32 accumulate"),
              "fp16x3": (native.PREC_FP16X3, "split-fp16 (hi,lo) MFMA operands x3, fp32 accumulate")}
PEAK_BF16_DENSE_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FS, SECONDS, N_STEPS, CORR, SNR, T_EPS = 16000, 4, 30, 1, 0.5, 0.03
DIT_OUT_GAIN, DIT_SKIP_GAIN, DEC_IN_GAIN, NCSN_OUT_GAIN = 0.002, 0.02, 0.08, 0.01


def build_engine(device, precision, dcfg, vcfg, dsd, vsd):
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


def cpu_baseline(dcfg, vcfg, dsd, vsd, L, n_mix, y=None):
    """The CPU restatement of the reference path (oracle/, kind = "port") timed on
    this host: sampler + decode on `n_mix` mixtures of the same workload."""
    from oracle import dit as odit
    from oracle import ncsnpp as oncs
    from oracle import oobleck as ovae
    from oracle import sampler as osmp

    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("DITSEP_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    T = (L + (vcfg.hop - L % vcfg.hop)) // vcfg.hop
    g = torch.Generator().manual_seed(99)
    y = torch.randn((n_mix, 1, vcfg.latent_dim, T), generator=g) if y is None else y.detach().cpu().float()
    noise = osmp.draw_noise(g, 1 + N_STEPS * (CORR + 1), (n_mix, dcfg.n_src, vcfg.latent_dim, T))
    score = odit.DiTScore(dsd, dcfg) if isinstance(dcfg, synthetic.DiTConfig) else oncs.NCSNppScore(dsd, dcfg)
    t0 = time.perf_counter()
    x, nfe = osmp.pc_sample(score, y, noise, osmp.OUVE(N=N_STEPS), eps=T_EPS, snr=SNR, corrector_steps=CORR,
                            denoise=True, n_spkrs=dcfg.n_src)
    wav = ovae.decode_sources(vsd, vcfg, x, L, "decoder.")
    dt = time.perf_counter() - t0
    return {"value": n_mix / dt, "unit": "utt/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_mix} mixtures of the same C2 workload (N=30, 60 NFE, sampler+decode), "
                      f"PyTorch-CPU fp32 oracle, {dt:.1f} s"}, y, noise, wav


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="mixtures per GPU")
    ap.add_argument("--precision", choices=list(PRECISIONS), default="fp16")
    ap.add_argument("--alt", default="bf16x3,bf16", help="comma list of secondary precisions to also measure")
    ap.add_argument("--score", choices=["dit", "ncsnpp"], default="dit",
                    help="dit: the north-star DiT score network (ditsep.json dims); ncsnpp: the NCSN++ the reference wires in")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="issue decode(i) and sampler(i+1) on separate streams (measured: no gain, off by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary-precision measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("DITSEP_FORCE_DIST"):     # DITSEP_FORCE_DIST: rehearse the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
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
    gather_buf = None
    if dist is not None and rank == 0:
        gather_buf = [torch.empty((B, dcfg.n_src, L), device=dev) for _ in range(world)]

    # Two HIP streams: the sampler of batch i+1 (MFMA / L2 bound) is issued while the decode of batch i
    # (HBM bound) and its gather are still running -- batches are independent, every step still does
    # all of its work inside the timed region (both streams are drained before the clock stops).
    s_samp, s_dec = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    pipelined = args.pipeline

    def step(i, engine=eng):
        if not pipelined:
            x, nfe = engine.pc_sample(y, None, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS,
                                      denoise=True, seed=(1000 * rank + i) & 0x7FFFFFFF)
            wav = engine.decode(x, L)
            if dist is not None:
                dist.gather(wav, gather_buf, dst=0)
            return wav, nfe
        with torch.cuda.stream(s_samp):
            x, nfe = engine.pc_sample(y, None, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS,
                                      denoise=True, seed=(1000 * rank + i) & 0x7FFFFFFF)
            ready = s_samp.record_event()
        s_dec.wait_event(ready)
        with torch.cuda.stream(s_dec):
            x.record_stream(s_dec)
            wav = engine.decode(x, L)
            if dist is not None:
                dist.gather(wav, gather_buf, dst=0)
        return wav, nfe

    def timed(k, w, engine=eng):
        # engine setup, not warm-up: the first call per shape sizes the workspace (eager), the second
        # captures the hipGraphs; W warm-up replays follow, then exactly K timed steps
        for i in range(2):
            step(-1 - i, engine)
        for i in range(w):
            step(i, engine)
        torch.cuda.synchronize()          # device-wide: drains both pipeline streams
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
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    elapsed = timed(args.steps, args.warmup)
    value = world * B * args.steps / elapsed
    log(f"timed region: {elapsed:.3f} s for {args.steps} steps -> {value:.2f} utt/s")

    # dominant kernel roofline: per-launch HIP events around every implicit-GEMM launch of one step
    eng.profile_begin()
    step(10_000)
    prof = eng.profile_end()
    ach = prof["gemm_flops"] / (prof["gemm_ms"] * 1e-3) / 1e12
    # HBM traffic of the same kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs over scripts/pmc_workload.py, gfx950 corrections applied by scripts/pmc_summary.py),
    # weighted like one bench step: N_STEPS*(CORR+1) score calls + one decode.
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_{args.precision}_{args.score}.json")
    if os.path.exists(pmc_file) and B == 64:
        pm = json.load(open(pmc_file))
        nfe = N_STEPS * (CORR + 1)
        launches = nfe * pm["igemm_launches_per_score_call"] + pm["igemm_launches_per_decode"]
        traffic = round((nfe * pm["score_call_hbm_bytes"] + pm["decode_hbm_bytes"]) / launches)
    roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": traffic,
                "kernel": "igemm_panel_kernel / igemm2_kernel (implicit-GEMM MFMA family, all launches, %s)" % args.precision, "launches_per_step": prof["gemm_launches"],
                "avg_launch_us": round(1e3 * prof["gemm_ms"] / max(1, prof["gemm_launches"]), 2),
                "algorithmic_tflop_per_step": round(prof["gemm_flops"] / 1e12, 3),
                "gemm_ms_per_step": round(prof["gemm_ms"], 2)}
    if prof["hbm_launches"]:
        # second-largest kernel class: the fused ResidualUnit of the 128-channel decoder layers, HBM bound
        gbs = prof["hbm_bytes"] / (prof["hbm_ms"] * 1e-3) / 1e9
        roofline["secondary"] = {"kernel": "ru_fused2_kernel (fused Oobleck ResidualUnit, 128 ch)", "bound": "hbm",
                                 "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(gbs / 8000.0, 4), "launches_per_step": prof["hbm_launches"],
                                 "avg_launch_us": round(1e3 * prof["hbm_ms"] / prof["hbm_launches"], 1),
                                 "algorithmic_gb_per_step": round(prof["hbm_bytes"] / 1e9, 2)}

    out = {
        "metric": "separated utterances/sec @ N=30, 2-spk 4 s mixtures",
        "value": round(value, 3), "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": f"{args.precision} ({PRECISIONS[args.precision][1]})",
        "data": "synthetic",
        "config": {"workload": "C2 Libri2Mix-shape: 2-spk 16 kHz 4 s mixtures, N=30 PC sampler "
                               "(reverse_diffusion + ald, 1 corrector step, 60 NFE) + Oobleck decode, "
                               f"batch={B} per GPU; score function: {score_desc}; "
                               "decoder: Oobleck 128ch x(1,2,4,8,16), strides (2,4,4,8,8), ELU",
                   "global_batch": world * B, "latent_frames": int(y.shape[-1]), "graphs": not args.no_graphs,
                   "pipelined_decode": pipelined,
                   "parallelism": f"dp{world}: batch sharded, one RCCL gather of waveforms per step"},
        "roofline": roofline,
    }

    if rank == 0 and world == 1:
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
        if not args.no_cpu_baseline:
            n_cpu = 2
            log("cpu baseline (oracle) ...")
            # the CPU path starts from the same encoded latents of the first synthetic mixtures
            cb, y_c, noise_c, wav_c = cpu_baseline(dcfg, vcfg, dsd, vsd, L, n_cpu, y=y[:n_cpu])
            log("cpu baseline done:", cb["value"], "utt/s on", cb["cores"], "threads")
            out["cpu_baseline"] = cb
            # live parity of the native path on the very sample the CPU just computed
            xg, _ = eng.pc_sample(y_c, noise_c, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS)
            wg = eng.decode(xg, L).cpu()
            # BASELINE's second metric: |SI-SDR(build, s) - SI-SDR(CPU path, s)| per mixture under PIT against the
            # synthetic sources s (random-init weights: the absolute SI-SDR means nothing, the delta is the gate)
            from oracle import metrics as omet
            sdr_g, _ = eng.si_sdr_pit(src[:n_cpu], wg)
            sdr_c, _ = omet.si_sdr_pit(src[:n_cpu], wav_c)
            out["parity"] = {"rel_l2_waveform_vs_cpu_fp32": float((wg.double() - wav_c.double()).norm()
                                                                   / wav_c.double().norm()),
                             "tolerance": 1e-3,
                             "si_sdr_delta_db_vs_cpu_fp32": float((sdr_g.mean(-1) - sdr_c).abs().max()),
                             "si_sdr_tolerance_db": 0.05, "mixtures": n_cpu}
        if not args.no_alt:
            noise = torch.randn((1 + N_STEPS * (CORR + 1), 4, dcfg.n_src, 64, int(y.shape[-1])), device=dev)
            wa = eng.decode(eng.pc_sample(y[:4], noise, N=N_STEPS, corrector_steps=CORR, snr=SNR, t_eps=T_EPS)[0], L)
            out["alt_precision"] = []
            for name in [a for a in args.alt.split(",") if a and a != args.precision]:
                log("secondary precision", name, "...")
                eng2 = build_engine(local, PRECISIONS[name][0], dcfg, vcfg, dsd, vsd)
                eng2.enable_graphs(not args.no_graphs)
                el2 = timed(max(1, args.steps), 3, eng2)
                wb = eng2.decode(eng2.pc_sample(y[:4], noise, N=N_STEPS, corrector_steps=CORR, snr=SNR,
                                                t_eps=T_EPS)[0], L)
                out["alt_precision"].append({
                    "dtype": name, "value": round(B * max(1, args.steps) / el2, 3), "unit": "utt/s",
                    "rel_l2_waveform_vs_headline_mode": float((wb.double() - wa.double()).norm()
                                                              / wa.double().norm())})
                eng2.close()
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
