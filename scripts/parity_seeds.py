"""Full-size C2 parity of the fp16 mode over several weight / mixture / noise seeds (development aid): the native
path and the CPU oracle start from the same encoded synthetic mixture; waveform rel-L2 and SI-SDR delta."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ditsep_amd import native, synthetic
from oracle import dit as odit, metrics as omet, oobleck as ovae, sampler as osmp
torch.set_num_threads(16)
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
L = bench.FS * bench.SECONDS
for wseed in (1, 11):
    dsd = synthetic.random_dit_weights(dcfg, wseed, out_gain=bench.DIT_OUT_GAIN, skip_gain=bench.DIT_SKIP_GAIN)
    vsd = synthetic.vae_weights(vcfg, wseed + 1, dec_in_gain=bench.DEC_IN_GAIN)
    eng = bench.build_engine(0, native.PREC_FP16, dcfg, vcfg, dsd, vsd)
    for nseed in (99, 7):
        src = synthetic.synthetic_sources(1, dcfg.n_src, L, bench.FS, seed=nseed)
        y = eng.encode(src.sum(1, keepdim=True), seed=nseed).cpu()
        g = torch.Generator().manual_seed(nseed)
        noise = osmp.draw_noise(g, 61, (1, 2, 64, y.shape[-1]))
        x, _ = osmp.pc_sample(odit.DiTScore(dsd, dcfg), y, noise, osmp.OUVE(N=30), eps=0.03, snr=0.5, corrector_steps=1)
        wav = ovae.decode_sources(vsd, vcfg, x, L, "decoder.")
        xg, _ = eng.pc_sample(y, noise, N=30, corrector_steps=1, snr=0.5, t_eps=0.03)
        wg = eng.decode(xg, L).cpu()
        rx = float((xg.cpu().double() - x.double()).norm() / x.double().norm())
        rw = float((wg.double() - wav.double()).norm() / wav.double().norm())
        dsdr = float((eng.si_sdr_pit(src, wg)[0].mean(-1) - omet.si_sdr_pit(src, wav)[0]).abs().max())
        print(f"weights seed {wseed} mixture/noise seed {nseed}: latent rel-L2 {rx:.3e}  waveform rel-L2 {rw:.3e}  "
              f"SI-SDR delta {dsdr:.4f} dB", flush=True)
    eng.close()
