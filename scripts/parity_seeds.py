"""Full-size C2 parity of the fp16 mode over several seeds (development aid): waveform rel-L2 vs the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ditsep_amd import native, synthetic
torch.set_num_threads(16)
dcfg, vcfg = synthetic.DiTConfig(), synthetic.OobleckConfig()
for wseed in (1, 11):
    dsd = synthetic.random_dit_weights(dcfg, wseed, out_gain=bench.DIT_OUT_GAIN, skip_gain=bench.DIT_SKIP_GAIN)
    vsd = synthetic.vae_weights(vcfg, wseed + 1, dec_in_gain=bench.DEC_IN_GAIN)
    eng = bench.build_engine(0, native.PREC_FP16, dcfg, vcfg, dsd, vsd)
    for nseed in (99, 7):
        from oracle import dit as odit, oobleck as ovae, sampler as osmp
        g = torch.Generator().manual_seed(nseed)
        y = torch.randn((1, 1, 64, 32), generator=g)
        noise = osmp.draw_noise(g, 61, (1, 2, 64, 32))
        x, _ = osmp.pc_sample(odit.DiTScore(dsd, dcfg), y, noise, osmp.OUVE(N=30), eps=0.03, snr=0.5, corrector_steps=1)
        wav = ovae.decode_sources(vsd, vcfg, x, 64000, "decoder.")
        xg, _ = eng.pc_sample(y, noise, N=30, corrector_steps=1, snr=0.5, t_eps=0.03)
        wg = eng.decode(xg, 64000).cpu()
        rx = float((xg.cpu().double() - x.double()).norm() / x.double().norm())
        rw = float((wg.double() - wav.double()).norm() / wav.double().norm())
        print(f"weights seed {wseed} noise seed {nseed}: latent rel-L2 {rx:.3e}  waveform rel-L2 {rw:.3e}", flush=True)
    eng.close()
