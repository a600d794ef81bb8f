"""CPU oracle for the latent-diffusion separation hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU fp32 restatement
of the reference algorithm (eduardburlacu/DiTSep) for the path
``LatentDiffSep.separate()``: OUVE predictor-corrector sampler, score network
(DiT adapter and NCSN++), Oobleck VAE decoder/encoder, SI-SDR/PIT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker -- the product path in
``ditsep_amd/`` never imports ``oracle`` and fails loudly if the HIP library is
missing.

Parity pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against the reference classes
themselves, imported in the build container with the stub recipe of SURVEY.md
section 8c by ``oracle/make_golden.py``; the resulting input/output vectors are
committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``
without the reference present.
"""
