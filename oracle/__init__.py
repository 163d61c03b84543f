"""CPU oracle for the TrajectoryCrafter denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the reported CPU baseline.  The
product path (``trajectorycrafter_amd``) never imports this package and fails
loudly when the HIP extension is missing.

What it is: a pure-torch, CPU, fp32 restatement of the reference's algorithm for

* ``CrossTransformer3DModel.forward``   (reference models/crosstransformer3d.py:711-871)
* ``AutoencoderKLCogVideoX.decode/encode`` (reference models/autoencoder_magvit.py:1176-1280)
* ``TrajCrafter_Pipeline.__call__``      (reference models/pipeline_trajectorycrafter.py:673-1216)

operating on plain state dicts with the reference's parameter names.

Pinning status (see DESIGN.md "Oracle"):

* The reference's *own* pure-torch classes (patch embeds, CogVideoXBlock control
  flow, PerceiverCrossAttention, CrossTransformer3DModel.forward, every VAE class
  in autoencoder_magvit.py) are pinned: ``tests/golden/make_golden.py`` imports
  them from /root/reference in the build container and the committed fixtures
  under ``tests/golden/`` hold their inputs/outputs.
* The arithmetic that lives in the third-party ``diffusers`` package
  (requirements.txt:26 ``diffusers>=0.30.1``, unpinned, NOT vendored, NOT
  installed) is restated in ``oracle/diffusers_restated.py`` from its published
  algorithm.  The reference holds no tests or golden vectors for it, so that
  part is **parity unpinned**; each restated function carries an analytic
  known-answer test in ``tests/test_oracle_kat.py``.

Precision modes (``oracle.prec.Prec``):

* ``fp32``      – the reference's maths in float32 everywhere.
* ``bf16``      – the *rounding contract* of the HIP path: bf16 storage, fp32
                  arithmetic inside every fused op, one rounding at each tensor
                  the HIP path materialises in HBM.
* ``bf16_ref``  – rounds after every torch op the way the reference's eager
                  bf16 execution does (used to quantify contract-vs-reference
                  deviation; never used as the checker for the kernels).
"""
