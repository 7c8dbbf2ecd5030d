"""CPU oracle for the diffusion-sampling hot path.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

This package is a plain PyTorch fp32 *restatement* (functional, over a reference-format
state dict) of the reference's algorithm for the path named in BASELINE.json
(`north_star`): the ContinuousTransformer/DiT denoiser, the k-diffusion
DPM-Solver++(3M) SDE loop around it and the Oobleck VAE encode/decode.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the *checker* / reported baseline.  The product package
(``friendly-stable-audio-tools_amd/stable_audio_tools``) never imports it and has no CPU
fallback: it raises when the HIP extension is missing.

Pinning status
--------------
* The arithmetic here is floating point (the reference computes in fp32 on CPU), so the
  oracle is torch fp32, not C/numpy.
* The reference holds NO tests, golden vectors or fixtures for this path (SURVEY.md
  section 4).  The oracle is therefore pinned against outputs of the reference itself, run
  in the build container by ``tests/golden/make_golden.py`` (imports ``/root/reference``
  with ``sys.modules`` placeholders for un-installed third-party packages) and committed
  as small fixtures under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every
  oracle function against them.
* **Parity unpinned** at two third-party boundaries that are absent from
  ``/root/reference`` and from this image: ``k-diffusion==0.1.1`` (+ ``torchsde``) --
  ``oracle/sampler.py`` restates the published DPM-Solver++(3M) SDE / VDenoiser /
  polyexponential-schedule algorithms and is pinned only by analytic checks -- and
  ``descript-audio-codec==1.0.0`` whose ``WNConv1d`` is, by its published definition,
  ``torch.nn.utils.weight_norm(nn.Conv1d)``.
"""
