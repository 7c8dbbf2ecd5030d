"""Package-wide default of the 16-bit operand format of the matrix kernels (DiT GEMMs / attention, Oobleck convolutions).

"fp16" (default since round 4): IEEE fp16 operands, fp32 accumulation -- the arithmetic the reference itself runs on a GPU
(``torch.cuda.amp.autocast`` in ``inference/sampling.py:210``, fp16 flash attention in ``models/transformer.py:496-504``,
``model_half`` in ``models/pretransforms.py:39-59``) and the one with which this build meets the 1e-3 rel-L2 of its target against the
reference's fp32 outputs (full-size DiT 4.9e-4, codec 7e-4).  "bf16": the same kernels on the bf16 MFMAs -- 3-4 % faster end to end on
MI355X (the matrix pipes are power-limited and fp16 operands toggle three more mantissa bits) at 8x the operand rounding error
(4e-3 per DiT forward).  Per model: ``DiffusionTransformer.set_gemm_dtype`` / ``AudioAutoencoder.set_gemm_dtype``; process-wide:
``set_default_gemm_dtype`` before the models are built, or the environment variable ``SAT_GEMM_DTYPE``.
"""
import os

_CHOICES = ("fp16", "bf16")
_default = os.environ.get("SAT_GEMM_DTYPE", "fp16")
if _default not in _CHOICES:
    raise ValueError(f"SAT_GEMM_DTYPE must be one of {_CHOICES}, got {_default!r}")


def default_gemm_dtype() -> str:
    return _default


def set_default_gemm_dtype(dtype: str) -> str:
    """Sets the operand format newly built models start with; returns the previous one."""
    global _default
    if dtype not in _CHOICES:
        raise ValueError(f"default gemm dtype must be one of {_CHOICES}")
    prev, _default = _default, dtype
    return prev


def codec_gemm_dtype(dit_gemm_dtype: str) -> str:
    """The operand format the Oobleck codec runs in next to a DiT in ``dit_gemm_dtype`` -- ONE rule for generate.py, bench.py and the tests:
    "fp16" -> "fp16" (the reference's ``model_half``), "bf16" -> "bf16", the e4m3 modes -> "bf16" (what their description promises: everything
    that is not e4m3 stays bf16), "fp32x" -> "fp16" (there is no fp32 codec build; the verification mode is about the DiT)."""
    return {"fp16": "fp16", "bf16": "bf16", "fp8": "bf16", "fp8-all": "bf16", "fp32x": "fp16"}[dit_gemm_dtype]
