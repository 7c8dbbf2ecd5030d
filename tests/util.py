"""Shared helpers for the parity tests."""
import torch


def rel_l2(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def max_abs(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def assert_close(name, got, want, tol):
    err = rel_l2(got, want)
    assert torch.isfinite(got.float()).all(), f"{name}: non-finite values in the HIP result"
    assert err <= tol, f"{name}: rel-L2 {err:.3e} > tol {tol:.1e} (max abs {max_abs(got, want):.3e})"
    return err


def fp16_round(x):
    return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


class OperandFormat:
    """The two 16-bit operand formats of the matrix kernels (sat_dit_cfg.gemm_dtype 0 / 3): torch dtype, the rounding the kernels
    apply at their store points, the C-ABI name mapping (sat_*_bf16* -> sat_*_f16*), and the share of a bf16 tolerance that the
    format's unit roundoff leaves (bf16 2^-9, fp16 2^-12: gates for fp16 are the bf16 gates / 4, a 2x margin on the 8x)."""

    def __init__(self, name):
        self.name = name
        self.f16 = name == "f16"
        self.dtype = torch.float16 if self.f16 else torch.bfloat16
        self.round = fp16_round if self.f16 else bf16_round
        self.tol_scale = 0.25 if self.f16 else 1.0
        self.gemm_dtype = "fp16" if self.f16 else "bf16"          # the name set_gemm_dtype / SAT_GEMM_DTYPE use

    def fn(self, lib, name):
        return getattr(lib, name.replace("bf16", "f16") if self.f16 else name)

    def tol(self, bf16_tol):
        return bf16_tol * self.tol_scale

    def __repr__(self):
        return self.name


FORMATS = [OperandFormat("bf16"), OperandFormat("f16")]

# The operand format every test that is not parametrised over FORMATS runs in: the PACKAGE DEFAULT ("fp16", what a user gets) unless
# SAT_TEST_DTYPE=bf16 selects the other build (tools/gpu_session.sh runs the suite under both).  Gates written as T(x) = SUITE.tol(x) are
# the bf16 gates of rounds 1-3 (~2x the error measured on MI355X), divided by 4 under fp16.
import os as _os

_suite_name = _os.environ.get("SAT_TEST_DTYPE", "fp16")
assert _suite_name in ("fp16", "bf16"), f"SAT_TEST_DTYPE must be fp16 or bf16, got {_suite_name!r}"
SUITE = FORMATS[1] if _suite_name == "fp16" else FORMATS[0]
