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
