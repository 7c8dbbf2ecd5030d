"""k-diffusion pieces are un-vendored third-party code ("parity unpinned", oracle/sampler.py): the restatement
and the product's host-side coefficient form are pinned here by analytic properties of the published
algorithms, and against each other."""
import math

import pytest
import torch

from oracle import sampler as osamp
from stable_audio_tools.inference.sampling import dpmpp2m_coefficients, dpmpp3m_coefficients, get_sigmas_polyexponential


def test_polyexponential_schedule():
    s = osamp.get_sigmas_polyexponential(100, 0.3, 500.0, 1.0)
    assert s.shape == (101,) and s[-1] == 0
    assert abs(s[0].item() - 500.0) < 1e-3 and abs(s[99].item() - 0.3) < 1e-6
    r = (s[1:100] / s[:99])      # rho = 1 -> geometric
    assert (r.max() - r.min()).item() < 1e-5
    assert torch.allclose(torch.tensor(get_sigmas_polyexponential(100, 0.3, 500.0, 1.0)), s)
    s2 = osamp.get_sigmas_polyexponential(7, 0.5, 50.0, 2.0)
    assert all(s2[i] > s2[i + 1] for i in range(7))


def test_vdenoiser_scalings():
    sig = torch.tensor([0.1, 1.0, 30.0])
    c_skip, c_out, c_in = osamp.vdenoiser_scalings(sig)
    assert torch.allclose(c_skip + c_out ** 2, torch.ones(3))           # 1/(s^2+1) + s^2/(s^2+1) = 1
    assert torch.allclose(c_in ** 2, c_skip)
    assert torch.allclose(osamp.sigma_to_t(torch.tensor([1.0])), torch.tensor([0.5]))
    # a model that predicts v exactly for x = x0 + sigma*n recovers x0:  v = (n - sigma*x0) / sqrt(1+sigma^2)  (alpha-sigma form)
    x0, n = torch.randn(3, 4, 5), torch.randn(3, 4, 5)
    s = sig.view(-1, 1, 1)
    x = x0 + s * n
    alpha, sg = 1 / (1 + s * s).sqrt(), s / (1 + s * s).sqrt()
    v = alpha * n - sg * x0
    den = osamp.vdenoise(lambda xin, t: v, x, sig)
    assert torch.allclose(den, x0, atol=1e-5)


def _gaussian_denoiser(s_data):
    return lambda x, sigma: x * (s_data ** 2 / (s_data ** 2 + sigma.view(-1, 1, 1) ** 2))


def test_dpmpp3m_deterministic_limit_matches_probability_flow():
    """eta = 0: DPM-Solver++(3M); for data ~ N(0, s^2) the exact ODE solution is x(sigma) = x0 sqrt((s^2+sigma^2)/(s^2+sigma0^2))."""
    s_data = 0.7
    sig = osamp.get_sigmas_polyexponential(200, 0.05, 80.0, 1.0)
    x0 = torch.randn(2, 3, 50, dtype=torch.float64) * 80.0
    den = _gaussian_denoiser(s_data)
    out = osamp.sample_dpmpp_3m_sde(den, x0.clone(), sig[:-1].double(), lambda i, a, b: torch.zeros_like(x0), eta=0.0)
    exact = x0 * math.sqrt((s_data ** 2 + sig[-2].item() ** 2) / (s_data ** 2 + sig[0].item() ** 2))
    assert ((out - exact).norm() / exact.norm()).item() < 1e-4


def test_dpmpp3m_sde_keeps_the_marginal_variance():
    """eta = 1 on the Gaussian toy problem: the marginal std at the last positive sigma must be sqrt(s^2 + sigma^2)."""
    s_data = 1.3
    steps = 60
    sig = osamp.get_sigmas_polyexponential(steps, 0.3, 100.0, 1.0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 16, 256, generator=g) * sig[0]
    out = osamp.sample_dpmpp_3m_sde(_gaussian_denoiser(s_data), x, sig[:-1], lambda i, a, b: torch.randn(x.shape, generator=g))
    want = math.sqrt(s_data ** 2 + sig[-2].item() ** 2)
    assert abs(out.std().item() / want - 1.0) < 0.02


def test_last_step_returns_denoised_and_history_orders():
    sig = torch.tensor([2.0, 1.0, 0.0])
    calls = []

    def den(x, s):
        calls.append(float(s[0]))
        return x * 0.5 + 1.0

    x = torch.ones(1, 1, 4)
    out = osamp.sample_dpmpp_3m_sde(den, x, sig, lambda i, a, b: torch.zeros_like(x))
    assert calls == [2.0, 1.0]                       # one model evaluation per step
    h = math.log(2.0)
    x1 = math.exp(-2 * h) * 1.0 + (1 - math.exp(-2 * h)) * 1.5       # first step: 1st-order, eta=1, zero noise
    assert torch.allclose(out, torch.full_like(x, x1 * 0.5 + 1.0))  # sigma_next == 0 -> x = denoised


@pytest.mark.parametrize("eta", [1.0, 0.0, 0.5])
def test_fused_coefficient_form_equals_multistep_form(eta):
    """Host logic of the product: x <- a x + b D + c1 (D-D1) + c2 (D1-D2) + cn noise  ==  oracle's k-diffusion form."""
    steps = 25
    sig_list = get_sigmas_polyexponential(steps, 0.3, 500.0, 1.0)
    sig = torch.tensor(sig_list, dtype=torch.float64)
    coeffs = dpmpp3m_coefficients(sig_list, eta=eta)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(2, 4, 33, generator=g, dtype=torch.float64) * 500
    tgt = torch.randn(2, 4, 33, generator=g, dtype=torch.float64)
    noises = [torch.randn(2, 4, 33, generator=g, dtype=torch.float64) for _ in range(steps)]

    def den(x, s):
        s = s.view(-1, 1, 1)
        return tgt + (x - tgt) / (1 + s * s) + 0.05 * torch.sin(x / (1 + s))

    want = osamp.sample_dpmpp_3m_sde(den, x0.clone(), sig, lambda i, a, b: noises[i], eta=eta)
    x = x0.clone()
    d1 = d2 = None
    for i in range(steps):
        d = den(x, sig[i] * torch.ones(2, dtype=torch.float64))
        a, b, c1, c2, cn = coeffs[i]
        new = a * x + b * d
        if d1 is not None:
            new = new + c1 * (d - d1)
            if d2 is not None:
                new = new + c2 * (d1 - d2)
        assert (d1 is not None) or c1 == 0.0
        assert (d2 is not None) or c2 == 0.0
        new = new + cn * noises[i]
        x, d1, d2 = new, d, d1
    assert ((x - want).norm() / want.norm()).item() < 1e-12
    assert coeffs[-1] == (0.0, 1.0, 0.0, 0.0, 0.0)


@pytest.mark.parametrize("eta", [0.0, 1.0])
@pytest.mark.parametrize("solver_type", ["midpoint", "heun"])
def test_dpmpp2m_fused_form_equals_multistep_form(eta, solver_type):
    """dpmpp-2m-sde (the reference's default sampler_type): the product's host scalars (a, b, c1, 0, cn) reproduce the
    restated multistep recursion on an arbitrary smooth denoiser with injected noise."""
    steps = 17
    sig = osamp.get_sigmas_polyexponential(steps, 0.4, 120.0, 1.0).double()
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(2, 3, 11, generator=g, dtype=torch.float64) * sig[0]
    target = torch.randn(2, 3, 11, generator=g, dtype=torch.float64)
    noises = [torch.randn(2, 3, 11, generator=g, dtype=torch.float64) for _ in range(steps)]
    den = lambda x, s: target + (x - target) / (1 + s.view(-1, 1, 1) ** 2) + 0.1 * torch.tanh(x / (1 + s.view(-1, 1, 1)))
    want = osamp.sample_dpmpp_2m_sde(den, x0.clone(), sig, lambda i, a, b: noises[i], eta=eta, solver_type=solver_type)
    coeffs = dpmpp2m_coefficients([float(v) for v in sig], eta=eta, solver_type=solver_type)
    x, d1 = x0.clone(), None
    for i in range(steps):
        d = den(x, sig[i].expand(2))
        a, b, c1, c2, cn = coeffs[i]
        assert c2 == 0.0
        x = a * x + b * d + (c1 * (d - d1) if d1 is not None else 0) + cn * noises[i]
        d1 = d
    assert ((x - want).norm() / want.norm()).item() < 1e-12


def test_dpmpp2m_deterministic_limit_matches_probability_flow():
    s_data = 0.7
    sig = osamp.get_sigmas_polyexponential(300, 0.05, 80.0, 1.0)
    x0 = torch.randn(2, 3, 50, dtype=torch.float64) * 80.0
    out = osamp.sample_dpmpp_2m_sde(_gaussian_denoiser(s_data), x0.clone(), sig[:-1].double(), lambda i, a, b: torch.zeros_like(x0), eta=0.0)
    exact = x0 * math.sqrt((s_data ** 2 + sig[-2].item() ** 2) / (s_data ** 2 + sig[0].item() ** 2))
    assert ((out - exact).norm() / exact.norm()).item() < 1e-3


def test_inpainting_mask_schedule_and_build_mask():
    """build_mask against the reference's own outputs (tests/golden/host.npz) and the shrinking binary mask."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from stable_audio_tools.inference.generation import build_mask
    from stable_audio_tools.inference.sampling import get_bmask_strength
    g = cases.load("host")
    for i, ma in enumerate(cases.MASK_ARGS):
        assert torch.equal(build_mask(1024, ma), g[f"mask_{i}"]), f"mask_args case {i}"
    m = build_mask(1024, cases.MASK_ARGS[0])
    steps = 10
    kept = [int(osamp.get_bmask(i, steps, m).sum()) for i in range(steps)]
    assert kept == sorted(kept) and kept[-1] == 1024          # the keep-region only grows; the last step keeps everything
    assert all(torch.equal(osamp.get_bmask(i, steps, m), (m <= get_bmask_strength(i, steps)).long()) for i in range(steps))


def test_single_step_samplers_converge_to_the_probability_flow_solution():
    """k-heun / k-dpm-2 / k-lms / k-dpmpp-2s-ancestral(eta=0) / k-dpm-fast restatements (k-diffusion is un-vendored): on the
    Gaussian toy problem every one must converge to the exact ODE solution at its order (2, 2, 4, 2, ~3)."""
    s_data = 0.7
    den = _gaussian_denoiser(s_data)
    errs = {}
    for steps in (50, 100):
        sig = osamp.get_sigmas_polyexponential(steps, 0.05, 80.0, 1.0).double()
        x0 = torch.randn(2, 3, 20, dtype=torch.float64, generator=torch.Generator().manual_seed(1)) * 80.0
        exact = x0 * math.sqrt((s_data ** 2 + sig[-2].item() ** 2) / (s_data ** 2 + sig[0].item() ** 2))
        exact_f = x0 * math.sqrt((s_data ** 2 + 0.05 ** 2) / (s_data ** 2 + 80.0 ** 2))
        zero = lambda i, a, b: torch.zeros_like(x0)
        out = {"heun": osamp.sample_heun(den, x0.clone(), sig[:-1]), "dpm2": osamp.sample_dpm_2(den, x0.clone(), sig[:-1]),
               "lms": osamp.sample_lms(den, x0.clone(), sig[:-1]),
               "2s": osamp.sample_dpmpp_2s_ancestral(den, x0.clone(), sig[:-1], zero, eta=0.0)}
        errs[steps] = {k: ((v - exact).norm() / exact.norm()).item() for k, v in out.items()}
        errs[steps]["fast"] = ((osamp.sample_dpm_fast(den, x0.clone(), 0.05, 80.0, steps) - exact_f).norm() / exact_f.norm()).item()
    for k, order in (("heun", 2), ("dpm2", 2), ("2s", 2), ("lms", 3.5)):
        assert errs[100][k] < 1e-3 and errs[50][k] / errs[100][k] > 2 ** order * 0.8, (k, errs)
    assert errs[100]["fast"] < 5e-4


def test_ancestral_sampler_keeps_the_marginal_variance():
    """k-dpmpp-2s-ancestral with eta = 1 on the Gaussian toy problem: std at the last positive sigma = sqrt(s^2 + sigma^2)."""
    s_data, steps = 1.3, 80
    sig = osamp.get_sigmas_polyexponential(steps, 0.3, 100.0, 1.0).double()
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(64, 8, 256, generator=g, dtype=torch.float64) * math.sqrt(s_data ** 2 + 100.0 ** 2)
    out = osamp.sample_dpmpp_2s_ancestral(_gaussian_denoiser(s_data), x0, sig[:-1], lambda i, a, b: torch.randn(x0.shape, generator=g, dtype=torch.float64))
    want = math.sqrt(s_data ** 2 + sig[-2].item() ** 2)
    assert abs(out.std().item() / want - 1) < 2e-2
    sd, su = osamp.get_ancestral_step(2.0, 1.0)
    assert abs(sd ** 2 + su ** 2 - 1.0) < 1e-12


def test_lms_coefficients_exact_vs_quadrature():
    """The product integrates the Lagrange basis exactly; k-diffusion (and the oracle) use scipy quad with epsrel 1e-4."""
    from stable_audio_tools.inference.sampling import get_ancestral_step, lms_coefficient
    sig = [float(v) for v in osamp.get_sigmas_polyexponential(12, 0.3, 80.0, 1.0)]
    for i in range(12):
        cur = min(i + 1, 4)
        for j in range(cur):
            a, b = lms_coefficient(cur, sig, i, j), osamp.linear_multistep_coeff(cur, sig, i, j)
            assert abs(a - b) <= 2e-4 * max(abs(a), abs(b), 1e-9), (i, j, a, b)
    assert get_ancestral_step(2.0, 1.0) == pytest.approx(osamp.get_ancestral_step(2.0, 1.0))
    assert get_ancestral_step(2.0, 0.0)[0] == 0.0


def test_dpm_adaptive_controls_the_error():
    """k-dpm-adaptive restatement: on the Gaussian toy problem it must land on the exact probability-flow solution at
    sigma_min within a few rtol, take fewer steps at a looser tolerance, and reject steps when started with a huge h."""
    s_data = 0.7
    den = _gaussian_denoiser(s_data)
    x0 = torch.randn(2, 3, 40, dtype=torch.float64, generator=torch.Generator().manual_seed(5)) * 80.0
    exact = x0 * math.sqrt((s_data ** 2 + 0.05 ** 2) / (s_data ** 2 + 80.0 ** 2))
    info_t, info_l, info_h = {}, {}, {}
    tight = osamp.sample_dpm_adaptive(den, x0.clone(), 0.05, 80.0, rtol=0.01, atol=0.01, info=info_t)
    loose = osamp.sample_dpm_adaptive(den, x0.clone(), 0.05, 80.0, rtol=0.2, atol=0.2, info=info_l)
    err = lambda v: ((v - exact).norm() / exact.norm()).item()
    tighter = osamp.sample_dpm_adaptive(den, x0.clone(), 0.05, 80.0, rtol=0.001, atol=0.001)
    assert err(tight) < 5e-2 and err(tighter) < 0.3 * err(tight) and err(loose) > err(tight)    # local control, global error follows
    assert info_l["steps"] < info_t["steps"] and info_t["nfe"] == 3 * info_t["steps"]
    osamp.sample_dpm_adaptive(den, x0.clone(), 0.05, 80.0, h_init=5.0, info=info_h)
    assert info_h["n_reject"] >= 1
