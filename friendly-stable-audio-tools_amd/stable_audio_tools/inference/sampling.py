"""``sample_k`` (reference ``inference/sampling.py:144-228``): the multistep SDE samplers ``dpmpp-3m-sde`` /
``dpmpp-2m-sde`` (fused per-step update), the single-step k-diffusion samplers ``k-heun``, ``k-lms``,
``k-dpmpp-2s-ancestral``, ``k-dpm-2``, ``k-dpm-fast``, ``k-dpm-adaptive`` (generic linear-combination update), plain sampling, variations
(``init_data``) and inpainting (``init_data`` + soft ``mask``); ``sample_rf`` / ``sample_discrete_euler`` for
rectified-flow models (:28-60, :236-270).

The reference delegates to the un-vendored ``k-diffusion==0.1.1``: ``VDenoiser`` (:159),
``get_sigmas_polyexponential`` (:165), ``sample_dpmpp_3m_sde`` (:228).  Here the host computes
the per-step scalars of the published DPM-Solver++(3M) SDE update from the sigma schedule
(float64, a few flops per step) and the device does the tensor work in two C-ABI calls per
step: ``sat_dit_denoise_cfg`` (DiT with batched CFG + VDenoiser scalings) and
``sat_dpmpp3m_update`` (fused elementwise state update).  No host<->device sync inside the loop.

Noise: k-diffusion's default ``BrownianTreeNoiseSampler`` (torchsde) is replaced by i.i.d.
``randn`` per step, which has the same distribution on a fixed sigma grid; ``noise_sampler=``
(callable ``(sigma, sigma_next) -> tensor``) overrides it (tests inject the noise).
"""
import math

import torch

from .. import _hip
from ..models.diffusion import DiTWrapper

SUPPORTED_SAMPLERS = ("dpmpp-3m-sde", "dpmpp-2m-sde", "k-heun", "k-lms", "k-dpmpp-2s-ancestral", "k-dpm-2", "k-dpm-fast",
                      "k-dpm-adaptive")


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0):
    """k_diffusion.sampling.get_sigmas_polyexponential: fp32 ramp/exp, then append 0 (host list of python floats)."""
    ramp = torch.linspace(1, 0, n, dtype=torch.float32) ** rho
    sigmas = torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))
    return [float(s) for s in sigmas] + [0.0]


def dpmpp3m_coefficients(sigmas, eta=1.0, s_noise=1.0):
    """Per-step scalars (a, b, c1, c2, cn) of
         x <- a*x + b*D + c1*(D - D1) + c2*(D1 - D2) + cn*noise
    equivalent to the multistep form of k-diffusion's ``sample_dpmpp_3m_sde``:
         x = e^{-h_eta} x + (1 - e^{-h_eta}) D + phi2*d1 - phi3*d2 (+ noise term)."""
    coeffs = []
    h_1 = h_2 = None
    h = None
    for i in range(len(sigmas) - 1):
        s_i, s_n = sigmas[i], sigmas[i + 1]
        if s_n == 0:
            coeffs.append((0.0, 1.0, 0.0, 0.0, 0.0))
        else:
            t, s = -math.log(s_i), -math.log(s_n)
            h = s - t
            h_eta = h * (eta + 1)
            a = math.exp(-h_eta)
            b = -math.expm1(-h_eta)
            c1 = c2 = 0.0
            if h_2 is not None:
                r0, r1 = h_1 / h, h_2 / h
                phi_2 = math.expm1(-h_eta) / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                bc = (phi_2 * r0 - phi_3) / (r0 + r1)
                c1 = (phi_2 + bc) / r0
                c2 = -bc / r1
            elif h_1 is not None:
                r = h_1 / h
                phi_2 = math.expm1(-h_eta) / h_eta + 1
                c1 = phi_2 / r
            cn = s_n * math.sqrt(-math.expm1(-2 * h * eta)) * s_noise if eta else 0.0
            coeffs.append((a, b, c1, c2, cn))
        h_1, h_2 = h, h_1
    return coeffs


def dpmpp2m_coefficients(sigmas, eta=1.0, s_noise=1.0, solver_type="midpoint"):
    """Per-step scalars (a, b, c1, 0, cn) of k-diffusion's ``sample_dpmpp_2m_sde`` (the reference's DEFAULT
    ``sampler_type``, sampling.py:150/226) in the same fused form as the 3M solver:
         x = (s_next/s) e^{-eta h} x + (1 - e^{-h - eta h}) D + corr * (1/r) (D - D1) (+ noise term),
    corr = 0.5 (1 - e^{-h-eta h}) for 'midpoint' (k-diffusion's default) or (1 - e^{-h-eta h})/(-h-eta h) + 1 for 'heun'."""
    if solver_type not in ("midpoint", "heun"):
        raise ValueError("solver_type must be 'heun' or 'midpoint'")
    coeffs = []
    h_last = None
    for i in range(len(sigmas) - 1):
        s_i, s_n = sigmas[i], sigmas[i + 1]
        if s_n == 0:
            coeffs.append((0.0, 1.0, 0.0, 0.0, 0.0))
            h = None
        else:
            h = math.log(s_i) - math.log(s_n)
            eta_h = eta * h
            a = s_n / s_i * math.exp(-eta_h)
            b = -math.expm1(-h - eta_h)
            c1 = 0.0
            if h_last is not None:
                r = h_last / h
                c1 = (0.5 * b if solver_type == "midpoint" else (b / (-h - eta_h) + 1)) / r
            cn = s_n * math.sqrt(-math.expm1(-2 * eta_h)) * s_noise if eta else 0.0
            coeffs.append((a, b, c1, 0.0, cn))
        h_last = h
    return coeffs


def get_bmask_strength(i, steps):
    """sampling.py:98-103: the binary mask of step i keeps init data where ``mask <= (i + 1) / steps``."""
    return (i + 1) / steps


def _lin(out, terms):
    """out <- sum c_i * t_i (``sat_lincomb``; up to five terms, terms may alias ``out``)."""
    assert 1 <= len(terms) <= 5
    args = []
    for c, t in terms:
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == out.numel()
        args += [_hip.ptr(t), float(c)]
    args += [None, 0.0] * (5 - len(terms))
    _hip.check(_hip.lib().sat_lincomb(_hip.ptr(out), *args, out.numel(), _hip.stream()))
    return out


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """k_diffusion.sampling.get_ancestral_step: (sigma_down, sigma_up)."""
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * math.sqrt(sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2))
    return math.sqrt(sigma_to ** 2 - sigma_up ** 2), sigma_up


def lms_coefficient(order, t, i, j):
    """k_diffusion.sampling.linear_multistep_coeff: integral over [t_i, t_{i+1}] of the j-th Lagrange basis polynomial through
    t_i, t_{i-1}, ..., t_{i-order+1}.  k-diffusion integrates numerically (scipy quad, epsrel 1e-4); the polynomial is
    integrated exactly here."""
    import numpy as np
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")
    poly = np.polynomial.Polynomial([1.0])
    for k in range(order):
        if k != j:
            poly = poly * np.polynomial.Polynomial([-t[i - k], 1.0]) / (t[i - j] - t[i - k])
    integ = poly.integ()
    return float(integ(t[i + 1]) - integ(t[i]))


class PIDStepSizeController:
    """k_diffusion.sampling.PIDStepSizeController (host scalars only)."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        self.h = h
        self.b1 = (pcoeff + icoeff + dcoeff) / order
        self.b2 = -(pcoeff + 2 * dcoeff) / order
        self.b3 = dcoeff / order
        self.accept_safety = accept_safety
        self.eps = eps
        self.errs = []

    def propose_step(self, error):
        inv_error = 1 / (float(error) + self.eps)
        if not self.errs:
            self.errs = [inv_error, inv_error, inv_error]
        self.errs[0] = inv_error
        factor = self.errs[0] ** self.b1 * self.errs[1] ** self.b2 * self.errs[2] ** self.b3
        factor = 1 + math.atan(factor - 1)
        accept = factor >= self.accept_safety
        if accept:
            self.errs[2] = self.errs[1]
            self.errs[1] = self.errs[0]
        self.h *= factor
        return accept


def _dpm_adaptive(denoise, after_denoise, x, sigma_min, sigma_max, rtol=0.01, atol=0.01, h_init=0.05, pcoeff=0.0, icoeff=1.0,
                  dcoeff=0.0, accept_safety=0.81, info=None):
    """k_diffusion.sampling.sample_dpm_adaptive -> DPMSolver.dpm_solver_adaptive(order=3, eta=0) with the reference's rtol = atol =
    0.01 (sampling.py:223-224): embedded 2nd/3rd-order pair in t = -log(sigma), PID step-size control on the host.  Three denoiser
    evaluations per attempted step; ONE host synchronisation per step (the error norm decides accept / reject)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    lib = _hip.lib()
    t_end = -math.log(sigma_min)
    s = -math.log(sigma_max)
    sig = lambda t: math.exp(-t)
    pid = PIDStepSizeController(abs(h_init), pcoeff, icoeff, dcoeff, 3, accept_safety)
    den, den1, den2 = (torch.empty_like(x) for _ in range(3))
    u1, u2, eps1, x_low, x_high = (torch.empty_like(x) for _ in range(5))
    x_prev = x.clone()
    n_part = 1024
    partial = torch.empty(n_part, dtype=torch.float32, device=x.device)
    stats = {"steps": 0, "nfe": 0, "n_accept": 0, "n_reject": 0}
    r1, r2 = 1 / 3, 2 / 3
    while s < t_end - 1e-5:
        t = min(t_end, s + pid.h)
        h = t - s
        ss = sig(s)
        denoise(x, ss, den)                                              # eps = (x - den) / sigma(s)
        s1, s2 = s + r1 * h, s + r2 * h
        c = -sig(s1) * math.expm1(r1 * h) / ss
        _lin(u1, [(1 + c, x), (-c, den)])
        denoise(u1, sig(s1), den1)                                       # eps_r1 = (u1 - den1) / sigma(s1)
        _lin(eps1, [(1 / sig(s1), u1), (-1 / sig(s1), den1)])
        a = -sig(t) * math.expm1(h)
        b_low = -sig(t) / (2 * r1) * math.expm1(h)                       # dpm_solver_2_step(r1 = 1/3)
        _lin(x_low, [(1 + (a - b_low) / ss, x), (-(a - b_low) / ss, den), (b_low, eps1)])
        a2 = -sig(s2) * math.expm1(r2 * h)
        b2 = -sig(s2) * (r2 / r1) * (math.expm1(r2 * h) / (r2 * h) - 1)
        _lin(u2, [(1 + (a2 - b2) / ss, x), (-(a2 - b2) / ss, den), (b2, eps1)])
        denoise(u2, sig(s2), den2)                                       # eps_r2 = (u2 - den2) / sigma(s2)
        b3 = -sig(t) / r2 * (math.expm1(h) / h - 1)
        _lin(x_high, [(1 + (a - b3) / ss, x), (-(a - b3) / ss, den), (b3 / sig(s2), u2), (-b3 / sig(s2), den2)])
        _hip.check(lib.sat_dpm_error_partials(_hip.ptr(x_low), _hip.ptr(x_high), _hip.ptr(x_prev), atol, rtol, x.numel(),
                                              _hip.ptr(partial), n_part, _hip.stream()))
        error = math.sqrt(float(partial.double().sum().item()) / x.numel())     # the one host sync of the step
        if pid.propose_step(error):
            x_prev, x_low = x_low, x_prev
            x, x_high = x_high, x
            s = t
            stats["n_accept"] += 1
        else:
            stats["n_reject"] += 1
        stats["nfe"] += 3
        stats["steps"] += 1
        # DPMSolver.dpm_solver_adaptive calls info_callback at the END of the attempted step, with the (possibly advanced) state
        # and time and i = index of that step; `den` still holds the denoised estimate of the step's starting point
        after_denoise(stats["steps"] - 1, x, sig(s), den)
    if info is not None:
        info.update(stats)
    return x


def _single_step_sampler(sampler_type, denoise, after_denoise, x, sigmas, sigma_min, sigma_max, steps, noise_sampler, eta, s_noise,
                         order=4):
    """The k-diffusion samplers that evaluate the denoiser more than once per step or keep a derivative history
    (sampling.py:212-225).  ``denoise(x, sigma, out)`` is the fused CFG + VDenoiser DiT evaluation; ``after_denoise(i, x,
    sigma, denoised)`` runs the inpainting re-injection and the user callback exactly where k-diffusion calls ``callback``.
    Every state update is one ``sat_lincomb`` launch with scalars computed here in float64."""
    den = torch.empty_like(x)
    den2 = torch.empty_like(x)
    x2 = torch.empty_like(x)
    d0 = torch.empty_like(x)

    def draw(s_from, s_to):
        nz = noise_sampler(s_from, s_to) if noise_sampler is not None else torch.randn_like(x)
        return nz.float().contiguous()

    if sampler_type == "k-dpm-fast":
        # k_diffusion.sampling.sample_dpm_fast -> DPMSolver.dpm_solver_fast(x, t_start, t_end, nfe=steps, eta=0): orders 3/2/1 in
        # log-sigma time t = -log(sigma); eps(x, t) = (x - D(x, sigma(t))) / sigma(t); no final step to sigma = 0
        if sigma_min <= 0 or sigma_max <= 0:
            raise ValueError("sigma_min and sigma_max must not be 0")
        nfe = steps
        t_start, t_end = -math.log(sigma_max), -math.log(sigma_min)
        m = nfe // 3 + 1
        ts = [t_start + (t_end - t_start) * i / m for i in range(m + 1)]
        orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
        sig = lambda t: math.exp(-t)
        eps1 = torch.empty_like(x)
        u = torch.empty_like(x)
        e = torch.empty_like(x)
        for i, order_i in enumerate(orders):
            t, t_next = ts[i], ts[i + 1]
            h = t_next - t
            denoise(x, sig(t), den)
            st = sig(t)
            # k-diffusion's DPMSolver caches eps(x, t) BEFORE it reports to the callback; the inpainting callback then mutates x in
            # place, and the step functions combine the mutated x with the cached eps
            _lin(e, [(1 / st, x), (-1 / st, den)])
            after_denoise(i, x, st, den)
            if order_i == 1:
                _lin(x, [(1.0, x), (-sig(t_next) * math.expm1(h), e)])
            elif order_i == 2:
                r1 = 0.5
                s1 = t + r1 * h
                _lin(u, [(1.0, x), (-sig(s1) * math.expm1(r1 * h), e)])  # u1 = x - sigma(s1) expm1(r1 h) eps
                denoise(u, sig(s1), den2)                              # eps_r1 = (u1 - den2) / sigma(s1)
                a = -sig(t_next) * math.expm1(h)
                b = -sig(t_next) / (2 * r1) * math.expm1(h)
                # x_2 = x + a eps + b (eps_r1 - eps)
                _lin(x, [(1.0, x), (a - b, e), (b / sig(s1), u), (-b / sig(s1), den2)])
            else:
                r1, r2 = 1 / 3, 2 / 3
                s1, s2 = t + r1 * h, t + r2 * h
                _lin(u, [(1.0, x), (-sig(s1) * math.expm1(r1 * h), e)])  # u1
                denoise(u, sig(s1), den2)                              # eps_r1 = (u - den2)/sigma(s1)
                _lin(eps1, [(1 / sig(s1), u), (-1 / sig(s1), den2)])
                a2 = -sig(s2) * math.expm1(r2 * h)
                b2 = -sig(s2) * (r2 / r1) * (math.expm1(r2 * h) / (r2 * h) - 1)
                # u2 = x + a2 eps + b2 (eps_r1 - eps)
                _lin(u, [(1.0, x), (a2 - b2, e), (b2, eps1)])
                denoise(u, sig(s2), den2)                              # eps_r2 = (u2 - den2)/sigma(s2)
                a3 = -sig(t_next) * math.expm1(h)
                b3 = -sig(t_next) / r2 * (math.expm1(h) / h - 1)
                _lin(x, [(1.0, x), (a3 - b3, e), (b3 / sig(s2), u), (-b3 / sig(s2), den2)])
        return x

    hist = []                                                          # k-lms: derivative history, newest last
    for i in range(len(sigmas) - 1):
        s_i, s_n = sigmas[i], sigmas[i + 1]
        denoise(x, s_i, den)
        # k-diffusion's sample_lms / sample_heun / sample_dpm_2 form d = to_d(x, sigma, denoised) BEFORE calling `callback`; the
        # reference's inpainting callback (sampling.py:175-190) then mutates x in place, so d must come from the pre-callback x.
        # sample_dpmpp_2s_ancestral calls back first and derives everything from the mutated x.
        if sampler_type == "k-lms":
            d = hist.pop(0) if len(hist) == order else torch.empty_like(x)
            _lin(d, [(1 / s_i, x), (-1 / s_i, den)])                    # to_d
            hist.append(d)
            after_denoise(i, x, s_i, den)
            cur = min(i + 1, order)
            cs = [lms_coefficient(cur, sigmas, i, j) for j in range(cur)]
            _lin(x, [(1.0, x)] + [(c, dd) for c, dd in zip(cs, reversed(hist))])
        elif sampler_type in ("k-heun", "k-dpm-2"):
            _lin(d0, [(1 / s_i, x), (-1 / s_i, den)])                   # to_d
            after_denoise(i, x, s_i, den)
            dt = s_n - s_i
            if s_n == 0:
                _lin(x, [(1.0, x), (dt, d0)])                           # Euler
            elif sampler_type == "k-heun":
                _lin(x2, [(1.0, x), (dt, d0)])
                denoise(x2, s_n, den2)
                # x + dt/2 * [d + (x2 - D2)/s_n]
                _lin(x, [(1.0, x), (dt / 2, d0), (dt / (2 * s_n), x2), (-dt / (2 * s_n), den2)])
            else:
                s_mid = math.exp(0.5 * (math.log(s_i) + math.log(s_n)))
                dt1, dt2 = s_mid - s_i, s_n - s_i
                _lin(x2, [(1.0, x), (dt1, d0)])
                denoise(x2, s_mid, den2)
                _lin(x, [(1.0, x), (dt2 / s_mid, x2), (-dt2 / s_mid, den2)])
        elif sampler_type == "k-dpmpp-2s-ancestral":
            after_denoise(i, x, s_i, den)
            s_down, s_up = get_ancestral_step(s_i, s_n, eta)
            if s_down == 0:
                dt = s_down - s_i
                _lin(x, [(1 + dt / s_i, x), (-dt / s_i, den)])
            else:
                t, t_next = -math.log(s_i), -math.log(s_down)
                h = t_next - t
                s = t + 0.5 * h
                _lin(x2, [(math.exp(-s) / math.exp(-t), x), (-math.expm1(-h * 0.5), den)])
                denoise(x2, math.exp(-s), den2)
                _lin(x, [(math.exp(-t_next) / math.exp(-t), x), (-math.expm1(-h), den2)])
            if s_n > 0:
                _lin(x, [(1.0, x), (s_noise * s_up, draw(s_i, s_n))])
        else:
            raise NotImplementedError(sampler_type)
    return x


@torch.no_grad()
def sample_k(model_fn, noise, init_data=None, mask=None, steps=100, sampler_type="dpmpp-2m-sde", sigma_min=0.5, sigma_max=50,
             rho=1.0, device="cuda", callback=None, cond_fn=None, disable_tqdm: bool = False, noise_sampler=None,
             inpaint_noise=None, eta=1.0,
             s_noise=1.0, cfg_scale=1.0, scale_phi=0.0, batch_cfg=True, rescale_cfg=False, cross_attn_cond=None,
             cross_attn_mask=None, global_cond=None, negative_cross_attn_cond=None, negative_cross_attn_mask=None,
             input_concat_cond=None, prepend_cond=None, prepend_cond_mask=None, negative_global_cond=None,
             negative_input_concat_cond=None, **extra_args):
    if sampler_type not in SUPPORTED_SAMPLERS:
        raise NotImplementedError(f"sampler_type '{sampler_type}' is not implemented by the HIP path; supported: {SUPPORTED_SAMPLERS}")
    if cond_fn is not None:
        raise NotImplementedError("cond_fn (gradient guidance through the denoiser) is outside the supported hot path")
    if mask is not None and init_data is None:
        mask = None                                                     # sampling.py:166-201: a mask without init data is ignored
    if input_concat_cond is not None or prepend_cond is not None:
        raise NotImplementedError("input_concat_cond / prepend_cond are outside the supported hot path")
    if not isinstance(model_fn, DiTWrapper):
        raise NotImplementedError("sample_k drives the HIP DiT (DiTWrapper) only; there is no eager/CPU denoiser path")
    assert batch_cfg, "batch_cfg must be True for DiTWrapper"
    dit = model_fn.model

    sigmas = get_sigmas_polyexponential(steps, sigma_min, sigma_max, rho)
    lib = _hip.lib()
    unit_noise = noise.float().contiguous()
    noise = unit_noise * sigmas[0]
    rows = noise.numel() // noise.shape[-1]
    if init_data is not None and mask is not None:
        # INPAINTING (sampling.py:166-172): x = (init + noise) where the step-0 binary mask keeps the input, noise elsewhere
        init_data = init_data.float().contiguous()
        mask = mask.to(noise.device, torch.float32).contiguous()
        assert mask.ndim == 1 and mask.numel() == noise.shape[-1], "mask must be [latent_length]"
        x = noise.clone()
        _hip.check(lib.sat_inpaint_mix(_hip.ptr(x), _hip.ptr(init_data), _hip.ptr(unit_noise), _hip.ptr(mask), sigmas[0],
                                       get_bmask_strength(0, steps), rows, x.shape[-1], _hip.stream()))
    elif init_data is not None:
        x = (init_data.float() + noise).contiguous()                    # VARIATION (sampling.py:162-165)
    else:
        x = noise                                                       # SAMPLING

    dit.prepare_generation(cross_attn_cond, global_cond, cfg_scale, negative_cross_attn_cond, negative_cross_attn_mask)

    def after_denoise(i, x, sigma, denoised):
        if mask is not None:
            # sampling.py:178-190: right after the denoiser evaluation, x is overwritten in the keep-region of this step's
            # (shrinking) binary mask with the init data re-noised to the current sigma (fresh Gaussian draw per step)
            rn = inpaint_noise(i) if inpaint_noise is not None else torch.randn_like(x)
            _hip.check(lib.sat_inpaint_mix(_hip.ptr(x), _hip.ptr(init_data), _hip.ptr(rn.float().contiguous()), _hip.ptr(mask),
                                           sigma, get_bmask_strength(i, steps), rows, x.shape[-1], _hip.stream()))
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigma, "sigma_hat": sigma, "denoised": denoised})

    denoise = lambda xin, sigma, out: dit.denoise(xin, sigma, cfg_scale=cfg_scale, scale_phi=scale_phi, out=out)
    if sampler_type == "k-dpm-adaptive":
        return _dpm_adaptive(denoise, after_denoise, x.clone() if x is noise else x, sigma_min, sigma_max, rtol=extra_args.pop("rtol", 0.01),
                             atol=extra_args.pop("atol", 0.01), info=extra_args.pop("info", None))
    if sampler_type not in ("dpmpp-3m-sde", "dpmpp-2m-sde"):
        return _single_step_sampler(sampler_type, denoise, after_denoise, x, sigmas, sigma_min, sigma_max, steps, noise_sampler,
                                    eta, s_noise, order=extra_args.pop("order", 4))

    if sampler_type == "dpmpp-3m-sde":
        coeffs = dpmpp3m_coefficients(sigmas, eta=eta, s_noise=s_noise)
    else:
        coeffs = dpmpp2m_coefficients(sigmas, eta=eta, s_noise=s_noise, solver_type=extra_args.pop("solver_type", "midpoint"))
    n = x.numel()
    d = torch.empty_like(x)
    d1 = torch.empty_like(x)
    d2 = torch.empty_like(x)
    have = 0
    for i in range(steps):
        dit.denoise(x, sigmas[i], cfg_scale=cfg_scale, scale_phi=scale_phi, out=d)
        after_denoise(i, x, sigmas[i], d)
        a, b, c1, c2, cn = coeffs[i]
        nz = None
        if cn != 0.0:
            nz = noise_sampler(sigmas[i], sigmas[i + 1]) if noise_sampler is not None else torch.randn_like(x)
            nz = nz.float().contiguous()
        _hip.check(lib.sat_dpmpp3m_update(_hip.ptr(x), _hip.ptr(d), _hip.ptr(d1) if have >= 1 else None,
                                          _hip.ptr(d2) if have >= 2 else None, _hip.ptr(nz), a, b, c1, c2, cn, n, _hip.stream()))
        d, d1, d2 = d2, d, d1      # D2 <- D1, D1 <- D; the old D2 buffer is recycled for the next D
        have = min(have + 1, 2)
    return x


@torch.no_grad()
def sample_discrete_euler(model, x, steps, sigma_max=1, verbose=False, callback=None, **extra_args):
    """Rectified-flow Euler integration (reference sampling.py:28-60): t from sigma_max down to 0, x <- x + dt * model(x, t)."""
    xf = x.float().contiguous()
    x = xf.clone() if xf.data_ptr() == x.data_ptr() else xf          # updated in place below: never the caller's tensor
    t = torch.linspace(sigma_max, 0, steps + 1)
    for i in range(steps):
        t_curr, t_prev = float(t[i]), float(t[i + 1])
        v = model(x, torch.full((x.shape[0],), t_curr, dtype=torch.float32, device=x.device), **extra_args).float().contiguous()
        _lin(x, [(1.0, x), (t_prev - t_curr, v)])
        if callback is not None:
            callback({"x": x, "i": i, "t": t_curr})
    return x


@torch.no_grad()
def sample_rf(model_fn, noise, init_data=None, steps=100, sigma_max=1, device="cuda", callback=None, cond_fn=None,
              disable_tqdm: bool = False, **extra_args):
    """reference sampling.py:236-270: rectified-flow sampling / variations through the discrete Euler loop."""
    if cond_fn is not None:
        raise NotImplementedError("cond_fn (gradient guidance through the denoiser) is outside the supported hot path")
    sigma_max = min(sigma_max, 1)
    noise = noise.float()
    x = init_data.float() * (1 - sigma_max) + noise * sigma_max if init_data is not None else noise
    return sample_discrete_euler(model_fn, x, steps, sigma_max, verbose=not disable_tqdm, callback=callback, **extra_args)
