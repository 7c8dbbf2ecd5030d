"""``sample_k`` with the ``dpmpp-3m-sde`` / ``dpmpp-2m-sde`` samplers (reference ``inference/sampling.py:144-228``),
plain sampling, variations (``init_data``) and inpainting (``init_data`` + soft ``mask``).

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

SUPPORTED_SAMPLERS = ("dpmpp-3m-sde", "dpmpp-2m-sde")


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
    if sampler_type == "dpmpp-3m-sde":
        coeffs = dpmpp3m_coefficients(sigmas, eta=eta, s_noise=s_noise)
    else:
        coeffs = dpmpp2m_coefficients(sigmas, eta=eta, s_noise=s_noise, solver_type=extra_args.pop("solver_type", "midpoint"))
    lib = _hip.lib()
    unit_noise = noise.float().contiguous()
    noise = unit_noise * sigmas[0]
    if init_data is not None and mask is not None:
        # INPAINTING (sampling.py:166-172): x = (init + noise) where the step-0 binary mask keeps the input, noise elsewhere
        init_data = init_data.float().contiguous()
        mask = mask.to(noise.device, torch.float32).contiguous()
        assert mask.ndim == 1 and mask.numel() == noise.shape[-1], "mask must be [latent_length]"
        x = noise.clone()
        rows = x.numel() // x.shape[-1]
        _hip.check(lib.sat_inpaint_mix(_hip.ptr(x), _hip.ptr(init_data), _hip.ptr(unit_noise), _hip.ptr(mask), sigmas[0],
                                       get_bmask_strength(0, steps), rows, x.shape[-1], _hip.stream()))
    elif init_data is not None:
        x = (init_data.float() + noise).contiguous()                    # VARIATION (sampling.py:162-165)
    else:
        x = noise                                                       # SAMPLING

    dit.prepare_generation(cross_attn_cond, global_cond, cfg_scale, negative_cross_attn_cond, negative_cross_attn_mask)
    n = x.numel()
    d = torch.empty_like(x)
    d1 = torch.empty_like(x)
    d2 = torch.empty_like(x)
    have = 0
    for i in range(steps):
        dit.denoise(x, sigmas[i], cfg_scale=cfg_scale, scale_phi=scale_phi, out=d)
        if mask is not None:
            # sampling.py:178-190: right after the denoiser evaluation, x is overwritten in the keep-region of this step's
            # (shrinking) binary mask with the init data re-noised to the current sigma (fresh Gaussian draw per step)
            rn = inpaint_noise(i) if inpaint_noise is not None else torch.randn_like(x)
            _hip.check(lib.sat_inpaint_mix(_hip.ptr(x), _hip.ptr(init_data), _hip.ptr(rn.float().contiguous()), _hip.ptr(mask),
                                           sigmas[i], get_bmask_strength(i, steps), rows, x.shape[-1], _hip.stream()))
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": d})
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
