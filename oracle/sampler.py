"""Oracle: k-diffusion pieces used by the reference's ``sample_k`` -- RESTATED, PARITY UNPINNED.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The arithmetic lives in the un-vendored third-party package ``k-diffusion==0.1.1``
(reference ``setup.py:21``), called at ``inference/sampling.py:159`` (``K.external.VDenoiser``),
``:165`` (``K.sampling.get_sigmas_polyexponential``) and ``:228``
(``K.sampling.sample_dpmpp_3m_sde``).  It is not installed in this image and there is no
network, so what follows restates the *published* algorithms (Karras et al. 2022 for the
v-prediction pre-conditioning; Lu et al. 2022 "DPM-Solver++" multistep, 3rd order, SDE
variant as shipped in k-diffusion 0.1.1 -- SURVEY.md Appendix A).  The reference holds no
tests for this boundary, so bit-parity with k-diffusion is *unpinned*; the restatement is
pinned by analytic checks in tests/test_sampler.py.  ``BrownianTreeNoiseSampler`` (torchsde)
cannot be reproduced: the noise sampler is pluggable and tests inject the noise.
"""
import math

import torch


# K.sampling.get_sigmas_polyexponential(n, sigma_min, sigma_max, rho)  (sampling.py:165)
def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0):
    ramp = torch.linspace(1, 0, n, dtype=torch.float32) ** rho
    sigmas = torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))
    return torch.cat([sigmas, sigmas.new_zeros([1])])


# K.external.VDenoiser (sigma_data = 1)  (sampling.py:159)
def vdenoiser_scalings(sigma, sigma_data=1.0):
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = -sigma * sigma_data / (sigma ** 2 + sigma_data ** 2) ** 0.5
    c_in = 1 / (sigma ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out, c_in


def sigma_to_t(sigma):
    return sigma.atan() / math.pi * 2


def vdenoise(model_fn, x, sigma):
    """D(x, sigma) = model(x*c_in, t(sigma)) * c_out + x * c_skip ; sigma is [B]."""
    c_skip, c_out, c_in = (s.view(-1, *([1] * (x.dim() - 1))) for s in vdenoiser_scalings(sigma))
    return model_fn(x * c_in, sigma_to_t(sigma)) * c_out + x * c_skip


# K.sampling.sample_dpmpp_3m_sde(model, x, sigmas, eta=1, s_noise=1, noise_sampler)  (sampling.py:228)
def sample_dpmpp_3m_sde(denoiser, x, sigmas, noise_sampler, eta=1.0, s_noise=1.0, callback=None):
    """denoiser(x, sigma[B]) -> denoised; noise_sampler(i, sigma, sigma_next) -> unit-variance noise like x."""
    s_in = x.new_ones([x.shape[0]])
    denoised_1 = denoised_2 = None
    h_1 = h_2 = None
    for i in range(len(sigmas) - 1):
        denoised = denoiser(x, sigmas[i] * s_in)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        if sigmas[i + 1] == 0:
            x = denoised
        else:
            t, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - t
            h_eta = h * (eta + 1)
            x = torch.exp(-h_eta) * x + (-h_eta).expm1().neg() * denoised
            if h_2 is not None:
                r0 = h_1 / h
                r1 = h_2 / h
                d1_0 = (denoised - denoised_1) / r0
                d1_1 = (denoised_1 - denoised_2) / r1
                d1 = d1_0 + (d1_0 - d1_1) * r0 / (r0 + r1)
                d2 = (d1_0 - d1_1) / (r0 + r1)
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                x = x + phi_2 * d1 - phi_3 * d2
            elif h_1 is not None:
                r = h_1 / h
                d = (denoised - denoised_1) / r
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                x = x + phi_2 * d
            if eta:
                x = x + noise_sampler(i, sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * h * eta).expm1().neg().sqrt() * s_noise
        denoised_1, denoised_2 = denoised, denoised_1
        h_1, h_2 = h, h_1
    return x


# K.sampling.sample_dpmpp_2m_sde(model, x, sigmas, eta=1, s_noise=1, noise_sampler, solver_type='midpoint')
# (sampling.py:226; the reference's DEFAULT sampler_type, sampling.py:150) -- restated from the published algorithm, unpinned
def sample_dpmpp_2m_sde(denoiser, x, sigmas, noise_sampler, eta=1.0, s_noise=1.0, solver_type="midpoint", callback=None):
    s_in = x.new_ones([x.shape[0]])
    old_denoised = None
    h_last = None
    for i in range(len(sigmas) - 1):
        denoised = denoiser(x, sigmas[i] * s_in)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        h = None
        if sigmas[i + 1] == 0:
            x = denoised
        else:
            t, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - t
            eta_h = eta * h
            x = sigmas[i + 1] / sigmas[i] * (-eta_h).exp() * x + (-h - eta_h).expm1().neg() * denoised
            if old_denoised is not None:
                r = h_last / h
                if solver_type == "heun":
                    x = x + ((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / r) * (denoised - old_denoised)
                else:
                    x = x + 0.5 * (-h - eta_h).expm1().neg() * (1 / r) * (denoised - old_denoised)
            if eta:
                x = x + noise_sampler(i, sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * s_noise
        old_denoised = denoised
        h_last = h
    return x


# inference/sampling.py:98-103 (get_bmask) and :166-201 (inpainting start + callback that mutates x)
def get_bmask(i, steps, mask):
    return torch.where(mask <= (i + 1) / steps, 1, 0)


def inpainting_start_and_callback(init_data, noise_scaled, mask, steps, renoise):
    """Returns (x0, callback).  ``renoise(i)`` supplies the unit Gaussian the reference draws with randn_like per step."""
    b0 = get_bmask(0, steps, mask)
    x0 = (init_data + noise_scaled) * b0 + noise_scaled * (1 - b0)

    def callback(args):
        i, x, sigma = args["i"], args["x"], args["sigma"]
        bm = get_bmask(i, steps, mask)
        x[:, :, :] = ((init_data + renoise(i) * sigma) * bm + x * (1 - bm))[:, :, :]
    return x0, callback
