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


# ---------------------------------------------------------------------------------------------------------------------
# The single-step k-diffusion samplers selectable at inference/sampling.py:212-225 (k-diffusion 0.1.1, restated, unpinned)
# ---------------------------------------------------------------------------------------------------------------------
def to_d(x, sigma, denoised):
    return (x - denoised) / sigma


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def sample_heun(denoiser, x, sigmas, callback=None):                      # K.sampling.sample_heun, s_churn = 0
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = denoiser(x, sigmas[i] * s_in)
        d = to_d(x, sigmas[i], denoised)          # k-diffusion forms d BEFORE the callback (which may mutate x in place)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        dt = sigmas[i + 1] - sigmas[i]
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            d_2 = to_d(x_2, sigmas[i + 1], denoiser(x_2, sigmas[i + 1] * s_in))
            x = x + (d + d_2) / 2 * dt
    return x


def sample_dpm_2(denoiser, x, sigmas, callback=None):                     # K.sampling.sample_dpm_2, s_churn = 0
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = denoiser(x, sigmas[i] * s_in)
        d = to_d(x, sigmas[i], denoised)          # before the callback, as in k-diffusion
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        if sigmas[i + 1] == 0:
            x = x + d * (sigmas[i + 1] - sigmas[i])
        else:
            sigma_mid = sigmas[i].log().lerp(sigmas[i + 1].log(), 0.5).exp()
            x_2 = x + d * (sigma_mid - sigmas[i])
            d_2 = to_d(x_2, sigma_mid, denoiser(x_2, sigma_mid * s_in))
            x = x + d_2 * (sigmas[i + 1] - sigmas[i])
    return x


def linear_multistep_coeff(order, t, i, j):
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def fn(tau):
        prod = 1.0
        for k in range(order):
            if j != k:
                prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod
    return integrate.quad(fn, t[i], t[i + 1], epsrel=1e-4)[0]


def sample_lms(denoiser, x, sigmas, order=4, callback=None):               # K.sampling.sample_lms
    s_in = x.new_ones([x.shape[0]])
    sig = [float(v) for v in sigmas]
    ds = []
    for i in range(len(sigmas) - 1):
        denoised = denoiser(x, sigmas[i] * s_in)
        ds.append(to_d(x, sigmas[i], denoised))   # before the callback, as in k-diffusion
        if len(ds) > order:
            ds.pop(0)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        cur_order = min(i + 1, order)
        coeffs = [linear_multistep_coeff(cur_order, sig, i, j) for j in range(cur_order)]
        x = x + sum(c * d for c, d in zip(coeffs, reversed(ds)))
    return x


def sample_dpmpp_2s_ancestral(denoiser, x, sigmas, noise_sampler, eta=1.0, s_noise=1.0, callback=None):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = denoiser(x, sigmas[i] * s_in)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        sigma_down, sigma_up = get_ancestral_step(float(sigmas[i]), float(sigmas[i + 1]), eta)
        if sigma_down == 0:
            x = x + to_d(x, sigmas[i], denoised) * (sigma_down - sigmas[i])
        else:
            t, t_next = -math.log(float(sigmas[i])), -math.log(sigma_down)
            h = t_next - t
            s = t + 0.5 * h
            x_2 = (math.exp(-s) / math.exp(-t)) * x - math.expm1(-h * 0.5) * denoised
            denoised_2 = denoiser(x_2, math.exp(-s) * s_in)
            x = (math.exp(-t_next) / math.exp(-t)) * x - math.expm1(-h) * denoised_2
        if sigmas[i + 1] > 0:
            x = x + noise_sampler(i, sigmas[i], sigmas[i + 1]) * s_noise * sigma_up
    return x


def sample_dpm_fast(denoiser, x, sigma_min, sigma_max, n, callback=None):  # K.sampling.sample_dpm_fast, eta = 0 (DPMSolver)
    s_in = x.new_ones([x.shape[0]])
    sigma = lambda t: math.exp(-t)
    eps = lambda xx, t: (xx - denoiser(xx, sigma(t) * s_in)) / sigma(t)
    t_start, t_end = -math.log(sigma_max), -math.log(sigma_min)
    m = n // 3 + 1
    ts = [t_start + (t_end - t_start) * i / m for i in range(m + 1)]
    orders = [3] * (m - 2) + [2, 1] if n % 3 == 0 else [3] * (m - 1) + [n % 3]
    for i, order in enumerate(orders):
        t, t_next = ts[i], ts[i + 1]
        h = t_next - t
        e = eps(x, t)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigma(t), "sigma_hat": sigma(t), "denoised": x - sigma(t) * e})
        if order == 1:
            x = x - sigma(t_next) * math.expm1(h) * e
        elif order == 2:
            r1 = 0.5
            s1 = t + r1 * h
            u1 = x - sigma(s1) * math.expm1(r1 * h) * e
            e1 = eps(u1, s1)
            x = x - sigma(t_next) * math.expm1(h) * e - sigma(t_next) / (2 * r1) * math.expm1(h) * (e1 - e)
        else:
            r1, r2 = 1 / 3, 2 / 3
            s1, s2 = t + r1 * h, t + r2 * h
            u1 = x - sigma(s1) * math.expm1(r1 * h) * e
            e1 = eps(u1, s1)
            u2 = x - sigma(s2) * math.expm1(r2 * h) * e - sigma(s2) * (r2 / r1) * (math.expm1(r2 * h) / (r2 * h) - 1) * (e1 - e)
            e2 = eps(u2, s2)
            x = x - sigma(t_next) * math.expm1(h) * e - sigma(t_next) / r2 * (math.expm1(h) / h - 1) * (e2 - e)
    return x


class PIDStepSizeController:                                               # K.sampling.PIDStepSizeController
    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        self.h, self.accept_safety, self.eps, self.errs = h, accept_safety, eps, []
        self.b1 = (pcoeff + icoeff + dcoeff) / order
        self.b2 = -(pcoeff + 2 * dcoeff) / order
        self.b3 = dcoeff / order

    def propose_step(self, error):
        inv_error = 1 / (float(error) + self.eps)
        if not self.errs:
            self.errs = [inv_error, inv_error, inv_error]
        self.errs[0] = inv_error
        factor = 1 + math.atan(self.errs[0] ** self.b1 * self.errs[1] ** self.b2 * self.errs[2] ** self.b3 - 1)
        accept = factor >= self.accept_safety
        if accept:
            self.errs[2], self.errs[1] = self.errs[1], self.errs[0]
        self.h *= factor
        return accept


# K.sampling.sample_dpm_adaptive(model, x, sigma_min, sigma_max, rtol=0.01, atol=0.01) as called at sampling.py:222-224
# -> DPMSolver.dpm_solver_adaptive(order=3, eta=0): embedded DPM-Solver-2 (r1 = 1/3) / DPM-Solver-3 pair, PID step control
def sample_dpm_adaptive(denoiser, x, sigma_min, sigma_max, rtol=0.01, atol=0.01, h_init=0.05, accept_safety=0.81, info=None,
                        callback=None):
    s_in = x.new_ones([x.shape[0]])
    sigma = lambda t: math.exp(-t)
    eps = lambda xx, t: (xx - denoiser(xx, sigma(t) * s_in)) / sigma(t)
    s, t_end = -math.log(sigma_max), -math.log(sigma_min)
    pid = PIDStepSizeController(h_init, 0.0, 1.0, 0.0, 3, accept_safety)
    x_prev = x
    stats = {"steps": 0, "nfe": 0, "n_accept": 0, "n_reject": 0}
    r1, r2 = 1 / 3, 2 / 3
    while s < t_end - 1e-5:
        t = min(t_end, s + pid.h)
        h = t - s
        e = eps(x, s)
        x_start, s_start = x, s
        s1, s2 = s + r1 * h, s + r2 * h
        u1 = x - sigma(s1) * math.expm1(r1 * h) * e
        e1 = eps(u1, s1)
        x_low = x - sigma(t) * math.expm1(h) * e - sigma(t) / (2 * r1) * math.expm1(h) * (e1 - e)
        u2 = x - sigma(s2) * math.expm1(r2 * h) * e - sigma(s2) * (r2 / r1) * (math.expm1(r2 * h) / (r2 * h) - 1) * (e1 - e)
        e2 = eps(u2, s2)
        x_high = x - sigma(t) * math.expm1(h) * e - sigma(t) / r2 * (math.expm1(h) / h - 1) * (e2 - e)
        delta = torch.maximum(torch.tensor(atol, dtype=x.dtype), rtol * torch.maximum(x_low.abs(), x_prev.abs()))
        error = torch.linalg.norm((x_low - x_high) / delta) / x.numel() ** 0.5
        if pid.propose_step(error):
            x_prev, x, s = x_low, x_high, t
            stats["n_accept"] += 1
        else:
            stats["n_reject"] += 1
        stats["nfe"] += 3
        stats["steps"] += 1
        if callback is not None:      # DPMSolver.dpm_solver_adaptive reports at the END of the attempted step: the (possibly
            # advanced) state and time, i = index of the step just attempted, denoised of the step's starting point
            callback({"x": x, "i": stats["steps"] - 1, "sigma": sigma(s), "sigma_hat": sigma(s), "denoised": x_start - sigma(s_start) * e})
    if info is not None:
        info.update(stats)
    return x


# inference/sampling.py:28-60 (sample_discrete_euler): rectified flow
def sample_discrete_euler(model, x, steps, sigma_max=1.0):
    t = torch.linspace(sigma_max, 0, steps + 1)
    for t_curr, t_prev in zip(t[:-1], t[1:]):
        x = x + (t_prev - t_curr) * model(x, t_curr * torch.ones((x.shape[0],), dtype=x.dtype))
    return x
