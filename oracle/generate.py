"""Oracle: the orchestration of ``generate_diffusion_cond`` / ``sample_k`` -- RESTATED from the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Functional CPU restatement of ``inference/generation.py:95-261`` and ``inference/sampling.py:144-228`` over a
reference-format state dict, composed from oracle.dit / oracle.oobleck / oracle.sampler.  Every Gaussian draw is an
explicit argument (the reference draws them from the global generator in this order: initial noise, VAE bottleneck
noise of the init audio, then per step the inpainting re-noise and the sampler noise).

Pinned by ``tests/golden/generate.npz``: outputs of the REFERENCE's own generate_diffusion_cond / sample_k run in the
build container on top of a stand-in ``k_diffusion`` (tests/golden/make_golden.py: gen_generate), with the draws recorded.
"""
import math

import torch
import torch.nn.functional as F

from . import dit as odit
from . import oobleck as oob
from . import sampler as osamp


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# inference/generation.py:269-290
def build_mask(sample_size, mask_args):
    maskstart = math.floor(mask_args["maskstart"] / 100.0 * sample_size)
    maskend = math.ceil(mask_args["maskend"] / 100.0 * sample_size)
    soft_l = round(mask_args["softnessL"] / 100.0 * sample_size)
    soft_r = round(mask_args["softnessR"] / 100.0 * sample_size)
    hann_l = torch.hann_window(soft_l * 2, periodic=False)[:soft_l]
    hann_r = torch.hann_window(soft_r * 2, periodic=False)[soft_r:]
    mask = torch.zeros(sample_size)
    mask[maskstart:maskend] = 1
    mask[maskstart:maskstart + soft_l] = hann_l
    mask[maskend - soft_r:maskend] = hann_r
    if mask_args["marination"] > 0:
        mask = mask * (1 - mask_args["marination"])
    return mask


# inference/utils.py:20-40 + data/modification.py:11-23 (PadCrop, randomize=False), same sample rate only
def prepare_audio(audio, target_length, target_channels):
    n, s = audio.shape
    out = audio.new_zeros([n, target_length])
    out[:, :min(s, target_length)] = audio[:, :target_length]
    out = out.unsqueeze(0)
    if target_channels == 1:
        return out.mean(1, keepdim=True)
    if out.shape[1] == 1:
        return out.repeat(1, 2, 1)
    return out[:, :2, :]


_SAMPLERS_WITH_NOISE = {"dpmpp-3m-sde": osamp.sample_dpmpp_3m_sde, "dpmpp-2m-sde": osamp.sample_dpmpp_2m_sde,
                        "k-dpmpp-2s-ancestral": osamp.sample_dpmpp_2s_ancestral}
_SAMPLERS_PLAIN = {"k-heun": osamp.sample_heun, "k-lms": osamp.sample_lms, "k-dpm-2": osamp.sample_dpm_2}


# inference/sampling.py:144-228
def sample_k(model_fn, noise, init_data, mask, steps, sampler_type, sigma_min, sigma_max, step_noise=None, renoise=None, callback=None,
             rho=1.0):
    """model_fn(x, t) -> v-prediction (CFG included); step_noise(i) / renoise(i) -> unit Gaussians like x."""
    denoiser = lambda x, sigma: osamp.vdenoise(model_fn, x, sigma)
    sigmas = osamp.get_sigmas_polyexponential(steps, sigma_min, sigma_max, rho)
    noise = noise * sigmas[0]
    wrapped = callback
    if mask is None and init_data is not None:
        x = init_data + noise                                            # variation
    elif mask is not None and init_data is not None:
        x, inpaint_cb = osamp.inpainting_start_and_callback(init_data, noise, mask, steps, renoise)
        x = x.clone()
        wrapped = inpaint_cb if callback is None else (lambda a: (inpaint_cb(a), callback(a)))
    else:
        x = noise
    if sampler_type in _SAMPLERS_WITH_NOISE:
        return _SAMPLERS_WITH_NOISE[sampler_type](denoiser, x, sigmas, lambda i, a, b: step_noise(i), callback=wrapped)
    if sampler_type in _SAMPLERS_PLAIN:
        return _SAMPLERS_PLAIN[sampler_type](denoiser, x, sigmas, callback=wrapped)
    if sampler_type == "k-dpm-fast":
        return osamp.sample_dpm_fast(denoiser, x, sigma_min, sigma_max, steps, callback=wrapped)
    if sampler_type == "k-dpm-adaptive":
        return osamp.sample_dpm_adaptive(denoiser, x, sigma_min, sigma_max, rtol=0.01, atol=0.01, callback=wrapped)
    raise ValueError(sampler_type)


# inference/generation.py:95-261 (diffusion_objective "v", latent diffusion with an Oobleck VAE pretransform)
def generate_diffusion_cond(sd, cfg, cross_attn_cond, global_cond, steps, cfg_scale, sample_size, noise, sampler_type, sigma_min,
                            sigma_max, init_audio=None, init_noise_level=1.0, mask_args=None, vae_noise=None, step_noise=None,
                            renoise=None, return_latents=False, rnd=None):
    """sd: state dict of the whole ConditionedDiffusionModelWrapper; cfg: its model config; cross_attn_cond / global_cond: output
    of get_conditioning_inputs; noise [B, C, sample_size // ratio]: the initial draw."""
    dc = cfg["model"]["diffusion"]["config"]
    pr = cfg["model"]["pretransform"]["config"]
    ratio = pr["downsampling_ratio"]
    audio_sample_size = sample_size
    sample_size //= ratio                                                # :139-140
    num_sample = cross_attn_cond.shape[0]
    init = None
    mask = None
    if init_audio is not None:
        a = prepare_audio(init_audio, audio_sample_size, pr["io_channels"])          # :170-183
        ms = oob.oobleck_encoder(_sub(sd, "pretransform.model.encoder."), a, strides=pr["encoder"]["config"]["strides"], rnd=rnd)
        init = oob.vae_sample(ms, vae_noise).repeat(num_sample, 1, 1)                # :186-188
        if mask_args is not None:                                                    # :195-213
            cropfrom = math.floor(mask_args["cropfrom"] / 100.0 * sample_size)
            pastefrom = math.floor(mask_args["pastefrom"] / 100.0 * sample_size)
            pasteto = math.ceil(mask_args["pasteto"] / 100.0 * sample_size)
            croplen = pasteto - pastefrom
            if cropfrom + croplen > sample_size:
                croplen = sample_size - cropfrom
            cutpaste = init.new_zeros(init.shape)
            cutpaste[:, :, pastefrom:pastefrom + croplen] = init[:, :, cropfrom:cropfrom + croplen]
            init = cutpaste
            mask = build_mask(sample_size, mask_args)
        else:
            sigma_max = init_noise_level                                             # :214-217
    dsd = _sub(sd, "model.model.")
    model_fn = lambda x, t: odit.dit_forward(dsd, x, t, cross_attn_cond, global_cond, dc["depth"], dc["num_heads"], cfg_scale=cfg_scale,
                                             rnd=rnd)
    sampled = sample_k(model_fn, noise, init, mask, steps, sampler_type, sigma_min, sigma_max, step_noise=step_noise, renoise=renoise)
    if return_latents:
        return sampled
    return oob.oobleck_decoder(_sub(sd, "pretransform.model.decoder."), sampled, strides=pr["decoder"]["config"]["strides"], rnd=rnd)
