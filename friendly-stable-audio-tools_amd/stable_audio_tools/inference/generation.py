"""``generate_diffusion_cond`` (reference ``inference/generation.py:95-261``) -- same signature,
same RNG order (manual_seed -> initial noise -> init-audio VAE noise -> sampler noise), same
return values; the sampler loop and the decode run on the HIP C ABI."""
import math
import typing as tp

import numpy as np
import torch

from .sampling import sample_k, sample_rf
from .utils import prepare_audio


def _init_latents(model, init_audio, audio_length, num_sample, device):
    """(sample_rate, waveform) -> init tensor in the domain the denoiser works in, one copy per batch item (generation.py:170-188).
    With a pretransform this samples the VAE bottleneck once: the second Gaussian draw after the initial noise."""
    in_sr, waveform = init_audio
    channels = model.pretransform.io_channels if model.pretransform else model.io_channels
    init = prepare_audio(waveform, in_sr=in_sr, target_sr=model.sample_rate, target_length=audio_length, target_channels=channels,
                         device=device)
    if model.pretransform:
        init = model.pretransform.encode(init)
    return init.repeat(num_sample, 1, 1)


def _cut_and_paste(init, mask_args, length):
    """Out-painting layout (generation.py:195-209): the span of the init data that starts at ``cropfrom`` % is pasted at
    ``pastefrom`` % .. ``pasteto`` % of an otherwise empty canvas (shortened if it would run past the end of the source)."""
    crop_from = math.floor(mask_args["cropfrom"] / 100.0 * length)
    paste_from = math.floor(mask_args["pastefrom"] / 100.0 * length)
    paste_to = math.ceil(mask_args["pasteto"] / 100.0 * length)
    assert paste_from < paste_to, "Paste From should be less than Paste To"
    span = min(paste_to - paste_from, length - crop_from)
    canvas = torch.zeros_like(init)
    canvas[:, :, paste_from:paste_from + span] = init[:, :, crop_from:crop_from + span]
    return canvas


@torch.no_grad()
def generate_diffusion_cond(model, steps: int = 250, cfg_scale: float = 6, conditioning: tp.Optional[tp.List[dict]] = None,
                            conditioning_tensors: tp.Optional[dict] = None, negative_conditioning: tp.Optional[tp.List[dict]] = None,
                            negative_conditioning_tensors: tp.Optional[dict] = None, sample_size: int = 2097152, seed: int = -1,
                            device: str = "cuda", init_audio: tp.Optional[tp.Tuple[int, torch.Tensor]] = None,
                            init_noise_level: float = 1.0, mask_args: dict = None, return_latents: bool = False,
                            disable_tqdm: bool = False, noise: tp.Optional[torch.Tensor] = None, **sampler_kwargs) -> torch.Tensor:
    """Same arguments, Gaussian-draw order (seed -> initial noise -> VAE noise of the init audio -> sampler draws) and return value
    as the reference's generation.py:95-261: audio ``[B, channels, sample_size]`` (or latents with ``return_latents``).
    ``noise=`` (build extension) overrides the initial draw, for parity tests.  Pinned against the reference's own runs by
    tests/test_reference_generate.py."""
    if model.conditioner is not None:
        model.conditioner.set_device(device)
    assert conditioning or conditioning_tensors, "Must provide either conditioning or conditioning_tensors"
    if conditioning_tensors is None:
        conditioning_tensors = model.conditioner(conditioning)
    cond_inputs = {k: (None if v is None else v.float()) for k, v in model.get_conditioning_inputs(conditioning_tensors).items()}
    # Negative prompts.  The reference resets `negative_conditioning_tensors` to {} before looking at it (generation.py:148-155), so
    # its own call ends in a KeyError; what it evidently means to do -- hand the negative set to the denoiser, where the
    # unconditional CFG half attends to it instead of the null embed (dit.py:294-300) -- is what happens here.
    if negative_conditioning_tensors is None and negative_conditioning:
        negative_conditioning_tensors = model.conditioner(negative_conditioning)
    if negative_conditioning_tensors:
        negative = model.get_conditioning_inputs(negative_conditioning_tensors, negative=True)
        cond_inputs.update({k: (None if v is None else v.float()) for k, v in negative.items()})
    num_sample = next(iter(conditioning_tensors.values()))[0].shape[0]           # batch size = that of the first conditioning tensor

    ratio = model.pretransform.downsampling_ratio if model.pretransform else 1
    length = sample_size // ratio                                                # what the denoiser sees (latent frames)

    torch.manual_seed(int(seed if seed != -1 else np.random.randint(0, 2**32 - 1, dtype=np.uint32)))
    noise = torch.randn([num_sample, model.io_channels, length], device=device) if noise is None else noise.to(device)

    init, mask = None, None
    if init_audio is not None:
        init = _init_latents(model, init_audio, sample_size, num_sample, device)
        if mask_args is not None:                       # in-/out-painting: generation.py:195-213
            init = _cut_and_paste(init, mask_args, length)
            mask = build_mask(length, mask_args).to(device)
        else:                                           # variation: start from init + noise at init_noise_level (generation.py:214-217)
            sampler_kwargs["sigma_max"] = init_noise_level

    shared = dict(cfg_scale=cfg_scale, batch_cfg=True, rescale_cfg=True, device=device, disable_tqdm=disable_tqdm)
    if model.diffusion_objective == "v":
        sampled = sample_k(model.model, noise, init, mask, steps, **sampler_kwargs, **cond_inputs, **shared)
    elif model.diffusion_objective == "rectified_flow":                # generation.py:235-244
        for k_diffusion_only in ("sigma_min", "sampler_type"):
            sampler_kwargs.pop(k_diffusion_only, None)
        sampled = sample_rf(model.model, noise, init_data=init, steps=steps, **sampler_kwargs, **cond_inputs, **shared)
    else:
        raise ValueError(f"unknown diffusion objective {model.diffusion_objective}")
    if model.pretransform and not return_latents:
        sampled = model.pretransform.decode(sampled)
    return sampled


def build_mask(sample_size, mask_args):
    """Soft inpainting mask of length ``sample_size`` (reference generation.py:269-290): 1 = keep the input, 0 = generate,
    Hann ramps of ``softnessL`` / ``softnessR`` percent at the edges of [maskstart, maskend) percent, everything scaled by
    ``1 - marination`` so that the kept region is released before the last steps."""
    start = math.floor(mask_args["maskstart"] / 100.0 * sample_size)
    end = math.ceil(mask_args["maskend"] / 100.0 * sample_size)
    soft_l = round(mask_args["softnessL"] / 100.0 * sample_size)
    soft_r = round(mask_args["softnessR"] / 100.0 * sample_size)
    mask = torch.zeros(sample_size)
    mask[start:end] = 1
    mask[start:start + soft_l] = torch.hann_window(soft_l * 2, periodic=False)[:soft_l]      # rising half
    mask[end - soft_r:end] = torch.hann_window(soft_r * 2, periodic=False)[soft_r:]          # falling half
    if mask_args["marination"] > 0:
        mask = mask * (1 - mask_args["marination"])
    return mask
