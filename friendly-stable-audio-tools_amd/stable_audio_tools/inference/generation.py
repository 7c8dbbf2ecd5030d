"""``generate_diffusion_cond`` (reference ``inference/generation.py:95-261``) -- same signature,
same RNG order (manual_seed -> initial noise -> init-audio VAE noise -> sampler noise), same
return values; the sampler loop and the decode run on the HIP C ABI."""
import math
import typing as tp

import numpy as np
import torch

from .sampling import sample_k, sample_rf
from .utils import prepare_audio


@torch.no_grad()
def generate_diffusion_cond(model, steps: int = 250, cfg_scale: float = 6, conditioning: tp.Optional[tp.List[dict]] = None,
                            conditioning_tensors: tp.Optional[dict] = None, negative_conditioning: tp.Optional[tp.List[dict]] = None,
                            negative_conditioning_tensors: tp.Optional[dict] = None, sample_size: int = 2097152, seed: int = -1,
                            device: str = "cuda", init_audio: tp.Optional[tp.Tuple[int, torch.Tensor]] = None,
                            init_noise_level: float = 1.0, mask_args: dict = None, return_latents: bool = False,
                            disable_tqdm: bool = False, noise: tp.Optional[torch.Tensor] = None, **sampler_kwargs) -> torch.Tensor:
    """``noise=`` (build extension) overrides the initial Gaussian draw, for parity tests."""
    if model.conditioner is not None:
        model.conditioner.set_device(device)
    audio_sample_size = sample_size
    if model.pretransform:
        sample_size //= model.pretransform.downsampling_ratio          # generation.py:139-140

    assert conditioning or conditioning_tensors, "Must provide either conditioning or conditioning_tensors"
    if conditioning_tensors is None:
        conditioning_tensors = model.conditioner(conditioning)
    conditioning_inputs = model.get_conditioning_inputs(conditioning_tensors)

    if negative_conditioning or negative_conditioning_tensors:
        # The reference overwrites negative_conditioning_tensors with {} before testing it
        # (generation.py:148-155), so negative prompts raise KeyError there; refuse loudly instead.
        raise NotImplementedError("negative conditioning is broken in the reference API (generation.py:148-155) and not offered here")

    num_sample = list(conditioning_tensors.values())[0][0].shape[0]    # generation.py:158

    seed = seed if seed != -1 else np.random.randint(0, 2**32 - 1, dtype=np.uint32)
    torch.manual_seed(int(seed))
    if noise is None:
        noise = torch.randn([num_sample, model.io_channels, sample_size], device=device)
    else:
        noise = noise.to(device)

    if init_audio is not None:
        in_sr, init_audio = init_audio
        io_channels = model.pretransform.io_channels if model.pretransform else model.io_channels
        init_audio = prepare_audio(init_audio, in_sr=in_sr, target_sr=model.sample_rate, target_length=audio_sample_size,
                                   target_channels=io_channels, device=device)
        if model.pretransform:
            init_audio = model.pretransform.encode(init_audio)         # samples the VAE
        init_audio = init_audio.repeat(num_sample, 1, 1)
    else:
        init_audio = None
        init_noise_level = None
        mask_args = None

    mask = None
    if init_audio is not None and mask_args is not None:
        # inpainting / outpainting (generation.py:195-213), in latent units: cut & paste the init latents, then the soft mask
        cropfrom = math.floor(mask_args["cropfrom"] / 100.0 * sample_size)
        pastefrom = math.floor(mask_args["pastefrom"] / 100.0 * sample_size)
        pasteto = math.ceil(mask_args["pasteto"] / 100.0 * sample_size)
        assert pastefrom < pasteto, "Paste From should be less than Paste To"
        croplen = min(pasteto - pastefrom, sample_size - cropfrom)
        cutpaste = init_audio.new_zeros(init_audio.shape)
        cutpaste[:, :, pastefrom:pastefrom + croplen] = init_audio[:, :, cropfrom:cropfrom + croplen]
        init_audio = cutpaste
        mask = build_mask(sample_size, mask_args).to(device)
    elif init_audio is not None:
        sampler_kwargs["sigma_max"] = init_noise_level                 # variations: generation.py:214-217

    conditioning_inputs = {k: (v.float() if v is not None else v) for k, v in conditioning_inputs.items()}

    if model.diffusion_objective == "v":
        sampled = sample_k(model.model, noise, init_audio, mask, steps, **sampler_kwargs, **conditioning_inputs, cfg_scale=cfg_scale,
                           batch_cfg=True, rescale_cfg=True, device=device, disable_tqdm=disable_tqdm)
    elif model.diffusion_objective == "rectified_flow":                # generation.py:235-244
        sampler_kwargs.pop("sigma_min", None)
        sampler_kwargs.pop("sampler_type", None)
        sampled = sample_rf(model.model, noise, init_data=init_audio, steps=steps, **sampler_kwargs, **conditioning_inputs,
                            cfg_scale=cfg_scale, batch_cfg=True, rescale_cfg=True, device=device, disable_tqdm=disable_tqdm)
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
