"""``generate_diffusion_cond`` (reference ``inference/generation.py:95-261``) -- same signature,
same RNG order (manual_seed -> initial noise -> init-audio VAE noise -> sampler noise), same
return values; the sampler loop and the decode run on the HIP C ABI."""
import typing as tp

import numpy as np
import torch

from .sampling import sample_k
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

    if init_audio is not None and mask_args is not None:
        raise NotImplementedError("inpainting (mask_args) is outside the supported hot path")
    if init_audio is not None:
        sampler_kwargs["sigma_max"] = init_noise_level                 # variations: generation.py:214-217

    conditioning_inputs = {k: (v.float() if v is not None else v) for k, v in conditioning_inputs.items()}

    if model.diffusion_objective != "v":
        raise NotImplementedError("only the v-objective / k-diffusion path is implemented (rectified flow is out of scope)")
    sampled = sample_k(model.model, noise, init_audio, None, steps, **sampler_kwargs, **conditioning_inputs, cfg_scale=cfg_scale,
                       batch_cfg=True, rescale_cfg=True, device=device, disable_tqdm=disable_tqdm)

    if model.pretransform and not return_latents:
        sampled = model.pretransform.decode(sampled)
    return sampled
