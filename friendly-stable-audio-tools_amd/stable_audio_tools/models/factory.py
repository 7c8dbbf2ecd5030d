"""Model construction from the reference's JSON configs (counterpart of ``models/factory.py:4-142``), for what this build
accelerates: ``diffusion_cond`` models with a DiT denoiser and ``autoencoder`` models / pretransforms with an Oobleck VAE.
Every other ``model_type`` / ``type`` the reference knows is refused with a reason instead of being half-built.
"""
import json

_REFERENCE_ONLY_MODELS = ("diffusion_uncond", "diffusion_prior", "diffusion_autoencoder", "lm")


def create_model_from_config(model_config):
    kind = model_config["model_type"]
    if kind == "autoencoder":
        from .autoencoders import create_autoencoder_from_config as build
    elif kind in ("diffusion_cond", "diffusion_cond_inpaint"):
        from .diffusion import create_diffusion_cond_from_config as build
    elif kind in _REFERENCE_ONLY_MODELS:
        raise NotImplementedError(f"model type '{kind}' exists in the reference but is outside this build's hot path")
    else:
        raise NotImplementedError(f"Unknown model type: {kind}")
    return build(model_config)


def create_model_from_config_path(model_config_path):
    with open(model_config_path) as handle:
        return create_model_from_config(json.load(handle))


def create_pretransform_from_config(pretransform_config, sample_rate):
    if pretransform_config["type"] != "autoencoder":
        raise NotImplementedError(f"pretransform type '{pretransform_config['type']}' is outside this build's hot path")
    from .autoencoders import create_autoencoder_from_config
    from .pretransforms import AutoencoderPretransform
    # the autoencoder factory wants a top-level config: wrap the pretransform's own one together with the sample rate
    codec = create_autoencoder_from_config({"sample_rate": sample_rate, "model": pretransform_config["config"]})
    options = {key: pretransform_config.get(key, default)
               for key, default in (("scale", 1.0), ("model_half", False), ("iterate_batch", False), ("chunked", False))}
    pretransform = AutoencoderPretransform(codec, **options)
    pretransform.enable_grad = pretransform_config.get("enable_grad", False)
    return pretransform.eval().requires_grad_(pretransform.enable_grad)


def create_bottleneck_from_config(bottleneck_config):
    if bottleneck_config["type"] != "vae":
        raise NotImplementedError(f"bottleneck type '{bottleneck_config['type']}' is outside this build's hot path")
    from .bottleneck import VAEBottleneck
    bottleneck = VAEBottleneck()
    if not bottleneck_config.get("requires_grad", True):
        bottleneck.requires_grad_(False)
    return bottleneck
