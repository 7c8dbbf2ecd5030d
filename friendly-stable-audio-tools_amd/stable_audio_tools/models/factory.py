"""``create_model_from_config`` & friends (reference ``models/factory.py:4-142``) for the model
types on this build's hot path: ``diffusion_cond`` (DiT) and ``autoencoder`` (Oobleck VAE)."""
import json


def create_model_from_config(model_config):
    model_type = model_config["model_type"]
    if model_type == "autoencoder":
        from .autoencoders import create_autoencoder_from_config
        return create_autoencoder_from_config(model_config)
    if model_type in ("diffusion_cond", "diffusion_cond_inpaint"):
        from .diffusion import create_diffusion_cond_from_config
        return create_diffusion_cond_from_config(model_config)
    if model_type in ("diffusion_uncond", "diffusion_prior", "diffusion_autoencoder", "lm"):
        raise NotImplementedError(f"model type '{model_type}' exists in the reference but is outside this build's hot path")
    raise NotImplementedError(f"Unknown model type: {model_type}")


def create_model_from_config_path(model_config_path):
    with open(model_config_path) as f:
        return create_model_from_config(json.load(f))


def create_pretransform_from_config(pretransform_config, sample_rate):
    pretransform_type = pretransform_config["type"]
    if pretransform_type != "autoencoder":
        raise NotImplementedError(f"pretransform type '{pretransform_type}' is outside this build's hot path")
    from .autoencoders import create_autoencoder_from_config
    from .pretransforms import AutoencoderPretransform
    # fake top-level config to hand the sample rate to the autoencoder factory (factory.py:41-44)
    autoencoder = create_autoencoder_from_config({"sample_rate": sample_rate, "model": pretransform_config["config"]})
    pretransform = AutoencoderPretransform(autoencoder, scale=pretransform_config.get("scale", 1.0),
                                           model_half=pretransform_config.get("model_half", False),
                                           iterate_batch=pretransform_config.get("iterate_batch", False),
                                           chunked=pretransform_config.get("chunked", False))
    pretransform.enable_grad = pretransform_config.get("enable_grad", False)
    pretransform.eval().requires_grad_(pretransform.enable_grad)
    return pretransform


def create_bottleneck_from_config(bottleneck_config):
    bottleneck_type = bottleneck_config["type"]
    if bottleneck_type == "vae":
        from .bottleneck import VAEBottleneck
        bottleneck = VAEBottleneck()
    else:
        raise NotImplementedError(f"bottleneck type '{bottleneck_type}' is outside this build's hot path")
    if not bottleneck_config.get("requires_grad", True):
        for p in bottleneck.parameters():
            p.requires_grad = False
    return bottleneck
