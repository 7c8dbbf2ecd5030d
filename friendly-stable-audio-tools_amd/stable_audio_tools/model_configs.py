"""Shape contracts of the shipped Stable Audio model configs, as Python dicts.

Values restate ``configs/model_configs/txt2audio/stable_audio_open_1_0.json``,
``.../stable_audio_2_0.json`` and ``.../autoencoders/stable_audio_2_0_vae.json`` of the reference
(architecture keys only; training sections omitted).  ``with_text_encoder=False`` drops the
T5 / CLAP entry, whose embeddings are supplied through ``conditioning_tensors=`` (BASELINE.json:
"random T5 embeds").  ``reduced()`` shrinks a config for fast parity tests while keeping every
structural feature (prepend token, GQA cross-attention, partial RoPE, 5-stage codec).
"""
import copy


def _oobleck_vae(channels=128, c_mults=(1, 2, 4, 8, 16), strides=(2, 4, 4, 8, 8), latent_dim=64):
    ratio = 1
    for s in strides:
        ratio *= s
    return {
        "encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": channels, "c_mults": list(c_mults),
                                                  "strides": list(strides), "latent_dim": 2 * latent_dim, "use_snake": True}},
        "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": channels, "c_mults": list(c_mults),
                                                  "strides": list(strides), "latent_dim": latent_dim, "use_snake": True,
                                                  "final_tanh": False}},
        "bottleneck": {"type": "vae"},
        "latent_dim": latent_dim,
        "downsampling_ratio": ratio,
        "io_channels": 2,
    }


def stable_audio_vae():
    return {"model_type": "autoencoder", "sample_size": 65536, "sample_rate": 44100, "audio_channels": 2, "model": _oobleck_vae()}


def _dit_cond(sample_size, text_cfg, with_text_encoder):
    configs = []
    if with_text_encoder:
        configs.append(text_cfg)
    configs += [{"id": "seconds_start", "type": "number", "config": {"min_val": 0, "max_val": 512}},
                {"id": "seconds_total", "type": "number", "config": {"min_val": 0, "max_val": 512}}]
    return {
        "model_type": "diffusion_cond",
        "sample_size": sample_size,
        "sample_rate": 44100,
        "audio_channels": 2,
        "model": {
            "pretransform": {"type": "autoencoder", "iterate_batch": True, "config": _oobleck_vae()},
            "conditioning": {"configs": configs, "cond_dim": 768},
            "diffusion": {
                "cross_attention_cond_ids": ["prompt", "seconds_start", "seconds_total"],
                "global_cond_ids": ["seconds_start", "seconds_total"],
                "type": "dit",
                "config": {"io_channels": 64, "embed_dim": 1536, "depth": 24, "num_heads": 24, "cond_token_dim": 768,
                           "global_cond_dim": 1536, "project_cond_tokens": False, "transformer_type": "continuous_transformer"},
            },
            "io_channels": 64,
        },
    }


def stable_audio_open_1_0(with_text_encoder=False):
    t5 = {"id": "prompt", "type": "t5", "config": {"t5_model_name": "t5-base", "max_length": 128}}
    return _dit_cond(2097152, t5, with_text_encoder)


def stable_audio_2_0(with_text_encoder=False):
    clap = {"id": "prompt", "type": "clap_text", "config": {"audio_model_type": "HTSAT-base", "enable_fusion": True,
                                                             "clap_ckpt_path": "ckpt/clap/music_audioset_epoch_15_esc_90.14.pt",
                                                             "use_text_features": True, "feature_layer_ix": -2}}
    return _dit_cond(12582912, clap, with_text_encoder)


def reduced(config, embed_dim=256, depth=2, num_heads=4, cond_dim=128, channels=64, c_mults=(1, 2, 4), strides=(2, 4, 4)):
    """Small variant of a diffusion_cond or autoencoder config (dim_heads stays 64; cross-attention
    stays GQA: cond_dim/64 kv heads < num_heads)."""
    cfg = copy.deepcopy(config)
    vae = _oobleck_vae(channels=channels, c_mults=c_mults, strides=strides)
    if cfg["model_type"] == "autoencoder":
        cfg["model"] = vae
        return cfg
    cfg["model"]["pretransform"]["config"] = vae
    cfg["model"]["conditioning"]["cond_dim"] = cond_dim
    d = cfg["model"]["diffusion"]["config"]
    d.update({"embed_dim": embed_dim, "depth": depth, "num_heads": num_heads, "cond_token_dim": cond_dim,
              "global_cond_dim": 2 * cond_dim})
    cfg["sample_size"] = 64 * vae["downsampling_ratio"]
    return cfg
