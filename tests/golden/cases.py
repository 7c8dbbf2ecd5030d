"""Seed-defined parity cases shared by make_golden.py (reference side, build container only) and the tests
(oracle / HIP side).  Weights and inputs are regenerated from (seed, name) by
``stable_audio_tools.synthetic`` -- only the reference OUTPUTS are stored in tests/golden/*.npz."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "friendly-stable-audio-tools_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from stable_audio_tools import model_configs as MC  # noqa: E402
from stable_audio_tools import synthetic  # noqa: E402

GOLDEN_DIR = HERE

# reduced DiT used by the op-level goldens
SMALL_DIT = dict(io_channels=64, embed_dim=256, depth=3, num_heads=4, cond_token_dim=128, global_cond_dim=96,
                 project_cond_tokens=False, transformer_type="continuous_transformer")
FULL_DIT = dict(io_channels=64, embed_dim=1536, depth=24, num_heads=24, cond_token_dim=768, global_cond_dim=1536,
                project_cond_tokens=False, transformer_type="continuous_transformer")
SMALL_VAE = dict(channels=16, c_mults=[1, 2, 4, 8, 16], strides=[2, 4, 4, 8, 8])
FULL_VAE = dict(channels=128, c_mults=[1, 2, 4, 8, 16], strides=[2, 4, 4, 8, 8])


# generation.py:269-290 (build_mask): percent-valued inpainting arguments as the Gradio UI passes them
MASK_ARGS = [
    dict(cropfrom=0, pastefrom=0, pasteto=100, maskstart=25, maskend=75, softnessL=10, softnessR=5, marination=0),
    dict(cropfrom=10, pastefrom=30, pasteto=90, maskstart=0, maskend=50.5, softnessL=0, softnessR=20, marination=0.3),
    dict(cropfrom=50, pastefrom=0, pasteto=80, maskstart=33.3, maskend=100, softnessL=3.7, softnessR=0, marination=0),
]


# reference-orchestration goldens (make_golden.py gen_generate -> generate.npz): reduced SA-Open model (MC.reduced), 2 prompts
GEN = {
    "t_len": 24,
    "calls": {      # generate_diffusion_cond(model, conditioning_tensors=..., sample_size=t_len*ratio, device="cpu", **kw)
        "plain": dict(steps=6, cfg_scale=7.0, seed=5, sampler_type="dpmpp-3m-sde", sigma_min=0.3, sigma_max=500),
        "a2a": dict(steps=5, cfg_scale=7.0, seed=6, init_audio=True, init_noise_level=4.0, sampler_type="dpmpp-3m-sde", sigma_min=0.3,
                    sigma_max=500),
        "inpaint": dict(steps=5, cfg_scale=7.0, seed=7, init_audio=True, sampler_type="dpmpp-2m-sde", sigma_min=0.3, sigma_max=50,
                        mask_args=dict(cropfrom=0, pastefrom=25, pasteto=100, maskstart=25, maskend=75, softnessL=10, softnessR=10,
                                       marination=0)),
    },
    "sample_k_mask": dict(maskstart=20, maskend=80, softnessL=15, softnessR=10, marination=0.1),
    "sample_k": {   # sample_k(model.model, noise, init, mask, device="cpu", cfg_scale=7, **kw, **conditioning_inputs)
        "heun_inpaint": dict(steps=4, sampler_type="k-heun", sigma_min=0.3, sigma_max=80.0, init=True, mask=True),
        "lms_inpaint": dict(steps=5, sampler_type="k-lms", sigma_min=0.3, sigma_max=80.0, init=True, mask=True),
        "dpm2_variation": dict(steps=4, sampler_type="k-dpm-2", sigma_min=0.3, sigma_max=6.0, init=True),
        "ancestral_plain": dict(steps=4, sampler_type="k-dpmpp-2s-ancestral", sigma_min=0.3, sigma_max=80.0),
        "fast_inpaint": dict(steps=6, sampler_type="k-dpm-fast", sigma_min=0.3, sigma_max=80.0, init=True, mask=True),
    },
}


# full-size multi-step fixture (make_traj_golden.py -> traj_full.npz): 12 steps of DPM-Solver++(3M) SDE, CFG 7, noise injected
TRAJ = dict(steps=12, sigma_min=0.3, sigma_max=500.0, cfg_scale=7.0, snapshots=(4, 8, 12))


# the headline's own length (make_traj100_golden.py -> traj100_full.npz): 100 steps, same model / conditioning / schedule family
TRAJ100 = dict(steps=100, sigma_min=0.3, sigma_max=500.0, cfg_scale=7.0, snapshots=(12, 25, 50, 100),
               audio_windows={"start": 0, "middle": 1048576}, audio_window_len=65536)


# further (prompt, seed) pairs of the same 100-step fixture: variant v has its own conditioning and its own noise
# draws; stored as traj100_full_v{v}.npz with the latents after 50 / 100 steps only (no audio windows)
TRAJ100_VARIANTS = (0, 1, 2)
TRAJ100_VARIANT_SNAPSHOTS = (50, 100)


def traj100_inputs(variant=0):
    """(cross_attn_cond, global_cond, unit initial noise, per-step unit noise) of the 100-step full-size fixture; `variant` > 0 =
    another (prompt, seed) pair"""
    _, _, c, g = dit_inputs(1, 1024, 768, 1536, 1 + 10 * variant)
    base = 700 + 1000 * variant
    tag = "" if variant == 0 else f"_v{variant}"
    noise = synthetic.synth_input(f"traj100_noise{tag}", (1, 64, 1024), base)
    step_noise = [synthetic.synth_input(f"traj100_sn{i}{tag}", (1, 64, 1024), base + 10 + i) for i in range(TRAJ100["steps"])]
    return c, g, noise, step_noise


def traj_inputs():
    """(cross_attn_cond, global_cond, unit initial noise, per-step unit noise) of the full-size trajectory fixture"""
    _, _, c, g = dit_inputs(1, 1024, 768, 1536, 1)
    noise = synthetic.synth_input("traj_noise", (1, 64, 1024), 500)
    step_noise = [synthetic.synth_input(f"traj_sn{i}", (1, 64, 1024), 510 + i) for i in range(TRAJ["steps"])]
    return c, g, noise, step_noise


def dit_inputs(b, t_len, cond_dim, global_dim, seed, lc=130):
    x = synthetic.synth_input("x", (b, 64, t_len), seed)
    c = synthetic.synth_input("c", (b, lc, cond_dim), seed + 1)
    g = synthetic.synth_input("g", (b, global_dim), seed + 2)
    t = (torch.arange(b, dtype=torch.float32) + 1) / (b + 1)
    return x, t, c, g


def vae_kwargs(v, decoder):
    if decoder:
        return dict(out_channels=2, channels=v["channels"], c_mults=v["c_mults"], strides=v["strides"], latent_dim=64, use_snake=True,
                    final_tanh=False)
    return dict(in_channels=2, channels=v["channels"], c_mults=v["c_mults"], strides=v["strides"], latent_dim=128, use_snake=True)


def load(name):
    import numpy as np
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}
