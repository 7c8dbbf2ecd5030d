"""Deterministic synthetic weights / inputs keyed by (seed, tensor name).

Build-owned helper (no counterpart in the reference): BASELINE.json asks for
"synthetic random-init weights and random T5 embeddings".  A freshly constructed reference
DiT is an identity stack (zero-initialised ``to_out`` / FF output / pre- and post-conv,
SURVEY.md F9), so benchmarks and parity fixtures re-draw *every* tensor from this
generator instead.  Values depend only on ``(seed, key, shape)`` -- the GPU box regenerates
bit-identical weights without any file from the build container.
"""
import hashlib
import math

import torch

_KEEP = ("inv_freq",)


def _gen(seed, key):
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:7], "little"))
    return g


def _uniform(shape, bound, g):
    return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound


def synth_tensor(key, shape, seed, template=None, norm_of_v=None):
    """One tensor of the synthetic state dict (fp32, CPU)."""
    g = _gen(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    if any(key.endswith(k) for k in _KEEP):
        return template.clone().float()
    if leaf == "gamma":
        return 0.5 * (1.0 + _uniform(shape, 0.2, g))
    if leaf == "beta" and "norm" in key:
        return _uniform(shape, 0.02, g)
    if leaf in ("alpha", "beta"):          # SnakeBeta log-scale parameters
        return torch.randn(shape, generator=g) * 0.2
    if leaf == "weight_g":
        return norm_of_v.view(shape) * (1.0 + _uniform(shape, 0.2, g))
    if leaf == "weights":                   # LearnedPositionalEmbedding
        return torch.randn(shape, generator=g)
    if key.endswith("timestep_features.weight"):
        return torch.randn(shape, generator=g) * 0.5
    if leaf == "bias":
        return _uniform(shape, 0.05, g)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if any(t in key for t in ("to_qkv", "to_q.", "to_kv")):
            c = 2.0
        elif any(t in key for t in ("to_out", "ff.2", "preprocess_conv", "postprocess_conv")):
            c = 1.0
        elif "transformer" in key or key.startswith(("to_", "model.")):
            c = 0.5
        else:
            c = 1.0
        return _uniform(shape, c / math.sqrt(fan_in), g)
    return _uniform(shape, 0.05, g)


def synth_state_dict(template_sd, seed):
    """Returns a new state dict with the same keys/shapes as ``template_sd``."""
    out = {}
    for key, t in template_sd.items():
        if key.endswith("weight_g"):
            continue
        out[key] = synth_tensor(key, tuple(t.shape), seed, template=t)
    for key, t in template_sd.items():
        if key.endswith("weight_g"):
            v = out[key[:-1] + "v"]
            out[key] = synth_tensor(key, tuple(t.shape), seed, norm_of_v=v.flatten(1).norm(dim=1))
    return {k: out[k] for k in template_sd}


def synth_input(name, shape, seed, scale=1.0):
    return torch.randn(shape, generator=_gen(seed, "input:" + name), dtype=torch.float32) * scale
