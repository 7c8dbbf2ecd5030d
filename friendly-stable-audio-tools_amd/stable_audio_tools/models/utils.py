"""``load_ckpt_state_dict`` / ``remove_weight_norm_from_model`` (reference ``models/utils.py:6-21``)."""
import torch


def exists(x):
    return x is not None


def load_ckpt_state_dict(ckpt_path):
    if ckpt_path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(ckpt_path)
    return torch.load(ckpt_path, map_location="cpu")["state_dict"]


def remove_weight_norm_from_model(model):
    """The HIP plans fold ``weight_g``/``weight_v`` when they are (re)built, so there is nothing
    to strip; kept for API compatibility and returns the model unchanged."""
    return model
