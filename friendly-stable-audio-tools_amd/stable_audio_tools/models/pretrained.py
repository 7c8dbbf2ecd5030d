"""``get_pretrained_model(name)``: fetch ``model_config.json`` and the weights of a Hugging Face model repository, build the
model with ``create_model_from_config`` and load the checkpoint (counterpart of the reference's ``models/pretrained.py``).
Needs network access; the benchmark and the tests use synthetic weights instead."""
import json

from .factory import create_model_from_config
from .utils import load_ckpt_state_dict

_WEIGHT_FILES = ("model.safetensors", "model.ckpt")      # tried in this order


def get_pretrained_model(name: str):
    from huggingface_hub import hf_hub_download

    def fetch(filename):
        return hf_hub_download(name, filename=filename, repo_type="model")

    with open(fetch("model_config.json")) as handle:
        model_config = json.load(handle)
    model = create_model_from_config(model_config)

    last_error = None
    for filename in _WEIGHT_FILES:
        try:
            checkpoint = fetch(filename)
        except Exception as err:                          # not in the repository: try the next format
            last_error = err
            continue
        model.load_state_dict(load_ckpt_state_dict(checkpoint))
        return model, model_config
    raise last_error
