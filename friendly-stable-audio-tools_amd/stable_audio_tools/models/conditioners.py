"""Conditioner front-end needed to *build* the conditioning tensors without T5/CLAP
(reference ``models/conditioners.py:18-102, 506-599`` and ``models/adp.py:680-701, 1495-1514``).

Scope note: the metric is quoted on *random T5 embeddings* passed through
``conditioning_tensors=``; T5/CLAP text encoders need a Hugging Face download and are out of
scope.  ``NumberConditioner`` is a 257-wide Fourier feature + one Linear on a handful of scalars
per prompt, evaluated once per generation -- host-side plumbing in torch, not on the hot path.
"""
import math
import typing as tp

import torch
from torch import nn


class Conditioner(nn.Module):
    def __init__(self, dim: int, output_dim: int, project_out: bool = False):
        super().__init__()
        self.dim = dim
        self.output_dim = output_dim
        self.proj_out = nn.Linear(dim, output_dim) if (dim != output_dim or project_out) else nn.Identity()

    def set_device(self, device: tp.Any) -> None:
        raise NotImplementedError()


class LearnedPositionalEmbedding(nn.Module):
    """adp.py:680-694: cat(x, sin(2 pi x w), cos(2 pi x w))"""

    def __init__(self, dim: int):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))

    def forward(self, x):
        x = x[:, None]
        freqs = x * self.weights[None, :] * 2 * math.pi
        return torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)


class NumberEmbedder(nn.Module):
    """adp.py:1495-1514 (TimePositionalEmbedding = LearnedPositionalEmbedding + Linear(dim+1, features))"""

    def __init__(self, features: int, dim: int = 256):
        super().__init__()
        self.features = features
        self.embedding = nn.Sequential(LearnedPositionalEmbedding(dim), nn.Linear(dim + 1, features))

    def forward(self, x):
        if not torch.is_tensor(x):
            x = torch.tensor(x, device=next(self.embedding.parameters()).device)
        shape = x.shape
        return self.embedding(x.reshape(-1)).view(*shape, self.features)


class NumberConditioner(Conditioner):
    """conditioners.py:64-102: clamp -> normalise to [0,1] -> NumberEmbedder -> [B,1,output_dim], ones mask"""

    def __init__(self, output_dim: int, min_val: float = 0, max_val: float = 1):
        super().__init__(output_dim, output_dim)
        self.min_val = min_val
        self.max_val = max_val
        self.embedder = NumberEmbedder(features=output_dim)
        self.device = next(self.embedder.parameters()).device

    def set_device(self, device):
        self.to(device)
        self.device = device

    @torch.no_grad()
    def forward(self, floats: tp.List[float]) -> tp.Any:
        self.device = next(self.embedder.parameters()).device
        floats = torch.tensor([float(x) for x in floats]).to(self.device)
        floats = floats.clamp(self.min_val, self.max_val)
        normalized = ((floats - self.min_val) / (self.max_val - self.min_val)).to(next(self.embedder.parameters()).dtype)
        embeds = self.embedder(normalized).unsqueeze(1)
        return [embeds, torch.ones(embeds.shape[0], 1).to(self.device)]


class MultiConditioner(nn.Module):
    """conditioners.py:506-549"""

    def __init__(self, conditioners: tp.Dict[str, Conditioner], default_keys: tp.Dict[str, str] = {}):
        super().__init__()
        self.conditioners = nn.ModuleDict(conditioners)
        self.default_keys = default_keys

    def set_device(self, device):
        for mod in self.conditioners.values():
            mod.set_device(device)

    def forward(self, batch_metadata: tp.List[tp.Dict[str, tp.Any]]) -> tp.Dict[str, tp.Any]:
        output = {}
        for key, conditioner in self.conditioners.items():
            condition_key = key
            inputs = []
            for x in batch_metadata:
                if condition_key not in x:
                    if condition_key in self.default_keys:
                        condition_key = self.default_keys[condition_key]
                    else:
                        raise ValueError(f"Conditioner key {condition_key} not found in batch metadata")
                v = x[condition_key]
                # single-element list/tuple unwrap, with the reference's operator precedence (conditioners.py:541)
                if isinstance(v, list) or isinstance(v, tuple) and len(v) == 1:
                    inputs.append(v[0])
                else:
                    inputs.append(v)
            output[key] = conditioner(inputs)
        return output


def create_multi_conditioner_from_conditioning_config(config: tp.Dict[str, tp.Any]) -> MultiConditioner:
    """conditioners.py:552-599 for the conditioner types on this build's path.  Text/audio encoders
    ("t5", "clap_text", "clap_audio", ...) need downloaded checkpoints: they are skipped with their id
    recorded in ``MultiConditioner.external_ids`` -- callers supply those entries through
    ``conditioning_tensors=`` (exactly the 'random T5 embeds' configuration of BASELINE.json)."""
    conditioners = {}
    external = []
    cond_dim = config["cond_dim"]
    default_keys = config.get("default_keys", {})
    for info in config["configs"]:
        cid, ctype = info["id"], info["type"]
        cfg = {"output_dim": cond_dim}
        cfg.update(info["config"])
        if ctype == "number":
            conditioners[cid] = NumberConditioner(**cfg)
        elif ctype in ("t5", "clap_text", "clap_audio", "phoneme", "lut", "pretransform", "int"):
            external.append(cid)
        else:
            raise ValueError(f"Unknown conditioner type: {ctype}")
    mc = MultiConditioner(conditioners, default_keys=default_keys)
    mc.external_ids = external
    return mc
