"""The conditioner front-end of a conditioned diffusion model, as far as this build computes it itself.

Counterparts of ``Conditioner`` / ``NumberConditioner`` / ``MultiConditioner`` / ``create_multi_conditioner_from_conditioning_config``
(reference ``models/conditioners.py:18-102, 506-599``) and of the ``NumberEmbedder`` they embed numbers with
(``models/adp.py:680-701, 1495-1514``): same module tree and parameter names, so reference checkpoints load.

``NumberConditioner`` runs on the HIP C ABI (``sat_number_embed``: clamp, normalise, Fourier features and the Linear in one
launch per call); like the rest of the package it has no CPU path.  Text / audio encoders (T5, CLAP, ...) need downloaded
checkpoints and are out of scope: their ids are recorded in ``MultiConditioner.external_ids`` and the
caller supplies those entries through ``conditioning_tensors=`` -- the "random T5 embeds" configuration of BASELINE.json.
"""
import typing as tp

import torch
from torch import nn

from .. import _hip

# conditioner types whose tensors must come from outside (encoders this build does not ship)
_EXTERNAL_TYPES = ("t5", "clap_text", "clap_audio", "phoneme", "lut", "pretransform", "int")


class Conditioner(nn.Module):
    def __init__(self, dim: int, output_dim: int, project_out: bool = False):
        super().__init__()
        self.dim, self.output_dim = dim, output_dim
        self.proj_out = nn.Linear(dim, output_dim) if (project_out or dim != output_dim) else nn.Identity()

    def set_device(self, device: tp.Any) -> None:
        raise NotImplementedError()


class LearnedPositionalEmbedding(nn.Module):
    """Parameter holder of adp.py:680-694 (``weights`` [dim / 2]); evaluated inside ``sat_number_embed``."""

    def __init__(self, dim: int):
        super().__init__()
        if dim % 2:
            raise ValueError("LearnedPositionalEmbedding needs an even dim")
        self.weights = nn.Parameter(torch.randn(dim // 2))


class NumberEmbedder(nn.Module):
    """Parameter holder of adp.py:1495-1514: ``embedding.0`` = LearnedPositionalEmbedding(dim), ``embedding.1`` = Linear(dim + 1, features)."""

    def __init__(self, features: int, dim: int = 256):
        super().__init__()
        self.features = features
        self.embedding = nn.Sequential(LearnedPositionalEmbedding(dim), nn.Linear(dim + 1, features))

    def forward(self, values: torch.Tensor, min_val: float = 0.0, max_val: float = 1.0) -> torch.Tensor:
        """values [B] (raw numbers, clamped to [min_val, max_val] and normalised by the kernel) -> [B, features]."""
        pos, lin = self.embedding[0].weights, self.embedding[1]
        values = values.to(pos.device, torch.float32).contiguous()
        out = torch.empty((values.numel(), self.features), dtype=torch.float32, device=pos.device)
        _hip.check(_hip.lib().sat_number_embed(_hip.ptr(values), values.numel(), float(min_val), float(max_val),
                                               _hip.ptr(pos.detach().float().contiguous()), pos.numel(),
                                               _hip.ptr(lin.weight.detach().float().contiguous()), _hip.ptr(lin.bias.detach().float().contiguous()),
                                               self.features, _hip.ptr(out), _hip.stream()))
        return out


class NumberConditioner(Conditioner):
    """A list of numbers -> ``[B, 1, output_dim]`` embeddings and an all-ones ``[B, 1]`` mask (conditioners.py:64-102)."""

    def __init__(self, output_dim: int, min_val: float = 0, max_val: float = 1):
        super().__init__(output_dim, output_dim)
        self.min_val, self.max_val = min_val, max_val
        self.embedder = NumberEmbedder(features=output_dim)

    @property
    def device(self):
        return next(self.embedder.parameters()).device

    def set_device(self, device):
        self.to(device)

    @torch.no_grad()
    def forward(self, floats: tp.List[float]) -> tp.Any:
        values = torch.tensor([float(v) for v in floats], dtype=torch.float32)
        embeds = self.embedder(values, self.min_val, self.max_val).unsqueeze(1)
        return [embeds, torch.ones(embeds.shape[0], 1, device=embeds.device)]


class MultiConditioner(nn.Module):
    """Applies each conditioner to its entry of the per-item metadata dicts (conditioners.py:506-549)."""

    def __init__(self, conditioners: tp.Dict[str, Conditioner], default_keys: tp.Dict[str, str] = {}):
        super().__init__()
        self.conditioners = nn.ModuleDict(conditioners)
        self.default_keys = default_keys
        self.external_ids: tp.List[str] = []

    def set_device(self, device):
        for conditioner in self.conditioners.values():
            conditioner.set_device(device)

    def forward(self, batch_metadata: tp.List[tp.Dict[str, tp.Any]]) -> tp.Dict[str, tp.Any]:
        out = {}
        for key, conditioner in self.conditioners.items():
            lookup, inputs = key, []
            for item in batch_metadata:
                if lookup not in item:
                    if lookup not in self.default_keys:
                        raise ValueError(f"Conditioner key {lookup} not found in batch metadata")
                    lookup = self.default_keys[lookup]        # sticks for the rest of the batch, as in the reference (:533-538)
                value = item[lookup]
                # the reference unwraps ANY list, but a tuple only when it has exactly one element (operator precedence, :541)
                if isinstance(value, list) or (isinstance(value, tuple) and len(value) == 1):
                    value = value[0]
                inputs.append(value)
            out[key] = conditioner(inputs)
        return out


def create_multi_conditioner_from_conditioning_config(config: tp.Dict[str, tp.Any]) -> MultiConditioner:
    """conditioners.py:552-599 for the conditioner types this build evaluates ("number"); encoder-backed types are listed in
    ``external_ids`` instead of being instantiated."""
    built, external = {}, []
    for entry in config["configs"]:
        kind = entry["type"]
        if kind == "number":
            built[entry["id"]] = NumberConditioner(**{"output_dim": config["cond_dim"], **entry["config"]})
        elif kind in _EXTERNAL_TYPES:
            external.append(entry["id"])
        else:
            raise ValueError(f"Unknown conditioner type: {kind}")
    multi = MultiConditioner(built, default_keys=config.get("default_keys", {}))
    multi.external_ids = external
    return multi
