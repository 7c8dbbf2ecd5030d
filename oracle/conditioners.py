"""Oracle: the number conditioners of the diffusion model -- RESTATED from the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).

``NumberConditioner.forward`` (reference ``models/conditioners.py:64-102``) with its ``NumberEmbedder``
(``models/adp.py:1495-1514``: ``LearnedPositionalEmbedding`` :680-694 + ``Linear``) and ``MultiConditioner.forward``
(:520-549) for number-only conditioner sets, functional over a reference-format state dict.  Pinned by the reference's own
outputs in tests/golden/ops.npz (``number_cond``) and tests/golden/host.npz (``get_conditioning_inputs``).
"""
import math

import torch


def number_conditioner(sd, prefix, floats, min_val, max_val):
    """-> (embeds [B, 1, features], mask [B, 1]); sd keys: ``{prefix}embedder.embedding.0.weights`` / ``.1.weight`` / ``.1.bias``."""
    x = torch.tensor([float(v) for v in floats]).clamp(min_val, max_val)
    x = ((x - min_val) / (max_val - min_val))[:, None]
    freqs = x * sd[prefix + "embedder.embedding.0.weights"][None, :] * 2 * math.pi
    fouriered = torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)
    emb = torch.nn.functional.linear(fouriered, sd[prefix + "embedder.embedding.1.weight"], sd[prefix + "embedder.embedding.1.bias"])
    return emb.unsqueeze(1), torch.ones(emb.shape[0], 1)


def multi_conditioner(sd, prefix, number_ids, batch_metadata, min_val=0.0, max_val=512.0):
    """{id: (embeds, mask)} for the number conditioners ``number_ids`` of a MultiConditioner stored under ``prefix``."""
    return {cid: number_conditioner(sd, f"{prefix}conditioners.{cid}.", [m[cid] for m in batch_metadata], min_val, max_val) for cid in number_ids}
