"""Multi-GPU layout of the sampling path: one process per GPU, full model replica per rank, prompts sharded
rank-strided exactly like the reference's ``generate.py:119-120`` (``items[rank::world_size]``) -- no exchange
during compute.  The only collective is ONE all-gather of the finished int16 audio (RCCL over xGMI when the
process group backend is "nccl"; "gloo" in the CPU tests).  Ragged shards are padded to the largest shard.
"""
import torch
import torch.distributed as dist


def shard_items(items, rank, world_size):
    return list(items)[rank::world_size]


def shard_sizes(n_items, world_size):
    return [len(range(r, n_items, world_size)) for r in range(world_size)]


def gather_sharded(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """local [n_local, ...] (this rank's items, rank-strided order) -> [n_items, ...] in the ORIGINAL item order,
    on every rank.  One all_gather_into_tensor; shards shorter than the longest are zero-padded for the transfer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert local.shape[0] == n_items
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_items, world)
    assert local.shape[0] == sizes[rank], f"rank {rank} holds {local.shape[0]} items, expected {sizes[rank]}"
    n_max = max(sizes)
    if local.shape[0] < n_max:
        pad = local.new_zeros((n_max - local.shape[0],) + tuple(local.shape[1:]))
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    gathered = local.new_empty((world * n_max,) + tuple(local.shape[1:]))
    # an all-gather only moves bytes; int16 is not a collective dtype in RCCL/gloo, so ship a uint8 view
    dist.all_gather_into_tensor(gathered.view(torch.uint8).view(-1), local.view(torch.uint8).view(-1), group=group)
    gathered = gathered.view((world, n_max) + tuple(local.shape[1:]))
    out = local.new_empty((n_items,) + tuple(local.shape[1:]))
    for r in range(world):
        out[r::world] = gathered[r, :sizes[r]]
    return out
