"""N > 1 path on CPU: world_size-2 gloo process group, rank-strided sharding + the single all-gather
(the same helper bench.py and a generate.py-style driver use with the nccl/RCCL backend)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_audio(item, n=64):
    g = torch.Generator().manual_seed(1000 + item)
    return torch.randint(-32768, 32767, (2, n), generator=g, dtype=torch.int32).to(torch.int16)


def _worker(rank, world, n_items, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stable_audio_tools.inference.distributed import gather_sharded, shard_items
        from stable_audio_tools.utils.torch_common import get_rank, get_world_size
        assert get_rank() == rank and get_world_size() == world
        mine = shard_items(range(n_items), rank, world)
        local = torch.stack([_fake_audio(i) for i in mine]) if mine else torch.zeros((0, 2, 64), dtype=torch.int16)
        full = gather_sharded(local, n_items)
        want = torch.stack([_fake_audio(i) for i in range(n_items)])
        q.put((rank, bool(torch.equal(full, want)), mine))
    except Exception as e:   # fail fast instead of letting the parent wait for the queue timeout
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 5, 2])
def test_rank_strided_shards_and_single_allgather(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + n_items
    procs = [ctx.Process(target=_worker, args=(r, world, n_items, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = []
    for rank, ok, mine in results:
        assert ok, f"rank {rank}: gathered audio differs from the original item order"
        assert mine == list(range(n_items))[rank::world]
        seen += mine
    assert sorted(seen) == list(range(n_items)), "every prompt must be generated exactly once"


def test_single_process_is_identity():
    from stable_audio_tools.inference.distributed import gather_sharded, shard_sizes
    x = torch.arange(12, dtype=torch.int16).view(3, 2, 2)
    assert gather_sharded(x, 3) is x
    assert shard_sizes(5, 2) == [3, 2] and shard_sizes(8, 8) == [1] * 8
