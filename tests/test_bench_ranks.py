"""bench.py's own multi-rank logic on CPU: the exact code an 8-GPU launch runs -- prompt sharding, warm-up,
barrier-bracketed timed region, MAX-reduced clock, the single all-gather, the one JSON line on rank 0 -- with the generation stubbed
out (SAT_BENCH_STUB=1) and gloo in place of RCCL; plus the argument check of the 8-GPU command line (--dry-run touches no GPU)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 must print exactly ONE JSON line, got {len(lines)}:\n{r.stdout[-1000:]}"
    return json.loads(lines[0])


def test_bench_rank_logic_world2_gloo():
    world, batch, steps, warmup = 2, 3, 2, 1
    line = _run(["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port",
                 str(_free_port()), "bench.py", "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup), "--batch", str(batch)],
                {"SAT_BENCH_STUB": "1"})
    assert line["stub"] and line["n_gpus"] == world and line["steps"] == steps and line["warmup"] == warmup
    assert line["rccl_ranks"] == world
    assert line["prompt_ids_rank0"] == [0, 2, 4]                       # rank-strided (reference generate.py:119-120)
    assert line["gathered_shape"] == [world * batch, 2, 64]
    # the gather restores the ORIGINAL prompt order; the stub encodes (prompt id, seed of the last timed step = 2000 + steps - 1)
    seed = 2000 + steps - 1
    assert line["gathered_first_samples"] == [(pid * 131 + seed) % 30000 for pid in range(world * batch)]
    assert line["ms_per_step"] > 0


def test_bench_self_launch_world2_gloo():
    """plain `python bench.py --gpus 2`: bench.py starts its own ranks (the path a user takes; the driver passes torch.distributed.run)"""
    line = _run(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1"], {"SAT_BENCH_STUB": "1"})
    assert line["n_gpus"] == 2 and line["gathered_first_samples"] == [(pid * 131 + 2000) % 30000 for pid in range(2)]


def test_bench_dry_run_8_gpus():
    line = _run(["bench.py", "--gpus", "8", "--batch", "8", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert line["dry_run"] and line["global_batch"] == 64 and line["collectives_per_step"] == 1
    ids = line["prompt_ids_per_rank"]
    assert len(ids) == 8 and all(len(r) == 8 for r in ids)
    assert sorted(i for r in ids for i in r) == list(range(64))
    assert ids[3] == list(range(3, 64, 8))


def test_bench_defaults_to_config3_at_n_gt_1():
    """without --batch, N > 1 means BASELINE config 3 (8 prompts per GPU: 64 on 8 GPUs); one GPU means config 2 (one prompt).
    The stub run also checks that every rank's own prompts survive the gather + host copy that the timed step ends with."""
    line = _run(["bench.py", "--gpus", "8", "--dry-run"])
    assert line["prompts_per_gpu"] == 8 and line["global_batch"] == 64
    line = _run(["bench.py", "--gpus", "1", "--dry-run"])
    assert line["prompts_per_gpu"] == 1 and line["global_batch"] == 1
    line = _run(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], {"SAT_BENCH_STUB": "1"})
    assert line["n_gpus"] == 2 and line["gathered_shape"] == [16, 2, 64]
    assert line["gathered_first_samples"] == [(pid * 131 + 2000) % 30000 for pid in range(16)]


def test_bench_rejects_mismatched_world():
    env = dict(os.environ, SAT_BENCH_STUB="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
