"""Small host-side helpers under the reference's names (``utils/torch_common.py``): process-group queries, seeding,
parameter counting, shape-matched partial state-dict loading."""
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def exists(x):
    return x is not None


def _group_ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    """1 outside a process group (single-GPU runs never initialise torch.distributed)."""
    return dist.get_world_size() if _group_ready() else 1


def get_rank():
    """0 outside a process group."""
    return dist.get_rank() if _group_ready() else 0


def print_once(*args):
    """print on rank 0 only."""
    if get_rank() == 0:
        print(*args)


def set_seed(seed: int = 0):
    """Seed every generator the scripts touch: torch (host + all devices), numpy, ``random`` and the hash seed."""
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def count_parameters(model: torch.nn.Module):
    """Elements of parameters plus buffers (the reference counts both)."""
    return sum(t.numel() for group in (model.parameters(), model.buffers()) for t in group)


def copy_state_dict(model, state_dict):
    """Load the entries of ``state_dict`` whose key exists in ``model`` with the same shape; ignore everything else."""
    target = model.state_dict()
    usable = {k: (v.data if isinstance(v, torch.nn.Parameter) else v)
              for k, v in state_dict.items() if k in target and v.shape == target[k].shape}
    target.update(usable)
    model.load_state_dict(target, strict=False)
