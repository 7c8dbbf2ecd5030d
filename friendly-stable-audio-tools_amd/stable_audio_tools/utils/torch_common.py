"""Host-side helpers with the reference's names (reference ``utils/torch_common.py``)."""
import os
import random

import numpy as np
import torch


def exists(x):
    return x is not None


def get_world_size():
    # reference utils/torch_common.py:12-16
    if not torch.distributed.is_available() or not torch.distributed.is_initialized():
        return 1
    return torch.distributed.get_world_size()


def get_rank():
    # reference utils/torch_common.py:19-24
    if not torch.distributed.is_available() or not torch.distributed.is_initialized():
        return 0
    return torch.distributed.get_rank()


def print_once(*args):
    if get_rank() == 0:
        print(*args)


def set_seed(seed: int = 0):
    # reference utils/torch_common.py:32-38
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def count_parameters(model: torch.nn.Module):
    # parameters AND buffers, as the reference does (utils/torch_common.py:41-43)
    return sum(p.numel() for p in model.parameters()) + sum(p.numel() for p in model.buffers())


def copy_state_dict(model, state_dict):
    """Shape-matched partial load (reference utils/torch_common.py:46-61)."""
    model_state_dict = model.state_dict()
    for key in state_dict:
        if key in model_state_dict and state_dict[key].shape == model_state_dict[key].shape:
            value = state_dict[key]
            if isinstance(value, torch.nn.Parameter):
                value = value.data
            model_state_dict[key] = value
    model.load_state_dict(model_state_dict, strict=False)
