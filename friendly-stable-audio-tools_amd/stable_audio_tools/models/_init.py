"""Parameter construction helpers.

The reference initialises with PyTorch defaults and zero-inits some layers
(``transformer.py:274-277,318-319``, ``dit.py:130-133``).  Constructing the 1.06 B-parameter
SA-Open DiT with random init takes tens of seconds on the host for values that a checkpoint
(or the synthetic generator) overwrites immediately, so ``skip_init()`` lets callers build
the module tree with uninitialised storage.
"""
import contextlib

import torch
from torch import nn

_SKIP = False


@contextlib.contextmanager
def skip_init():
    global _SKIP
    old, _SKIP = _SKIP, True
    try:
        yield
    finally:
        _SKIP = old


def linear(fan_in, fan_out, bias=True, zero=False):
    if _SKIP:
        m = torch.nn.utils.skip_init(nn.Linear, fan_in, fan_out, bias=bias)
    else:
        m = nn.Linear(fan_in, fan_out, bias=bias)
        if zero:
            nn.init.zeros_(m.weight)
            if bias:
                nn.init.zeros_(m.bias)
    return m


def conv1d(cin, cout, k, bias=True, zero=False, **kw):
    if _SKIP:
        m = torch.nn.utils.skip_init(nn.Conv1d, cin, cout, k, bias=bias, **kw)
    else:
        m = nn.Conv1d(cin, cout, k, bias=bias, **kw)
        if zero:
            nn.init.zeros_(m.weight)
    return m


def conv_transpose1d(cin, cout, k, **kw):
    if _SKIP:
        return torch.nn.utils.skip_init(nn.ConvTranspose1d, cin, cout, k, **kw)
    return nn.ConvTranspose1d(cin, cout, k, **kw)


def params_version(module):
    """Changes whenever any parameter/buffer of ``module`` is modified in place or replaced."""
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))
