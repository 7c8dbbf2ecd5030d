"""``VAEBottleneck`` (reference ``models/bottleneck.py:10-65``).  ``encode`` SAMPLES, as the
reference does (``randn_like``, :48); the draw comes from torch's generator on the tensor's
device (plumbing), the arithmetic ``noise * (softplus(scale) + 1e-4) + mean`` is the HIP kernel
``sat_vae_sample``.  ``noise=`` lets tests inject the Gaussian."""
import torch
from torch import nn

from .. import _hip


class Bottleneck(nn.Module):
    def __init__(self, is_discrete: bool):
        super().__init__()
        self.is_discrete = is_discrete

    def encode(self, x, return_info: bool = False, **kwargs):
        raise NotImplementedError

    def decode(self, x):
        raise NotImplementedError


class VAEBottleneck(Bottleneck):
    def __init__(self):
        super().__init__(is_discrete=False)

    @torch.no_grad()
    def encode(self, x, return_info=False, noise=None, **kwargs):
        x = x.contiguous().float()
        b, c2, t = x.shape
        c = c2 // 2
        if noise is None:
            noise = torch.randn((b, c, t), device=x.device, dtype=torch.float32)
        z = torch.empty((b, c, t), device=x.device, dtype=torch.float32)
        _hip.check(_hip.lib().sat_vae_sample(_hip.ptr(x), _hip.ptr(noise.contiguous().float()), _hip.ptr(z), b, c, t, _hip.stream()))
        # the KL term (bottleneck.py:49-51) is a training loss and is not computed on this path
        return (z, {}) if return_info else z

    def decode(self, x):
        return x
