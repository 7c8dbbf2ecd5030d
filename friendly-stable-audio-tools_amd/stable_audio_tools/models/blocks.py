"""``FourierFeatures`` and ``SnakeBeta`` parameter containers (reference ``models/blocks.py:88-97,
318-358``).  FourierFeatures is evaluated inside ``sat_dit_forward``; SnakeBeta is fused into
the conv epilogues of the Oobleck kernels.  ``SnakeBeta.forward`` exposes the standalone HIP
kernel (``sat_snake_beta``) for parity tests."""
import torch
from torch import nn

from .. import _hip


class FourierFeatures(nn.Module):
    def __init__(self, in_features, out_features, std=1.0):
        super().__init__()
        assert out_features % 2 == 0
        self.weight = nn.Parameter(torch.randn([out_features // 2, in_features]) * std)


class SnakeBeta(nn.Module):
    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=True):
        super().__init__()
        if not alpha_logscale:
            raise NotImplementedError("SnakeBeta: only alpha_logscale=True (the Oobleck setting) is supported")
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        self.alpha = nn.Parameter(torch.zeros(in_features) * alpha)
        self.beta = nn.Parameter(torch.zeros(in_features) * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.beta.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001

    @torch.no_grad()
    def forward(self, x):
        x = x.contiguous().float()
        y = torch.empty_like(x)
        b, c, t = x.shape
        _hip.check(_hip.lib().sat_snake_beta(_hip.ptr(x), _hip.ptr(self.alpha.data), _hip.ptr(self.beta.data), _hip.ptr(y),
                                             b, c, t, _hip.stream()))
        return y
