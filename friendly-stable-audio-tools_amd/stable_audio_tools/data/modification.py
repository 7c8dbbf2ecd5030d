"""Channel / length normalisation of raw audio tensors ``[channels, samples]`` on the host.

Counterparts of ``PadCrop``, ``Mono`` and ``Stereo`` in the reference's ``data/modification.py`` -- same call behaviour (pinned
by tests/golden/host.npz through ``prepare_audio``), used by ``inference.utils.prepare_audio`` and ``reconstruct_audios.py``.
"""
import torch
from torch import nn
from torch.nn import functional as F


class PadCrop(nn.Module):
    """Exactly ``n_samples`` samples: a window of the signal (random start if ``randomize``, else the beginning), zero-padded
    on the right when the signal is shorter."""

    def __init__(self, n_samples: int, randomize: bool = True):
        super().__init__()
        self.n_samples, self.randomize = n_samples, randomize

    def forward(self, signal: torch.Tensor) -> torch.Tensor:
        length = signal.shape[-1]
        surplus = max(length - self.n_samples, 0)
        first = int(torch.randint(0, surplus + 1, [])) if self.randomize else 0
        window = signal[:, first:first + self.n_samples]
        return F.pad(window, (0, self.n_samples - window.shape[-1]))


class Mono(nn.Module):
    """Average the channels of ``[c, s]`` into ``[1, s]``; a bare ``[s]`` signal passes through."""

    def forward(self, signal: torch.Tensor) -> torch.Tensor:
        return signal if signal.dim() < 2 else signal.mean(dim=0, keepdim=True)


class Stereo(nn.Module):
    """Two channels: duplicate a mono signal (``[s]`` or ``[1, s]``), keep the first two channels of anything wider."""

    def forward(self, signal: torch.Tensor) -> torch.Tensor:
        if signal.dim() == 1:
            signal = signal[None]
        if signal.dim() == 2 and signal.shape[0] != 2:
            signal = signal.expand(2, -1).clone() if signal.shape[0] == 1 else signal[:2]
        return signal
