"""``PadCrop`` / ``Mono`` / ``Stereo`` (reference ``data/modification.py``); host-side tensor
reshaping used by ``prepare_audio`` and ``reconstruct_audios.py``."""
import torch
from torch import nn


class PadCrop(nn.Module):
    # reference data/modification.py:11-23
    def __init__(self, n_samples, randomize=True):
        super().__init__()
        self.n_samples = n_samples
        self.randomize = randomize

    def __call__(self, signal):
        n, s = signal.shape
        start = 0 if (not self.randomize) else torch.randint(0, max(0, s - self.n_samples) + 1, []).item()
        end = start + self.n_samples
        output = signal.new_zeros([n, self.n_samples])
        output[:, :min(s, self.n_samples)] = signal[:, start:end]
        return output


class Mono(nn.Module):
    def __call__(self, signal):
        return torch.mean(signal, dim=0, keepdims=True) if len(signal.shape) > 1 else signal


class Stereo(nn.Module):
    def __call__(self, signal):
        shape = signal.shape
        if len(shape) == 1:      # s -> 2, s
            signal = signal.unsqueeze(0).repeat(2, 1)
        elif len(shape) == 2:
            if shape[0] == 1:    # 1, s -> 2, s
                signal = signal.repeat(2, 1)
            elif shape[0] > 2:   # ?, s -> 2, s
                signal = signal[:2, :]
        return signal
