"""TEST INFRASTRUCTURE (oracle): CPU restatement of ``torchaudio.transforms.Resample`` with its defaults, the sample-rate conversion
the reference applies in front of the encoder (inference/utils.py:25-27, models/autoencoders.py:394-397, reconstruct_audios.py:34-35).

PARITY UNPINNED: torchaudio (a third-party dependency of the reference, pinned by its setup.py as ``torchaudio>=2.0.2``) is absent
from /root/reference and from this image, and the reference holds no test or fixture for this step; the algorithm below restates the
published definition (torchaudio/functional/functional.py: ``_get_sinc_resample_kernel``, ``_apply_sinc_resample_kernel``,
``resample``) in an independent form -- an explicit per-output-sample dot product in float64 -- so that it does not share code with
the product's filter-bank builder or its convolution.
"""
import math

import numpy as np


def resample(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """x [..., T] (numpy, any float) -> [..., ceil(T * new / orig)] float64, Hann-windowed sinc interpolation."""
    x = np.asarray(x, dtype=np.float64)
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if orig == new:
        return x
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    length = x.shape[-1]
    out_len = -(-new * length // orig)
    flat = x.reshape(-1, length)
    padded = np.concatenate([np.zeros((flat.shape[0], width)), flat, np.zeros((flat.shape[0], width + orig))], axis=1)
    out = np.zeros((flat.shape[0], out_len))
    taps = np.arange(-width, width + orig, dtype=np.float64)
    for phase in range(new):
        # filter of output phase p: h[k] = sinc(t) * hann(t) * base / orig with t = ((k - width) / orig - p / new) * base, clamped to the
        # window's support; rounded to float32 like the filter bank torchaudio convolves with
        t = np.clip((taps / orig - phase / new) * base, -lowpass_filter_width, lowpass_filter_width)
        win = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        tp = t * math.pi
        h = np.where(tp == 0, 1.0, np.sin(tp) / np.where(tp == 0, 1.0, tp)) * win * (base / orig)
        h = h.astype(np.float32).astype(np.float64)
        js = np.arange(phase, out_len, new)
        frames = js // new
        for col, f in zip(js, frames):
            out[:, col] = padded[:, f * orig:f * orig + taps.size] @ h
    return out.reshape(x.shape[:-1] + (out_len,))
