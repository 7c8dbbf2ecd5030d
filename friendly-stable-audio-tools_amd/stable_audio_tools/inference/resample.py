"""Sample-rate conversion in front of the encoder: the HIP counterpart of ``torchaudio.transforms.Resample(orig, new)`` with its
defaults (``lowpass_filter_width=6, rolloff=0.99, resampling_method="sinc_interp_hann"``), which the reference applies at
``inference/utils.py:25-27``, ``models/autoencoders.py:394-397`` and ``reconstruct_audios.py:34-35``.

torchaudio is not part of this image, so the algorithm is restated from its published definition (``torchaudio.functional``:
``_get_sinc_resample_kernel`` / ``_apply_sinc_resample_kernel``): the two rates are divided by their gcd; a bank of ``new`` FIR
filters of ``2 * width + orig`` taps (``width = ceil(lowpass_filter_width * orig / (min(orig, new) * rolloff))``) -- Hann-windowed
sinc at cutoff ``min(orig, new) * rolloff``, computed in float64 and rounded to float32 -- is applied as a convolution with stride
``orig`` to the input padded by ``(width, width + orig)``; the output is cropped to ``ceil(new * length / orig)`` samples.
The bank is built here (a few thousand values, once per rate pair); the convolution runs in ``sat_resample_sinc``.
"""
import math

import torch

from .. import _hip

_BANKS = {}


def sinc_resample_bank(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (bank [new, 2 * width + orig] float32 (CPU), width, orig, new) with orig / new reduced by their gcd."""
    if orig_freq <= 0 or new_freq <= 0 or int(orig_freq) != orig_freq or int(new_freq) != new_freq:
        raise ValueError("sample rates must be positive integers")
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, :] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    bank = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * scale
    return bank.to(torch.float32).contiguous(), width, orig, new


def resample(audio: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """[..., T] float32 on a HIP device -> [..., ceil(T * new / orig)]; the identity when the rates agree (as torchaudio)."""
    if int(orig_freq) == int(new_freq):
        return audio
    if not audio.is_cuda:
        # the same error type as every other op of this package; the reference would resample on the CPU here (T.Resample follows the
        # audio's device), this build deliberately has no CPU arithmetic anywhere: move the audio to the model's device first
        raise _hip.SatError("resample: audio is on the CPU; this package computes on a HIP device only (no CPU path) -- "
                            "call audio.to(model_device) before prepare_audio / preprocess_audio_list_for_encoder")
    key = (int(orig_freq), int(new_freq), audio.device)
    if key not in _BANKS:
        bank, width, orig, new = sinc_resample_bank(orig_freq, new_freq)
        _BANKS[key] = (bank.to(audio.device), width, orig, new)
    bank, width, orig, new = _BANKS[key]
    shape = audio.shape
    x = audio.reshape(-1, shape[-1]).float().contiguous()
    out_len = -(-new * shape[-1] // orig)                       # ceil
    y = torch.empty((x.shape[0], out_len), dtype=torch.float32, device=audio.device)
    _hip.check(_hip.lib().sat_resample_sinc(_hip.ptr(x), _hip.ptr(bank), _hip.ptr(y), x.shape[0], shape[-1], out_len, orig, new, width,
                                            _hip.stream()))
    return y.reshape(shape[:-1] + (out_len,))
