"""``float_to_int16_audio`` / ``is_silence`` (reference ``utils/audio_utils.py``)."""
import torch

from .. import _hip

EPS = 1e-8


def is_silence(audio: torch.Tensor, thresh: float = -60.0):
    # reference utils/audio_utils.py:6-18 (host-side check, one reduction)
    return 20 * torch.log10(torch.flatten(audio.abs()).max() + EPS).item() < thresh


def float_to_int16_audio(x: torch.Tensor, maximize: bool = False):
    """peak-normalise (peak floored at 1 unless ``maximize``), scale by 32767, truncate to
    int16, return on CPU (reference utils/audio_utils.py:21-26).  HIP: one abs-max
    reduction + one quantisation pass (``sat_float_to_int16``), no host sync for the peak."""
    x = x.contiguous().float()
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    scratch = torch.empty(1, dtype=torch.int32, device=x.device)
    _hip.check(_hip.lib().sat_float_to_int16(_hip.ptr(x), _hip.ptr(out), x.numel(), int(bool(maximize)), _hip.ptr(scratch),
                                             _hip.stream()))
    return out.cpu()
