"""Host-side shape bookkeeping in front of the encoder: channel count, length, batch dimension.
Behavioural counterpart of the reference's ``inference/utils.py:7-39`` (``set_audio_channels`` / ``prepare_audio``)."""
from ..data.modification import PadCrop


def set_audio_channels(audio, target_channels):
    """[B, C, T] -> [B, target_channels, T]: down-mix by averaging, up-mix a mono signal by duplication, keep the first two
    channels of anything wider.  Other channel targets pass through untouched (as in the reference)."""
    have = audio.shape[1]
    if target_channels == 1:
        return audio.mean(1, keepdim=True)
    if target_channels == 2 and have == 1:
        return audio.repeat(1, 2, 1)
    if target_channels == 2 and have > 2:
        return audio[:, :2, :]
    return audio


def prepare_audio(audio, in_sr, target_sr, target_length, target_channels, device):
    """Move to ``device``, resample to ``target_sr``, pad / crop (from the start) to ``target_length``, lift to [B, C, T], fix the channel count."""
    if target_channels not in (1, 2):
        raise AssertionError("target_channels must be 1 or 2")
    audio = audio.to(device)
    if in_sr != target_sr:                      # torchaudio.transforms.Resample in the reference (utils.py:25-27): the HIP polyphase kernel
        from .resample import resample
        audio = resample(audio, in_sr, target_sr)
    fitted = PadCrop(target_length, randomize=False)(audio)
    while fitted.dim() < 3:                     # [T] -> [1, 1, T]; [C, T] -> [1, C, T]
        fitted = fitted.unsqueeze(0)
    return set_audio_channels(fitted, target_channels)
