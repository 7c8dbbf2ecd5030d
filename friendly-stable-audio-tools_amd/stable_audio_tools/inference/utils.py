"""``set_audio_channels`` / ``prepare_audio`` (reference ``inference/utils.py:7-39``): host-side
shape bookkeeping before the encoder (pad/crop, mono/stereo, batch dim)."""
from ..data.modification import PadCrop


def set_audio_channels(audio, target_channels):
    if target_channels == 1:
        audio = audio.mean(1, keepdim=True)
    elif target_channels == 2:
        if audio.shape[1] == 1:
            audio = audio.repeat(1, 2, 1)
        elif audio.shape[1] > 2:
            audio = audio[:, :2, :]
    return audio


def prepare_audio(audio, in_sr, target_sr, target_length, target_channels, device):
    assert target_channels in [1, 2]
    audio = audio.to(device)
    if in_sr != target_sr:
        raise NotImplementedError("resampling needs torchaudio, which this image does not provide; pass audio at the model sample rate")
    audio = PadCrop(target_length, randomize=False)(audio)
    if audio.dim() == 1:
        audio = audio.unsqueeze(0).unsqueeze(0)
    elif audio.dim() == 2:
        audio = audio.unsqueeze(0)
    return set_audio_channels(audio, target_channels)
