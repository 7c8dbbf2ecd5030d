"""Minimal PCM WAV reader/writer on the stdlib ``wave`` module (torchaudio is not available in the target image;
the reference scripts use ``torchaudio.load/save``: ``generate.py:151``, ``reconstruct_audios.py:28-36,146-147``)."""
import wave

import numpy as np
import torch


def save_wav_int16(path, audio_int16: torch.Tensor, sample_rate: int):
    """audio_int16 [channels, n] int16 on CPU."""
    a = audio_int16.detach().cpu().to(torch.int16).contiguous()
    if a.dim() == 1:
        a = a.unsqueeze(0)
    data = a.t().contiguous().numpy().astype("<i2").tobytes()      # interleave channels
    with wave.open(str(path), "wb") as w:
        w.setnchannels(a.shape[0])
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(data)


def save_wav_float(path, audio: torch.Tensor, sample_rate: int):
    """float audio in [-1, 1] -> 16-bit PCM (clipped), like torchaudio.save's default for float input to .wav int16."""
    a = (audio.detach().cpu().float().clamp(-1.0, 1.0) * 32767.0).round().to(torch.int16)
    save_wav_int16(path, a, sample_rate)


def load_wav(path):
    """-> (float32 tensor [channels, n] in [-1, 1), sample_rate).  8/16/24/32-bit integer PCM."""
    with wave.open(str(path), "rb") as w:
        ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    return torch.from_numpy(x.reshape(-1, ch).T.copy()), sr
