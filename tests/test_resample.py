"""Sample-rate conversion: the HIP polyphase kernel + its host-built filter bank against the oracle's independent
float64 restatement of torchaudio.transforms.Resample (oracle/resample.py; PARITY UNPINNED -- torchaudio is absent, the published
algorithm is restated twice, in different forms), plus the properties a resampler must have."""
import math

import numpy as np
import pytest
import torch


def test_filter_bank_matches_oracle_definition():
    """CPU: the product's vectorised float64 filter bank == the oracle's per-phase formula, for up- and down-sampling"""
    from stable_audio_tools.inference.resample import sinc_resample_bank
    for orig_sr, new_sr in [(48000, 44100), (22050, 44100), (44100, 16000), (32000, 44100)]:
        bank, width, orig, new = sinc_resample_bank(orig_sr, new_sr)
        g = math.gcd(orig_sr, new_sr)
        assert (orig, new) == (orig_sr // g, new_sr // g)
        base = min(orig, new) * 0.99
        assert width == math.ceil(6 * orig / base) and bank.shape == (new, 2 * width + orig)
        taps = np.arange(-width, width + orig, dtype=np.float64)
        for phase in (0, 1, new // 2, new - 1):
            t = np.clip((taps / orig - phase / new) * base, -6, 6)
            tp = t * math.pi
            h = np.where(tp == 0, 1.0, np.sin(tp) / np.where(tp == 0, 1.0, tp)) * np.cos(t * math.pi / 12) ** 2 * (base / orig)
            assert np.array_equal(bank[phase].numpy(), h.astype(np.float32))
        # unity DC gain: every phase's taps sum to ~1 (the low-pass passes a constant)
        assert np.allclose(bank.double().sum(1).numpy(), 1.0, atol=2e-3)


def test_oracle_resample_properties():
    """CPU: the oracle itself -- length rule, identity, a sine keeps frequency and amplitude"""
    from oracle import resample as ores
    x = np.random.default_rng(0).standard_normal((2, 1000))
    assert ores.resample(x, 44100, 44100) is not None and np.array_equal(ores.resample(x, 44100, 44100), x)
    assert ores.resample(x, 48000, 44100).shape == (2, math.ceil(1000 * 147 / 160))
    n, f0 = 4800, 1000.0
    s = np.sin(2 * math.pi * f0 * np.arange(n) / 48000)
    y = ores.resample(s, 48000, 44100)
    want = np.sin(2 * math.pi * f0 * np.arange(y.shape[-1]) / 44100)
    assert np.abs(y[200:-200] - want[200:-200]).max() < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("orig_sr,new_sr,length", [(48000, 44100, 30011), (22050, 44100, 8000), (44100, 16000, 20000), (32000, 44100, 1), (48000, 44100, 159)])
def test_resample_hip_vs_oracle(dev, orig_sr, new_sr, length):
    from oracle import resample as ores
    from stable_audio_tools.inference.resample import resample
    g = torch.Generator().manual_seed(orig_sr + length)
    x = torch.randn(3, 2, length, generator=g)
    got = resample(x.to(dev), orig_sr, new_sr)
    want = torch.from_numpy(ores.resample(x.numpy(), orig_sr, new_sr)).float()
    assert got.shape == want.shape == (3, 2, -(-length * (new_sr // math.gcd(orig_sr, new_sr)) // (orig_sr // math.gcd(orig_sr, new_sr))))
    err = ((got.cpu() - want).norm() / want.norm().clamp_min(1e-12)).item()
    assert err <= 1e-5, f"resample {orig_sr}->{new_sr}: rel-L2 {err:.2e} vs the float64 oracle (fp32 accumulation: tolerance 1e-5)"


@pytest.mark.gpu
def test_prepare_audio_resamples(dev):
    """inference/utils.py:21-39 of the reference: resample -> pad / crop -> batch dim -> channels"""
    from oracle import resample as ores
    from stable_audio_tools.inference.utils import prepare_audio
    x = torch.randn(1, 9600, generator=torch.Generator().manual_seed(3))
    out = prepare_audio(x, in_sr=48000, target_sr=44100, target_length=10000, target_channels=2, device=dev)
    assert out.shape == (1, 2, 10000)
    want = torch.from_numpy(ores.resample(x.numpy(), 48000, 44100)).float()          # 8820 samples, then zero padding, then mono -> stereo
    assert torch.allclose(out[0, 0, :8820].cpu(), want[0], atol=2e-5) and torch.equal(out[0, 0], out[0, 1])
    assert (out[0, :, 8820:] == 0).all()
