"""Developer probe: full-size Oobleck decode (1024 latent frames) and encode (the same audio back) alone, for rocprofv3 / A-B runs
of the codec kernels (SAT_HIP_EXP=1 SAT_OOBLECK_UNFUSED=1: experiments build, two launches per ResidualUnit).  Not part of the product or the tests."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch

import stable_audio_tools as S
from stable_audio_tools import model_configs as MC, synthetic
from stable_audio_tools.models import _init

from stable_audio_tools import _hip

if os.environ.get("SAT_HIP_EXP"):       # experiments build: honours SAT_OOBLECK_UNFUSED=1
    _hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsat_hip_exp.so")
dev = torch.device("cuda:0")
with _init.skip_init():
    vae = S.create_model_from_config(MC.stable_audio_vae())
vae.load_state_dict(synthetic.synth_state_dict(vae.state_dict(), 3))
vae = vae.to(dev).eval()
z = torch.randn(1, 64, int(os.environ.get("FRAMES", "1024")), device=dev)


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters * 1e3, out


ms, audio = timeit(lambda: vae.decode(z), 5)
print(f"decode {z.shape[-1]} frames: {ms:.2f} ms", flush=True)
ms, lat = timeit(lambda: vae.encode(audio), 5)
print(f"encode {audio.shape[-1]} samples: {ms:.2f} ms", flush=True)
