"""Full-size parity fixture at the HEADLINE's own length: tests/golden/traj100_full.npz.

The metric is a 100-step generation (reference generate.py:27-32, inference/generation.py:95-261).  This script runs exactly that
trajectory on the CPU oracle in fp32: 100 steps of DPM-Solver++(3M) SDE (sigma 500 -> 0.3, polyexponential, rho 1) with batched CFG 7
on the FULL-size SA-Open DiT (24 layers, D = 1536, T = 1024, synthetic weights seed 0), initial and per-step noise injected, followed
by the full-size Oobleck decode of the final latents (synthetic weights seed 0).  Stored: the latents after 12 / 25 / 50 / 100 steps and
two 65 536-sample windows (start, middle) of the decoded stereo audio.  One CFG evaluation takes 20-45 s on 8 cores, the whole job
about an hour, which is why this is a committed fixture; partial results are checkpointed next to the output after every snapshot.

    python tests/golden/make_traj100_golden.py [threads] [variant]     (any machine with the repo; no GPU, no reference needed)

`variant` 1, 2, ... = another (prompt, seed) pair (the <= 1e-3 claim on more than one trajectory): other conditioning
tensors, other initial and per-step noise (cases.traj100_inputs(variant)); written to traj100_full_v{variant}.npz with the latents after
50 / 100 steps only (no decode).  NOTE: these fixtures are outputs of the repo's CPU ORACLE (pinned against the reference by the
short-run goldens of make_golden.py), not of the reference itself.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402
from oracle import dit as odit, oobleck as oob, sampler as osamp  # noqa: E402
from stable_audio_tools import synthetic  # noqa: E402
from stable_audio_tools.models import _init  # noqa: E402
from stable_audio_tools.models.autoencoders import OobleckDecoder  # noqa: E402
from stable_audio_tools.models.dit import DiffusionTransformer  # noqa: E402

T100 = cases.TRAJ100


@torch.no_grad()
def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count())
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    snapshots = T100["snapshots"] if variant == 0 else cases.TRAJ100_VARIANT_SNAPSHOTS
    with _init.skip_init():
        dit = DiffusionTransformer(**cases.FULL_DIT)
    sd = synthetic.synth_state_dict(dit.state_dict(), 0)
    del dit
    c, g, noise, step_noise = cases.traj100_inputs(variant)
    sig = osamp.get_sigmas_polyexponential(T100["steps"], T100["sigma_min"], T100["sigma_max"], 1.0)
    path = os.path.join(cases.GOLDEN_DIR, "traj100_full.npz" if variant == 0 else f"traj100_full_v{variant}.npz")
    out = {}
    t0 = time.time()

    def cb(info):
        i = info["i"]
        if i in snapshots:          # x at the start of step i = the latents after i steps
            out[f"fp32_step{i}"] = info["x"].numpy().astype(np.float32).copy()
            np.savez_compressed(path + ".partial.npz", **out)
        print(f"step {i:3d} sigma {float(info['sigma']):9.4f}  |x| {float(info['x'].std()):.4f}  {time.time() - t0:6.0f} s", flush=True)

    fn = lambda xin, tt: odit.dit_forward(sd, xin, tt, c, g, 24, 24, cfg_scale=T100["cfg_scale"], rnd=None)
    x = osamp.sample_dpmpp_3m_sde(lambda x_, s_: osamp.vdenoise(fn, x_, s_), noise * sig[0], sig, lambda i, s, sn: step_noise[i], callback=cb)
    out[f"fp32_step{T100['steps']}"] = x.numpy().astype(np.float32)
    np.savez_compressed(path + ".partial.npz", **out)
    print(f"trajectory: {time.time() - t0:.0f} s, final std {x.std():.4f}", flush=True)
    del sd
    if variant != 0:
        np.savez_compressed(path, **out)
        os.remove(path + ".partial.npz")
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")
        return
    with _init.skip_init():
        dec = OobleckDecoder(**cases.vae_kwargs(cases.FULL_VAE, True))
    vsd = synthetic.synth_state_dict(dec.state_dict(), 0)
    del dec
    audio = oob.oobleck_decoder(vsd, x)
    print(f"decode: audio {tuple(audio.shape)} std {audio.std():.4f}  {time.time() - t0:.0f} s", flush=True)
    for name, s0 in T100["audio_windows"].items():
        out[f"audio_{name}"] = audio[:, :, s0:s0 + T100["audio_window_len"]].numpy().astype(np.float32)
    np.savez_compressed(path, **out)
    os.remove(path + ".partial.npz")
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
