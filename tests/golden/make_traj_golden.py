"""Full-size multi-step parity fixture: tests/golden/traj_full.npz.

12 steps of DPM-Solver++(3M) SDE (sigma 500 -> 0.3, polyexponential) with batched CFG 7 on the FULL-size SA-Open DiT (24 layers,
D = 1536, T = 1024, synthetic weights seed 0), initial and per-step noise injected, evaluated by the CPU oracle (oracle/dit.py +
oracle/sampler.py) four times: fp32, with the bf16 matched-rounding hook of the default plan (LnFoldRounding), with the e4m3 /
MXFP8 hooks of BASELINE config 5 (Fp8Rounding: "fp8" = cross to_q + FF-in + FF-out, "fp8all" = every block GEMM) and with the fp16 hook of gemm_dtype "fp16" (LnFoldRoundingF16).  Stored: the latents after steps 4, 8 and 12 of each run.  The oracle takes ~20 s
per CFG evaluation on 8 cores, which is why this is a committed fixture and not recomputed in the -m gpu test.

    python tests/golden/make_traj_golden.py            (any machine with the repo; no GPU, no reference needed)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402
from oracle import dit as odit, sampler as osamp  # noqa: E402
from stable_audio_tools import synthetic  # noqa: E402
from stable_audio_tools.models import _init  # noqa: E402
from stable_audio_tools.models.dit import DiffusionTransformer  # noqa: E402

STEPS, SIGMA_MIN, SIGMA_MAX, CFG = cases.TRAJ["steps"], cases.TRAJ["sigma_min"], cases.TRAJ["sigma_max"], cases.TRAJ["cfg_scale"]
SNAP = cases.TRAJ["snapshots"]
inputs = cases.traj_inputs


@torch.no_grad()
def main():
    torch.set_num_threads(os.cpu_count())
    with _init.skip_init():
        dit = DiffusionTransformer(**cases.FULL_DIT)
    sd = synthetic.synth_state_dict(dit.state_dict(), 0)
    del dit
    c, g, noise, step_noise = inputs()
    sig = osamp.get_sigmas_polyexponential(STEPS, SIGMA_MIN, SIGMA_MAX, 1.0)
    out = {}
    hooks = (("fp32", None), ("bf16", odit.LnFoldRounding()), ("fp8", odit.Fp8Rounding()), ("fp8all", odit.Fp8Rounding(odit.FP8_FAMILIES)),
             ("fp16", odit.LnFoldRoundingF16()))
    path = os.path.join(cases.GOLDEN_DIR, "traj_full.npz")
    only = sys.argv[1:]                 # e.g. `make_traj_golden.py fp16`: add these runs to the existing file, leave the others as they are
    if only:
        out.update({k: v for k, v in np.load(path).items()})
        hooks = tuple(h for h in hooks if h[0] in only)
    for tag, rnd in hooks:
        t0 = time.time()
        snaps = {}

        def cb(info, snaps=snaps):
            if info["i"] in SNAP:          # x at the start of step i = the latents after i steps
                snaps[info["i"]] = info["x"].clone()

        fn = lambda xin, tt, rnd=rnd: odit.dit_forward(sd, xin, tt, c, g, 24, 24, cfg_scale=CFG, rnd=rnd)
        x = osamp.sample_dpmpp_3m_sde(lambda x_, s_: osamp.vdenoise(fn, x_, s_), noise * sig[0], sig, lambda i, s, sn: step_noise[i], callback=cb)
        snaps[STEPS] = x
        for i in SNAP:
            out[f"{tag}_step{i}"] = snaps[i].numpy().astype(np.float32)
        print(f"{tag}: {time.time() - t0:.0f} s, final std {x.std():.4f}", flush=True)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    for i in SNAP:
        a = torch.from_numpy(out[f"fp32_step{i}"])
        print(f"step {i}: " + "   ".join(f"{t}-matched vs fp32 {((torch.from_numpy(out[f'{t}_step{i}']) - a).norm() / a.norm()).item():.3e}"
                                          for t in ("bf16", "fp16", "fp8", "fp8all") if f"{t}_step{i}" in out))


if __name__ == "__main__":
    main()
