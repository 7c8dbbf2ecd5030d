"""Developer experiment (CPU, oracle only): the error budget of BASELINE config 5.  The full-size 12-step CFG-7
trajectory of tests/golden/traj_full.npz is re-run with e4m3 operands in ONE subset of the block's GEMM families at a time
(oracle.dit.Fp8Rounding(families=...)) and compared with the fp32 trajectory of the fixture after 4 / 8 / 12 steps.
usage: python tools/fp8_budget.py qkv cq ff1 ff2 o qkv+ff1 qkv+cq+ff1 ...      (one run per argument, ~7 min each on 8 cores)
Round 5: a family name with the suffix "@mx" quantises its LayerNorm-fed ACTIVATION with MX block scales (one power-of-two
scale per 32 channels, oracle.dit.mxfp8_blocks) instead of one scale per token, "@mxw" the weights as well: `qkv@mx`, `qkv@mxw`, `qkv@mx+cq+ff1+ff2`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "friendly-stable-audio-tools_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import cases  # noqa: E402
from oracle import dit as odit, sampler as osamp  # noqa: E402
from stable_audio_tools import synthetic  # noqa: E402
from stable_audio_tools.models import _init  # noqa: E402
from stable_audio_tools.models.dit import DiffusionTransformer  # noqa: E402


@torch.no_grad()
def main():
    torch.set_num_threads(os.cpu_count())
    tj = cases.TRAJ
    with _init.skip_init():
        dit = DiffusionTransformer(**cases.FULL_DIT)
    sd = synthetic.synth_state_dict(dit.state_dict(), 0)
    del dit
    c, g, noise, step_noise = cases.traj_inputs()
    gold = cases.load("traj_full")
    sig = osamp.get_sigmas_polyexponential(tj["steps"], tj["sigma_min"], tj["sigma_max"], 1.0)
    for arg in sys.argv[1:]:
        spec = arg.split("+")
        fams = tuple(f.split("@")[0] for f in spec)
        mx_act = {f.split("@")[0] for f in spec if "@mx" in f}
        mx_w = {f.split("@")[0] for f in spec if f.endswith("@mxw")}

        class Rnd(odit.Fp8Rounding):
            def act(self, x, fam="qkv"):
                return odit.mxfp8_blocks(x) if fam in mx_act else super().act(x, fam)

            def weight(self, w, fam="qkv"):
                return odit.mxfp8_blocks(w) if fam in mx_w else super().weight(w, fam)

        rnd = Rnd(families=fams)
        snaps = {}

        def cb(info, snaps=snaps):
            if info["i"] in tj["snapshots"]:
                snaps[info["i"]] = info["x"].clone()

        t0 = time.time()
        fn = lambda xin, tt: odit.dit_forward(sd, xin, tt, c, g, 24, 24, cfg_scale=tj["cfg_scale"], rnd=rnd)
        x = osamp.sample_dpmpp_3m_sde(lambda x_, s_: osamp.vdenoise(fn, x_, s_), noise * sig[0], sig, lambda i, s, sn: step_noise[i], callback=cb)
        snaps[tj["steps"]] = x
        errs = [((snaps[i] - gold[f"fp32_step{i}"]).norm() / gold[f"fp32_step{i}"].norm()).item() for i in tj["snapshots"]]
        print(f"fp8 families {arg:16s}: rel-L2 vs fp32 after 4 / 8 / 12 steps: " + " / ".join(f"{e:.2e}" for e in errs) + f"   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
