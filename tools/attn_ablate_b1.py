"""One-prompt self-attention (S = 1025, 2 x 24 heads, two-key-range layout, MODE 1): where does its time go?  Ablation modes of the experiments
build (SAT_ATTN_DBG, wrong results): 1 no exp / max / sum, 2 no LDS-DMA in the loop, 3 no MFMA, 4 = 2 + no barrier, 5 = 4 + fragments read once.
The one launch of a block that is NOT at the power cap (profiles/r05_power_per_kernel.txt).  Developer tool.   python tools/attn_ablate_b1.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch
from stable_audio_tools import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsat_hip_exp.so")
lib = _hip.lib(); dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(torch.cuda.get_device_name(0))
for dt, fn_name in ((torch.float16, "sat_attention_prescaled_f16"), (torch.bfloat16, "sat_attention_prescaled_bf16")):
    for name, b, h, s in (("one prompt", 2, 24, 1025), ("eight prompts", 16, 24, 1025)):
        sp = 1152
        torch.manual_seed(3)
        q = (torch.randn(b, h, sp, 64, device=dev) * 0.18).to(dt)          # pre-scaled by log2(e) / 8
        k = torch.randn(b, h, sp, 64, device=dev).to(dt)
        vt = torch.randn(b, h, 64, sp, device=dev).to(dt)
        o = torch.empty(b * s, h * 64, device=dev, dtype=dt)
        f = lambda: _hip.check(getattr(lib, fn_name)(_hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(o), b, h, h, s, s, sp, sp, _hip.stream()))
        res = {}
        for mode in (0, 1, 2, 3, 4, 5):
            if mode: os.environ["SAT_ATTN_DBG"] = str(mode)
            else: os.environ.pop("SAT_ATTN_DBG", None)
            res[mode] = statistics.median([timeit(f) for _ in range(3)])
        os.environ.pop("SAT_ATTN_DBG", None)
        os.environ["SAT_ATTN_GROUPS"] = "1"          # the single-range layout (256 queries per workgroup, MODE 2) forced: 240 workgroups at one prompt
        one_range = statistics.median([timeit(f) for _ in range(3)])
        os.environ["SAT_ATTN_GROUPS"] = "2"
        two_range = statistics.median([timeit(f) for _ in range(3)])
        os.environ.pop("SAT_ATTN_GROUPS", None)
        print(f"{str(dt):15s} {name:14s}: forced single-range layout {one_range:6.1f} us, forced two-range layout {two_range:6.1f} us", flush=True)
        print(f"{str(dt):15s} {name:14s}: shipped {res[0]:6.1f} us | no softmax arithmetic {res[1]:6.1f} | no LDS-DMA {res[2]:6.1f} | no MFMA {res[3]:6.1f} | "
              f"no DMA + no barrier {res[4]:6.1f} | + fragments read once {res[5]:6.1f}", flush=True)
