"""Which self-attention layout for which number of sequences?  Single key range (256 queries per workgroup, all KV tiles, MODE 2: the softmax
reference in the matrix pipe) against two key ranges (128 queries x 2 key halves merged through LDS, MODE 1), S = 1025, 24 heads, forced through
SAT_ATTN_GROUPS in the experiments build (re-read per launch).  Round 5: the rule dates from before MODE 2 existed.   python tools/attn_layout_sweep.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch
from stable_audio_tools import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsat_hip_exp.so")
lib = _hip.lib(); dev = torch.device("cuda:0")


def timeit(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(torch.cuda.get_device_name(0))
for s, sp in ((1025, 1152), (6145, 6272)):
    for b in ((1, 2, 3, 4, 6, 8, 12, 16) if s == 1025 else (1, 2, 4)):
        h = 24
        torch.manual_seed(3)
        q = (torch.randn(b, h, sp, 64, device=dev) * 0.18).to(torch.float16)
        k = torch.randn(b, h, sp, 64, device=dev).to(torch.float16)
        vt = torch.randn(b, h, 64, sp, device=dev).to(torch.float16)
        o = torch.empty(b * s, h * 64, device=dev, dtype=torch.float16)
        f = lambda: _hip.check(lib.sat_attention_prescaled_f16(_hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(o), b, h, h, s, s, sp, sp, _hip.stream()))
        res = {}
        for rep in range(3):
            for grp in ("1", "2"):
                os.environ["SAT_ATTN_GROUPS"] = grp
                res.setdefault(grp, []).append(timeit(f))
        os.environ.pop("SAT_ATTN_GROUPS", None)
        one, two = statistics.median(res["1"]), statistics.median(res["2"])
        wg1 = -(-s // 256) * h * b
        print(f"S={s} sequences={b:2d} (single-range workgroups {wg1:4d}): single range {one:7.1f} us   two ranges {two:7.1f} us   -> {'single' if one < two else 'two'} ({100 * (two / one - 1):+.1f} %)", flush=True)
        del q, k, vt, o
