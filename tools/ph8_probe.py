"""Developer probe (GPU box): calibration of the 8-wave / 8-phase 256x256 GEMM tile (variant 80, csrc/gemm_ph8.hip) against the
16-wave tile 22 and hipBLASLt (torch.matmul) on the same random operands, all arms interleaved in ONE process; correctness
against an fp32 product of the same bf16 operands; repeated-run race screen.  Not part of the product or the tests.
usage: python tools/ph8_probe.py [calib] [shapes] [race] [ablate]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch

from stable_audio_tools import _hip

dev = torch.device("cuda:0")
if os.environ.get("SAT_HIP_EXP"):
    _hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsat_hip_exp.so")
lib = _hip.lib()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_fn(a, w, c, m, n, k, v, accumulate=0, bias=None):
    return lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias) if bias is not None else None, _hip.ptr(c), m, n, k,
                                                     accumulate, v, _hip.stream()))


def check(m, n, k, v, fill="randn"):
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    if fill == "rows":          # transpose / permutation detecting: every row and column distinct
        a = (torch.arange(m, device=dev).float()[:, None] * 0.001 + torch.randn(m, k, device=dev) * 0.01).to(torch.bfloat16)
    c = torch.full((m, n), float("nan"), device=dev)
    gemm_fn(a, w, c, m, n, k, v)()
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t()
    err = ((c - ref).norm() / ref.norm()).item()
    bad = (~torch.isfinite(c)).sum().item()
    mx = (c - ref).abs().max().item()
    print(f"check v{v} M={m} N={n} K={k} fill={fill}: rel-L2 {err:.3e} max-abs {mx:.3e} non-finite {bad}", flush=True)
    return err, bad


def arms_bench(label, m, n, k, variants, rounds=5, iters=10, blas=True, scale_w=0.05, uniform=False):
    if uniform:
        a = (torch.rand(m, k, device=dev) * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(n, k, device=dev) * 2 - 1).to(torch.bfloat16)
    else:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * scale_w).to(torch.bfloat16)
    c = torch.zeros(m, n, device=dev)
    fs = {}
    for v in variants:
        fs[f"v{v & 0xffff}" + (f"o{v >> 16}" if v >> 16 else "")] = gemm_fn(a, w, c, m, n, k, v)
    if blas:
        wt = w.t()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        fs["hipblaslt(bf16 out)"] = lambda: torch.matmul(a, wt, out=out)
    res = {kk: [] for kk in fs}
    for _ in range(rounds):
        for kk, f in fs.items():
            try:
                res[kk].append(timeit(f, iters=iters, warm=2))
            except Exception as e:       # an arm that is not built for this shape
                res[kk].append(float("nan"))
    fl = 2.0 * m * n * k
    print(f"{label} M={m} N={n} K={k}: " + " | ".join(f"{kk} {statistics.median(vv)*1e3:8.1f} us {fl/statistics.median(vv)/1e9:7.1f} TF (min {min(vv)*1e3:.1f})"
                                                        for kk, vv in res.items()), flush=True)


def calib():
    for (m, n, k) in [(256, 256, 256), (512, 512, 512), (300, 512, 384), (2050, 1536, 1536)]:
        check(m, n, k, 80)
        check(m, n, k, 80, fill="rows")
    check(4096, 4096, 4096, 80)
    arms_bench("calib uniform[-1,1)", 4096, 4096, 4096, [22, 80], uniform=True)
    arms_bench("calib randn x 0.05 randn", 4096, 4096, 4096, [22, 80])
    arms_bench("calib uniform[-1,1)", 8192, 8192, 8192, [22, 80], rounds=3, iters=4, uniform=True)


def shapes():
    for name, m, n, k in [("ff_in B1", 2050, 12288, 1536), ("qkv B1", 2050, 4608, 1536), ("ff_out B1", 2050, 1536, 6144), ("ff_in B8", 16400, 12288, 1536),
                          ("qkv B8", 16400, 4608, 1536), ("ff_out B8", 16400, 1536, 6144), ("to_out B8", 16400, 1536, 1536), ("ff_in sa2", 12290, 12288, 1536)]:
        arms_bench(name, m, n, k, [22, 80], blas=False, rounds=4)


def race():
    # the same launch 40 times: every output must be bit-identical to the first (a DMA / ds_read race shows up as rare differing tiles)
    for (m, n, k) in [(256, 256, 256), (512, 512, 512), (4096, 4096, 4096), (2050, 12288, 1536)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c0 = torch.empty(m, n, device=dev)
        gemm_fn(a, w, c0, m, n, k, 80)()
        ref = a.float() @ w.float().t()
        e0 = ((c0 - ref).norm() / ref.norm()).item()
        diff = 0
        for i in range(40):
            c = torch.empty(m, n, device=dev)
            gemm_fn(a, w, c, m, n, k, 80)()
            diff += int(not torch.equal(c, c0))
        print(f"race M={m} N={n} K={k}: rel-L2 {e0:.2e}, {diff} of 40 repeats differ", flush=True)


def opts():
    vs = [80] + [80 | (o << 16) for o in (1, 2, 3, 4, 5, 8, 9)]
    for o in (1, 4, 5, 9):
        check(512, 512, 512, 80 | (o << 16))
    for name, m, n, k in [("4096^3", 4096, 4096, 4096), ("ff_out B8", 16400, 1536, 6144), ("qkv B8", 16400, 4608, 1536)]:
        arms_bench("opts " + name, m, n, k, vs, blas=False, rounds=5)


def ablate():
    for name, m, n, k in [("4096^3", 4096, 4096, 4096), ("ff_in B8", 16400, 12288, 1536)]:
        arms_bench("ablate " + name, m, n, k, [80, 180, 280, 380], blas=False, rounds=3)


if __name__ == "__main__":
    what = sys.argv[1:] or ["calib", "race", "shapes"]
    print(torch.cuda.get_device_name(0), flush=True)
    for wname in what:
        print(f"==== {wname}", flush=True)
        globals()[wname]()
