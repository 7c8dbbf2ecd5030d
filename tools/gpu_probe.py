"""Developer probe for a GPU box: micro-benchmarks of the hot kernels at SA-Open shapes (all variants in ONE
process, interleaved), one full-size DiT step and a full-size decode.  Not part of the product or the tests."""
import ctypes
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch

from stable_audio_tools import _hip

dev = torch.device("cuda:0")
if os.environ.get("SAT_HIP_EXP"):       # developer build with the experimental tiles / ablation modes (make -C csrc exp)
    _hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsat_hip_exp.so")
if os.environ.get("SAT_HIP_LIB"):       # another build of the library (tools/ab/*.so: same-box comparisons against an earlier round)
    _hip.LIB_PATH = os.path.join(ROOT, os.environ["SAT_HIP_LIB"])
    _hip.ABI_VERSION = int(os.environ.get("SAT_HIP_ABI", _hip.ABI_VERSION))
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
    return e0.elapsed_time(e1) / iters   # ms


def section(name, fn):
    print(f"==== {name}", flush=True)
    try:
        fn()
    except Exception:
        traceback.print_exc()
    sys.stdout.flush()


def gemm_bench():
    if os.environ.get("PROBE_DATA"):
        # how much of the MFMA "peak" is a function of operand data (power): zeros vs N(0,1) operands, production kernel (22)
        # and its MFMA-only ablation (522: fragments stay in registers, no LDS traffic, no barriers)
        m, n, k = 16400, 12288, 1536
        c = torch.zeros(m, n, device=dev)
        for label, mk in (("zeros", lambda *sh: torch.zeros(*sh, device=dev)), ("randn", lambda *sh: torch.randn(*sh, device=dev)),
                          ("randn*1e-3", lambda *sh: torch.randn(*sh, device=dev) * 1e-3)):
            a = mk(m, k).to(torch.bfloat16)
            w = mk(n, k).to(torch.bfloat16)
            for v in (22, 522, 622):
                f = lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), None, _hip.ptr(c), m, n, k, 0, v, _hip.stream()))
                ms = timeit(f)
                print(f"data {label:10s} v{v} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:8.1f} TFLOP/s", flush=True)
        return
    if os.environ.get("PROBE_QUANT"):
        # tile quantisation: how the time of one launch depends on the number of workgroups (256 CUs), and what the near-empty
        # tiles of the 2 leftover rows of M = 2050 cost
        shapes = [("256 tiles", 2048, 8192, 1536, 22), ("384 tiles", 2048, 12288, 1536, 22), ("432 tiles (2050)", 2050, 12288, 1536, 22),
                  ("512 tiles", 2048, 16384, 1536, 22), ("288 tiles (2050)", 2050, 8192, 1536, 22), ("128 tiles", 2048, 4096, 1536, 22),
                  ("v15 192 tiles", 2048, 1536, 6144, 15), ("v15 204 tiles (2050)", 2050, 1536, 6144, 15), ("v15 256 tiles", 2048, 2048, 6144, 15),
                  ("v15 272 tiles (2050)", 2050, 2048, 6144, 15), ("v15 192 tiles K1536", 2048, 1536, 1536, 15), ("v15 256 tiles K1536", 2048, 2048, 1536, 15),
                  ("v30 192 tiles", 2048, 4608, 1536, 30), ("v30 216 tiles (2050)", 2050, 4608, 1536, 30), ("v30 256 tiles", 2048, 6144, 1536, 30),
                  ("v16 192 tiles", 1024, 1536, 1536, 16), ("v16 216 tiles (1025)", 1025, 1536, 1536, 16), ("v16 256 tiles", 1024, 2048, 1536, 16)]
        for name, m, n, k, v in shapes:
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            c = torch.zeros(m, n, device=dev)
            f = lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), None, _hip.ptr(c), m, n, k, 1, v, _hip.stream()))
            ms = min(timeit(f) for _ in range(3))
            print(f"quant {name:22s} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:8.1f} TFLOP/s", flush=True)
        return
    shapes = [("ff_in(swiglu)", 2050, 12288, 1536), ("ff_out", 2050, 1536, 6144), ("qkv", 2050, 4608, 1536), ("proj", 2050, 1536, 1536),
              ("cross q/out", 1025, 1536, 1536), ("ff_in B8", 16400, 12288, 1536), ("ff_out B8", 16400, 1536, 6144)]
    for name, m, n, k in shapes:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        for v in (15, 16, 22, 26, 30):
            if v % 100 in (3, 4, 7, 8, 11, 13, 21, 22, 24, 25, 26) and n % 256:
                continue
            if v == 30 and n % 192:
                continue
            f = lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), None, _hip.ptr(c), m, n, k, 0, v, _hip.stream()))
            ms = timeit(f)
            print(f"gemm {name:14s} v{v} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:8.1f} TFLOP/s", flush=True)
        del a, w, c


def epi_ab():
    """A/B of the accumulator orientation (transposed: lane = token, 16-byte stores | legacy: lane = channel, bit 12 of the variant)
    on the shipped tiles at the shapes the plan launches, all three epilogues, interleaved rounds, median of per-round means."""
    import statistics
    LEG = 0x1000
    rounds = 5

    def ab(label, make, flops):
        fs = {"T": make(0), "L": make(LEG)}
        res = {k: [] for k in fs}
        for _ in range(rounds):
            for k, f in fs.items():
                res[k].append(timeit(f, iters=10, warm=2))
        t, l = statistics.median(res["T"]), statistics.median(res["L"])
        print(f"{label:44s} transposed {t*1e3:7.1f} us {flops/t/1e9:7.1f} TF | legacy {l*1e3:7.1f} us {flops/l/1e9:7.1f} TF | x{l/t:.3f}", flush=True)

    # fp32 output + residual accumulate (to_out, FF-out, cross out)
    for name, m, n, k, v in [("to_out B1 v15", 2050, 1536, 1536, 15), ("ff_out B1 v15", 2050, 1536, 6144, 15), ("cross B1 v16", 1025, 1536, 1536, 16),
                             ("to_out B8 v26", 16400, 1536, 1536, 26), ("ff_out B8 v26", 16400, 1536, 6144, 26), ("ff_out B1 v22", 2050, 1536, 6144, 22)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        bias = torch.randn(n, device=dev)
        ab(f"f32+resid {name} {m}x{n}x{k}", lambda flag: (lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(c), m, n, k, 1,
                                                                                                  v | flag, _hip.stream()))), 2.0 * m * n * k)
    # SwiGLU (FF-in)
    for name, m, v in [("ff_in B1 v22", 2050, 22), ("ff_in B1 v26", 2050, 26), ("ff_in B8 v26", 16400, 26), ("ff_in B8 v22", 16400, 22)]:
        n, k = 12288, 1536
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = torch.randn(n, k, device=dev) * 0.05
        bias = torch.randn(n, device=dev) * 0.1
        wp = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
        bp = torch.empty((n,), dtype=torch.float32, device=dev)
        out = torch.empty((m, n // 2), dtype=torch.bfloat16, device=dev)
        # pack once, then time the GEMM alone through the same entry point (the repack is two tiny launches: subtract by timing it)
        ab(f"swiglu {name} {m}x{n}x{k} (+pack)", lambda flag: (lambda: _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp),
                                                                                                    _hip.ptr(out), m, n, k, v | flag, _hip.stream()))), 2.0 * m * n * k)
    # heads + RoPE (to_qkv)
    for name, b, v in [("qkv B1 v30", 2, 30), ("qkv B1 v22", 2, 22), ("qkv B8 v26", 16, 26)]:
        s_len, s_pad, d = 1025, 1152, 1536
        a = torch.randn(b * s_len, d, device=dev).to(torch.bfloat16)
        w = (torch.randn(3 * d, d, device=dev) * 0.05).to(torch.bfloat16)
        inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
        q = torch.empty((b, 24, s_pad, 64), dtype=torch.bfloat16, device=dev)
        kk = torch.empty_like(q)
        vt = torch.empty((b, 24, 64, s_pad), dtype=torch.bfloat16, device=dev)
        scratch = torch.empty((2 * s_len * 16,), dtype=torch.float32, device=dev)
        ab(f"heads {name} {b*s_len}x{3*d}x{d} (+memsets)", lambda flag: (lambda: _hip.check(lib.sat_qkv_rope_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(kk),
                                                                                                          _hip.ptr(vt), _hip.ptr(scratch), b, s_len, s_pad, d, v | flag,
                                                                                                          _hip.stream()))), 2.0 * b * s_len * 3 * d * d)


def b8_tiles():
    """8 prompts per GPU (M = 16400): the 2-stage 256x256x64 tile (22) against the 4-stage 256x256x32 one with cross-tile fragment
    prefetch and grouped raster (26), every GEMM of the block."""
    import statistics
    m = 16400
    def ab(label, mk):
        res = {22: [], 26: []}
        for _ in range(4):
            for v in res:
                res[v].append(timeit(mk(v), iters=5, warm=2))
        print(f"B8 {label:10s} tile22 {statistics.median(res[22])*1e3:7.1f} us   tile26 {statistics.median(res[26])*1e3:7.1f} us", flush=True)
    for name, n, k in [("to_out", 1536, 1536), ("ff_out", 1536, 6144)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        ab(name, lambda v: (lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), None, _hip.ptr(c), m, n, k, 1, v, _hip.stream()))))
    n, k = 12288, 1536
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = torch.randn(n, k, device=dev) * 0.05
    bias = torch.randn(n, device=dev) * 0.1
    wp = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
    bp = torch.empty((n,), dtype=torch.float32, device=dev)
    out = torch.empty((m, n // 2), dtype=torch.bfloat16, device=dev)
    _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp), _hip.ptr(out), m, n, k, 22, _hip.stream()))
    ab("ff_in", lambda v: (lambda: _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp), _hip.ptr(out), m, n, k,
                                                                      v | 0x4000, _hip.stream()))))
    b, s_len, s_pad, d = 16, 1025, 1152, 1536
    a = torch.randn(b * s_len, d, device=dev).to(torch.bfloat16)
    w = (torch.randn(3 * d, d, device=dev) * 0.05).to(torch.bfloat16)
    inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
    q = torch.empty((b, 24, s_pad, 64), dtype=torch.bfloat16, device=dev)
    kk = torch.empty_like(q)
    vt = torch.empty((b, 24, 64, s_pad), dtype=torch.bfloat16, device=dev)
    scratch = torch.empty((2 * s_len * 16,), dtype=torch.float32, device=dev)
    ab("qkv", lambda v: (lambda: _hip.check(lib.sat_qkv_rope_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(kk), _hip.ptr(vt),
                                                                  _hip.ptr(scratch), b, s_len, s_pad, d, v, _hip.stream()))))


def f32_epi_ab():
    """fp32 residual epilogue: LDS-staged 16-byte coalesced (default) vs direct dword (bit 15) vs transposed 16-byte (bit 13)."""
    import statistics
    for name, m, n, k, v in [("to_out B1 v15", 2050, 1536, 1536, 15), ("ff_out B1 v15", 2050, 1536, 6144, 15), ("cross B1 v16", 1025, 1536, 1536, 16),
                             ("to_out B8 v26", 16400, 1536, 1536, 26), ("ff_out B8 v26", 16400, 1536, 6144, 26), ("ff_out B8 v22", 16400, 1536, 6144, 22)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        bias = torch.randn(n, device=dev)
        mk = lambda flag: (lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(c), m, n, k, 1, v | flag, _hip.stream())))
        fs = {"staged": mk(0), "direct": mk(0x8000), "transposed": mk(0x2000)}
        res = {kk: [] for kk in fs}
        for _ in range(5):
            for kk, f in fs.items():
                res[kk].append(timeit(f, iters=10, warm=2))
        print(f"f32+resid {name:16s} " + "  ".join(f"{kk}: {statistics.median(vv)*1e3:6.1f} us" for kk, vv in res.items()), flush=True)


def splitk_probe():
    """What a K-split of the narrow fp32-output GEMMs could buy: time of ONE part (fewer, bigger tiles over a fraction of K) against
    the shipped single-pass launch.  A split launch would take about max(part) + one epilogue hand-off."""
    import statistics
    cases = [("ff_out  full K=6144", 2050, 1536, 6144, 15), ("ff_out  256x128 K=3072", 2050, 1536, 3072, 12), ("ff_out  256x128 K=2816", 2050, 1536, 2816, 12),
             ("ff_out  256x192 K=2048", 2050, 1536, 2048, 30), ("ff_out  256x256 K=1536", 2050, 1536, 1536, 22),
             ("to_out  full K=1536", 2050, 1536, 1536, 15), ("to_out  256x128 K=768", 2050, 1536, 768, 12),
             ("cross   full K=1536 (128x64)", 1025, 1536, 1536, 16), ("cross   128x128 K=768", 1025, 1536, 768, 15)]
    fs = {}
    for name, m, n, k, v in cases:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        bias = torch.randn(n, device=dev)
        fs[name] = (lambda a=a, w=w, c=c, bias=bias, m=m, n=n, k=k, v=v: _hip.check(
            lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(c), m, n, k, 1, v, _hip.stream())), 2.0 * m * n * k)
    res = {kk: [] for kk in fs}
    for _ in range(5):
        for kk, (f, _fl) in fs.items():
            res[kk].append(timeit(f, iters=10, warm=2))
    for kk, (f, fl) in fs.items():
        ms = statistics.median(res[kk])
        print(f"splitk {kk:30s} {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)


def small_tiles():
    """The narrow fp32-output GEMMs (to_out, FF-out, cross to_q / to_out): wave tile 32x64 on 8 waves (shipped 15) against 64x64 on 4
    waves (10: 3 stages, 42: 4 stages, 43: 2 stages / two workgroups per CU) and BK = 128 (39)."""
    import statistics
    shapes = [("ff_out", 2050, 1536, 6144), ("to_out", 2050, 1536, 1536), ("cross", 1025, 1536, 1536)]
    for name, m, n, k in shapes:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        bias = torch.randn(n, device=dev)
        fs = {v: (lambda v=v: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(c), m, n, k, 1, v, _hip.stream())))
              for v in ([15, 16, 10, 42, 43, 39, 12] if not os.environ.get("PROBE_DEPTH") else [15, 44, 45, 16, 46, 47])}
        res = {v: [] for v in fs}
        for _ in range(5):
            for v, f in fs.items():
                res[v].append(timeit(f, iters=10, warm=2))
        print(f"tiles {name:8s} " + "  ".join(f"v{v}: {statistics.median(r)*1e3:6.1f} us" for v, r in res.items()), flush=True)


def hybrid_probe():
    """FF-in (SwiGLU) at 1 prompt is 384 full 256x256 tiles + 48 two-row tiles on 256 CUs = 1.5 rounds run as 2.  How long do the pieces
    of a two-launch split take: N = 8192 on 256x256 tiles (one full round) + N = 4096 on smaller tiles (a second full round of less work)?"""
    import statistics
    k = 1536
    cases = [("full  N=12288 v22", 2050, 12288, 22), ("A     N=8192  v22", 2050, 8192, 22), ("B     N=4096  v12 (256x128)", 2050, 4096, 12),
             ("B     N=4096  v15 (128x128)", 2050, 4096, 15), ("B     N=4096  v22", 2050, 4096, 22), ("B     N=4096  v10 (128x128, 4 waves)", 2050, 4096, 10),
             ("A'    N=6144  v22", 2050, 6144, 22), ("B'    N=6144  v12", 2050, 6144, 12), ("B'    N=6144  v30", 2050, 6144, 30),
             ("M2048 N=12288 v22", 2048, 12288, 22), ("M2048 N=8192 v22", 2048, 8192, 22)]
    fs = {}
    for name, m, n, v in cases:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = torch.randn(n, k, device=dev) * 0.05
        bias = torch.randn(n, device=dev)
        wp = torch.empty(n, k, dtype=torch.bfloat16, device=dev)
        bp = torch.empty(n, device=dev)
        h = torch.empty(m, n // 2, dtype=torch.bfloat16, device=dev)
        _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp), _hip.ptr(h), m, n, k, v, _hip.stream()))
        fs[name] = (lambda a=a, w=w, bias=bias, wp=wp, bp=bp, h=h, m=m, n=n, v=v: _hip.check(
            lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp), _hip.ptr(h), m, n, k, v | 0x4000, _hip.stream())),
            2.0 * m * n * k)
    res = {kk: [] for kk in fs}
    for _ in range(5):
        for kk, (f, _fl) in fs.items():
            res[kk].append(timeit(f, iters=10, warm=2))
    for kk, (f, fl) in fs.items():
        ms = statistics.median(res[kk])
        print(f"hybrid {kk:40s} {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)


def two_streams():
    """Would running the two CFG halves (conditional / unconditional sequences) as independent kernel streams fill the CUs that every
    kernel's last round leaves idle?  (a) the shipped CFG-batched step (2 sequences per launch) vs (b) two single-sequence forwards of two
    model instances on two HIP streams at once -- the conditional one with its context, the unconditional one with the null-context skip."""
    import copy
    import stable_audio_tools as S
    from stable_audio_tools import model_configs as MC, synthetic
    from stable_audio_tools.models import _init
    mods = []
    for _ in range(2):
        with _init.skip_init():
            model = S.create_model_from_config(MC.stable_audio_open_1_0())
        model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 0))
        mods.append(model.to(dev).eval().model.model)
    m0, m1 = mods
    b = int(os.environ.get("PROBE_B", "1"))
    c = torch.randn(b, 130, 768, device=dev)
    g = torch.randn(b, 1536, device=dev)
    x = torch.randn(b, 64, 1024, device=dev)
    out = torch.empty_like(x)
    m0.prepare_generation(c, g, 7.0)
    ms_a = min(timeit(lambda: m0.denoise(x, 3.0, cfg_scale=7.0, out=out), iters=10, warm=2) for _ in range(3))
    m0.prepare_generation(c, g, 1.0)
    m1.prepare_context(torch.zeros_like(c), g, 0)
    o0, o1 = torch.empty_like(x), torch.empty_like(x)
    ms_c = min(timeit(lambda: m0.denoise(x, 3.0, out=o0), iters=10, warm=2) for _ in range(3))
    ms_u = min(timeit(lambda: m1.denoise(x, 3.0, out=o1), iters=10, warm=2) for _ in range(3))
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        cur = torch.cuda.current_stream()
        s0.wait_stream(cur)
        s1.wait_stream(cur)
        with torch.cuda.stream(s0):
            m0.denoise(x, 3.0, out=o0)
        with torch.cuda.stream(s1):
            m1.denoise(x, 3.0, out=o1)
        cur.wait_stream(s0)
        cur.wait_stream(s1)
    ms_b = min(timeit(both, iters=10, warm=2) for _ in range(3))
    print(f"two streams B={b}: batched CFG step {ms_a:.3f} ms | conditional alone {ms_c:.3f} + unconditional alone {ms_u:.3f} = {ms_c + ms_u:.3f} ms | "
          f"both on two streams {ms_b:.3f} ms", flush=True)


def ablate():
    """Where does the time of each shipped GEMM go?  Ablation modes of the experiments build (SAT_HIP_EXP=1): 2 = no LDS-DMA in the
    loop, 4 = + no barrier, 5 = + no ds_read (MFMA on register fragments), 6 = + no epilogue, 7 = 5 with the epilogue arithmetic but
    no stores, 8 = production loop without stores.  Interleaved, median of 5 rounds."""
    import statistics
    NOPACK = 0x4000

    def run(label, fns, flops):
        res = {k: [] for k in fns}
        for _ in range(5):
            for k, f in fns.items():
                res[k].append(timeit(f, iters=10, warm=2))
        print(f"{label:34s} " + "  ".join(f"{k}:{statistics.median(v)*1e3:6.1f}" for k, v in res.items()), flush=True)

    for name, m, n, k, tile in [("ff_out B1", 2050, 1536, 6144, 15), ("to_out B1", 2050, 1536, 1536, 15), ("cross B1", 1025, 1536, 1536, 16),
                                ("ff_out B1 tile22", 2050, 1536, 6144, 22), ("ff_out B8 tile22", 16400, 1536, 6144, 22)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        bias = torch.randn(n, device=dev)
        mk = lambda v: (lambda: _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(c), m, n, k, 1, v, _hip.stream())))
        run(f"f32+resid {name} (us)", {"prod": mk(tile), "noload": mk(200 + tile), "nobar": mk(400 + tile), "nolds": mk(500 + tile), "noepi": mk(600 + tile)},
            2.0 * m * n * k)
    for name, m in [("ff_in B1", 2050), ("ff_in B8", 16400)]:
        n, k = 12288, 1536
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = torch.randn(n, k, device=dev) * 0.05
        bias = torch.randn(n, device=dev) * 0.1
        wp = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
        bp = torch.empty((n,), dtype=torch.float32, device=dev)
        out = torch.empty((m, n // 2), dtype=torch.bfloat16, device=dev)
        mk = lambda v: (lambda: _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp), _hip.ptr(out), m, n, k,
                                                                    v, _hip.stream())))
        mk(22)()        # pack once
        run(f"swiglu {name} tile22 (us)", {"prod": mk(22 | NOPACK), "legacy": mk(22 | NOPACK | 0x1000), "nostore": mk(822 | NOPACK), "noload": mk(222 | NOPACK),
                                           "nobar": mk(422 | NOPACK), "nolds": mk(522 | NOPACK), "nolds+math": mk(722 | NOPACK), "noepi": mk(622 | NOPACK),
                                           "tile26": mk(26 | NOPACK)}, 2.0 * m * n * k)
    for name, b in [("qkv B1", 2), ("qkv B8", 16)]:
        s_len, s_pad, d = 1025, 1152, 1536
        a = torch.randn(b * s_len, d, device=dev).to(torch.bfloat16)
        w = (torch.randn(3 * d, d, device=dev) * 0.05).to(torch.bfloat16)
        inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
        q = torch.empty((b, 24, s_pad, 64), dtype=torch.bfloat16, device=dev)
        kk = torch.empty_like(q)
        vt = torch.empty((b, 24, 64, s_pad), dtype=torch.bfloat16, device=dev)
        scratch = torch.empty((2 * s_len * 16,), dtype=torch.float32, device=dev)
        mk = lambda v: (lambda: _hip.check(lib.sat_qkv_rope_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(kk), _hip.ptr(vt),
                                                                 _hip.ptr(scratch), b, s_len, s_pad, d, v, _hip.stream())))
        run(f"heads {name} tile30 (+memsets, us)", {"prod": mk(30), "noload": mk(230), "nobar": mk(430), "nolds": mk(530), "noepi": mk(630), "tile22": mk(22),
                                                    "tile26": mk(26)}, 0)


def gemm_pmc():
    """few launches of selected variants for rocprofv3 --pmc runs"""
    for name, m, n, k, vs in [("ff_inB8", 16400, 12288, 1536, (7, 22, 26, 13)), ("ff_in", 2050, 12288, 1536, (22, 26))]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        for v in vs:
            for _ in range(3):
                _hip.check(lib.sat_gemm_bf16_f32(_hip.ptr(a), _hip.ptr(w), None, _hip.ptr(c), m, n, k, 0, v, _hip.stream()))
        torch.cuda.synchronize()


def attn_bench():
    for name, b, h, kvh, sq, sk in [("self B1", 2, 24, 24, 1025, 1025), ("cross B1", 2, 24, 12, 1025, 130), ("self B8", 16, 24, 24, 1025, 1025),
                                    ("self SA2", 2, 24, 24, 6145, 6145)]:
        sqp, skp = (sq + 127) // 128 * 128, (sk + 3 + 63) // 64 * 64
        q = (torch.randn(b, h, sqp, 64, device=dev) * 0.18).to(torch.bfloat16)
        k = torch.randn(b, kvh, skp, 64, device=dev).to(torch.bfloat16)
        vt = torch.randn(b, kvh, 64, skp, device=dev).to(torch.bfloat16)
        o = torch.empty(b * sq, h * 64, device=dev, dtype=torch.bfloat16)
        # the plan's entry: Q pre-scaled by its producer (single-KV-group shapes then run the reference-in-the-matrix-pipe kernel)
        f = lambda: _hip.check(lib.sat_attention_prescaled_bf16(_hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(o), b, h, kvh, sq, sk, sqp, skp, _hip.stream()))
        ms = timeit(f)
        print(f"attention {name:9s}: {ms*1e3:8.1f} us  {4.0*b*h*sq*sk*64/ms/1e9:8.1f} TFLOP/s", flush=True)


def attn_ablate():
    """experiments build: SAT_ATTN_DBG = 1 no softmax arithmetic, 2 no LDS-DMA in the loop, 3 no MFMA, 4 no loads + no barrier,
    5 = 4 + no K / V^T fragment reads"""
    import statistics
    for name, b, h, kvh, sq, sk in [("self B1", 2, 24, 24, 1025, 1025), ("self B8", 16, 24, 24, 1025, 1025), ("self SA2", 2, 24, 24, 6145, 6145)]:
        sqp, skp = (sq + 127) // 128 * 128, (sk + 3 + 63) // 64 * 64
        q = torch.randn(b, h, sqp, 64, device=dev).to(torch.bfloat16)
        k = torch.randn(b, kvh, skp, 64, device=dev).to(torch.bfloat16)
        vt = torch.randn(b, kvh, 64, skp, device=dev).to(torch.bfloat16)
        o = torch.empty(b * sq, h * 64, device=dev, dtype=torch.bfloat16)
        f = lambda: _hip.check(lib.sat_attention_bf16(_hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(o), b, h, kvh, sq, sk, sqp, skp, _hip.stream()))
        res = {}
        for rnd in range(3):
            for mode in ("0", "1", "2", "3", "4", "5"):
                if mode == "0":
                    os.environ.pop("SAT_ATTN_DBG", None)
                else:
                    os.environ["SAT_ATTN_DBG"] = mode
                res.setdefault(mode, []).append(timeit(f, iters=10, warm=2))
        os.environ.pop("SAT_ATTN_DBG", None)
        names = {"0": "prod", "1": "nosoftmax", "2": "noload", "3": "nomfma", "4": "noload+nobar", "5": "+nolds"}
        print(f"attention {name:9s} (us) " + "  ".join(f"{names[m]}: {statistics.median(v)*1e3:6.1f}" for m, v in res.items()), flush=True)


def ln_bench():
    m, d = 2050, 1536
    x = torch.randn(m, d, device=dev)
    g = torch.ones(d, device=dev)
    bb = torch.zeros(d, device=dev)
    y = torch.empty(m, d, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: _hip.check(lib.sat_layernorm_bf16(_hip.ptr(x), _hip.ptr(g), _hip.ptr(bb), _hip.ptr(y), m, d, _hip.stream())), iters=50)
    print(f"layernorm {m}x{d}: {ms*1e3:.1f} us  {(m*d*6)/ms/1e6:.1f} GB/s")


def full_model():
    import stable_audio_tools as S
    from stable_audio_tools import model_configs as MC, synthetic
    from stable_audio_tools.models import _init
    t0 = time.time()
    with _init.skip_init():
        model = S.create_model_from_config(MC.stable_audio_open_1_0())
    sd = synthetic.synth_state_dict(model.state_dict(), 0)
    model.load_state_dict(sd)
    del sd
    model = model.to(dev).eval()
    print(f"model built + synthetic weights in {time.time()-t0:.1f}s", flush=True)
    dit = model.model.model
    b = int(os.environ.get("PROBE_B", "1"))
    c = torch.randn(b, 130, 768, device=dev)
    g = torch.randn(b, 1536, device=dev)
    x = torch.randn(b, 64, 1024, device=dev)
    t0 = time.time()
    dit.prepare_generation(c, g, 7.0)
    torch.cuda.synchronize()
    print(f"plan finalize + context: {time.time()-t0:.2f}s", flush=True)
    out = torch.empty_like(x)
    ms = timeit(lambda: dit.denoise(x, 3.0, cfg_scale=7.0, out=out), iters=10, warm=2)
    print(f"DiT CFG step B={b}: {ms:.3f} ms  -> {4.544*b/ms:.1f} TFLOP/s effective; 100 steps = {ms/10:.2f} s", flush=True)
    if os.environ.get("PROBE_WIDE_AB"):          # A/B of the wide-tile policy inside the model: 80 (shipped) vs the listed settings, interleaved
        import statistics
        tiles = [80] + [int(t) for t in os.environ["PROBE_WIDE_AB"].split(",")]
        res = {t: [] for t in tiles}
        for _ in range(4):
            for t in tiles:
                dit.set_tile_policy(t)          # (per plan since ABI version 5: the plan is rebuilt and the context re-prepared)
                dit.prepare_generation(c, g, 7.0)
                res[t].append(timeit(lambda: dit.denoise(x, 3.0, cfg_scale=7.0, out=out), iters=6, warm=2))
        dit.set_tile_policy(80)
        dit.prepare_generation(c, g, 7.0)
        print("wide-tile A/B, DiT CFG step ms: " + " | ".join(f"{t}: {statistics.median(v):.3f} (min {min(v):.3f})" for t, v in res.items()), flush=True)
    print("denoise finite:", torch.isfinite(out).all().item(), "std", out.std().item())
    z = torch.randn(b, 64, 1024, device=dev)
    dec = model.pretransform.model
    ms = timeit(lambda: dec.decode(z[:1]), iters=3, warm=1)
    print(f"Oobleck decode 1x1024 frames: {ms:.2f} ms -> {5.16/ms*1e3:.1f} TFLOP/s", flush=True)
    a = dec.decode(z[:1])
    print("decode finite:", torch.isfinite(a).all().item(), "std", a.std().item(), "mem GB", torch.cuda.max_memory_allocated() / 1e9)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "ln", "full"]
    print(torch.cuda.get_device_name(0), flush=True)
    if "gemm" in which:
        section("gemm", gemm_bench)
    if "epi" in which:
        section("epilogue A/B", epi_ab)
    if "b8tiles" in which:
        section("8 prompts: tile 22 vs 26", b8_tiles)
    if "f32epi" in which:
        section("fp32 epilogue A/B", f32_epi_ab)
    if "twostreams" in which:
        section("two CFG halves on two streams", two_streams)
    if "hybrid" in which:
        section("FF-in two-launch split", hybrid_probe)
    if "smalltiles" in which:
        section("small tiles", small_tiles)
    if "splitk" in which:
        section("split-K parts", splitk_probe)
    if "ablate" in which:
        section("ablation", ablate)
    if "gemm_pmc" in which:
        section("gemm_pmc", gemm_pmc)
    if "attn" in which:
        section("attention", attn_bench)
    if "attn_ablate" in which:
        section("attention ablation", attn_ablate)
    if "ln" in which:
        section("layernorm", ln_bench)
    if "full" in which:
        section("full model", full_model)
