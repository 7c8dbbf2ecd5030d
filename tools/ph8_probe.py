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


_WS = {}


def gemm_fn(a, w, c, m, n, k, v, accumulate=0, bias=None):
    if v & 0x10000 or v == 0:          # K-split of the remainder round: slabs in caller-supplied scratch
        import ctypes
        need = ctypes.c_size_t()
        _hip.check(lib.sat_gemm_f32_workspace_bytes(m, n, k, v, ctypes.byref(need)))
        if need.value:
            ws = _WS.setdefault(need.value, torch.empty(need.value, dtype=torch.uint8, device=dev))
            return lambda: _hip.check(lib.sat_gemm_bf16_f32_ws(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias) if bias is not None else None, _hip.ptr(c), m, n, k,
                                                                accumulate, v, _hip.ptr(ws), need.value, _hip.stream()))
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
        fs[f"v{v & 0xffff}" + (f"+{v >> 16:x}" if v >> 16 else "")] = gemm_fn(a, w, c, m, n, k, v)
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


def libcal():
    """the vendor library (hipBLASLt through torch.matmul, bf16 output -- its cheapest epilogue) against this build's
    plain fp32-output GEMM (launcher's choice, the 8-phase tile, and the narrow tiles) on the PRODUCT's shapes, interleaved in one process"""
    shapes = []
    for mname, m in (("B1", 2050), ("B8", 16400), ("sa2", 12290), ("cross B1", 1025)):
        for nname, n, k in (("ff_in", 12288, 1536), ("qkv", 4608, 1536), ("to_out", 1536, 1536), ("ff_out", 1536, 6144)):
            if mname == "cross B1" and nname != "to_out":
                continue
            shapes.append((f"{nname} {mname}", m, n, k))
    for name, m, n, k in shapes:
        vs = [0, 80 | 0x20000]
        if m <= 2050:
            vs += [15, 16, 30] if n % 192 == 0 else [15, 16]
            if k >= 4096:
                vs += [44, 80 | 0x10000]
        arms_bench("libcal " + name, m, n, k, vs, rounds=4, iters=10)
    arms_bench("libcal 4096^3", 4096, 4096, 4096, [80], rounds=3, iters=6)
    arms_bench("libcal 8192^3", 8192, 8192, 8192, [80], rounds=3, iters=3)


def quant6():
    """Round 6, pricing the M-tail BEFORE building anything: what would every block GEMM cost if M were a whole number of
    tiles -- M = 2048 / 16384 (the tokens without the 2 / 16 rows that spill into a ninth / 65th row of tiles) against the model's M = 2050 /
    16400 -- with the shipped tile choice (v0), the 8-phase tile (80), the 12-wave 256 x 192 (30), the 256 x 128 8-wave tile (12, experiments
    build: 768 tiles = 3.0 rounds at M = 16384) and the vendor library, all interleaved; plain fp32 output (the epilogue is the same on both M)."""
    for name, n, k, vs1, vs8 in (("ff_in", 12288, 1536, [0], [0]), ("qkv", 4608, 1536, [0, 80 | 0x20000], [0]),
                                 ("to_out", 1536, 1536, [0, 15], [0, 22, 80 | 0x20000, 12]), ("ff_out", 1536, 6144, [0, 44], [0, 22, 12])):
        for m in (2050, 2048):
            arms_bench(f"quant6 {name} B1", m, n, k, vs1, rounds=4, iters=10)
        for m in (16400, 16384):
            arms_bench(f"quant6 {name} B8", m, n, k, vs8, rounds=3, iters=6)
    for m in (1025, 1024):
        arms_bench("quant6 cross B1", m, 1536, 1536, [0, 15], rounds=4, iters=10)
    for m in (8200, 8192):
        arms_bench("quant6 cross B8", m, 1536, 1536, [0, 22, 12], rounds=3, iters=6)


def balance(bit=0x200000):
    """schedule A/B on the real epilogues: every CU a workgroup (256 + remainder) vs balanced rounds on fewer workgroups (variant bit 21)"""
    def ab(label, mk, flops, arms, rounds=5):
        fs = {k_: mk(v) for k_, v in arms.items()}
        res = {k_: [] for k_ in fs}
        for _ in range(rounds):
            for k_, f in fs.items():
                res[k_].append(timeit(f, iters=10, warm=2))
        print(f"balance {label:42s} " + " | ".join(f"{k_} {statistics.median(t)*1e3:7.1f} us {flops/statistics.median(t)/1e9:7.1f} TF" for k_, t in res.items()), flush=True)

    for name, m in [("B1", 2050), ("B4", 8200), ("B8", 16400), ("sa2", 12290)]:
        n, k = 12288, 1536
        xb = torch.randn(m, k, device=dev).to(torch.bfloat16)
        part = torch.stack([xb.float().view(m, k // 64, 64).sum(-1), (xb.float() ** 2).view(m, k // 64, 64).sum(-1)], -1).contiguous()
        w = torch.randn(n, k, device=dev) * 0.05
        gamma = 0.8 + 0.2 * torch.rand(k, device=dev)
        beta = 0.1 * torch.randn(k, device=dev)
        bias = torch.randn(n, device=dev) * 0.1
        wp = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
        c12 = torch.empty((2 * n,), dtype=torch.float32, device=dev)
        out = torch.empty((m, n // 2), dtype=torch.bfloat16, device=dev)
        _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias), _hip.ptr(wp),
                                               _hip.ptr(c12), _hip.ptr(out), m, n, k, 80, _hip.stream()))
        ab(f"ff_in swiglu+ln {name} {m}x{n}x{k}",
           lambda v: (lambda: _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias),
                                                                       _hip.ptr(wp), _hip.ptr(c12), _hip.ptr(out), m, n, k, v | 0x4000, _hip.stream()))),
           2.0 * m * n * k, {"all CUs": 80, "balanced": 80 | bit})
        for nm, kk in [("ff_out", 6144), ("to_out", 1536)]:
            nn = 1536
            a = torch.randn(m, kk, device=dev).to(torch.bfloat16)
            w2 = (torch.randn(nn, kk, device=dev) * 0.05).to(torch.bfloat16)
            b2 = torch.randn(nn, device=dev)
            c = torch.zeros(m, nn, device=dev)
            xo = torch.empty((m, nn), dtype=torch.bfloat16, device=dev)
            po = torch.empty((m, nn // 64, 2), dtype=torch.float32, device=dev)
            ab(f"{nm} resid+ln {name} {m}x{nn}x{kk}",
               lambda v: (lambda: _hip.check(lib.sat_gemm_resid_ln_bf16(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po), m, nn, kk, v,
                                                                          _hip.stream()))), 2.0 * m * nn * kk,
               {"launcher": 0, "v22": 22, "v80 all CUs": 80 | 0x20000, "v80 balanced": 80 | 0x20000 | bit})
    for name, b in [("B1", 2), ("B8", 16), ("sa2", 2)]:
        s_len = 6145 if name == "sa2" else 1025
        s_pad, d = (s_len + 3 + 127) // 128 * 128, 1536
        a = torch.randn(b * s_len, d, device=dev).to(torch.bfloat16)
        w = (torch.randn(3 * d, d, device=dev) * 0.05).to(torch.bfloat16)
        inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
        q = torch.empty((b, 24, s_pad, 64), dtype=torch.bfloat16, device=dev)
        kk = torch.empty_like(q)
        vt = torch.empty((b, 24, 64, s_pad), dtype=torch.bfloat16, device=dev)
        scratch = torch.empty((2 * s_len * 16,), dtype=torch.float32, device=dev)
        ab(f"qkv heads+rope {name} {b*s_len}x{3*d}x{d} (+memsets)",
           lambda v: (lambda: _hip.check(lib.sat_qkv_rope_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(kk), _hip.ptr(vt), _hip.ptr(scratch),
                                                               b, s_len, s_pad, d, v, _hip.stream()))), 2.0 * b * s_len * 3 * d * d,
           {"launcher": 0, "v30": 30, "v80 all CUs": 80, "v80 balanced": 80 | bit})


def kg():
    """the two-K-group 128 x 128 tile (variant 49) against the shipped 8-wave tiles on the one-prompt fp32-output GEMMs, product epilogue"""
    f16 = bool(os.environ.get("PROBE_F16"))
    odt = torch.float16 if f16 else torch.bfloat16
    fn = lib.sat_gemm_resid_ln_f16 if f16 else lib.sat_gemm_resid_ln_bf16
    print("operands:", odt)
    for name, m, n, k, arms in [("ff_out B1", 2050, 1536, 6144, (44, 49)), ("to_out B1", 2050, 1536, 1536, (15, 49)), ("cross out B1", 1025, 1536, 1536, (16, 15))]:
        nset = int(os.environ.get("PROBE_SETS", "1"))          # > 1: rotate through operand sets (cold weights, as in the model: 24 layers x 57 MB)
        sets = []
        for i in range(nset):
            sets.append((torch.randn(m, k, device=dev).to(odt), (torch.randn(n, k, device=dev) * 0.05).to(odt)))
        b2 = torch.randn(n, device=dev)
        c = torch.zeros(m, n, device=dev)
        xo = torch.empty((m, n), dtype=odt, device=dev)
        po = torch.empty((m, n // 64, 2), dtype=torch.float32, device=dev)
        cnt = [0]

        def call(v):
            a, w2 = sets[cnt[0] % nset]
            cnt[0] += 1
            _hip.check(fn(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po), m, n, k, v, _hip.stream()))
        if 50 in arms:
            outs = {}
            for v in (49, 50):
                c.copy_(torch.arange(m * n, device=dev, dtype=torch.float32).view(m, n) * 1e-6)
                cnt[0] = 0
                call(v)
                torch.cuda.synchronize()
                outs[v] = (c.clone(), xo.float().clone(), po.clone())
            print("   v50 vs v49:", ["%.2e" % ((x50 - x49).abs().max().item()) for x49, x50 in zip(outs[49], outs[50])], flush=True)
        res = {v: [] for v in arms}
        for _ in range(7):
            for v in arms:
                res[v].append(timeit(lambda: call(v), iters=10, warm=2))
        fl = 2.0 * m * n * k
        print(f"kg {name} {m}x{n}x{k}: " + "  ".join(f"v{v} {statistics.median(t)*1e3:.1f} us ({fl/statistics.median(t)/1e9:.0f} TF, min {min(t)*1e3:.1f})" for v, t in res.items()), flush=True)


def narrow():
    """the narrow one-prompt GEMMs (fp32 residual + LayerNorm-fold producer epilogue): shipped tiles against the experiments build's 4-wave / BK = 128 tiles"""
    for name, m, n, k, arms in [("ff_out B1", 2050, 1536, 6144, (44, 15, 49, 39, 42, 43, 10, 48)), ("to_out B1", 2050, 1536, 1536, (15, 49, 16, 39, 42, 43, 10, 48)),
                                ("cross out B1", 1025, 1536, 1536, (16, 15, 49, 42, 48)),
                                # two workgroups of tile 43 (4 waves of 64 x 64, 64 KiB) share a CU once there are >= 512 tiles: 8 waves per CU with half
                                # the LDS reads per MFMA of tile 15 / 44 -- does the fragment traffic bound the narrow tiles?
                                ("ff_out 4 prompts", 8200, 1536, 6144, (44, 49, 43, 80)), ("to_out 4 prompts", 8200, 1536, 1536, (15, 49, 43, 22, 80)),
                                ("ff_out 8 prompts", 16400, 1536, 6144, (44, 49, 43, 80)), ("to_out 8 prompts", 16400, 1536, 1536, (15, 49, 43, 22, 80)),
                                ("cross out 8 prompts", 8200, 1536, 1536, (15, 49, 43, 22, 80))]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w2 = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        b2 = torch.randn(n, device=dev)
        c = torch.zeros(m, n, device=dev)
        xo = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        po = torch.empty((m, n // 64, 2), dtype=torch.float32, device=dev)
        if 49 in arms and m < 4000:          # the K-group tile against the shipped one: same sums up to the order of the two k-halves
            outs = {}
            for v in (15, 49):
                c.copy_(torch.arange(m * n, device=dev, dtype=torch.float32).view(m, n) * 1e-6)
                xo.fill_(float("nan")); po.fill_(float("nan"))
                _hip.check(lib.sat_gemm_resid_ln_bf16(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po), m, n, k, v, _hip.stream()))
                torch.cuda.synchronize()
                outs[v] = (c.clone(), xo.float().clone(), po.clone())
            errs = [((x49 - x15).norm() / x15.norm()).item() for x15, x49 in zip(outs[15], outs[49])]
            print(f"   v49 vs v15 ({name}): rel-L2 C {errs[0]:.2e}  image {errs[1]:.2e}  partial sums {errs[2]:.2e}  non-finite {sum((~torch.isfinite(x)).sum().item() for x in outs[49])}", flush=True)
            c.zero_()
        res = {v: [] for v in arms}
        for _ in range(5):
            for v in arms:
                try:
                    res[v].append(timeit(lambda: _hip.check(lib.sat_gemm_resid_ln_bf16(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po), m, n, k,
                                                                                      v, _hip.stream())), iters=10, warm=2))
                except Exception as e:
                    res[v].append(float("nan"))
        fl = 2.0 * m * n * k
        print(f"narrow resid+ln {name} {m}x{n}x{k}: " + "  ".join(f"v{v} {statistics.median(t)*1e3:.1f} us ({fl/statistics.median(t)/1e9:.0f} TF)" for v, t in res.items()), flush=True)


def balance_any():
    """experiments build: balanced rounds at ANY round count (variant bit 22, the 'balanced' arm) against the shipped policy (only 1 < rounds <= 2)"""
    balance(bit=0x400000)


def race():
    # The same launch 40 times, alternating between two different A operands: every output must be bit-identical to the first run on
    # that operand and equal to its fp32 reference (a DMA / ds_read race shows up as rare differing tiles; a stale stream-K slab or
    # ticket as the OTHER operand's partial sums)
    for (m, n, k) in [(256, 256, 256), (512, 512, 512), (512, 512, 6144), (257, 768, 6144), (4096, 4096, 4096), (2050, 12288, 1536), (2050, 1536, 6144),
                      (2050, 4608, 1536), (16400, 1536, 1536), (16400, 4608, 1536)]:
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        As = [torch.randn(m, k, device=dev).to(torch.bfloat16), (torch.randn(m, k, device=dev) * 2 + 0.5).to(torch.bfloat16)]
        refs = [a.float() @ w.float().t() for a in As]
        first = [None, None]
        diff, worst = 0, 0.0
        for i in range(40):
            j = i & 1
            c = torch.full((m, n), float("nan"), device=dev)
            gemm_fn(As[j], w, c, m, n, k, 80)()
            if first[j] is None:
                first[j] = c
                worst = max(worst, ((c - refs[j]).norm() / refs[j].norm()).item())
            else:
                diff += int(not torch.equal(c, first[j]))
        print(f"race M={m} N={n} K={k}: worst rel-L2 {worst:.2e}, {diff} of 38 repeats differ", flush=True)


def opts():
    vs = [80] + [80 | (o << 16) for o in (1, 2, 3, 4, 5, 8, 9)]
    for o in (1, 4, 5, 9):
        check(512, 512, 512, 80 | (o << 16))
    for name, m, n, k in [("4096^3", 4096, 4096, 4096), ("ff_out B8", 16400, 1536, 6144), ("qkv B8", 16400, 4608, 1536)]:
        arms_bench("opts " + name, m, n, k, vs, blas=False, rounds=5)


def epi():
    """the real epilogues at the shapes the plan launches: 16-wave tile 22 vs the 8-phase tile 80, interleaved"""
    def ab(label, mk, flops, rounds=5):
        fs = {"v22": mk(22), "v80": mk(80 | 0x20000), "v80sk": mk(80 | 0x10000)}
        res = {k: [] for k in fs}
        for _ in range(rounds):
            for k, f in fs.items():
                res[k].append(timeit(f, iters=10, warm=2))
        a, b, c = (statistics.median(res[k]) for k in ("v22", "v80", "v80sk"))
        print(f"epi {label:40s} v22 {a*1e3:7.1f} us {flops/a/1e9:7.1f} TF | v80 {b*1e3:7.1f} us {flops/b/1e9:7.1f} TF | v80+streamK {c*1e3:7.1f} us {flops/c/1e9:7.1f} TF", flush=True)

    for name, m in [("B1", 2050), ("B8", 16400), ("sa2", 12290)]:
        # SwiGLU with the LayerNorm fold (FF-in as the plan runs it)
        n, k = 12288, 1536
        xb = torch.randn(m, k, device=dev).to(torch.bfloat16)
        part = torch.stack([xb.float().view(m, k // 64, 64).sum(-1), (xb.float() ** 2).view(m, k // 64, 64).sum(-1)], -1).contiguous()
        w = torch.randn(n, k, device=dev) * 0.05
        gamma = 0.8 + 0.2 * torch.rand(k, device=dev)
        beta = 0.1 * torch.randn(k, device=dev)
        bias = torch.randn(n, device=dev) * 0.1
        wp = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
        c12 = torch.empty((2 * n,), dtype=torch.float32, device=dev)
        out = torch.empty((m, n // 2), dtype=torch.bfloat16, device=dev)
        _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias), _hip.ptr(wp),
                                               _hip.ptr(c12), _hip.ptr(out), m, n, k, 22, _hip.stream()))
        ab(f"ff_in swiglu+ln {name} {m}x{n}x{k}",
           lambda v: (lambda: _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias),
                                                                       _hip.ptr(wp), _hip.ptr(c12), _hip.ptr(out), m, n, k, v | 0x4000, _hip.stream()))),
           2.0 * m * n * k)
        # fp32 residual update + LayerNorm-fold producer (FF-out, to_out)
        for nm, kk in [("ff_out", 6144), ("to_out", 1536)]:
            nn = 1536
            a = torch.randn(m, kk, device=dev).to(torch.bfloat16)
            w2 = (torch.randn(nn, kk, device=dev) * 0.05).to(torch.bfloat16)
            b2 = torch.randn(nn, device=dev)
            c = torch.zeros(m, nn, device=dev)
            xo = torch.empty((m, nn), dtype=torch.bfloat16, device=dev)
            po = torch.empty((m, nn // 64, 2), dtype=torch.float32, device=dev)
            ab(f"{nm} resid+ln {name} {m}x{nn}x{kk}",
               lambda v: (lambda: _hip.check(lib.sat_gemm_resid_ln_bf16(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po), m, nn, kk, v,
                                                                          _hip.stream()))), 2.0 * m * nn * kk)
    for name, b in [("B1", 2), ("B8", 16)]:
        s_len, s_pad, d = 1025, 1152, 1536
        a = torch.randn(b * s_len, d, device=dev).to(torch.bfloat16)
        w = (torch.randn(3 * d, d, device=dev) * 0.05).to(torch.bfloat16)
        inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
        q = torch.empty((b, 24, s_pad, 64), dtype=torch.bfloat16, device=dev)
        kk = torch.empty_like(q)
        vt = torch.empty((b, 24, 64, s_pad), dtype=torch.bfloat16, device=dev)
        scratch = torch.empty((2 * s_len * 16,), dtype=torch.float32, device=dev)
        ab(f"qkv heads+rope {name} {b*s_len}x{3*d}x{d} (+memsets)",
           lambda v: (lambda: _hip.check(lib.sat_qkv_rope_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(kk), _hip.ptr(vt), _hip.ptr(scratch),
                                                               b, s_len, s_pad, d, v, _hip.stream()))), 2.0 * b * s_len * 3 * d * d)


def ksweep():
    """one full round (256 tiles) at growing K: time = fixed per-tile overhead (prologue + epilogue) + slope * K"""
    m, n = 2048, 8192
    for epi in ("f32", "f32+resid+ln", "swiglu", "swiglu+ln"):
        pts = []
        for k in (256, 512, 1024, 1536, 3072, 6144):
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            if epi.startswith("f32"):
                w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
                c = torch.zeros(m, n, device=dev)
                if epi == "f32":
                    f = gemm_fn(a, w, c, m, n, k, 80)
                else:
                    xo = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                    po = torch.empty((m, n // 64, 2), dtype=torch.float32, device=dev)
                    b2 = torch.randn(n, device=dev)
                    f = (lambda a=a, w=w, c=c, xo=xo, po=po, b2=b2, k=k: _hip.check(lib.sat_gemm_resid_ln_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo),
                                                                                                                    _hip.ptr(po), m, n, k, 80, _hip.stream())))
            else:
                w = torch.randn(n, k, device=dev) * 0.05
                bias = torch.randn(n, device=dev) * 0.1
                wp = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
                out = torch.empty((m, n // 2), dtype=torch.bfloat16, device=dev)
                if epi == "swiglu":
                    bp = torch.empty((n,), dtype=torch.float32, device=dev)
                    _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp), _hip.ptr(out), m, n, k, 80, _hip.stream()))
                    f = (lambda a=a, w=w, bias=bias, wp=wp, bp=bp, out=out, k=k: _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(bp),
                                                                                                                      _hip.ptr(out), m, n, k, 80 | 0x4000, _hip.stream())))
                else:
                    part = torch.stack([a.float().view(m, k // 64, 64).sum(-1), (a.float() ** 2).view(m, k // 64, 64).sum(-1)], -1).contiguous()
                    gamma = 0.8 + 0.2 * torch.rand(k, device=dev)
                    beta = 0.1 * torch.randn(k, device=dev)
                    c12 = torch.empty((2 * n,), dtype=torch.float32, device=dev)
                    _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(a), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias), _hip.ptr(wp),
                                                           _hip.ptr(c12), _hip.ptr(out), m, n, k, 80, _hip.stream()))
                    f = (lambda a=a, part=part, w=w, gamma=gamma, beta=beta, bias=bias, wp=wp, c12=c12, out=out, k=k: _hip.check(lib.sat_gemm_swiglu_ln_bf16(
                        _hip.ptr(a), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias), _hip.ptr(wp), _hip.ptr(c12), _hip.ptr(out), m, n, k,
                        80 | 0x4000, _hip.stream())))
            t = min(timeit(f, iters=20, warm=3) for _ in range(3)) * 1e3
            pts.append((k, t))
        (k1, t1), (k2, t2) = pts[2], pts[-1]
        slope = (t2 - t1) / (k2 - k1)
        print(f"ksweep {epi:14s} " + "  ".join(f"K={k}: {t:6.1f} us" for k, t in pts) + f"   slope {slope*64:.3f} us per K-tile, intercept {t1 - slope*k1:.1f} us", flush=True)


def timeline():
    """DBG 9: per-workgroup, per-K-range timestamps (100 MHz) of the persistent kernel: where a workgroup's time goes"""
    import ctypes
    import numpy as np
    lib.sat_gemm_ph8_timestamps.restype = ctypes.c_int32
    lib.sat_gemm_ph8_timestamps.argtypes = [ctypes.c_void_p]
    for name, m, n, k in [("ff_out B8", 16400, 1536, 6144), ("to_out B8", 16400, 1536, 1536), ("to_out B1", 2050, 1536, 1536), ("ff_out B1", 2050, 1536, 6144), ("qkv-like B1 f32", 2050, 4608, 1536)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev)
        f = gemm_fn(a, w, c, m, n, k, 980, accumulate=1)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        buf = np.zeros((256, 4, 8), dtype=np.uint64)
        _hip.check(lib.sat_gemm_ph8_timestamps(buf.ctypes.data))
        v = buf[buf[:, :, 7] == 1].astype(np.int64)
        t0 = v[:, 0].min()
        tt = (v[:, :4] - t0) / 100.0
        print(f"timeline {name} {m}x{n}x{k}: {len(v)} K-ranges on {int((buf[:, 0, 7] == 1).sum())} workgroups, span {tt[:, 3].max():.1f} us")
        for lab, sel in (("whole tiles", v[:, 4] == 1), ("partial, not last", (v[:, 4] == 0) & (v[:, 6] == 0)), ("partial, last arriver", (v[:, 4] == 0) & (v[:, 6] == 1))):
            if sel.sum():
                x = tt[sel]
                print(f"  {lab:22s} n={int(sel.sum()):4d}  K-tiles {np.median(v[sel, 5]):5.1f}  main {np.median(x[:,1]-x[:,0]):6.2f} us  prepare+fixup {np.median(x[:,2]-x[:,1]):6.2f} (p90 {np.percentile(x[:,2]-x[:,1],90):6.2f})"
                      f"  epilogue {np.median(x[:,3]-x[:,2]):6.2f} (p90 {np.percentile(x[:,3]-x[:,2],90):6.2f})  end at {np.median(x[:,3]):6.1f} (max {x[:,3].max():6.1f})", flush=True)


def ph2():
    """A/B of the shipped two-phase K-tile (32 MFMAs per phase, 4 barriers) against a variant selected by PH2_BIT: 0x40000 = the first
    version's four phases of 16 MFMAs (8 barriers), 0x80000 = W-hi issued one phase earlier"""
    V2 = 80 | (int(os.environ.get("PH2_BIT", "0x40000"), 16))          # bit 18: the four-phase loop; bit 19: W-hi issued one phase earlier
    for (m, n, k) in [(256, 256, 256), (512, 512, 512), (300, 512, 384), (2050, 1536, 1536), (257, 768, 6144)]:
        check(m, n, k, V2)
    for (m, n, k) in [(512, 512, 512), (4096, 4096, 4096), (2050, 12288, 1536)]:
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        As = [torch.randn(m, k, device=dev).to(torch.bfloat16), (torch.randn(m, k, device=dev) * 2 + 0.5).to(torch.bfloat16)]
        first, diff = [None, None], 0
        for i in range(40):
            c = torch.full((m, n), float("nan"), device=dev)
            gemm_fn(As[i & 1], w, c, m, n, k, V2)()
            if first[i & 1] is None:
                first[i & 1] = c
            else:
                diff += int(not torch.equal(c, first[i & 1]))
        print(f"race(ph2) M={m} N={n} K={k}: {diff} of 38 repeats differ", flush=True)
    for name, m, n, k in [("4096^3", 4096, 4096, 4096), ("1 round K=1536", 2048, 8192, 1536), ("ff_out B8", 16400, 1536, 6144), ("f32 ff_in-shape B8", 16400, 12288, 1536)]:
        arms_bench("ph2 " + name, m, n, k, [80, V2], blas=False, rounds=5)
    # SwiGLU + LayerNorm fold at the plan's shapes
    for name, m in [("B1", 2050), ("B8", 16400)]:
        n, k = 12288, 1536
        xb = torch.randn(m, k, device=dev).to(torch.bfloat16)
        part = torch.stack([xb.float().view(m, k // 64, 64).sum(-1), (xb.float() ** 2).view(m, k // 64, 64).sum(-1)], -1).contiguous()
        w = torch.randn(n, k, device=dev) * 0.05
        gamma = 0.8 + 0.2 * torch.rand(k, device=dev)
        beta = 0.1 * torch.randn(k, device=dev)
        bias = torch.randn(n, device=dev) * 0.1
        wp = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
        c12 = torch.empty((2 * n,), dtype=torch.float32, device=dev)
        outs = {}
        res = {80: [], V2: []}
        for v in res:
            out = torch.empty((m, n // 2), dtype=torch.bfloat16, device=dev)
            _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias), _hip.ptr(wp),
                                                   _hip.ptr(c12), _hip.ptr(out), m, n, k, v, _hip.stream()))
            outs[v] = out
        print(f"ph2 swiglu {name}: outputs equal: {torch.equal(outs[80], outs[V2])}")
        out = outs[80]
        for _ in range(5):
            for v in res:
                res[v].append(timeit(lambda: _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(bias),
                                                                                   _hip.ptr(wp), _hip.ptr(c12), _hip.ptr(out), m, n, k, v | 0x4000, _hip.stream())), iters=10, warm=2))
        a, b = statistics.median(res[80]), statistics.median(res[V2])
        print(f"ph2 swiglu+ln {name}: v80 {a*1e3:.1f} us  v80+{V2 >> 16:x} {b*1e3:.1f} us  x{a/b:.3f}", flush=True)


def small():
    """the 128 x 128 geometry of the 8-phase kernel (variant 81: 4 waves, two workgroups per CU) against the 16-wave-family tiles the plan
    uses at one prompt: 44 / 15 (128 x 128, 8 waves), 16 (128 x 64), 30 (256 x 192)"""
    for (m, n, k) in [(128, 128, 128), (256, 256, 256), (300, 512, 384), (2050, 1536, 1536), (257, 768, 6144), (1025, 1536, 1536)]:
        check(m, n, k, 81)
        check(m, n, k, 81, fill="rows")
    for (m, n, k) in [(512, 512, 512), (2050, 1536, 6144), (1025, 1536, 1536)]:
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        As = [torch.randn(m, k, device=dev).to(torch.bfloat16), (torch.randn(m, k, device=dev) * 2 + 0.5).to(torch.bfloat16)]
        first, diff = [None, None], 0
        for i in range(40):
            c = torch.full((m, n), float("nan"), device=dev)
            gemm_fn(As[i & 1], w, c, m, n, k, 81)()
            if first[i & 1] is None:
                first[i & 1] = c
            else:
                diff += int(not torch.equal(c, first[i & 1]))
        print(f"race(81) M={m} N={n} K={k}: {diff} of 38 repeats differ", flush=True)
    for name, m, n, k, olds in [("ff_out B1", 2050, 1536, 6144, (44, 15)), ("to_out B1", 2050, 1536, 1536, (15, 16)), ("cross out B1", 1025, 1536, 1536, (16, 15))]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w2 = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        b2 = torch.randn(n, device=dev)
        c = torch.zeros(m, n, device=dev)
        xo = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        po = torch.empty((m, n // 64, 2), dtype=torch.float32, device=dev)
        arms = {f"v{v}": v for v in olds}
        arms["v81"] = 81
        res = {kk: [] for kk in arms}
        for _ in range(5):
            for kk, v in arms.items():
                res[kk].append(timeit(lambda: _hip.check(lib.sat_gemm_resid_ln_bf16(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po), m, n, k,
                                                                                   v, _hip.stream())), iters=10, warm=2))
        fl = 2.0 * m * n * k
        print(f"small resid+ln {name} {m}x{n}x{k}: " + "  ".join(f"{kk} {statistics.median(vv)*1e3:.1f} us ({fl/statistics.median(vv)/1e9:.0f} TF)" for kk, vv in res.items()), flush=True)
    b, s_len, s_pad, d = 2, 1025, 1152, 1536
    a = torch.randn(b * s_len, d, device=dev).to(torch.bfloat16)
    w = (torch.randn(3 * d, d, device=dev) * 0.05).to(torch.bfloat16)
    inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
    q = torch.empty((b, 24, s_pad, 64), dtype=torch.bfloat16, device=dev)
    kk_ = torch.empty_like(q)
    vt = torch.empty((b, 24, 64, s_pad), dtype=torch.bfloat16, device=dev)
    scratch = torch.empty((2 * s_len * 16,), dtype=torch.float32, device=dev)
    res = {30: [], 80: [], 81: []}
    for _ in range(5):
        for v in res:
            res[v].append(timeit(lambda: _hip.check(lib.sat_qkv_rope_bf16(_hip.ptr(a), _hip.ptr(w), _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(kk_), _hip.ptr(vt), _hip.ptr(scratch),
                                                                       b, s_len, s_pad, d, v, _hip.stream())), iters=10, warm=2))
    print("small qkv heads+rope B1 (+memsets): " + "  ".join(f"v{v} {statistics.median(t)*1e3:.1f} us" for v, t in res.items()), flush=True)


def ablate():
    for name, m, n, k in [("4096^3", 4096, 4096, 4096), ("ff_in B8", 16400, 12288, 1536)]:
        arms_bench("ablate " + name, m, n, k, [80, 180, 280, 380], blas=False, rounds=3)


def ffout():
    """FF-out (fp32 residual update + LayerNorm-fold producer, K = 6144): whole tiles vs per-tile K parts vs contiguous stream-K shares of
    the remainder round, interleaved; the split results are checked against the whole-tile result"""
    for name, m in [("B1", 2050), ("B4", 8200), ("B8", 16400), ("sa2", 12290)]:
        nn, kk = 1536, 6144
        a = torch.randn(m, kk, device=dev).to(torch.bfloat16)
        w2 = (torch.randn(nn, kk, device=dev) * 0.05).to(torch.bfloat16)
        b2 = torch.randn(nn, device=dev)
        c = torch.zeros(m, nn, device=dev)
        xo = torch.empty((m, nn), dtype=torch.bfloat16, device=dev)
        po = torch.empty((m, nn // 64, 2), dtype=torch.float32, device=dev)
        mk = lambda v: (lambda: _hip.check(lib.sat_gemm_resid_ln_bf16(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po), m, nn, kk, v,
                                                                       _hip.stream())))
        arms = {"v22": 22, "whole": 80 | 0x20000, "auto": 0}
        ref = None
        for k_, v in arms.items():
            c.zero_(); mk(v)(); torch.cuda.synchronize()
            got = (c.clone(), xo.float().clone(), po.clone())
            if k_ == "whole":
                ref = got
            elif ref is not None:
                errs = [((g_ - r_).norm() / r_.norm()).item() for g_, r_ in zip(got, ref)]
                assert max(errs) < 2e-3, (name, k_, errs)
        res = {k_: [] for k_ in arms}
        for _ in range(5):
            for k_, v in arms.items():
                res[k_].append(timeit(mk(v), iters=10, warm=2))
        flops = 2.0 * m * nn * kk
        print(f"ff_out {name:4s} {m}x{nn}x{kk}: " + " | ".join(f"{k_} {statistics.median(t)*1e3:6.1f} us {flops/statistics.median(t)/1e9:6.1f} TF" for k_, t in res.items()), flush=True)


def _ts_report(label):
    import ctypes
    import numpy as np
    lib.sat_gemm_ph8_timestamps.restype = ctypes.c_int32
    lib.sat_gemm_ph8_timestamps.argtypes = [ctypes.c_void_p]
    buf = np.zeros((256, 4, 8), dtype=np.uint64)
    _hip.check(lib.sat_gemm_ph8_timestamps(buf.ctypes.data))
    v = buf[buf[:, :, 7] == 1].astype(np.int64)
    t0 = v[:, 0].min()
    tt = (v[:, :4] - t0) / 100.0
    print(f"timeline {label}: first {len(v)} K-ranges (<= 4 per workgroup) on {int((buf[:, 0, 7] == 1).sum())} workgroups")
    for r in range(4):
        sel = buf[:, r, 7] == 1
        if sel.sum():
            x = (buf[sel, r, :4].astype(np.int64) - t0) / 100.0
            print(f"  range {r}: n={int(sel.sum()):4d}  start {np.median(x[:,0]):7.2f}  main {np.median(x[:,1]-x[:,0]):6.2f} us (p90 {np.percentile(x[:,1]-x[:,0],90):6.2f})  prepare-next {np.median(x[:,2]-x[:,1]):6.2f}"
                  f"  epilogue {np.median(x[:,3]-x[:,2]):6.2f} (p10 {np.percentile(x[:,3]-x[:,2],10):6.2f} p90 {np.percentile(x[:,3]-x[:,2],90):6.2f})  end {np.median(x[:,3]):7.2f} (max {x[:,3].max():7.2f})", flush=True)


def qkv8():
    """to_qkv + RoPE + head split with the LayerNorm fold at 8 and 1 prompts (the product call): time per launch by tile family and the
    per-workgroup timeline of the 8-phase kernel's heads epilogue (experiments build: DBG 9); FF-in SwiGLU next to it as the yardstick"""
    d, s, s_pad = 1536, 1025, 1152
    f16 = bool(os.environ.get("PROBE_F16"))
    odt = torch.float16 if f16 else torch.bfloat16
    fq = lib.sat_qkv_rope_ln_f16 if f16 else lib.sat_qkv_rope_ln_bf16
    fs = lib.sat_gemm_swiglu_ln_f16 if f16 else lib.sat_gemm_swiglu_ln_bf16
    print("operands:", odt)
    h = d // 64
    for name, b in (("B8", 16), ("B1", 2)):
        m = b * s
        xb = torch.randn(m, d, device=dev).to(odt)
        part = torch.stack([xb.float().view(m, d // 64, 64).sum(-1), xb.float().view(m, d // 64, 64).pow(2).sum(-1)], dim=-1).contiguous()
        w = torch.randn(3 * d, d, device=dev) * 0.05
        gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
        wp = torch.empty((3 * d, d), dtype=odt, device=dev)
        c12 = torch.empty((6 * d,), dtype=torch.float32, device=dev)
        nset = int(os.environ.get("PROBE_SETS", "1"))          # > 1: rotate through destination / source sets (defeats the 256-MB memory-side cache)
        sets = []
        for i in range(nset):
            q = torch.zeros((b, h, s_pad, 64), dtype=odt, device=dev)
            sets.append((xb if i == 0 else xb.clone(), q, torch.zeros_like(q), torch.zeros((b, h, 64, s_pad), dtype=odt, device=dev)))
        scratch = torch.empty((2 * s * 16,), dtype=torch.float32, device=dev)
        cnt = [0]

        def mk(v):
            def f():
                xx, q, k_, vt = sets[cnt[0] % nset]
                cnt[0] += 1
                _hip.check(fq(_hip.ptr(xx), _hip.ptr(part), _hip.ptr(w), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(wp), _hip.ptr(c12),
                              _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(k_), _hip.ptr(vt), _hip.ptr(scratch), b, s, s_pad, d, v, _hip.stream()))
            return f
        mk(0)()
        torch.cuda.synchronize()
        arms = {"auto": 0x4000, "v80": 80 | 0x4000, "v30": 30 | 0x4000, "v60": 60 | 0x4000}
        res = {k2: [] for k2 in arms}
        for _ in range(5):
            for k2, v in arms.items():
                try:
                    res[k2].append(timeit(mk(v), iters=10, warm=2))
                except Exception:
                    res[k2].append(float("nan"))
        fl = 2.0 * m * 3 * d * d
        print(f"qkv {name} {m}x{3*d}x{d}: " + " | ".join(f"{k2} {statistics.median(t)*1e3:6.1f} us {fl/statistics.median(t)/1e9:6.1f} TF" for k2, t in res.items()), flush=True)
        if os.environ.get("SAT_HIP_EXP") and not f16:
            f = mk(980 | 0x4000)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            _ts_report(f"qkv heads {name}")
        # yardstick: FF-in SwiGLU with the fold on the same rows
        n2 = 12288
        w2 = torch.randn(n2, d, device=dev) * 0.05
        b2 = torch.zeros(n2, device=dev)
        wp2 = torch.empty((n2, d), dtype=odt, device=dev)
        c122 = torch.empty((2 * n2,), dtype=torch.float32, device=dev)
        hh = torch.empty((m, n2 // 2), dtype=odt, device=dev)
        mk2 = lambda v: (lambda: _hip.check(fs(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w2), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(b2), _hip.ptr(wp2),
                                                                       _hip.ptr(c122), _hip.ptr(hh), m, n2, d, v, _hip.stream())))
        mk2(0)()
        torch.cuda.synchronize()
        t = statistics.median([timeit(mk2(80 | 0x4000), iters=10, warm=2) for _ in range(3)])
        print(f"ff_in {name} {m}x{n2}x{d}: v80 {t*1e3:6.1f} us {2.0*m*n2*d/t/1e9:6.1f} TF", flush=True)
        if os.environ.get("SAT_HIP_EXP") and not f16:
            f = mk2(980 | 0x4000)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            _ts_report(f"ff_in swiglu {name}")


if __name__ == "__main__":
    what = sys.argv[1:] or ["calib", "race", "shapes"]
    print(torch.cuda.get_device_name(0), flush=True)
    for wname in what:
        print(f"==== {wname}", flush=True)
        globals()[wname]()
