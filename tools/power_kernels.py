"""Which kernels of a transformer block run at the power cap?  Each of the block's big launches is run ALONE in a tight loop for a few
seconds at the one-prompt and the eight-prompt shape (fp16 operands, N(0,1)-scale data) while rocm-smi is sampled four times a second:
socket power, shader clock and the achieved TFLOP/s per kernel.  A kernel at the cap with the clock pulled down is energy-bound (removing its
stalls buys nothing); a kernel below the cap at full clock is stall-bound.  Developer tool (round 5).
usage: python tools/power_kernels.py [seconds per kernel, default 3]"""
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch  # noqa: E402

from stable_audio_tools import _hip  # noqa: E402

dev = torch.device("cuda:0")
lib = _hip.lib()
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
samples, stop = [], threading.Event()


def poll():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout
            card = next(iter(json.loads(out).values()))
            pw = float(card.get("Current Socket Graphics Package Power (W)", "nan"))
            sclk = float(str(card.get("sclk clock speed:", "nan")).strip("()").replace("Mhz", ""))
            samples.append((time.time(), pw, sclk))
        except Exception:          # noqa: BLE001
            pass
        stop.wait(0.25)


def run(label, flops, fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < SECS:
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        n += 200
    t1 = time.time()
    sel = [(p, c) for (t, p, c) in samples if t0 + 0.7 <= t <= t1]
    pw = statistics.median([p for p, _ in sel]) if sel else float("nan")
    ck = statistics.median([c for _, c in sel]) if sel else float("nan")
    us = (t1 - t0) / n * 1e6
    print(f"  {label:34s} {us:8.1f} us/launch  {flops / us / 1e6:7.1f} TFLOP/s   power {pw:7.0f} W   sclk {ck:6.0f} MHz   ({len(sel)} samples)", flush=True)
    time.sleep(1.0)


def main():
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    print(torch.cuda.get_device_name(0), f"-- {SECS:.0f} s per kernel, fp16 operands", flush=True)
    d, s, s_pad, h, inner = 1536, 1025, 1152, 24, 6144
    hp = torch.float16
    for name, b in (("one prompt", 2), ("eight prompts", 16)):
        m = b * s
        print(f"== {name} (M = {m})", flush=True)
        xb = torch.randn(m, d, device=dev).to(hp)
        part = torch.stack([xb.float().view(m, d // 64, 64).sum(-1), xb.float().view(m, d // 64, 64).pow(2).sum(-1)], dim=-1).contiguous()
        gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        # FF-in (SwiGLU, LayerNorm fold)
        w1 = torch.randn(2 * inner, d, device=dev) * 0.05
        b1 = torch.zeros(2 * inner, device=dev)
        wp1 = torch.empty((2 * inner, d), dtype=hp, device=dev)
        c12 = torch.empty((4 * inner,), dtype=torch.float32, device=dev)
        hh = torch.empty((m, inner), dtype=hp, device=dev)
        ffin = lambda v: _hip.check(lib.sat_gemm_swiglu_ln_f16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(w1), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(b1),
                                                               _hip.ptr(wp1), _hip.ptr(c12), _hip.ptr(hh), m, 2 * inner, d, v, _hip.stream()))
        ffin(0)
        run("FF-in SwiGLU (8-phase)", 2.0 * m * 2 * inner * d, lambda: ffin(0x4000))
        # FF-out / to_out (fp32 residual + LayerNorm-fold producer)
        c = torch.zeros(m, d, device=dev)
        xo = torch.empty((m, d), dtype=hp, device=dev)
        po = torch.empty((m, d // 64, 2), dtype=torch.float32, device=dev)
        for lbl, kk in (("FF-out (K = 6144)", inner), ("to_out (K = 1536)", d)):
            a = torch.randn(m, kk, device=dev).to(hp)
            w2 = (torch.randn(d, kk, device=dev) * 0.05).to(hp)
            b2 = torch.zeros(d, device=dev)
            f = lambda a=a, w2=w2, b2=b2, kk=kk: _hip.check(lib.sat_gemm_resid_ln_f16(_hip.ptr(a), _hip.ptr(w2), _hip.ptr(b2), _hip.ptr(c), _hip.ptr(xo), _hip.ptr(po),
                                                                                       m, d, kk, 0, _hip.stream()))
            run(lbl, 2.0 * m * d * kk, f)
        # to_qkv (heads + RoPE, LayerNorm fold)
        wq = torch.randn(3 * d, d, device=dev) * 0.05
        wpq = torch.empty((3 * d, d), dtype=hp, device=dev)
        c12q = torch.empty((6 * d,), dtype=torch.float32, device=dev)
        inv_freq = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
        q = torch.zeros((b, h, s_pad, 64), dtype=hp, device=dev)
        k = torch.zeros_like(q)
        vt = torch.zeros((b, h, 64, s_pad), dtype=hp, device=dev)
        scratch = torch.empty((2 * s * 16,), dtype=torch.float32, device=dev)
        fq = lambda v: _hip.check(lib.sat_qkv_rope_ln_f16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(wq), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(wpq), _hip.ptr(c12q),
                                                          _hip.ptr(inv_freq), _hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(scratch), b, s, s_pad, d, v, _hip.stream()))
        fq(0)
        run("to_qkv (heads epilogue)", 2.0 * m * 3 * d * d, lambda: fq(0x4000))
        # self-attention on what to_qkv just wrote
        o = torch.empty((m, d), dtype=hp, device=dev)
        run("self-attention", 4.0 * b * h * s * s * 64, lambda: _hip.check(lib.sat_attention_prescaled_f16(_hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(o), b, h, h, s, s,
                                                                                                             s_pad, s_pad, _hip.stream())))
        del xb, part, w1, wp1, hh, c, xo, po, wq, wpq, q, k, vt, o
        torch.cuda.empty_cache()
    stop.set()


if __name__ == "__main__":
    main()
