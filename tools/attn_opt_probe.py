"""A/B of the attention kernel's softmax recurrences (attn_core.h MODE 0 / 1 / 2; experiments build, SAT_ATTN_MODE read at every launch): time at the four shipped shapes and
error against an fp32 softmax(QK^T/8)V of the same bf16 inputs.    python tools/attn_opt_probe.py [opts...]"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/friendly-stable-audio-tools_amd")
import torch
from stable_audio_tools import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), os.environ.get("SAT_PROBE_LIB", "libsat_hip_exp.so"))
lib = _hip.lib(); dev = torch.device("cuda:0")
opts = [int(a) for a in sys.argv[1:]] or [0, 1, 2]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def make(b, h, kvh, sq, sk, gain=1.0):
    sqp, skp = (sq + 127) // 128 * 128, (sk + 3 + 63) // 64 * 64
    torch.manual_seed(5)
    q = (torch.randn(b, h, sqp, 64, device=dev) * gain).to(torch.bfloat16)
    k = torch.randn(b, kvh, skp, 64, device=dev).to(torch.bfloat16)
    vt = torch.randn(b, kvh, 64, skp, device=dev).to(torch.bfloat16)
    o = torch.empty(b * sq, h * 64, device=dev, dtype=torch.bfloat16)
    f = lambda: _hip.check(lib.sat_attention_bf16(_hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(o), b, h, kvh, sq, sk, sqp, skp, _hip.stream()))
    return q, k, vt, o, f


# correctness: keys of sequence b sit at columns [ob, ob + sk), ob = (b*sk) & 3; V^T columns in vt_pos order (bits 2/3 of the key index swapped)
def reference(q, k, vt, b, h, kvh, sq, sk):
    out = []
    for bi in range(b):
        ob = (bi * sk) & 3
        pos = torch.arange(ob, ob + sk, device=dev)
        vpos = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1)
        kk = k[bi, :, ob:ob + sk].float().repeat_interleave(h // kvh, 0)
        vv = vt[bi][:, :, vpos].float().transpose(1, 2).repeat_interleave(h // kvh, 0)
        s = torch.einsum("hqd,hkd->hqk", q[bi, :, :sq].float(), kk) * 0.125
        out.append(torch.einsum("hqk,hkd->qhd", torch.softmax(s, -1), vv).reshape(sq, h * 64))
    return torch.cat(out)


print(torch.cuda.get_device_name(0))
for gain, label in [(1.0, "unit-variance scores"), (6.0, "peaky scores (std 6)")]:
    for (b, h, kvh, sq, sk) in [(2, 4, 4, 1025, 1025), (3, 4, 2, 300, 130)]:
        q, k, vt, o, f = make(b, h, kvh, sq, sk, gain)
        want = reference(q, k, vt, b, h, kvh, sq, sk)
        for grp in (1, 2):
            os.environ["SAT_ATTN_GROUPS"] = str(grp)
            for opt in opts:
                os.environ["SAT_ATTN_MODE"] = str(opt)
                o.zero_(); f(); torch.cuda.synchronize()
                err = ((o.float() - want).norm() / want.norm()).item()
                print(f"{label:22s} b{b} h{h}/{kvh} sq{sq} sk{sk} groups={grp} opt={opt}: rel-L2 {err:.3e}", flush=True)
del os.environ["SAT_ATTN_GROUPS"]
for name, b, h, kvh, sq, sk in [("self B1", 2, 24, 24, 1025, 1025), ("cross B1", 2, 24, 12, 1025, 130), ("self B8", 16, 24, 24, 1025, 1025),
                                ("self SA2", 2, 24, 24, 6145, 6145)]:
    q, k, vt, o, f = make(b, h, kvh, sq, sk)
    line = f"attention {name:9s}:"
    for opt in opts:
        os.environ["SAT_ATTN_MODE"] = str(opt)
        ms = timeit(f)
        line += f"   opt={opt} {ms*1e3:7.1f} us {4.0*b*h*sq*sk*64/ms/1e9:7.1f} TF"
    print(line, flush=True)
