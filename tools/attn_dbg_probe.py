import ctypes, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/friendly-stable-audio-tools_amd")
import torch
from stable_audio_tools import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libsat_hip_exp.so")
lib = _hip.lib(); dev = torch.device("cuda:0")
lib.sat_attention_dbg_read.restype = ctypes.c_int32; lib.sat_attention_dbg_read.argtypes = [ctypes.c_void_p]
import numpy as np
for name, b, h, kvh, sq, sk in [("self B8", 16, 24, 24, 1025, 1025), ("self SA2", 2, 24, 24, 6145, 6145)]:
    sqp, skp = (sq + 127) // 128 * 128, (sk + 3 + 63) // 64 * 64
    q = torch.randn(b, h, sqp, 64, device=dev).to(torch.bfloat16); k = torch.randn(b, kvh, skp, 64, device=dev).to(torch.bfloat16)
    vt = torch.randn(b, kvh, 64, skp, device=dev).to(torch.bfloat16); o = torch.empty(b * sq, h * 64, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _hip.check(lib.sat_attention_bf16(_hip.ptr(q), _hip.ptr(k), _hip.ptr(vt), _hip.ptr(o), b, h, kvh, sq, sk, sqp, skp, _hip.stream()))
    torch.cuda.synchronize()
    out = np.zeros(4, dtype=np.uint64); _hip.check(lib.sat_attention_dbg_read(out.ctypes.data))
    ts, tw, nw, nit = [float(x) for x in out]
    print(f"{name}: per wave per tile: sync {ts/nit:.0f} cycles, work {tw/nit:.0f} cycles  (waves {nw/3:.0f}, tiles/wave {nit/nw:.1f}); share parked {100*ts/(ts+tw):.1f} %")
