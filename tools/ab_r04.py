"""Same-box A/B of the DiT denoiser step: this tree's libsat_hip.so against the ROUND-4 library (tools/ab/libsat_hip_r04.so, built from
commit 274bbd1 by `git worktree add /tmp/r04wt 274bbd1 && make -C /tmp/r04wt/friendly-stable-audio-tools_amd/csrc`; git-ignored like
every .so, but it travels to the GPU box).  Box-to-box spread of the bench line is +-3 %, larger than most single changes of round 5, so
the two libraries are timed INTERLEAVED in one process on one box: two full-size models (same synthetic weights), each bound to its own
library handle, `dit.denoise` (one CFG-7 denoiser step = 2 sequences per prompt, what generate_diffusion_cond calls 100 times).
Further libraries (e.g. an intermediate commit) can ride along: AB_LIBS="name=path,name=path".
Developer tool; not part of the product or the tests.     usage: python tools/ab_r04.py [batch ...]      (default: 1 8)"""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch  # noqa: E402

from stable_audio_tools import _hip  # noqa: E402

dev = torch.device("cuda:0")


def load(path):
    h = ctypes.CDLL(path)
    for name, (res, args) in _hip._SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def timeit(fn, iters=10, warm=2):
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


def main():
    import stable_audio_tools as S
    from stable_audio_tools import model_configs as MC, synthetic
    from stable_audio_tools.models import _init
    libs = {"r04": load(os.path.join(ROOT, "tools", "ab", "libsat_hip_r04.so"))}
    for item in filter(None, os.environ.get("AB_LIBS", "").split(",")):
        libs[item.split("=")[0]] = load(os.path.join(ROOT, item.split("=")[1]))
    libs["r05"] = _hip.lib()
    print(torch.cuda.get_device_name(0), {k: v.sat_version() for k, v in libs.items()}, flush=True)
    dits = {}
    for name, h in libs.items():
        _hip._lib = h
        with _init.skip_init():
            model = S.create_model_from_config(MC.stable_audio_open_1_0())
        model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 0))
        dits[name] = model.to(dev).eval().model.model
    for fmt in ("fp16", "bf16"):
        for b in [int(a) for a in sys.argv[1:]] or [1, 8]:
            c = torch.randn(b, 130, 768, device=dev)
            g = torch.randn(b, 1536, device=dev)
            x = torch.randn(b, 64, 1024, device=dev)
            outs = {}
            for name in libs:
                _hip._lib = libs[name]
                dits[name].set_gemm_dtype(fmt)
                dits[name].prepare_generation(c, g, 7.0)
                outs[name] = dits[name].denoise(x, 3.0, cfg_scale=7.0).clone()
            res = {name: [] for name in libs}
            for _ in range(6):
                for name in libs:
                    _hip._lib = libs[name]
                    out = torch.empty_like(x)
                    res[name].append(timeit(lambda: dits[name].denoise(x, 3.0, cfg_scale=7.0, out=out)))
            m4 = statistics.median(res["r04"])
            diff = ((outs["r05"] - outs["r04"]).norm() / outs["r04"].norm()).item()
            print(f"DiT CFG step, {fmt}, {b} prompt(s): " + "  ".join(f"{n} {statistics.median(v):.3f} ms (min {min(v):.3f}, {100 * (m4 / statistics.median(v) - 1):+.2f} %)"
                                                                       for n, v in res.items()) + f"; r05 vs r04 outputs differ by {diff:.2e} (rel-L2)", flush=True)
    for name, dit in dits.items():          # every plan goes back to the library that made it
        if dit._plan is not None:
            libs[name].sat_dit_plan_destroy(dit._plan)
            dit._plan = None
    _hip._lib = libs["r05"]


if __name__ == "__main__":
    main()
