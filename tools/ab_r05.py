"""Same-box A/B of the DiT denoiser step (round 6): the ROUND-5 library (tools/ab/libsat_hip_r05.so = `make` of commit 848fe42, byte-identical
to the library round 5 shipped; git-ignored like every .so, it travels to the GPU box) against this tree's library under named settings of its
per-plan switches, all timed INTERLEAVED in one process on one box (box-to-box spread of the bench line is +-3 %): one full-size model per
entry, `dit.denoise` = one CFG-7 denoiser step (2 sequences per prompt), what generate_diffusion_cond calls 100 times.
Entries: AB_SET="name:key=value;key=value,name:..." with keys tile_policy / old=1 (a second instance of the round-5 library:
instance-to-instance noise) / lib=path (another build of this tree).  Developer tool; not part of the product or the tests.     usage: python tools/ab_r05.py [batch ...]      (default: 1 8)"""
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


def load_old(path):
    """the round-5 library behind this tree's Python: same entry points minus the sized plan constructor (its struct is the version-5 prefix)"""
    h = ctypes.CDLL(path)
    for name, (res, args) in _hip._SIGNATURES.items():
        if not hasattr(h, name):
            continue
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    h.sat_dit_plan_create_sized = lambda cfg, size, out: h.sat_dit_plan_create(cfg, out)
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
    new = _hip.lib()
    entries = {"r05": (load_old(os.path.join(ROOT, "tools", "ab", "libsat_hip_r05.so")), {})}
    spec = os.environ.get("AB_SET", "r05b:old=1,r06:")
    for item in filter(None, spec.split(",")):
        name, _, kv = item.partition(":")
        raw = dict(p.split("=") for p in filter(None, kv.split(";")))
        libpath = raw.pop("lib", None)          # lib=tools/ab/x.so: another build of THIS tree (same ABI); old=1: a further instance of the round-5 library
        opts = {k: int(v) for k, v in raw.items()}
        if libpath:
            h = ctypes.CDLL(os.path.join(ROOT, libpath))
            for fname, (res, args) in _hip._SIGNATURES.items():
                fn = getattr(h, fname)
                fn.restype, fn.argtypes = res, args
            entries[name] = (h, opts)
        else:
            entries[name] = (entries["r05"][0] if opts.pop("old", 0) else new, opts)
    print(torch.cuda.get_device_name(0), {k: (v[0].sat_version(), v[1]) for k, v in entries.items()}, flush=True)
    dits = {}
    for name, (h, opts) in entries.items():
        _hip._lib = h
        with _init.skip_init():
            model = S.create_model_from_config(MC.stable_audio_open_1_0())
        model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 0))
        dit = model.to(dev).eval().model.model
        if "tile_policy" in opts:
            dit.set_tile_policy(opts["tile_policy"])
        dits[name] = dit
    fmts = os.environ.get("AB_FMT", "fp16").split(",")
    for fmt in fmts:
        for b in [int(a) for a in sys.argv[1:]] or [1, 8]:
            c = torch.randn(b, 130, 768, device=dev)
            g = torch.randn(b, 1536, device=dev)
            x = torch.randn(b, 64, 1024, device=dev)
            outs = {}
            for name, (h, _) in entries.items():
                _hip._lib = h
                dits[name].set_gemm_dtype(fmt)
                dits[name].prepare_generation(c, g, 7.0)
                outs[name] = dits[name].denoise(x, 3.0, cfg_scale=7.0).clone()
            res = {name: [] for name in entries}
            for _ in range(int(os.environ.get("AB_ROUNDS", "6"))):
                for name, (h, _) in entries.items():
                    _hip._lib = h
                    out = torch.empty_like(x)
                    res[name].append(timeit(lambda: dits[name].denoise(x, 3.0, cfg_scale=7.0, out=out)))
            base = statistics.median(res["r05"])
            diffs = {n: ((outs[n] - outs["r05"]).norm() / outs["r05"].norm()).item() for n in entries if n != "r05"}
            print(f"DiT CFG step, {fmt}, {b} prompt(s): " + "  ".join(f"{n} {statistics.median(v):.3f} ms (min {min(v):.3f}, {100 * (base / statistics.median(v) - 1):+.2f} %)"
                                                                       for n, v in res.items()) + "; rel-L2 vs r05: " + ", ".join(f"{n} {d:.1e}" for n, d in diffs.items()), flush=True)
    for name, dit in dits.items():          # every plan goes back to the library that made it
        if dit._plan is not None:
            entries[name][0].sat_dit_plan_destroy(dit._plan)
            dit._plan = None
    _hip._lib = new


if __name__ == "__main__":
    main()
