"""Round 6 probe (GPU box): one CFG-7 denoiser step of the full-size model, eager launches against a hipGraph replay of the same step
(torch.cuda.CUDAGraph around sat_dit_denoise_cfg), interleaved.  Eager writes a fresh kernel-argument block per launch; a graph's kernel arguments
sit at fixed device addresses.  usage: python tools/graph_probe.py [batch ...]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "friendly-stable-audio-tools_amd"))
import torch  # noqa: E402

from stable_audio_tools import _hip  # noqa: E402

dev = torch.device("cuda:0")


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
    with _init.skip_init():
        model = S.create_model_from_config(MC.stable_audio_open_1_0())
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 0))
    dit = model.to(dev).eval().model.model
    print(torch.cuda.get_device_name(0), flush=True)
    for b in [int(a) for a in sys.argv[1:]] or [1, 8]:
        c = torch.randn(b, 130, 768, device=dev)
        g = torch.randn(b, 1536, device=dev)
        x = torch.randn(b, 64, 1024, device=dev)
        dit.prepare_generation(c, g, 7.0)
        out = torch.empty_like(x)
        eager = lambda: dit.denoise(x, 3.0, cfg_scale=7.0, out=out)
        eager()
        want = out.clone()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph.capture_begin()
            dit.denoise(x, 3.0, cfg_scale=7.0, out=out)
            graph.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        same = torch.equal(out, want)
        res = {"eager": [], "graph": []}
        for _ in range(6):
            res["eager"].append(timeit(eager))
            res["graph"].append(timeit(graph.replay))
        me, mg = statistics.median(res["eager"]), statistics.median(res["graph"])
        print(f"DiT CFG step, fp16, {b} prompt(s): eager {me:.3f} ms (min {min(res['eager']):.3f})  graph replay {mg:.3f} ms (min {min(res['graph']):.3f}, {100 * (me / mg - 1):+.2f} %)  "
              f"replay bit-identical: {same}", flush=True)


if __name__ == "__main__":
    main()
