"""SURVEY section 8 row f1/f2 on the GPU: the counterparts of the reference's entry scripts run end to end.

* a synthetic checkpoint in the REFERENCE's key layout (tests/golden/state_dict_keys.json pins names and shapes against the
  reference's own ``state_dict()``) goes to ``.safetensors``, comes back through ``load_ckpt_state_dict`` (models/utils.py:6-12)
  and into the model with ``load_state_dict(strict=True)``;
* ``generate.py`` ``main()`` (reference generate.py:83-151: YAML tree -> items -> batches -> ``generate_diffusion_cond`` -> int16
  WAV per item, ``--clip-length``) on that checkpoint, text embeddings from a file: files, lengths, and bit-equality of the
  samples with a direct ``generate_diffusion_cond`` + ``float_to_int16_audio`` call on the same seed;
* ``reconstruct_audios.py`` ``main()`` (reference reconstruct_audios.py:71-149): chunked encode / decode of every WAV below a
  directory, against a direct ``reconstruct_audio`` call with the same seed.
"""
import json
import os
import runpy
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def package_default(request):
    """The operand format a freshly built model starts with: "fp16" is what a user of the scripts gets
    (stable_audio_tools/_config.py) and what the suite runs on by default (conftest.py), "bf16" the other build.  The scripts and the direct calls they are compared with are built under the same default."""
    from stable_audio_tools import _config
    prev = _config.set_default_gemm_dtype(request.param)
    yield request.param
    _config.set_default_gemm_dtype(prev)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "friendly-stable-audio-tools_amd")


def _run_script(name, argv):
    old = sys.argv
    sys.argv = [name] + argv
    try:
        runpy.run_path(os.path.join(PKG, name), run_name="__main__")
    finally:
        sys.argv = old


@pytest.mark.parametrize("package_default", ["bf16", "fp16"], indirect=True)
def test_checkpoint_roundtrip_and_generate_script(dev, tmp_path, package_default):
    import stable_audio_tools as S
    from safetensors.torch import save_file
    from stable_audio_tools import model_configs as MC, synthetic
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.utils import load_ckpt_state_dict
    from stable_audio_tools.utils.audio_utils import float_to_int16_audio
    from stable_audio_tools.utils.wav_io import load_wav

    cfg = MC.reduced(MC.stable_audio_open_1_0())
    cfg_path, ckpt_path = tmp_path / "model_config.json", tmp_path / "model.safetensors"
    json.dump(cfg, open(cfg_path, "w"))
    with _init.skip_init():
        donor = S.create_model_from_config(cfg)
    sd = synthetic.synth_state_dict(donor.state_dict(), 11)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt_path))

    # ---- checkpoint -> load_ckpt_state_dict -> strict load
    loaded = load_ckpt_state_dict(str(ckpt_path))
    assert set(loaded) == set(sd) and all(torch.equal(loaded[k], sd[k]) for k in sd)
    with _init.skip_init():
        model = S.create_model_from_config(cfg)
    missing = model.load_state_dict(loaded, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model = model.to(dev).eval()

    # ---- generate.py main()
    cond_dim = cfg["model"]["conditioning"]["cond_dim"]
    prompts = {"demo/break": "Amen break 174 BPM", "demo/pad": "warm analog pad", "fx/riser": "white noise riser"}
    seconds = {"demo/break": 0.01, "demo/pad": 0.02, "fx/riser": 0.04}          # the reduced model generates 64 * 32 samples = 46 ms
    tree = {}
    for path, text in prompts.items():
        grp, name = path.split("/")
        tree.setdefault(grp, {})[name] = {"prompt": text, "seconds_start": 0, "seconds_total": seconds[path]}
    import yaml
    yaml.safe_dump(tree, open(tmp_path / "cond.yaml", "w"))
    embeds = {text: synthetic.synth_input("emb:" + text, (5 + 3 * i, cond_dim), 50 + i) for i, text in enumerate(prompts.values())}
    torch.save(embeds, tmp_path / "embeds.pt")
    out_dir = tmp_path / "out"
    _run_script("generate.py", ["--output-dir", str(out_dir), "--cond-yaml-path", str(tmp_path / "cond.yaml"), "--model-config", str(cfg_path),
                                "--ckpt-path", str(ckpt_path), "--text-embeds", str(tmp_path / "embeds.pt"), "--sample-steps", "4",
                                "--batch-size", "4", "--n-sample-per-cond", "2", "--clip-length", "--seed", "3", "--cfg-scale", "7.0"])
    sr, sample_size = cfg["sample_rate"], cfg["sample_size"]
    files = sorted(str(p.relative_to(out_dir)) for p in out_dir.rglob("*.wav"))
    assert files == sorted(f"{p}_item-{i}.wav" for p in prompts for i in (1, 2)), files
    for path in prompts:
        a, got_sr = load_wav(out_dir / f"{path}_item-1.wav")
        assert got_sr == sr and a.shape == (2, min(int(seconds[path] * sr), sample_size)), (path, a.shape)

    # ---- the same first batch through the public API: items in file order, batch = max(batch_size // 2, 1) = 2, seed 3 + 0
    items = [(p, prompts[p]) for p in prompts for _ in (1, 2)]
    batch = items[:2]
    cond = model.conditioner([{"seconds_start": 0, "seconds_total": seconds[p]} for p, _ in batch])
    width = max(embeds[t].shape[0] for _, t in batch)
    emb = torch.zeros(len(batch), width, cond_dim)
    mask = torch.zeros(len(batch), width)
    for n, (_, t) in enumerate(batch):
        emb[n, : embeds[t].shape[0]] = embeds[t]
        mask[n, : embeds[t].shape[0]] = 1
    cond["prompt"] = (emb.to(dev), mask.to(dev))
    cond = {k: cond[k] for k in ("prompt", "seconds_start", "seconds_total")}
    audio = generate_diffusion_cond(model, steps=4, cfg_scale=7.0, conditioning_tensors=cond, sample_size=sample_size, sigma_min=0.3,
                                    sigma_max=500, sampler_type="dpmpp-3m-sde", device=str(dev), seed=3)
    for n, (p, _) in enumerate(batch):
        want = float_to_int16_audio(audio[n])[:, : int(seconds[p] * sr)]
        got, _ = load_wav(out_dir / f"{p}_item-{n + 1}.wav")
        assert torch.equal((got * 32768.0).round().to(torch.int16), want), f"{p}: script output differs from the direct call"

    # ---- real weights + random text embeddings must be refused
    with pytest.raises(SystemExit):
        _run_script("generate.py", ["--output-dir", str(tmp_path / "bad"), "--cond-yaml-path", str(tmp_path / "cond.yaml"), "--model-config",
                                    str(cfg_path), "--ckpt-path", str(ckpt_path), "--text-embeds", "random"])


@pytest.mark.parametrize("package_default", ["bf16", "fp16"], indirect=True)
def test_reconstruct_audios_script(dev, tmp_path, package_default):
    import stable_audio_tools as S
    from safetensors.torch import save_file
    from stable_audio_tools import model_configs as MC, synthetic
    from stable_audio_tools.models import _init
    from stable_audio_tools.utils.wav_io import load_wav, save_wav_float

    cfg = MC.reduced(MC.stable_audio_vae())
    cfg_path, ckpt_path = tmp_path / "vae.json", tmp_path / "vae.safetensors"
    json.dump(cfg, open(cfg_path, "w"))
    with _init.skip_init():
        model = S.create_model_from_config(cfg)
    sd = synthetic.synth_state_dict(model.state_dict(), 12)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt_path))
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    sr, ratio = model.sample_rate, model.downsampling_ratio

    in_dir = tmp_path / "in" / "nested"
    in_dir.mkdir(parents=True)
    sig = {"a.wav": synthetic.synth_input("wav_a", (2, 7000), 1, 0.2), "b.wav": synthetic.synth_input("wav_b", (1, 3100), 2, 0.2)}
    for name, x in sig.items():
        save_wav_float(in_dir / name, x, sr)
    (in_dir / ".hidden.wav").write_bytes(b"")                 # dot files are skipped by get_audio_filenames
    out_dir = tmp_path / "rec" / "reconstructed"
    torch.manual_seed(21)
    _run_script("reconstruct_audios.py", ["--audio-dir", str(tmp_path / "in"), "--output-dir", str(out_dir), "--model-config", str(cfg_path),
                                          "--ckpt-path", str(ckpt_path), "--frame-duration", str((16 * ratio + 0.5) / sr), "--overlap-rate", "0.1",
                                          "--batch-size", "3"])
    assert sorted(p.name for p in out_dir.glob("*.wav")) == ["a.wav", "b.wav"]
    assert sorted(p.name for p in (out_dir.parent / "original").glob("*.wav")) == ["a.wav", "b.wav"]
    # the same calls directly, same RNG state (VAE noise): files are processed in sorted order
    torch.manual_seed(21)
    from stable_audio_tools.data.modification import Stereo
    for name in ("a.wav", "b.wav"):
        x, _ = load_wav(in_dir / name)
        audio = Stereo()(x).unsqueeze(0).to(dev)
        rec = model.reconstruct_audio(audio, chunked=True, chunk_size=16, overlap=1, max_batch_size=3).squeeze(0)
        got, got_sr = load_wav(out_dir / name)
        assert got_sr == sr and got.shape == (2, x.shape[-1])
        want = (rec.cpu().float().clamp(-1, 1) * 32767.0).round().to(torch.int16)
        assert torch.equal((got * 32768.0).round().to(torch.int16), want), f"{name}: script output differs from the direct call"
        orig, _ = load_wav(out_dir.parent / "original" / name)
        assert orig.shape == (2, x.shape[-1])
