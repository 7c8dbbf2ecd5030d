"""Host-side logic of the drop-in package (no GPU): module tree / state-dict compatibility with the reference,
conditioning bookkeeping, audio preparation, config handling, error behaviour."""
import json
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import cases  # noqa: E402
import stable_audio_tools as S  # noqa: E402
from stable_audio_tools import model_configs as MC, synthetic  # noqa: E402
from stable_audio_tools.models import _init  # noqa: E402
from util import rel_l2  # noqa: E402


@pytest.fixture(scope="module")
def ref_keys():
    return json.load(open(os.path.join(cases.GOLDEN_DIR, "state_dict_keys.json")))


def test_state_dict_matches_the_reference_checkpoint_layout(ref_keys):
    with _init.skip_init():
        model = S.create_model_from_config(MC.stable_audio_open_1_0())
    ours = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert ours == ref_keys["sa_open_1_0"]                       # 745 tensors, same names, same shapes
    a = ref_keys["attrs"]
    assert (model.io_channels, model.sample_rate, model.min_input_length, model.diffusion_objective) == \
        (a["io_channels"], a["sample_rate"], a["min_input_length"], a["diffusion_objective"])
    assert model.pretransform.downsampling_ratio == a["pretransform_ratio"]
    assert model.model.model.patch_size == 1
    with _init.skip_init():
        vae = S.create_model_from_config(MC.stable_audio_vae())
    assert {k: list(v.shape) for k, v in vae.state_dict().items()} == ref_keys["vae"]
    assert (vae.latent_dim, vae.downsampling_ratio, vae.min_length) == (a["vae_latent_dim"], a["vae_ratio"], a["vae_min_length"])
    assert vae.in_channels == vae.out_channels == vae.io_channels == 2 and vae.sample_rate == 44100
    # SA-2.0 shares the architecture (longer context only)
    with _init.skip_init():
        m2 = S.create_model_from_config(MC.stable_audio_2_0())
    assert {k: list(v.shape) for k, v in m2.state_dict().items()} == ref_keys["sa_open_1_0"]


def test_reference_init_conventions():
    """zero-initialised branch outputs (transformer.py:274-277,318-319; dit.py:130-133), weight_g = ||v||, and the
    DiTWrapper x0.5 scaling of every PARAMETER but not of buffers (diffusion.py:487-489)."""
    model = S.create_model_from_config(MC.reduced(MC.stable_audio_open_1_0()))
    dit = model.model.model
    blk = dit.transformer.layers[0]
    for w in (blk.self_attn.to_out.weight, blk.cross_attn.to_out.weight, blk.ff.ff[2].weight, blk.ff.ff[2].bias,
              dit.preprocess_conv.weight, dit.postprocess_conv.weight):
        assert torch.count_nonzero(w) == 0
    assert torch.allclose(blk.pre_norm.gamma, torch.full_like(blk.pre_norm.gamma, 0.5))
    assert torch.count_nonzero(blk.pre_norm.beta) == 0 and "beta" in dict(blk.pre_norm.named_buffers())
    inv = dit.transformer.rotary_pos_emb.inv_freq
    assert inv.shape == (16,) and torch.allclose(inv, 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))
    conv = model.pretransform.model.decoder.layers[0]
    assert torch.allclose(conv.weight_g.flatten(), conv.weight_v.flatten(1).norm(dim=1))
    assert not any(p.requires_grad for p in model.pretransform.parameters())


def test_unsupported_configs_fail_loudly():
    cfg = MC.reduced(MC.stable_audio_open_1_0())
    for key, val in (("transformer_type", "x-transformers"), ("patch_size", 2)):
        bad = json.loads(json.dumps(cfg))
        bad["model"]["diffusion"]["config"][key] = val
        with pytest.raises(NotImplementedError):
            S.create_model_from_config(bad)
    bad = json.loads(json.dumps(cfg))
    bad["model"]["diffusion"]["config"]["global_cond_type"] = "film"
    with pytest.raises(ValueError):
        S.create_model_from_config(bad)
    # adaLN is supported: same keys as the reference (transformer.py:651-655)
    ada = json.loads(json.dumps(cfg))
    ada["model"]["diffusion"]["config"]["global_cond_type"] = "adaLN"
    m = S.create_model_from_config(ada)
    d = ada["model"]["diffusion"]["config"]["embed_dim"]
    assert tuple(m.state_dict()["model.model.transformer.layers.0.to_scale_shift_gate.1.weight"].shape) == (6 * d, d)
    bad = json.loads(json.dumps(cfg))
    bad["model"]["diffusion"]["type"] = "adp_cfg_1d"
    with pytest.raises(NotImplementedError):
        S.create_model_from_config(bad)
    with pytest.raises(NotImplementedError):
        S.create_model_from_config({"model_type": "lm"})
    with pytest.raises(NotImplementedError, match="Unknown model type"):
        S.create_model_from_config({"model_type": "nonsense"})
    bad = MC.reduced(MC.stable_audio_vae())
    bad["model"]["decoder"]["config"]["final_tanh"] = True
    with pytest.raises(NotImplementedError):
        S.create_model_from_config(bad)


def test_number_conditioner_and_conditioning_inputs_match_reference():
    """The oracle's NumberConditioner restatement against the reference's own outputs (ops.npz), and the product's
    ``get_conditioning_inputs`` (host-side concatenation) fed with oracle-computed conditioner tensors against the reference's
    (host.npz).  The product's NumberConditioner itself runs on the HIP C ABI: its parity test is
    tests/test_gpu_kernels.py::test_number_conditioner_hip."""
    from oracle import conditioners as ocond
    g_ops = cases.load("ops")
    from stable_audio_tools.models.conditioners import NumberConditioner
    nc = NumberConditioner(768, min_val=0, max_val=512)
    sd = synthetic.synth_state_dict(nc.state_dict(), 1)
    emb, mask = ocond.number_conditioner(sd, "", [0.0, 47.5, 600.0, -3.0], 0, 512)
    assert rel_l2(emb, g_ops["number_cond"]) < 1e-6 and torch.equal(mask, g_ops["number_mask"])
    assert torch.equal(emb[0], emb[3]), "values are clamped to [min_val, max_val]"
    with pytest.raises(Exception):          # no CPU path in the product
        nc([1.0])

    g = cases.load("host")
    cfg = MC.stable_audio_open_1_0()
    cfg["model"]["diffusion"]["config"].update(depth=1, embed_dim=128, num_heads=2)
    cfg["model"]["pretransform"]["config"]["encoder"]["config"]["channels"] = 8
    cfg["model"]["pretransform"]["config"]["decoder"]["config"]["channels"] = 8
    # the codec is only instantiated here (channels=8 is outside the kernels' supported set; the plan is never built)
    model = S.create_model_from_config(cfg)
    msd = synthetic.synth_state_dict(model.state_dict(), 4)
    assert model.conditioner.external_ids == []
    meta = [{"seconds_start": 0, "seconds_total": 47}, {"seconds_start": 3.5, "seconds_total": 700}]
    cond = ocond.multi_conditioner(msd, "conditioner.", ["seconds_start", "seconds_total"], meta)
    cond["prompt"] = [synthetic.synth_input("prompt", (2, 128, 768), 5), torch.ones(2, 128)]
    cond = {k: cond[k] for k in ("prompt", "seconds_start", "seconds_total")}
    ci = model.get_conditioning_inputs(cond)
    assert ci["cross_attn_cond"].shape == (2, 130, 768) and ci["global_cond"].shape == (2, 1536)
    assert rel_l2(ci["cross_attn_cond"], g["cross_attn_cond"]) < 1e-6
    assert rel_l2(ci["global_cond"], g["global_cond"]) < 1e-6
    assert torch.equal(ci["cross_attn_mask"], g["cross_attn_mask"])
    neg = model.get_conditioning_inputs(cond, negative=True)
    assert set(neg) == {"negative_cross_attn_cond", "negative_cross_attn_mask", "negative_global_cond", "negative_input_concat_cond"}
    with pytest.raises(ValueError):
        model.conditioner([{"seconds_total": 5}])        # the first conditioner (seconds_start) finds no entry
    full = S.create_model_from_config(MC.reduced(MC.stable_audio_open_1_0(with_text_encoder=True)))
    assert full.conditioner.external_ids == ["prompt"]


def test_prepare_audio_and_padcrop_match_reference():
    g = cases.load("host")
    from stable_audio_tools.data.modification import Mono, PadCrop, Stereo
    from stable_audio_tools.inference.utils import prepare_audio, set_audio_channels
    a = synthetic.synth_input("pa", (1, 1000), 6)
    assert torch.equal(prepare_audio(a, 44100, 44100, 1500, 2, "cpu"), g["prepare_mono_to_stereo_pad"])
    assert torch.equal(prepare_audio(synthetic.synth_input("pa3", (3, 1000), 7), 44100, 44100, 600, 2, "cpu"), g["prepare_crop"])
    with pytest.raises(RuntimeError, match="no CPU path"):      # (SatError is a RuntimeError) resampling runs on the device only (sat_resample_sinc; tests/test_resample.py)
        prepare_audio(a, 48000, 44100, 1500, 2, "cpu")
    x = torch.arange(12.0).view(2, 6)
    assert PadCrop(4, randomize=False)(x).tolist() == [[0, 1, 2, 3], [6, 7, 8, 9]]
    assert PadCrop(8, randomize=False)(x)[:, 6:].abs().sum() == 0
    assert Mono()(x).shape == (1, 6) and Stereo()(x[0]).shape == (2, 6) and Stereo()(torch.ones(3, 5)).shape == (2, 5)
    assert set_audio_channels(torch.ones(2, 1, 4), 2).shape == (2, 2, 4) and set_audio_channels(torch.ones(2, 2, 4), 1).shape == (2, 1, 4)


def test_generate_rejects_what_the_reference_api_cannot_do():
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    from stable_audio_tools.inference.sampling import sample_k
    model = S.create_model_from_config(MC.reduced(MC.stable_audio_open_1_0()))
    with pytest.raises(AssertionError):
        generate_diffusion_cond(model, steps=2, device="cpu")
    cond = {"prompt": (torch.zeros(1, 128, 128), torch.ones(1, 128)), "seconds_start": (torch.zeros(1, 1, 128), torch.ones(1, 1)),
            "seconds_total": (torch.zeros(1, 1, 128), torch.ones(1, 1))}
    # a negative prompt given as raw metadata needs a conditioner for every id: the number conditioners find no entry here
    with pytest.raises(ValueError, match="not found"):
        generate_diffusion_cond(model, steps=2, conditioning_tensors=cond, negative_conditioning=[{"prompt": "x"}], device="cpu")
    with pytest.raises(NotImplementedError, match="sampler_type"):
        sample_k(model.model, torch.zeros(1, 64, 8), sampler_type="k-euler-nonexistent")
    with pytest.raises(NotImplementedError):
        sample_k(lambda x, s: x, torch.zeros(1, 64, 8), sampler_type="dpmpp-3m-sde")


def test_utils_and_file_scan(tmp_path):
    from stable_audio_tools.data.dataset import get_audio_filenames
    from stable_audio_tools.utils.torch_common import copy_state_dict, count_parameters, get_rank, get_world_size
    assert get_rank() == 0 and get_world_size() == 1
    lin = torch.nn.Linear(3, 2)
    lin.register_buffer("buf", torch.zeros(5))
    assert count_parameters(lin) == 3 * 2 + 2 + 5
    copy_state_dict(lin, {"weight": torch.ones(2, 3), "bias": torch.ones(7), "other": torch.ones(1)})
    assert torch.equal(lin.weight.data, torch.ones(2, 3)) and lin.bias.shape == (2,)
    (tmp_path / "a").mkdir()
    for name in ("x.wav", "a/y.FLAC", "a/.hidden.wav", "z.txt", "a/drums_loop.mp3"):
        (tmp_path / name).write_bytes(b"")
    found = sorted(os.path.relpath(f, tmp_path) for f in get_audio_filenames(str(tmp_path)))
    assert found == ["a/drums_loop.mp3", "a/y.FLAC", "x.wav"]
    assert [os.path.basename(f) for f in get_audio_filenames([str(tmp_path)], keywords=["DRUMS"])] == ["drums_loop.mp3"]


def test_scripts_host_side(tmp_path):
    """generate.py condition-tree flattening (reference generate.py:38-50) and the stdlib WAV round trip."""
    import importlib.util
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "friendly-stable-audio-tools_amd")
    spec = importlib.util.spec_from_file_location("sat_generate", os.path.join(pkg, "generate.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    tree = {"drums": {"funk": {"prompt": "funk break", "seconds_start": 0, "seconds_total": 8},
                      "sub": {"slow": {"prompt": "slow", "seconds_start": 0, "seconds_total": 20}}},
            "pads": {"warm": {"prompt": "warm pad", "seconds_start": 0, "seconds_total": 30}}}
    flat = gen.flatten_conditions(tree)
    assert list(flat) == ["drums/funk", "drums/sub/slow", "pads/warm"] and flat["pads/warm"]["seconds_total"] == 30
    with pytest.raises(AssertionError):
        gen.flatten_conditions({"prompt": "x"})
    e1, e2 = gen.text_embedding("a", 768), gen.text_embedding("a", 768)
    assert e1.shape == (128, 768) and torch.equal(e1, e2) and not torch.equal(e1, gen.text_embedding("b", 768))
    from stable_audio_tools.utils.wav_io import load_wav, save_wav_float, save_wav_int16
    pcm = torch.randint(-32768, 32767, (2, 1000), dtype=torch.int32).to(torch.int16)
    save_wav_int16(tmp_path / "a.wav", pcm, 44100)
    back, sr = load_wav(tmp_path / "a.wav")
    assert sr == 44100 and back.shape == (2, 1000) and torch.equal((back * 32768).round().to(torch.int16), pcm)
    save_wav_float(tmp_path / "b.wav", torch.tensor([[0.5, -2.0, 1.0]]), 8000)
    b, sr = load_wav(tmp_path / "b.wav")
    assert sr == 8000 and torch.allclose(b, torch.tensor([[0.5, -1.0, 1.0]]), atol=1e-4)


def test_rope_angle_addition_recurrence_error():
    """The 8-phase heads epilogue (csrc/gemm_ph8.hip, round 5) fetches (cos, sin) of a lane's first row block and walks to the next -- 16
    positions on -- by the angle-addition rotation with the table row of position 16, in fp32, up to seven times.  The claims in the kernel's
    comment, for every start position and every frequency of rotary_freqs (transformer.py:158-183): against EXACT arithmetic the walked values
    are as accurate as the table; they differ from the table (= the reference's fp32 pos * inv_freq) by no more than its own angle rounding --
    <= 5e-5 at the SA-Open length, <= 3e-4 at the SA-2.0 length, rms an order below -- against a 16-bit rounding of q / k of 1.4e-4 rms."""
    import numpy as np
    for s_len, bound in ((1025, 5e-5), (6145, 3e-4)):
        inv_freq = (1.0 / (10000 ** (np.arange(0, 32, 2, dtype=np.float32) / np.float32(32)))).astype(np.float32)
        ang = (np.arange(s_len, dtype=np.float32)[:, None] * inv_freq[None, :]).astype(np.float32)          # fp32 product, as the table is built
        cos_t, sin_t = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
        pos64 = np.arange(s_len, dtype=np.float64)[:, None] * inv_freq[None, :].astype(np.float64)
        table_vs_exact = max(float(np.abs(cos_t - np.cos(pos64)).max()), float(np.abs(sin_t - np.sin(pos64)).max()))
        c16, s16 = cos_t[16], sin_t[16]
        start = np.arange(0, s_len - 7 * 16)
        c, s = cos_t[start].copy(), sin_t[start].copy()
        vs_table = vs_exact = 0.0
        for step in range(1, 8):
            c, s = (c * c16 - s * s16).astype(np.float32), (s * c16 + c * s16).astype(np.float32)
            idx = start + 16 * step
            vs_table = max(vs_table, float(np.abs(c - cos_t[idx]).max()), float(np.abs(s - sin_t[idx]).max()))
            vs_exact = max(vs_exact, float(np.abs(c - np.cos(pos64[idx])).max()), float(np.abs(s - np.sin(pos64[idx])).max()))
            rms = float(np.sqrt(np.mean((c - cos_t[idx]) ** 2)))
        assert vs_table <= bound and rms <= bound / 10, (s_len, vs_table, rms)
        assert vs_exact <= 1.05 * table_vs_exact + 1e-6, (s_len, vs_exact, table_vs_exact)


def test_codec_format_follows_the_dit_by_one_rule():
    """generate.py and bench.py picked the codec's operand format differently for the e4m3 modes.  One rule now
    (stable_audio_tools/_config.py: codec_gemm_dtype), used by both scripts and by the 100-step parity test."""
    from stable_audio_tools import _config
    assert [_config.codec_gemm_dtype(d) for d in ("fp16", "bf16", "fp8", "fp8-all", "fp32x")] == ["fp16", "bf16", "bf16", "bf16", "fp16"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script in ("bench.py", os.path.join("friendly-stable-audio-tools_amd", "generate.py")):
        assert "codec_gemm_dtype" in open(os.path.join(root, script)).read(), script


def test_package_default_is_fp16():
    """What a user gets without touching any switch: fp16 operands (the reference's own GPU arithmetic).  The suite runs on that default
    unless SAT_TEST_DTYPE=bf16 (conftest.py); the switch is per model and process-wide."""
    import conftest
    import stable_audio_tools as S
    from stable_audio_tools import _config, model_configs as MC
    from stable_audio_tools.models import _init
    assert conftest.PACKAGE_DEFAULT_GEMM_DTYPE == "fp16"
    from util import SUITE
    assert S.default_gemm_dtype() == SUITE.gemm_dtype      # what conftest selected for this run (fp16 unless SAT_TEST_DTYPE=bf16)
    prev = S.set_default_gemm_dtype("fp16")
    try:
        with _init.skip_init():
            model = S.create_model_from_config(MC.reduced(MC.stable_audio_open_1_0()))
        assert model.model.model.gemm_dtype == "fp16"
        assert model.pretransform.model.encoder.gemm_dtype == "fp16" and model.pretransform.model.decoder.gemm_dtype == "fp16"
        model.model.model.set_gemm_dtype("bf16")
        model.pretransform.model.set_gemm_dtype("bf16")
        assert model.model.model.gemm_dtype == "bf16" and model.pretransform.model.decoder.gemm_dtype == "bf16"
    finally:
        _config.set_default_gemm_dtype(prev)
    with pytest.raises(ValueError):
        S.set_default_gemm_dtype("fp8")
