"""the T5 text encoder of the conditioner on the device.

The reference's ``T5Conditioner.forward`` (models/conditioners.py:317-343) is tokenizer -> ``transformers.T5EncoderModel`` ->
``proj_out`` -> ``* attention_mask``.  The encoder is a third-party dependency (``transformers``, installed in this image, weights
not); the HIP stack (``sat_t5_*``, csrc/t5_encoder.hip) is compared with that very class, randomly initialised on the CPU in
fp32: same state-dict keys in, ``last_hidden_state`` out.  Tolerance 2e-5 rel-L2: both sides are fp32, only the summation order
differs (the reference itself runs the encoder under fp16 autocast, i.e. ~1e-3).
"""
import ctypes

import pytest
import torch

from util import assert_close

T5_CONFIGS = {
    # t5-base geometry, half the depth (CPU time); "relu" feed-forward, 32 buckets / distance 128
    "t5": dict(vocab_size=1000, d_model=768, d_kv=64, d_ff=3072, num_layers=6, num_heads=12, feed_forward_proj="relu"),
    # flan-t5 style: gated GELU, inner dim != d_model (6 heads x 64 = 384 vs 512), other bucket parameters
    "flan": dict(vocab_size=777, d_model=512, d_kv=64, d_ff=1024, num_layers=3, num_heads=6, feed_forward_proj="gated-gelu",
                 relative_attention_num_buckets=16, relative_attention_max_distance=64),
}


def _hf_encoder(name, seed):
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(seed)
    model = T5EncoderModel(T5Config(**T5_CONFIGS[name])).eval()
    with torch.no_grad():        # the default init leaves every T5LayerNorm weight at 1 and the bias table tiny: make them matter
        for k, p in model.named_parameters():
            if k.endswith("layer_norm.weight"):
                p.copy_(0.7 + 0.6 * torch.rand_like(p))
            if "relative_attention_bias" in k:
                p.copy_(torch.randn_like(p))
    return model


@pytest.mark.parametrize("num_buckets,max_distance", [(32, 128), (16, 64), (8, 20)])
@pytest.mark.parametrize("length", [1, 7, 128, 512])
def test_relative_position_buckets_match_transformers(num_buckets, max_distance, length):
    """Host helper of the C ABI against transformers' T5Attention._relative_position_bucket for every key - query offset."""
    from stable_audio_tools import _hip
    from transformers.models.t5.modeling_t5 import T5Attention
    out = (ctypes.c_int32 * (2 * length - 1))()
    _hip.check(_hip.lib().sat_t5_relative_buckets(length, num_buckets, max_distance, out))
    delta = torch.arange(-(length - 1), length)                      # memory_position - context_position
    want = T5Attention._relative_position_bucket(delta, bidirectional=True, num_buckets=num_buckets, max_distance=max_distance)
    assert list(out) == want.tolist()


def test_t5_conditioner_module_contract():
    """Same constructor, dims table and (empty) state dict as the reference class; no silent CPU path."""
    from stable_audio_tools import _hip
    from stable_audio_tools.models.conditioners import T5Conditioner
    c = T5Conditioner(768, "t5-base", max_length=128)
    assert (c.dim, c.output_dim, c.max_length) == (768, 768, 128) and list(c.state_dict()) == []
    p = T5Conditioner(1536, "google/flan-t5-large", project_out=True)
    assert sorted(p.state_dict()) == ["proj_out.bias", "proj_out.weight"] and p.proj_out.weight.shape == (1536, 1024)
    with pytest.raises(ValueError):
        T5Conditioner(768, "t5-huge")
    model = _hf_encoder("flan", 0)
    with pytest.raises(ValueError):          # d_model of the weights must match the named model
        c.load_encoder(model.state_dict(), model.config)
    c2 = T5Conditioner(512, "t5-small").load_encoder(model.state_dict(), model.config)
    with pytest.raises(_hip.SatError):       # not on a HIP device
        c2.encode_ids(torch.zeros(1, 4, dtype=torch.long), torch.ones(1, 4, dtype=torch.long))
    if not T5Conditioner.cached_locally("t5-base"):
        with pytest.raises(_hip.SatError, match="local Hugging Face cache"):
            c.set_device("cpu")
            c.forward(["a prompt"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,length", [("t5", 128), ("t5", 19), ("flan", 64)])
def test_t5_encoder_vs_transformers(dev, name, length):
    from stable_audio_tools.models.conditioners import T5Conditioner
    model = _hf_encoder(name, 3)
    cfg = model.config
    g = torch.Generator().manual_seed(5)
    b = 3
    ids = torch.randint(0, cfg.vocab_size, (b, length), generator=g)
    mask = torch.ones(b, length, dtype=torch.long)
    mask[1, max(1, length // 3):] = 0                      # padded prompts: the tokenizer pads with id 0, mask 0
    mask[2, 1:] = 0
    ids = ids * mask
    with torch.no_grad():
        want = model(input_ids=ids, attention_mask=mask.bool())["last_hidden_state"]
    cond = T5Conditioner(cfg.d_model, {768: "t5-base", 512: "t5-small"}[cfg.d_model]).load_encoder(model.state_dict(), cfg)
    cond.set_device(dev)
    got, got_mask = cond.encode_ids(ids, mask)
    assert got_mask.dtype == torch.bool and torch.equal(got_mask.cpu(), mask.bool())
    assert_close(f"T5 encoder ({name}, L={length}) x mask", got, want * mask.unsqueeze(-1).float(), 2e-5)
    assert (got.cpu()[mask == 0] == 0).all(), "padding rows must be exactly zero (conditioners.py:341)"


@pytest.mark.gpu
def test_t5_conditioner_forward_with_projection_and_multiconditioner(dev):
    """The whole conditioner call -- tokenizer interface, proj_out, mask -- inside a MultiConditioner next to the number embedders."""
    from stable_audio_tools.models.conditioners import MultiConditioner, NumberConditioner, T5Conditioner
    model = _hf_encoder("flan", 9)
    cfg = model.config

    class WordTokenizer:          # the transformers tokenizer interface the conditioner uses: ids from a word hash, EOS = 1, pad = 0
        def __call__(self, texts, truncation, max_length, padding, return_tensors):
            assert truncation and padding == "max_length" and return_tensors == "pt"
            ids = torch.zeros(len(texts), max_length, dtype=torch.long)
            mask = torch.zeros_like(ids)
            for n, text in enumerate(texts):
                toks = [2 + sum(map(ord, w)) % (cfg.vocab_size - 2) for w in text.split()][: max_length - 1] + [1]
                ids[n, : len(toks)] = torch.tensor(toks)
                mask[n, : len(toks)] = 1
            return {"input_ids": ids, "attention_mask": mask}

    torch.manual_seed(1)
    cond = T5Conditioner(192, "t5-small", max_length=24, project_out=True).load_encoder(model.state_dict(), cfg, WordTokenizer())
    multi = MultiConditioner({"prompt": cond, "seconds_total": NumberConditioner(192, 0, 512)})
    multi.to(dev)
    multi.set_device(dev)
    texts = ["Amen break 174 BPM", "warm analog pad with a slow filter sweep and tape hiss", ""]
    out = multi([{"prompt": t, "seconds_total": 30.0 + n} for n, t in enumerate(texts)])
    emb, mask = out["prompt"]
    enc = WordTokenizer()(texts, True, 24, "max_length", "pt")
    with torch.no_grad():
        hidden = model(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"].bool())["last_hidden_state"]
        want = torch.nn.functional.linear(hidden, cond.proj_out.weight.cpu(), cond.proj_out.bias.cpu()) * enc["attention_mask"].unsqueeze(-1).float()
    assert emb.shape == (3, 24, 192) and torch.equal(mask.cpu(), enc["attention_mask"].bool())
    assert_close("T5Conditioner.forward", emb, want, 2e-5)
    assert out["seconds_total"][0].shape == (3, 1, 192)
    # new proj_out weights (a checkpoint load) must reach the plan
    with torch.no_grad():
        cond.proj_out.weight.mul_(0.5)
    emb2, _ = cond(texts)
    assert_close("after a weight update", emb2, torch.nn.functional.linear(hidden, cond.proj_out.weight.cpu(), cond.proj_out.bias.cpu())
                 * enc["attention_mask"].unsqueeze(-1).float(), 2e-5)


@pytest.mark.gpu
def test_text_to_audio_through_generate_diffusion_cond(dev):
    """Prompts as TEXT through the public entry (generation.py:95-261 with ``conditioning=``): MultiConditioner -> T5 encoder on the
    device -> cross-attention context + mask -> sampler -> decoder, against the same call fed with pre-computed tensors."""
    import stable_audio_tools as S
    from stable_audio_tools import model_configs as MC, synthetic
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.conditioners import T5Conditioner
    cfg = MC.reduced(MC.stable_audio_open_1_0(with_text_encoder=True))       # conditioning config with the reference's "t5" entry
    with _init.skip_init():
        model = S.create_model_from_config(cfg)
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 5))
    # t5-base is not in this machine's Hugging Face cache: the factory lists the id as external instead of instantiating the encoder
    assert "prompt" in model.conditioner.external_ids or "prompt" in model.conditioner.conditioners
    hf = _hf_encoder("flan", 13)
    hcfg = hf.config

    class WordTokenizer:
        def __call__(self, texts, truncation, max_length, padding, return_tensors):
            ids = torch.zeros(len(texts), max_length, dtype=torch.long)
            mask = torch.zeros_like(ids)
            for n, text in enumerate(texts):
                toks = [2 + sum(map(ord, w)) % (hcfg.vocab_size - 2) for w in text.split()][: max_length - 1] + [1]
                ids[n, : len(toks)] = torch.tensor(toks)
                mask[n, : len(toks)] = 1
            return {"input_ids": ids, "attention_mask": mask}

    torch.manual_seed(2)
    cond_dim = cfg["model"]["conditioning"]["cond_dim"]
    t5 = T5Conditioner(cond_dim, "t5-small", max_length=16, project_out=True).load_encoder(hf.state_dict(), hcfg, WordTokenizer())
    model.conditioner.conditioners["prompt"] = t5              # what create_multi_conditioner does when the weights are cached locally
    model = model.to(dev).eval()
    meta = [{"prompt": "dry kick drum one shot", "seconds_start": 0, "seconds_total": 0.03},
            {"prompt": "a long evolving pad with shimmer", "seconds_start": 0, "seconds_total": 0.04}]
    kw = dict(steps=4, cfg_scale=6.0, sample_size=cfg["sample_size"], sigma_min=0.3, sigma_max=500, sampler_type="dpmpp-3m-sde",
              device=str(dev), seed=11)
    from_text = generate_diffusion_cond(model, conditioning=meta, **kw)
    tensors = model.conditioner(meta)
    emb, mask = tensors["prompt"]
    assert emb.shape == (2, 16, cond_dim) and mask.dtype == torch.bool and mask.sum().item() == (5 + 1) + (6 + 1)      # words + EOS
    from_tensors = generate_diffusion_cond(model, conditioning_tensors=tensors, **kw)
    assert torch.isfinite(from_text).all() and from_text.shape == (2, 2, cfg["sample_size"])
    assert torch.equal(from_text, from_tensors)
    # and the text matters: another prompt, same seed -> different audio
    meta[0]["prompt"] = "bright bell"
    assert not torch.equal(generate_diffusion_cond(model, conditioning=meta, **kw)[0], from_text[0])
