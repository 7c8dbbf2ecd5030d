"""Generates tests/golden/*.npz by running the REFERENCE (/root/reference, imported with the
placeholder modules of _ref_import.py) on seed-defined weights and inputs.

Runs only in the build container (the reference does not travel).  Usage:
    python tests/golden/make_golden.py [ops] [small] [adaln] [vae] [full] [full_long] [host]
Stored: reference OUTPUTS only (fp32 .npz); weights/inputs are regenerated from seeds by
``stable_audio_tools.synthetic`` (see cases.py).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import as R  # noqa: E402
import cases  # noqa: E402
from stable_audio_tools import synthetic  # noqa: E402

R.import_reference()
rdit = R.ref("models.dit")
rtr = R.ref("models.transformer")
rae = R.ref("models.autoencoders")
rblocks = R.ref("models.blocks")
rbott = R.ref("models.bottleneck")
rcond = R.ref("models.conditioners")
rdiff = R.ref("models.diffusion")
raudio = R.ref("utils.audio_utils")
rgen = R.ref("inference.generation")


def save(name, **arrays):
    path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **{k: v.detach().cpu().numpy() for k, v in arrays.items()})
    print(f"wrote {path}: {os.path.getsize(path)/1024:.0f} KiB  " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in arrays.items()))


def load_synth(module, seed):
    sd = synthetic.synth_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return module.eval()


@torch.no_grad()
def gen_ops():
    """Per-op outputs of the reference's own modules inside the reduced DiT / VAE."""
    m = load_synth(rdit.DiffusionTransformer(**cases.SMALL_DIT), 0)
    blk = m.transformer.layers[1]
    x = synthetic.synth_input("h", (2, 77, 256), 100)
    ctx = synthetic.synth_input("ctx", (2, 130, 128), 101)
    freqs = m.transformer.rotary_pos_emb.forward_from_seq_len(77)
    out = {}
    out["layernorm"] = blk.pre_norm(x)
    q = synthetic.synth_input("q", (2, 4, 77, 64), 102)
    out["rope_freqs"] = freqs[0]
    out["rope_q"] = rtr.apply_rotary_pos_emb(q, freqs[0])
    out["self_attn"] = blk.self_attn(x, rotary_pos_emb=freqs)
    out["cross_attn"] = blk.cross_attn(x, context=ctx)
    out["ff"] = blk.ff(x)
    out["block"] = blk(x, context=ctx, rotary_pos_emb=freqs)
    t = torch.tensor([0.13, 0.77])
    out["timestep_embed"] = m.to_timestep_embed(m.timestep_features(t[:, None]))
    nc = load_synth(rcond.NumberConditioner(768, min_val=0, max_val=512), 1)
    emb, mask = nc([0.0, 47.5, 600.0, -3.0])
    out["number_cond"] = emb
    out["number_mask"] = mask
    # codec ops
    sn = load_synth(rblocks.SnakeBeta(16), 2)
    xs = synthetic.synth_input("snake_x", (2, 16, 50), 103, 2.0)
    out["snake"] = sn(xs)
    for dil in (1, 3, 9):
        ru = load_synth(rae.ResidualUnit(16, 16, dil, use_snake=True), 3 + dil)
        out[f"resunit_d{dil}"] = ru(synthetic.synth_input("ru_x", (2, 16, 64), 104))
    for s in (2, 4, 8):
        db = load_synth(rae.DecoderBlock(32, 16, s, use_snake=True), 20 + s)
        out[f"decblock_s{s}"] = db(synthetic.synth_input("db_x", (1, 32, 11), 105))
        eb = load_synth(rae.EncoderBlock(16, 32, s, use_snake=True), 30 + s)
        out[f"encblock_s{s}"] = eb(synthetic.synth_input("eb_x", (1, 16, 16 * s), 106))
    ms = synthetic.synth_input("ms", (2, 8, 10), 107)
    torch.manual_seed(1234)
    out["vae_sample_seed1234"] = rbott.vae_sample(*ms.chunk(2, dim=1))[0]
    a = synthetic.synth_input("i16", (2, 4099), 108, 0.7)
    out["int16_quiet"] = raudio.float_to_int16_audio(a).float()
    out["int16_loud"] = raudio.float_to_int16_audio(a * 4).float()
    out["int16_max"] = raudio.float_to_int16_audio(a, maximize=True).float()
    save("ops", **out)


@torch.no_grad()
def gen_small():
    """Reduced DiT: DiffusionTransformer.forward with CFG 1 / 7 / 7+scale_phi."""
    m = load_synth(rdit.DiffusionTransformer(**cases.SMALL_DIT), 0)
    out = {}
    for t_len in (64, 77):
        x, t, c, g = cases.dit_inputs(2, t_len, 128, 96, 1)
        out[f"cfg1_T{t_len}"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=1.0)
        out[f"cfg7_T{t_len}"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=7.0, cross_attn_cond_mask=torch.ones(2, 130))
    x, t, c, g = cases.dit_inputs(2, 64, 128, 96, 1)
    out["cfg7_phi04_T64"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=7.0, scale_phi=0.4)
    out["zero_ctx_T64"] = m(x, t, cross_attn_cond=torch.zeros_like(c), global_embed=g, cfg_scale=1.0)
    # negative prompt (dit.py:294-300): the unconditional half attends to a second context, masked tokens fall back to the null embed
    c_neg = synthetic.synth_input("c_neg", tuple(c.shape), 77)
    neg_mask = torch.ones(c.shape[0], c.shape[1])
    neg_mask[1, 40:] = 0
    out["cfg7_negative_T64"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=7.0, negative_cross_attn_cond=c_neg)
    out["cfg7_negative_masked_T64"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=7.0, negative_cross_attn_cond=c_neg,
                                        negative_cross_attn_mask=neg_mask)
    # DiTWrapper path incl. the 0.5 scaling convention is exercised through get_conditioning_inputs below
    save("dit_small", **out)


@torch.no_grad()
def gen_vae():
    out = {}
    dec = load_synth(rae.OobleckDecoder(**cases.vae_kwargs(cases.SMALL_VAE, True)), 5)
    enc = load_synth(rae.OobleckEncoder(**cases.vae_kwargs(cases.SMALL_VAE, False)), 6)
    z = synthetic.synth_input("z", (2, 64, 9), 7)
    a = synthetic.synth_input("a", (2, 2, 2048 * 5), 8, 0.3)
    out["small_decode"] = dec(z)
    out["small_encode"] = enc(a)
    # BASELINE config 1: full-size decoder, z[1,64,43] -> [1,2,88064]
    t0 = time.time()
    decf = load_synth(rae.OobleckDecoder(**cases.vae_kwargs(cases.FULL_VAE, True)), 0)
    zf = synthetic.synth_input("z_full", (1, 64, 43), 1)
    out["full_decode_T43"] = decf(zf)
    encf = load_synth(rae.OobleckEncoder(**cases.vae_kwargs(cases.FULL_VAE, False)), 0)
    af = synthetic.synth_input("a_full", (1, 2, 2048 * 16), 2, 0.3)
    out["full_encode_T16"] = encf(af)
    print(f"full-size codec goldens in {time.time()-t0:.1f}s")
    save("vae", **out)
    # chunked reconstruct / encode / decode of AudioAutoencoder with the reference's own RNG order
    ae = rae.AudioAutoencoder(enc, dec, latent_dim=64, downsampling_ratio=2048, sample_rate=44100, io_channels=2,
                              bottleneck=rbott.VAEBottleneck())
    sig = synthetic.synth_input("sig", (1, 2, 2048 * 11 + 700), 9, 0.3)[..., : 2048 * 11]
    out2 = {}
    torch.manual_seed(77)
    out2["reconstruct_chunked"] = ae.reconstruct_audio(sig, chunked=True, chunk_size=4, overlap=1, max_batch_size=3)
    torch.manual_seed(78)
    out2["encode_chunked"] = ae.encode_audio(sig, chunked=True, chunk_size=4, overlap=1, max_batch_size=2)
    zz = synthetic.synth_input("zz", (1, 64, 11), 10)
    out2["decode_chunked"] = ae.decode_audio(zz, chunked=True, chunk_size=4, overlap=1, max_batch_size=2)
    out2["decode_unchunked"] = ae.decode_audio(zz, chunked=False)
    save("vae_chunked", **out2)


@torch.no_grad()
def gen_full(long=False):
    """Full-size SA-Open DiT (1.06 B parameters, seed 0): one _forward without CFG (CFG just doubles the batch)."""
    t0 = time.time()
    m = load_synth(rdit.DiffusionTransformer(**cases.FULL_DIT), 0)
    print(f"full DiT built in {time.time()-t0:.1f}s")
    t_len = 6144 if long else 1024
    x, t, c, g = cases.dit_inputs(1, t_len, 768, 1536, 1)
    t0 = time.time()
    y = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=1.0)
    print(f"full forward T={t_len} in {time.time()-t0:.1f}s, out std {y.std():.4f}")
    out = {"out": y}
    if not long:     # BASELINE config 2/3 arithmetic: the same prompt with batched CFG 7 (dit.py:270-345)
        t0 = time.time()
        out["cfg7"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=7.0)
        print(f"full forward T={t_len} CFG 7 in {time.time()-t0:.1f}s, out std {out['cfg7'].std():.4f}")
    save("dit_full_T%d" % t_len, **out)


@torch.no_grad()
def gen_host():
    """Host-side bookkeeping of the reference: get_conditioning_inputs, prepare_audio, build_mask."""
    import json
    cfg = json.load(open(os.path.join(R.REFERENCE_ROOT, "stable_audio_tools/configs/model_configs/txt2audio/stable_audio_open_1_0.json")))
    cfg["model"]["conditioning"]["configs"] = [c for c in cfg["model"]["conditioning"]["configs"] if c["type"] != "t5"]
    cfg["model"]["diffusion"]["config"].update(depth=1, embed_dim=128, num_heads=2)
    cfg["model"]["pretransform"]["config"]["encoder"]["config"]["channels"] = 8
    cfg["model"]["pretransform"]["config"]["decoder"]["config"]["channels"] = 8
    model = R.ref("models.factory").create_model_from_config(cfg)
    sd = synthetic.synth_state_dict(model.state_dict(), 4)
    model.load_state_dict(sd)
    cond = model.conditioner([{"seconds_start": 0, "seconds_total": 47}, {"seconds_start": 3.5, "seconds_total": 700}])
    cond["prompt"] = [synthetic.synth_input("prompt", (2, 128, 768), 5), torch.ones(2, 128)]
    cond = {k: cond[k] for k in ("prompt", "seconds_start", "seconds_total")}
    ci = model.get_conditioning_inputs(cond)
    out = {"cross_attn_cond": ci["cross_attn_cond"], "cross_attn_mask": ci["cross_attn_mask"], "global_cond": ci["global_cond"]}
    pa = R.ref("inference.utils").prepare_audio
    a = synthetic.synth_input("pa", (1, 1000), 6)
    out["prepare_mono_to_stereo_pad"] = pa(a, 44100, 44100, 1500, 2, "cpu")
    out["prepare_crop"] = pa(synthetic.synth_input("pa3", (3, 1000), 7), 44100, 44100, 600, 2, "cpu")
    for i, ma in enumerate(cases.MASK_ARGS):
        out[f"mask_{i}"] = rgen.build_mask(1024, ma)
    save("host", **out)


@torch.no_grad()
def gen_adaln():
    """Reduced DiT with global_cond_type='adaLN' (dit.py:205-206, transformer.py:665-689): no prepend token, per-layer
    scale/shift/gate from the global embedding.  The reference zero-initialises to_scale_shift_gate; the synthetic state
    dict re-draws it so that the modulation is exercised."""
    m = load_synth(rdit.DiffusionTransformer(**cases.SMALL_DIT, global_cond_type="adaLN"), 0)
    assert m.transformer.layers[1].to_scale_shift_gate[1].weight.abs().max() > 0
    out = {}
    for t_len in (64, 77):
        x, t, c, g = cases.dit_inputs(2, t_len, 128, 96, 1)
        out[f"cfg1_T{t_len}"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=1.0)
    x, t, c, g = cases.dit_inputs(2, 77, 128, 96, 1)
    out["cfg7_T77"] = m(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=7.0)
    out["noglobal_T77"] = m(x, t, cross_attn_cond=c, cfg_scale=1.0)
    save("dit_adaln_small", **out)


class _DrawRecorder:
    """Records, in order, every Gaussian draw the reference makes during one call (torch.randn for the initial noise,
    torch.randn_like for the VAE bottleneck and the inpainting callback) and every per-step draw of the stand-in noise sampler."""

    def __init__(self):
        self.randn, self.randn_like, self.step = [], [], []
        self._orig = (torch.randn, torch.randn_like)

    def __enter__(self):
        orig_randn, orig_like = self._orig

        def randn(*a, **k):
            t = orig_randn(*a, **k)
            self.randn.append(t.clone())
            return t

        def randn_like(*a, **k):
            t = orig_like(*a, **k)
            self.randn_like.append(t.clone())
            return t

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._orig

    def step_noise(self, x):
        t = self._orig[1](x)                 # i.i.d. unit Gaussian from the global generator (k-diffusion: BrownianTree, torchsde)
        self.step.append(t.clone())
        return t


def _install_kdiffusion_standin(rec):
    """k-diffusion 0.1.1 is not installed.  The REFERENCE's own generate_diffusion_cond / sample_k
    (generation.py:95-261, sampling.py:144-228) are run on top of a stand-in ``k_diffusion`` whose entry points are the
    restated algorithms of oracle/sampler.py with k-diffusion's call signatures.  What these goldens pin is therefore the
    reference-resident half of the path: RNG order, sample_size // ratio, sigma_max <- init_noise_level, cut & paste,
    build_mask, init_data / mask mixing, the in-place inpainting callback, the DiTWrapper / VDenoiser plumbing, decode."""
    from oracle import sampler as osamp
    K = sys.modules["k_diffusion"]

    class VDenoiser(torch.nn.Module):
        def __init__(self, inner_model):
            super().__init__()
            self.inner_model = inner_model
            self.sigma_data = 1.0

        def forward(self, x, sigma, **kwargs):
            return osamp.vdenoise(lambda xin, t: self.inner_model(xin, t, **kwargs), x, sigma)

    K.external.VDenoiser = VDenoiser
    K.utils.append_dims = lambda x, n: x[(...,) + (None,) * (n - x.ndim)]
    K.sampling.get_sigmas_polyexponential = lambda n, smin, smax, rho=1.0, device="cpu": \
        osamp.get_sigmas_polyexponential(n, smin, smax, rho).to(device)

    def wrap(fn, takes_noise):
        def sampler(model, x, sigmas, extra_args=None, callback=None, disable=None, **kw):
            den = lambda xx, sg: model(xx, sg, **(extra_args or {}))
            if takes_noise:
                return fn(den, x, sigmas, lambda i, a, b: rec().step_noise(x), callback=callback, **kw)
            return fn(den, x, sigmas, callback=callback, **kw)
        return sampler

    K.sampling.sample_dpmpp_3m_sde = wrap(osamp.sample_dpmpp_3m_sde, True)
    K.sampling.sample_dpmpp_2m_sde = wrap(osamp.sample_dpmpp_2m_sde, True)
    K.sampling.sample_dpmpp_2s_ancestral = wrap(osamp.sample_dpmpp_2s_ancestral, True)
    K.sampling.sample_heun = wrap(osamp.sample_heun, False)
    K.sampling.sample_lms = wrap(osamp.sample_lms, False)
    K.sampling.sample_dpm_2 = wrap(osamp.sample_dpm_2, False)
    K.sampling.sample_dpm_fast = lambda model, x, smin, smax, n, extra_args=None, callback=None, disable=None: \
        osamp.sample_dpm_fast(lambda xx, sg: model(xx, sg, **(extra_args or {})), x, smin, smax, n, callback=callback)
    K.sampling.sample_dpm_adaptive = lambda model, x, smin, smax, rtol=0.05, atol=0.0078, extra_args=None, callback=None, disable=None: \
        osamp.sample_dpm_adaptive(lambda xx, sg: model(xx, sg, **(extra_args or {})), x, smin, smax, rtol=rtol, atol=atol, callback=callback)


def gen_generate():
    """The reference's generate_diffusion_cond and sample_k on the reduced SA-Open model (CPU, fp32; torch.cuda.amp.autocast
    disables itself without CUDA), with every Gaussian draw recorded so that the oracle and the product can replay them."""
    from stable_audio_tools import model_configs as MC
    holder = {}
    _install_kdiffusion_standin(lambda: holder["rec"])
    cfg = MC.reduced(MC.stable_audio_open_1_0())
    model = R.ref("models.factory").create_model_from_config(cfg)
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 0))
    model.eval()
    dc = cfg["model"]["diffusion"]["config"]
    ratio = cfg["model"]["pretransform"]["config"]["downsampling_ratio"]
    b, t_len = 2, cases.GEN["t_len"]
    cond = model.conditioner([{"seconds_start": 0, "seconds_total": 10 + i} for i in range(b)])
    cond["prompt"] = [synthetic.synth_input("prompt", (b, 128, dc["cond_token_dim"]), 31), torch.ones(b, 128)]
    cond = {k: cond[k] for k in ("prompt", "seconds_start", "seconds_total")}
    init = synthetic.synth_input("init", (2, t_len * ratio - 100), 70, 0.3)
    out = {}

    def run(name, **kw):
        for latents in (True, False):
            with _DrawRecorder() as rec:
                holder["rec"] = rec
                y = rgen.generate_diffusion_cond(model, conditioning_tensors=dict(cond), sample_size=t_len * ratio, device="cpu",
                                                 return_latents=latents, **kw)
            key = "latents" if latents else "audio"
            out[f"{name}.{key}"] = y
            if latents:
                assert len(rec.randn) == 1
                out[f"{name}.noise"] = rec.randn[0]
                for i, t in enumerate(rec.randn_like):
                    out[f"{name}.randn_like{i}"] = t
                for i, t in enumerate(rec.step):
                    out[f"{name}.step{i}"] = t
                print(name, "draws: randn_like", len(rec.randn_like), "step", len(rec.step), "latents", tuple(y.shape), float(y.std()))

    for name, kw in cases.GEN["calls"].items():
        kw = dict(kw)
        if kw.pop("init_audio", False):
            kw["init_audio"] = (44100, init)
        run(name, **kw)
    # sample_k directly (sampling.py:144-228): inpainting under a single-step sampler whose derivative is formed before the
    # callback (k-heun), and the model-evaluation count / callback indices of plain sampling
    rsamp = R.ref("inference.sampling")
    ci = model.get_conditioning_inputs(cond)
    for name, kw in cases.GEN["sample_k"].items():
        noise = synthetic.synth_input("noise_" + name, (b, 64, t_len), 63)
        init_lat = synthetic.synth_input("init_" + name, (b, 64, t_len), 64)
        mask = rgen.build_mask(t_len, cases.GEN["sample_k_mask"]) if kw.get("mask") else None
        args = {k: v for k, v in kw.items() if k not in ("mask", "init")}
        seen = []
        with _DrawRecorder() as rec:
            holder["rec"] = rec
            y = rsamp.sample_k(model.model, noise, init_lat if kw.get("init") else None, mask, device="cpu", cfg_scale=7.0, batch_cfg=True,
                               rescale_cfg=True, callback=lambda a: seen.append(int(a["i"])), **args, **ci)
        out[f"sample_k.{name}.out"] = y
        out[f"sample_k.{name}.callback_i"] = torch.tensor(seen)
        for i, t in enumerate(rec.randn_like):
            out[f"sample_k.{name}.randn_like{i}"] = t
        for i, t in enumerate(rec.step):
            out[f"sample_k.{name}.step{i}"] = t
        print("sample_k", name, "draws: randn_like", len(rec.randn_like), "step", len(rec.step), "callbacks", seen)
    save("generate", **out)


def gen_keys():
    """State-dict keys + shapes of the reference's SA-Open-1.0 model (T5 entry removed) and of the VAE config:
    the checkpoint-compatibility contract."""
    import json
    base = os.path.join(R.REFERENCE_ROOT, "stable_audio_tools/configs/model_configs")
    cfg = json.load(open(os.path.join(base, "txt2audio/stable_audio_open_1_0.json")))
    cfg["model"]["conditioning"]["configs"] = [c for c in cfg["model"]["conditioning"]["configs"] if c["type"] != "t5"]
    m = R.ref("models.factory").create_model_from_config(cfg)
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    vcfg = json.load(open(os.path.join(base, "autoencoders/stable_audio_2_0_vae.json")))
    v = R.ref("models.factory").create_model_from_config(vcfg)
    vkeys = {k: list(t.shape) for k, t in v.state_dict().items()}
    info = {"sa_open_1_0": keys, "vae": vkeys,
            "attrs": {"io_channels": m.io_channels, "sample_rate": m.sample_rate, "min_input_length": m.min_input_length,
                      "diffusion_objective": m.diffusion_objective, "pretransform_ratio": m.pretransform.downsampling_ratio,
                      "vae_latent_dim": v.latent_dim, "vae_ratio": v.downsampling_ratio, "vae_min_length": v.min_length}}
    path = os.path.join(cases.GOLDEN_DIR, "state_dict_keys.json")
    json.dump(info, open(path, "w"))
    print("wrote", path, len(keys), len(vkeys))


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "small", "vae", "host", "full"]
    torch.set_num_threads(os.cpu_count())
    if "ops" in which:
        gen_ops()
    if "small" in which:
        gen_small()
    if "vae" in which:
        gen_vae()
    if "host" in which:
        gen_host()
    if "adaln" in which:
        gen_adaln()
    if "keys" in which:
        gen_keys()
    if "generate" in which:
        gen_generate()
    if "full" in which:
        gen_full(False)
    if "full_long" in which:
        gen_full(True)
