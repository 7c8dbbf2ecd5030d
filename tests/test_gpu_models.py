"""Model-level parity on the GPU: the product modules (HIP, through the C ABI) vs the oracle on the
same synthetic state dict and seeded inputs, at sizes the oracle finishes in seconds.

Tolerances (bf16 GEMM operands, fp32 accumulate / residual / LN / softmax):
  * vs the MATCHED-ROUNDING oracle (same bf16 rounding points): rel-L2 <= 3e-3 for the DiT -- what
    remains is accumulation order and P rounded against a running instead of the final max; <= 1e-2
    for the ~35-conv-deep codec, where a 1-ulp fp32 difference before a bf16 store flips the stored
    value (2^-8 relative) for a small fraction of elements at every layer, plus __sinf;
  * vs the pure-fp32 oracle / the reference's own fp32 outputs: gated at ~2x the measured value
    (round-1 GPU log: reduced DiT 1.1e-3 -> 2.5e-3, CFG 7 5.4e-3 -> 1.2e-2, full-size DiT 3.7e-3 ->
    8e-3, codec 5.6..7.7e-3 -> 1.5e-2), so that a regression of the bf16 path cannot hide behind
    the gate; the fp32-class mode (gemm_dtype="fp32x") is gated at north_star's 1e-3.
"""
import pytest
import torch

from util import SUITE, assert_close, bf16_round, fp16_round, rel_l2

T = SUITE.tol          # a bf16 gate -> the gate of the suite's operand format (fp16: / 4)

pytestmark = pytest.mark.gpu


def _build(cfg, seed, dev):
    import stable_audio_tools as S
    from stable_audio_tools import synthetic
    from stable_audio_tools.models import _init
    with _init.skip_init():
        model = S.create_model_from_config(cfg)
    sd = synthetic.synth_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd)
    return model.to(dev).eval(), sd


def _matched():
    """Rounding hook of the default DiT plan for bf16 / "prepend" models: bf16 operands, LayerNorms folded into the GEMMs
    (sat_dit_cfg.ln_fold; oracle/dit.py LnFoldRounding).  With ``set_layernorm_fusion(False)`` the hook is plain ``bf16_round``."""
    from oracle import dit as odit
    return odit.LnFoldRoundingF16() if SUITE.f16 else odit.LnFoldRounding()


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


@pytest.fixture(scope="module")
def small_dit(dev):
    from stable_audio_tools import model_configs as MC
    cfg = MC.reduced(MC.stable_audio_open_1_0())
    model, sd = _build(cfg, 0, dev)
    return cfg, model, sd


def _inputs(b, t_len, cond_dim, seed=1):
    from stable_audio_tools import synthetic
    x = synthetic.synth_input("x", (b, 64, t_len), seed)
    c = synthetic.synth_input("c", (b, 130, cond_dim), seed + 1)
    g = synthetic.synth_input("g", (b, 2 * cond_dim), seed + 2)
    return x, c, g


def test_cross_attention_fusion_on_off(dev, small_dit):
    """to_q + cross-attention as ONE launch (the plan's default where the projection's 128 x 64 tiles fit one round) against the same plan
    with two kernels (sat_dit_cfg.cross_attention = 1, per plan).  Both keep Q pre-scaled with one bf16 rounding and run the same per-wave step on
    the same K / V^T tiles (attn_core.h).  One sequence: a wave holds the same 32 queries either way, so the outputs are bit-identical.
    Three sequences of 78 rows: the fused tiles straddle sequences (second pass on the next sequence's keys) and group the queries
    into waves differently -- the wave-wide "redo this block" decision may differ, the results agree to rounding."""
    from stable_audio_tools import _hip
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    for b in (1, 3):
        x, c, g = _inputs(b, 77, dc["cond_token_dim"], seed=7)
        t = torch.tensor([0.31, 0.87, 0.5][:b])
        run = lambda: model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=1.0).cpu()
        fused = run()
        try:
            model.model.model.set_cross_attention_fusion(False)
            separate = run()
        finally:
            model.model.model.set_cross_attention_fusion(True)
        assert torch.isfinite(fused).all()
        assert torch.equal(fused, run()), "the fused path is not repeatable"
        if b == 1:
            assert torch.equal(fused, separate), f"fused vs separate cross-attention: max abs diff {(fused - separate).abs().max().item():.3e}"
        else:
            assert_close("fused vs separate cross-attention, 3 sequences", fused, separate, T(2e-3))


def test_residual_stream_report(dev, small_dit):
    """sat_dit_debug (round 6): the per-block statistics of the fp32 residual stream a forward leaves behind -- checked against the same quantities
    of the oracle's residual stream for the LAST update (FF-out of the last block = the transformer's output rows), and for what the report is for:
    a residual stream pushed beyond the fp16 range is counted, one with a common-mode offset is reported with its ratio."""
    from stable_audio_tools import _hip
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dit = model.model.model
    x, c, g = _inputs(2, 77, dc["cond_token_dim"], seed=3)
    t = torch.tensor([0.31, 0.87])
    run = lambda xs: model.model(xs.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=1.0)
    base = run(x).cpu()
    dit.residual_stream_report(True)
    try:
        same = run(x).cpu()
        rep = dit.residual_stream_report(False)
    finally:
        pass
    assert torch.equal(base, same), "the diagnostics changed the result"
    assert len(rep) == dc["depth"] * 3 and all(r["saturated"] == 0 for r in rep)
    assert all(0.0 < r["max_abs"] < 1e4 and r["crest"] >= 1.0 and r["common_mode"] >= 0.0 for r in rep), rep[:3]
    # an input 3e5 times larger: the input projection is linear, so the first residual rows leave the fp16 range and the report says so
    dit.residual_stream_report(True)
    run(x * 3e5)
    rep_big = dit.residual_stream_report(False)
    assert rep_big[0]["max_abs"] > 65504.0 and rep_big[0]["saturated"] > 0, rep_big[0]
    print(f"\n[residual stream report] first update: {rep[0]}\n  scaled input: {rep_big[0]}")


def test_layernorm_fusion_on_off(dev, small_dit):
    """The standalone-LayerNorm plan (sat_dit_cfg.ln_fold = 0) against ITS matched oracle (plain bf16 rounding points), and the two
    plans against each other: they differ only in where the activation is rounded (before / after the normalisation)."""
    from oracle import dit as odit
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    x, c, g = _inputs(2, 77, dc["cond_token_dim"])
    t = torch.tensor([0.31, 0.87])
    run = lambda: model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=1.0).cpu()
    fused = run()
    try:
        model.model.model.set_layernorm_fusion(False)
        plain = run()
    finally:
        model.model.model.set_layernorm_fusion(True)
    assert not torch.equal(fused, plain), "the switch did not change the plan"
    want_f = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"])
    e_p = assert_close("standalone LayerNorm plan vs matched oracle", plain, odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], rnd=SUITE.round), T(3e-3))
    e_f = assert_close("fused plan vs matched oracle", fused, odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], rnd=_matched()), T(3e-3))
    e_x = assert_close("fused vs standalone plan", fused, plain, T(3e-3))
    print(f"\n[ln_fold] standalone vs matched {e_p:.2e}, fused vs matched {e_f:.2e}, fused vs standalone {e_x:.2e}; "
          f"vs fp32: standalone {rel_l2(plain, want_f):.2e}, fused {rel_l2(fused, want_f):.2e}")
    assert rel_l2(fused, want_f) <= T(2.5e-3) and rel_l2(plain, want_f) <= T(2.5e-3)


@pytest.mark.parametrize("t_len", [64, 77])
def test_dit_forward_no_cfg(dev, small_dit, t_len):
    from oracle import dit as odit
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    x, c, g = _inputs(2, t_len, dc["cond_token_dim"])
    t = torch.tensor([0.31, 0.87])
    got = model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=1.0)
    want_m = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], rnd=_matched())
    want_f = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"])
    e_m = assert_close("dit forward vs matched oracle", got, want_m, T(3e-3))
    e_f = assert_close("dit forward vs fp32 oracle", got, want_f, T(2.5e-3))
    print(f"\n[dit T={t_len}] rel-L2 vs matched {e_m:.2e}, vs fp32 {e_f:.2e}")


def test_dit_forward_cfg_and_denoise(dev, small_dit):
    from oracle import dit as odit, sampler as osamp
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    x, c, g = _inputs(3, 50, dc["cond_token_dim"], seed=7)
    t = torch.tensor([0.5, 0.5, 0.5])
    for phi in (0.0, 0.4):
        got = model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=7.0, scale_phi=phi,
                          cross_attn_mask=torch.ones(3, 130, device=dev))
        want_m = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], cfg_scale=7.0, scale_phi=phi, rnd=_matched())
        want_f = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], cfg_scale=7.0, scale_phi=phi)
        e_m = assert_close(f"dit cfg7 phi={phi} vs matched", got, want_m, T(1e-2))
        e_f = assert_close(f"dit cfg7 phi={phi} vs fp32", got, want_f, T(1.2e-2))
        print(f"\n[dit cfg7 phi={phi}] rel-L2 vs matched {e_m:.2e}, vs fp32 {e_f:.2e}")
    # fused VDenoiser evaluation == oracle vdenoise around the CFG model
    sigma = 3.7
    dit = model.model.model
    dit.prepare_generation(c.to(dev), g.to(dev), 7.0)
    xs = x * sigma
    got = dit.denoise(xs.to(dev), sigma, cfg_scale=7.0)
    fn = lambda xin, tt: odit.dit_forward(dsd, xin, tt, c, g, dc["depth"], dc["num_heads"], cfg_scale=7.0, rnd=_matched())
    want = osamp.vdenoise(fn, xs, torch.full((3,), sigma))
    assert_close("denoise_cfg vs matched oracle", got, want, T(1e-2))
    # uncond half sees an all-zero context: its cross-attention must contribute exactly nothing
    got1 = model.model(x.to(dev), t.to(dev), cross_attn_cond=torch.zeros_like(c).to(dev), global_cond=g.to(dev), cfg_scale=1.0)
    want1 = odit.dit_forward(dsd, x, t, torch.zeros_like(c), g, dc["depth"], dc["num_heads"], rnd=_matched())
    assert_close("zero-context forward", got1, want1, T(3e-3))


def test_negative_prompt_vs_reference_golden(dev):
    """Negative prompts: the unconditional CFG half attends to a second context instead of the null embed
    (dit.py:294-300), optionally token-masked.  DiffusionTransformer.forward against the REFERENCE's outputs
    (tests/golden/dit_small.npz: cfg7_negative*), and generate_diffusion_cond(negative_conditioning_tensors=...) -- which the
    reference's own generate_diffusion_cond cannot run (generation.py:148-155 resets the argument and ends in a KeyError) --
    against the matched-rounding oracle trajectory."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from oracle import dit as odit, sampler as osamp
    from stable_audio_tools import synthetic
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.dit import DiffusionTransformer
    gold = cases.load("dit_small")
    with _init.skip_init():
        dit = DiffusionTransformer(**cases.SMALL_DIT)
    sd = synthetic.synth_state_dict(dit.state_dict(), 0)
    dit.load_state_dict(sd)
    dit = dit.to(dev).eval()
    x, t, c, g = cases.dit_inputs(2, 64, 128, 96, 1)
    c_neg = synthetic.synth_input("c_neg", tuple(c.shape), 77)
    neg_mask = torch.ones(c.shape[0], c.shape[1])
    neg_mask[1, 40:] = 0
    got = dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=7.0, negative_cross_attn_cond=c_neg.to(dev))
    e1 = assert_close("negative prompt vs reference", got, gold["cfg7_negative_T64"], T(1.6e-2))          # measured 7.9e-3
    got = dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=7.0, negative_cross_attn_cond=c_neg.to(dev),
              negative_cross_attn_mask=neg_mask.to(dev))
    e2 = assert_close("masked negative prompt vs reference", got, gold["cfg7_negative_masked_T64"], T(1.6e-2))   # measured 8.2e-3
    want_m = odit.dit_forward(sd, x, t, c, g, 3, 4, cfg_scale=7.0, negative_cross_attn_cond=c_neg, negative_cross_attn_mask=neg_mask, rnd=_matched())
    assert_close("masked negative prompt vs matched oracle", got, want_m, T(1e-2))
    print(f"\n[negative prompt] rel-L2 vs the reference {e1:.2e}, masked {e2:.2e}")


def test_generate_with_negative_conditioning(dev, small_dit):
    from oracle import dit as odit, sampler as osamp
    from stable_audio_tools import synthetic
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    b, steps, t_len = 2, 4, 24
    ratio = cfg["model"]["pretransform"]["config"]["downsampling_ratio"]

    def cond_of(seed, seconds):
        cond = model.conditioner([{"seconds_start": 0, "seconds_total": seconds + i} for i in range(b)])
        cond["prompt"] = (synthetic.synth_input("prompt", (b, 128, dc["cond_token_dim"]), seed).to(dev), torch.ones(b, 128, device=dev))
        return {k: cond[k] for k in ("prompt", "seconds_start", "seconds_total")}

    pos, neg = cond_of(31, 10), cond_of(33, 20)
    noise = synthetic.synth_input("noise", (b, 64, t_len), 32)
    step_noise = [synthetic.synth_input(f"sn{i}", (b, 64, t_len), 40 + i) for i in range(steps)]
    it = iter(step_noise)
    lat = generate_diffusion_cond(model, steps=steps, cfg_scale=7.0, conditioning_tensors=pos, negative_conditioning_tensors=neg,
                                  sample_size=t_len * ratio, seed=5, device=str(dev), sampler_type="dpmpp-3m-sde", sigma_min=0.3, sigma_max=500,
                                  return_latents=True, noise=noise, noise_sampler=lambda s, sn: next(it).to(dev))
    ci, ni = model.get_conditioning_inputs(pos), model.get_conditioning_inputs(neg, negative=True)
    cac, gc = ci["cross_attn_cond"].cpu().float(), ci["global_cond"].cpu().float()
    nc = ni["negative_cross_attn_cond"].cpu().float()
    sig = osamp.get_sigmas_polyexponential(steps, 0.3, 500.0, 1.0)
    fn = lambda xin, tt: odit.dit_forward(dsd, xin, tt, cac, gc, dc["depth"], dc["num_heads"], cfg_scale=7.0, negative_cross_attn_cond=nc,
                                          negative_cross_attn_mask=ni["negative_cross_attn_mask"].cpu(), rnd=_matched())
    want = osamp.sample_dpmpp_3m_sde(lambda x, s: osamp.vdenoise(fn, x, s), noise * sig[0], sig, lambda i, s, sn: step_noise[i])
    e = assert_close("trajectory with a negative prompt vs matched oracle", lat, want, T(2e-2))
    it = iter(step_noise)
    plain = generate_diffusion_cond(model, steps=steps, cfg_scale=7.0, conditioning_tensors=pos, sample_size=t_len * ratio, seed=5,
                                    device=str(dev), sampler_type="dpmpp-3m-sde", sigma_min=0.3, sigma_max=500, return_latents=True, noise=noise,
                                    noise_sampler=lambda s, sn: next(it).to(dev))
    assert rel_l2(lat, plain) > 1e-2, "the negative prompt must change the result"
    print(f"\n[negative conditioning] trajectory rel-L2 vs matched oracle {e:.2e}")


def test_dit_adaln_vs_reference_golden(dev):
    """global_cond_type='adaLN' (dit.py:205-206, transformer.py:665-689): no prepend token; LayerNorm modulated by
    (1 + scale, shift) and the self-attention / FF branch outputs gated by sigmoid(1 - gate), all from one stacked
    to_scale_shift_gate GEMV per forward.  Against the matched-rounding oracle (3e-3) and against the outputs of the
    REFERENCE itself (tests/golden/dit_adaln_small.npz, fp32; 2.5e-3 / 1.2e-2 with CFG 7, ~2x the measured 1.0e-3 / 4.9e-3)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from oracle import dit as odit
    from stable_audio_tools import synthetic
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.dit import DiffusionTransformer
    gold = cases.load("dit_adaln_small")
    with _init.skip_init():
        dit = DiffusionTransformer(**cases.SMALL_DIT, global_cond_type="adaLN")
    sd = synthetic.synth_state_dict(dit.state_dict(), 0)
    dit.load_state_dict(sd)
    dit = dit.to(dev).eval()
    for t_len in (64, 77):
        x, t, c, g = cases.dit_inputs(2, t_len, 128, 96, 1)
        got = dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=1.0)
        want_m = odit.dit_forward(sd, x, t, c, g, 3, 4, rnd=SUITE.round, adaln=True)
        e_m = assert_close(f"adaLN T={t_len} vs matched oracle", got, want_m, T(3e-3))
        e_f = assert_close(f"adaLN T={t_len} vs reference", got, gold[f"cfg1_T{t_len}"], T(2.5e-3))
        print(f"\n[adaLN T={t_len}] rel-L2 vs matched {e_m:.2e}, vs the reference {e_f:.2e}")
    x, t, c, g = cases.dit_inputs(2, 77, 128, 96, 1)
    got = dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=7.0)
    assert_close("adaLN cfg7 vs matched oracle", got, odit.dit_forward(sd, x, t, c, g, 3, 4, cfg_scale=7.0, rnd=SUITE.round, adaln=True), T(1e-2))
    e7 = assert_close("adaLN cfg7 vs reference", got, gold["cfg7_T77"], T(1.2e-2))
    print(f"\n[adaLN cfg7] rel-L2 vs the reference {e7:.2e}")
    got = dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), cfg_scale=1.0)
    assert_close("adaLN without global cond vs reference", got, gold["noglobal_T77"], T(2.5e-3))
    # adaLN-modulated LayerNorm fused with the e4m3 row quantisation (fp8 GEMM mode)
    for mode, fams in (("fp8", odit.FP8_DEFAULT_FAMILIES), ("fp8-all", odit.FP8_FAMILIES)):
        dit.set_gemm_dtype(mode)
        got8 = dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=1.0)
        assert_close(f"adaLN {mode} vs matched fp8 oracle", got8, odit.dit_forward(sd, x, t, c, g, 3, 4, rnd=odit.Fp8Rounding(fams), adaln=True), 5e-3)
    dit.set_gemm_dtype(SUITE.gemm_dtype)
    # fused sampler-step entry point
    dit.prepare_generation(c.to(dev), g.to(dev), 7.0)
    sigma = 2.3
    from oracle import sampler as osamp
    fn = lambda xin, tt: odit.dit_forward(sd, xin, tt, c, g, 3, 4, cfg_scale=7.0, rnd=SUITE.round, adaln=True)
    assert_close("adaLN denoise_cfg", dit.denoise((x * sigma).to(dev), sigma, cfg_scale=7.0), osamp.vdenoise(fn, x * sigma, torch.full((2,), sigma)), 1e-2)


def test_dit_other_operand_format(dev, small_dit):
    """The operand format the suite does NOT default to (SAT_TEST_DTYPE, tests/conftest.py; "fp16" = the package default, round 4: the fp16
    build of every block kernel, v_mfma_f32_*_f16 at the bf16 rate): switched on through set_gemm_dtype, against the oracle with that format's
    rounding at the same store points (LnFoldRounding[F16] / plain rounding without the LayerNorm fold) and against the fp32 oracle.  fp16
    carries 8x less operand rounding than bf16: its gates are the bf16 gates / 4."""
    from oracle import dit as odit
    from util import OperandFormat
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    dit = model.model.model
    x, c, g = _inputs(2, 77, dc["cond_token_dim"])
    t = torch.tensor([0.31, 0.87])
    run = lambda **kw: model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), **kw)
    base = run(cfg_scale=1.0)
    want_f = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"])
    other = OperandFormat("bf16" if SUITE.f16 else "f16")
    TO = other.tol
    dit.set_gemm_dtype(other.gemm_dtype)
    try:
        got = run(cfg_scale=1.0)
        fold_rnd = odit.LnFoldRoundingF16() if other.f16 else odit.LnFoldRounding()
        e_m = assert_close(f"{other.gemm_dtype} dit vs its matched oracle", got, odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], rnd=fold_rnd), TO(3e-3))
        e_f = assert_close(f"{other.gemm_dtype} dit vs fp32 oracle", got, want_f, TO(2.4e-3))
        e_b = rel_l2(got, base)
        assert e_b > 1e-4, "the switch must actually change the arithmetic"
        got7 = run(cfg_scale=7.0)
        want7 = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], cfg_scale=7.0)
        e_7 = assert_close(f"{other.gemm_dtype} dit cfg7 vs fp32 oracle", got7, want7, TO(1.2e-2))
        dit.set_layernorm_fusion(False)
        plain = run(cfg_scale=1.0)
        e_p = assert_close(f"{other.gemm_dtype} standalone-LayerNorm plan vs matched oracle", plain,
                           odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], rnd=other.round), TO(3e-3))
        print(f"\n[dit {other.gemm_dtype}] rel-L2 vs matched {e_m:.2e}, vs fp32 {e_f:.2e} ({SUITE.gemm_dtype} path vs fp32: {rel_l2(base, want_f):.2e}), CFG 7 vs fp32 {e_7:.2e}, "
              f"standalone LayerNorms vs matched {e_p:.2e}, vs the {SUITE.gemm_dtype} path {e_b:.2e}")
    finally:
        dit.set_layernorm_fusion(True)
        dit.set_gemm_dtype(SUITE.gemm_dtype)
    assert torch.equal(run(cfg_scale=1.0), base), "switching back must restore the default path bit for bit"


@pytest.mark.parametrize("mode", ["fp8", "fp8-all"])
def test_dit_fp8_gemm_mode(dev, small_dit, mode):
    """BASELINE config 5: e4m3 operands for the block GEMMs (per-token scales after a LayerNorm, MXFP8 block scales for the SwiGLU -- and
    in "fp8-all" the attention -- outputs, per-output-channel weight scales).  "fp8" = cross to_q + FF-in + FF-out, the families whose
    quantisation the sampler trajectory tolerates (tools/fp8_budget.py); "fp8-all" adds to_qkv and the to_out projections.  Against
    the matched-rounding oracle that quantises at the same points (gate 5e-3: accumulation order + the rare code flipped by
    x * (1/s) vs the kernel's own rounding), and against the fp32 oracle at the stated looser tolerance (3e-2 without
    CFG, measured 1.3e-2: e4m3 carries 3 mantissa bits; the bf16 path sits at ~1e-3 on the same case)."""
    from oracle import dit as odit
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    dit = model.model.model
    x, c, g = _inputs(2, 77, dc["cond_token_dim"])
    t = torch.tensor([0.31, 0.87])
    bf = model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=1.0)
    fams = odit.FP8_DEFAULT_FAMILIES if mode == "fp8" else odit.FP8_FAMILIES
    dit.set_gemm_dtype(mode)
    try:
        got = model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=1.0)
        want_m = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], rnd=odit.Fp8Rounding(fams))
        want_f = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"])
        e_m = assert_close("fp8 dit vs matched fp8 oracle", got, want_m, 5e-3)
        e_f = assert_close("fp8 dit vs fp32 oracle", got, want_f, 3e-2)
        e_b = rel_l2(got, bf)
        print(f"\n[dit {mode}] rel-L2 vs matched {e_m:.2e}, vs fp32 {e_f:.2e}, vs the bf16 path {e_b:.2e}")
        assert e_b > 1e-4, "fp8 mode must actually change the arithmetic"
        got7 = model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=7.0)
        want7 = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], cfg_scale=7.0, rnd=odit.Fp8Rounding(fams))
        # CFG 7 extrapolates the cond/uncond difference ~7x: 7 x (4e-3, a handful of e4m3 codes flipped by accumulation order) + margin
        assert_close("fp8 dit cfg7 vs matched fp8 oracle", got7, want7, 5e-2)
    finally:
        dit.set_gemm_dtype(SUITE.gemm_dtype)
    again = model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=1.0)
    assert torch.equal(again, bf), "switching back must restore the default path bit for bit"


@pytest.fixture(scope="module")
def small_vae(dev):
    from stable_audio_tools import model_configs as MC
    cfg = MC.reduced(MC.stable_audio_vae())
    model, sd = _build(cfg, 3, dev)
    return cfg, model, sd


# codec gates per operand format: (vs the fp32 oracle / reference, vs the matched-rounding oracle); fp16 = bf16 / 4 (8x less rounding, 2x margin)
CODEC = {"bf16": (bf16_round, 1.5e-2, 1e-2), "fp16": (fp16_round, 3.75e-3, 2.5e-3)}


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("b,t_len", [(1, 43), (2, 8), (1, 1)])
def test_oobleck_decode(dev, small_vae, b, t_len, fmt):
    from oracle import oobleck as oob
    from stable_audio_tools import synthetic
    cfg, model, sd = small_vae
    rnd, tol_f, tol_m = CODEC[fmt]
    strides = cfg["model"]["decoder"]["config"]["strides"]
    z = synthetic.synth_input("z", (b, 64, t_len), 11)
    model.set_gemm_dtype(fmt)
    try:
        got = model.decode(z.to(dev))
    finally:
        model.set_gemm_dtype(SUITE.gemm_dtype)
    dsd = _sub(sd, "decoder.")
    want_f = oob.oobleck_decoder(dsd, z, strides=strides)
    want_m = oob.oobleck_decoder(dsd, z, strides=strides, rnd=rnd)
    assert got.shape == want_f.shape
    e_f = assert_close("decode vs fp32 oracle", got, want_f, tol_f)
    e_m = assert_close("decode vs matched oracle", got, want_m, tol_m)
    print(f"\n[decode b={b} T={t_len}, {fmt}] rel-L2 vs matched {e_m:.2e}, vs fp32 {e_f:.2e}")


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("b,t_len", [(1, 21), (2, 4)])
def test_oobleck_encode_and_vae(dev, small_vae, b, t_len, fmt):
    from oracle import oobleck as oob
    from stable_audio_tools import synthetic
    cfg, model, sd = small_vae
    rnd, tol_f, tol_m = CODEC[fmt]
    strides = cfg["model"]["encoder"]["config"]["strides"]
    ratio = cfg["model"]["downsampling_ratio"]
    audio = synthetic.synth_input("a", (b, 2, t_len * ratio), 12, 0.4)
    esd = _sub(sd, "encoder.")
    want_f = oob.oobleck_encoder(esd, audio, strides=strides)
    want_m = oob.oobleck_encoder(esd, audio, strides=strides, rnd=rnd)
    noise = synthetic.synth_input("vn", (b, 64, t_len), 13)
    model.set_gemm_dtype(fmt)
    try:
        got = model.encoder(audio.to(dev))
        z = model.encode(audio.to(dev), noise=noise.to(dev))
    finally:
        model.set_gemm_dtype(SUITE.gemm_dtype)
    e_f = assert_close("encode vs fp32 oracle", got, want_f, tol_f)
    e_m = assert_close("encode vs matched oracle", got, want_m, tol_m)
    print(f"\n[encode b={b} T={t_len}, {fmt}] rel-L2 vs matched {e_m:.2e}, vs fp32 {e_f:.2e}")
    assert_close("encode+vae_sample", z, oob.vae_sample(want_f, noise), tol_f)


def test_generate_diffusion_cond_small(dev, small_dit):
    """Whole path (conditioning -> 6-step DPM++(3M) SDE with CFG -> decode) vs the oracle, with the initial and
    per-step noise injected; also checks that `seed` makes the call reproducible."""
    from oracle import dit as odit, oobleck as oob, sampler as osamp
    from stable_audio_tools import synthetic
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    b, steps, t_len = 2, 6, 24
    ratio = cfg["model"]["pretransform"]["config"]["downsampling_ratio"]
    prompt = synthetic.synth_input("prompt", (b, 128, dc["cond_token_dim"]), 31)
    cond = model.conditioner([{"seconds_start": 0, "seconds_total": 10 + i} for i in range(b)])
    cond["prompt"] = (prompt.to(dev), torch.ones(b, 128, device=dev))
    cond = {k: cond[k] for k in ("prompt", "seconds_start", "seconds_total")}
    noise = synthetic.synth_input("noise", (b, 64, t_len), 32)
    step_noise = [synthetic.synth_input(f"sn{i}", (b, 64, t_len), 40 + i) for i in range(steps)]
    it = iter(step_noise)
    lat = generate_diffusion_cond(model, steps=steps, cfg_scale=7.0, conditioning_tensors=cond, sample_size=t_len * ratio, seed=5,
                                  device=str(dev), sampler_type="dpmpp-3m-sde", sigma_min=0.3, sigma_max=500, return_latents=True,
                                  noise=noise, noise_sampler=lambda s, sn: next(it).to(dev))
    # oracle trajectory
    ci = model.get_conditioning_inputs(cond)
    cac, gc = ci["cross_attn_cond"].cpu().float(), ci["global_cond"].cpu().float()
    sig = osamp.get_sigmas_polyexponential(steps, 0.3, 500.0, 1.0)
    fn = lambda xin, tt: odit.dit_forward(dsd, xin, tt, cac, gc, dc["depth"], dc["num_heads"], cfg_scale=7.0, rnd=_matched())
    want = osamp.sample_dpmpp_3m_sde(lambda x, s: osamp.vdenoise(fn, x, s), noise * sig[0], sig, lambda i, s, sn: step_noise[i])
    e = assert_close("6-step latent trajectory vs matched oracle", lat, want, T(2e-2))
    print(f"\n[generate small] latent rel-L2 vs matched oracle {e:.2e}")
    # full call incl. decode; reproducible from the seed
    a1 = generate_diffusion_cond(model, steps=3, cfg_scale=7.0, conditioning_tensors=cond, sample_size=t_len * ratio, seed=9,
                                 device=str(dev), sampler_type="dpmpp-3m-sde", sigma_min=0.3, sigma_max=500)
    a2 = generate_diffusion_cond(model, steps=3, cfg_scale=7.0, conditioning_tensors=cond, sample_size=t_len * ratio, seed=9,
                                 device=str(dev), sampler_type="dpmpp-3m-sde", sigma_min=0.3, sigma_max=500)
    assert a1.shape == (b, 2, t_len * ratio) and torch.isfinite(a1).all()
    assert torch.equal(a1, a2), "same seed must give the same audio"
    vsd = _sub(sd, "pretransform.model.decoder.")
    strides = cfg["model"]["pretransform"]["config"]["decoder"]["config"]["strides"]
    assert_close("pretransform.decode", model.pretransform.decode(lat), oob.oobleck_decoder(vsd, lat.cpu(), strides=strides, rnd=SUITE.round), T(1.5e-2))


@pytest.mark.parametrize("sampler_type", ["dpmpp-2m-sde", "dpmpp-3m-sde"])
def test_sample_k_inpainting_and_2m(dev, small_dit, sampler_type):
    """sample_k with init data + soft mask (sampling.py:166-201: step-0 mix, then the per-step re-injection callback that
    mutates x right after the denoiser call) under both multistep SDE samplers, against the oracle trajectory with every
    Gaussian draw injected."""
    from oracle import dit as odit, sampler as osamp
    from stable_audio_tools import synthetic
    from stable_audio_tools.inference.generation import build_mask
    from stable_audio_tools.inference.sampling import sample_k
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    b, steps, t_len = 2, 5, 40
    c = synthetic.synth_input("c", (b, 130, dc["cond_token_dim"]), 61)
    g = synthetic.synth_input("g", (b, 2 * dc["cond_token_dim"]), 62)
    noise = synthetic.synth_input("noise", (b, 64, t_len), 63)
    init = synthetic.synth_input("init", (b, 64, t_len), 64)
    step_noise = [synthetic.synth_input(f"sn{i}", (b, 64, t_len), 70 + i) for i in range(steps)]
    renoise = [synthetic.synth_input(f"rn{i}", (b, 64, t_len), 80 + i) for i in range(steps)]
    mask = build_mask(t_len, dict(maskstart=20, maskend=80, softnessL=15, softnessR=10, marination=0.1))
    it = iter(step_noise)
    got = sample_k(model.model, noise.to(dev), init.to(dev), mask.to(dev), steps, sampler_type=sampler_type, sigma_min=0.3,
                   sigma_max=80.0, device=str(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=7.0,
                   noise_sampler=lambda s_, sn_: next(it).to(dev), inpaint_noise=lambda i: renoise[i].to(dev))
    sig = osamp.get_sigmas_polyexponential(steps, 0.3, 80.0, 1.0)
    fn = lambda xin, tt: odit.dit_forward(dsd, xin, tt, c, g, dc["depth"], dc["num_heads"], cfg_scale=7.0, rnd=_matched())
    x0, cb = osamp.inpainting_start_and_callback(init, noise * sig[0], mask, steps, lambda i: renoise[i])
    solver = osamp.sample_dpmpp_2m_sde if sampler_type == "dpmpp-2m-sde" else osamp.sample_dpmpp_3m_sde
    want = solver(lambda x, s_: osamp.vdenoise(fn, x, s_), x0.clone(), sig, lambda i, a_, b_: step_noise[i], callback=cb)
    e = assert_close(f"inpainting trajectory ({sampler_type}) vs matched oracle", got, want, T(2e-2))
    print(f"\n[inpaint {sampler_type}] rel-L2 {e:.2e}")
    with pytest.raises(NotImplementedError):
        sample_k(model.model, noise.to(dev), None, None, steps, sampler_type="k-euler-nonexistent", cross_attn_cond=c.to(dev),
                 global_cond=g.to(dev))


@pytest.mark.parametrize("sampler_type", ["k-heun", "k-lms", "k-dpmpp-2s-ancestral", "k-dpm-2", "k-dpm-fast"])
def test_sample_k_single_step_samplers(dev, small_dit, sampler_type):
    """The other sampler_type values of sample_k (sampling.py:212-225): product trajectory (fused CFG/VDenoiser DiT call + one
    sat_lincomb per state update) vs the oracle restatement driving the oracle DiT, ancestral noise injected."""
    from oracle import dit as odit, sampler as osamp
    from stable_audio_tools import synthetic
    from stable_audio_tools.inference.sampling import sample_k
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    b, steps, t_len = 2, 6, 24
    c = synthetic.synth_input("c", (b, 130, dc["cond_token_dim"]), 91)
    g = synthetic.synth_input("g", (b, 2 * dc["cond_token_dim"]), 92)
    noise = synthetic.synth_input("noise", (b, 64, t_len), 93)
    step_noise = [synthetic.synth_input(f"sn{i}", (b, 64, t_len), 95 + i) for i in range(steps)]
    it = iter(step_noise)
    seen = []
    got = sample_k(model.model, noise.to(dev), None, None, steps, sampler_type=sampler_type, sigma_min=0.3, sigma_max=80.0,
                   device=str(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=7.0,
                   noise_sampler=lambda s_, sn_: next(it).to(dev), callback=lambda a: seen.append(a["i"]))
    sig = osamp.get_sigmas_polyexponential(steps, 0.3, 80.0, 1.0)
    fn = lambda xin, tt: odit.dit_forward(dsd, xin, tt, c, g, dc["depth"], dc["num_heads"], cfg_scale=7.0, rnd=_matched())
    den = lambda x, s_: osamp.vdenoise(fn, x, s_)
    x0 = noise * sig[0]
    if sampler_type == "k-heun":
        want = osamp.sample_heun(den, x0, sig)
    elif sampler_type == "k-lms":
        want = osamp.sample_lms(den, x0, sig)
    elif sampler_type == "k-dpm-2":
        want = osamp.sample_dpm_2(den, x0, sig)
    elif sampler_type == "k-dpmpp-2s-ancestral":
        want = osamp.sample_dpmpp_2s_ancestral(den, x0, sig, lambda i, a_, b_: step_noise[i])
    else:
        want = osamp.sample_dpm_fast(den, x0, 0.3, 80.0, steps)
    e = assert_close(f"{sampler_type} trajectory vs matched oracle", got, want, T(2e-2))
    assert seen == list(range(len(seen))) and len(seen) >= 2
    print(f"\n[{sampler_type}] rel-L2 {e:.2e}")


def test_sample_k_dpm_adaptive(dev, small_dit):
    """k-dpm-adaptive (sampling.py:222-224): the product's PID-controlled embedded pair (3 fused DiT evaluations + one
    sat_dpm_error_partials reduction and ONE host sync per step) against the oracle restatement driving the oracle DiT: same
    accept/reject history and the same latents."""
    from oracle import dit as odit, sampler as osamp
    from stable_audio_tools import synthetic
    from stable_audio_tools.inference.sampling import sample_k
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    b, t_len = 2, 24
    c = synthetic.synth_input("c", (b, 130, dc["cond_token_dim"]), 121)
    g = synthetic.synth_input("g", (b, 2 * dc["cond_token_dim"]), 122)
    noise = synthetic.synth_input("noise", (b, 64, t_len), 123)
    info = {}
    got = sample_k(model.model, noise.to(dev), None, None, 10, sampler_type="k-dpm-adaptive", sigma_min=2.0, sigma_max=20.0,
                   device=str(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=3.0, info=info)
    fn = lambda xin, tt: odit.dit_forward(dsd, xin, tt, c, g, dc["depth"], dc["num_heads"], cfg_scale=3.0, rnd=_matched())
    winfo = {}
    sig0 = float(osamp.get_sigmas_polyexponential(10, 2.0, 20.0, 1.0)[0])
    want = osamp.sample_dpm_adaptive(lambda x, s_: osamp.vdenoise(fn, x, s_), noise * sig0, 2.0, 20.0, info=winfo)
    print(f"\n[k-dpm-adaptive] product {info}  oracle {winfo}")
    assert info["steps"] >= 3 and info["nfe"] == 3 * info["steps"]
    assert (info["n_accept"], info["n_reject"]) == (winfo["n_accept"], winfo["n_reject"])
    assert_close("k-dpm-adaptive latents vs matched oracle", got, want, T(2e-2))


def test_rectified_flow_euler(dev, small_dit):
    """diffusion_objective 'rectified_flow' (generation.py:235-244 -> sampling.py:236-270, :28-60): x <- x + dt * model(x, t)
    with the CFG-batched raw DiT output."""
    from oracle import dit as odit, sampler as osamp
    from stable_audio_tools import synthetic
    from stable_audio_tools.inference.sampling import sample_rf
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    b, steps, t_len = 2, 5, 24
    c = synthetic.synth_input("c", (b, 130, dc["cond_token_dim"]), 111)
    g = synthetic.synth_input("g", (b, 2 * dc["cond_token_dim"]), 112)
    noise = synthetic.synth_input("noise", (b, 64, t_len), 113)
    init = synthetic.synth_input("init", (b, 64, t_len), 114)
    got = sample_rf(model.model, noise.to(dev), init_data=init.to(dev), steps=steps, sigma_max=0.8, device=str(dev),
                    cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=3.0, batch_cfg=True, rescale_cfg=True)
    fn = lambda xin, tt: odit.dit_forward(dsd, xin, tt, c, g, dc["depth"], dc["num_heads"], cfg_scale=3.0, rnd=_matched())
    want = osamp.sample_discrete_euler(fn, init * (1 - 0.8) + noise * 0.8, steps, 0.8)
    assert_close("rectified-flow Euler trajectory", got, want, T(1e-2))


def _two_layer_full_width(dev, seed=0):
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from stable_audio_tools import synthetic
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.dit import DiffusionTransformer
    with _init.skip_init():
        dit = DiffusionTransformer(**dict(cases.FULL_DIT, depth=2))
    dit.load_state_dict(synthetic.synth_state_dict(dit.state_dict(), seed))
    return dit.to(dev).eval()


@pytest.mark.parametrize("gemm_dtype", ["bf16", "fp16"])
def test_two_plans_two_streams_split_k(dev, gemm_dtype):
    """VERDICT r3 item 2 / ADVICE r3: the K-split scratch of the 8-phase FF-out GEMM lives in the CALLER's workspace, per plan.  Two plans
    (two 2-layer full-width DiTs) at the SA-2.0 shape -- M = 2 x 6145 rows, FF-out K = 6144: 294 tiles = one whole round + 38 remainder
    tiles cut along K -- run on two streams at once on different inputs; each result must be bit-identical to the same plan run alone.
    (Round 3 kept ONE slab per device and process: concurrent launches added each other's partial sums.)"""
    import ctypes
    import cases
    from stable_audio_tools import _hip
    lib = _hip.lib()
    need = ctypes.c_size_t()
    _hip.check(lib.sat_gemm_f32_workspace_bytes(2 * 6145, 1536, 6144, 0, ctypes.byref(need)))
    assert need.value == 256 * 65536 * 4, "FF-out at the SA-2.0 shape is expected to split its remainder round"
    dits = [_two_layer_full_width(dev, 0).set_gemm_dtype(gemm_dtype), _two_layer_full_width(dev, 0).set_gemm_dtype(gemm_dtype)]
    ins = [cases.dit_inputs(2, 6144, 768, 1536, 11 + 7 * i) for i in range(2)]
    ins = [tuple(t.to(dev) for t in tup) for tup in ins]
    alone = [d(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=1.0).clone() for d, (x, t, c, g) in zip(dits, ins)]
    torch.cuda.synchronize()
    assert not torch.equal(alone[0], alone[1])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                x, t, c, g = ins[i]
                outs[i].append(dits[i](x, t, cross_attn_cond=c, global_embed=g, cfg_scale=1.0))
    torch.cuda.synchronize()
    for i in (0, 1):
        for rep, o in enumerate(outs[i]):
            assert torch.equal(o, alone[i]), f"plan {i}, repeat {rep}: concurrent run differs from the run alone by {rel_l2(o, alone[i]):.3e}"


def test_forward_is_graph_capturable(dev):
    """no allocation, host copy or device synchronisation is reachable from sat_dit_forward on a warmed plan, so a
    forward can be captured into a hipGraph (include/sat_hip.h conventions).  SA-2.0 shape on a 2-layer full-width DiT: covers the 8-phase
    GEMMs with their persistent schedule and the K-split FF-out with its reduce launch.  Replays must be bit-identical to the eager call."""
    import ctypes
    import cases
    from stable_audio_tools import _hip
    lib = _hip.lib()
    dit = _two_layer_full_width(dev, 0)
    x, t, c, g = (v.to(dev) for v in cases.dit_inputs(2, 6144, 768, 1536, 5))
    eager = dit(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=1.0).clone()       # builds the plan, the context and the workspace
    ws = dit._workspace(2, 6144)
    out = torch.zeros_like(x)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin()
        rc = lib.sat_dit_forward(dit._plan, _hip.ptr(x), _hip.ptr(t), _hip.ptr(out), 2, 6144, _hip.ptr(ws), ws.numel(), _hip.stream())
        graph.capture_end()
    _hip.check(rc)
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), f"graph replay {rep} differs from the eager forward by {rel_l2(out, eager):.3e}"


@pytest.fixture(scope="module")
def full_dit(dev):
    """Full-size SA-Open DiT (24 layers, D=1536, 1.06 B synthetic parameters, seed 0), built once for all full-size tests."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from stable_audio_tools import synthetic
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.dit import DiffusionTransformer
    with _init.skip_init():
        dit = DiffusionTransformer(**cases.FULL_DIT)
    dit.load_state_dict(synthetic.synth_state_dict(dit.state_dict(), 0))
    dit = dit.to(dev).eval()
    yield dit
    del dit
    torch.cuda.empty_cache()


@pytest.mark.parametrize("gemm_dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("t_len", [1024, 6144])
def test_full_size_dit_vs_reference_golden(dev, full_dit, t_len, gemm_dtype):
    """Full-size SA-Open DiT against the output of the REFERENCE itself (tests/golden/dit_full_T*.npz: fp32 CPU run of
    /root/reference in the build container) at the SA-Open (T=1024) and SA-2.0 (T=6144) context lengths.  bf16 GEMM operands
    vs the fp32 reference: gate 8e-3 (measured 3.7e-3 / 3.5e-3; SURVEY.md section 7 measured 1.5e-2 for a bf16 autocast of
    the reference itself).  fp16 operands (round 4; the reference's own GPU arithmetic): north_star's 1e-3."""
    import cases
    import os
    path = os.path.join(cases.GOLDEN_DIR, f"dit_full_T{t_len}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated")
    want = cases.load(f"dit_full_T{t_len}")["out"]
    x, t, c, g = cases.dit_inputs(1, t_len, 768, 1536, 1)
    full_dit.set_gemm_dtype(gemm_dtype)
    try:
        got = full_dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=1.0)
    finally:
        full_dit.set_gemm_dtype(SUITE.gemm_dtype)
    e = assert_close(f"full-size DiT T={t_len} vs reference [{gemm_dtype}]", got, want, 8e-3 if gemm_dtype == "bf16" else 1e-3)
    print(f"\n[full DiT T={t_len}, {gemm_dtype}] rel-L2 vs the reference's fp32 output {e:.2e} (out std {want.std():.3f})")


def _batch8_inputs():
    """item 0 = the inputs of the reference golden dit_full_T1024, items 1..7 = other seeds."""
    import cases
    xs, ts, cs, gs = zip(*[cases.dit_inputs(1, 1024, 768, 1536, 1 + 10 * i) for i in range(8)])
    t = torch.tensor([0.5, 0.11, 0.23, 0.37, 0.52, 0.68, 0.81, 0.93])
    return torch.cat(xs), t, torch.cat(cs), torch.cat(gs)


@pytest.mark.parametrize("gemm_dtype", ["bf16", "fp8", "fp8-all", "fp16"])
def test_full_size_batch8_config3_config5(dev, full_dit, gemm_dtype):
    """BASELINE config 3 (8 prompts per GPU: Bf = 16 sequences with CFG, M = 16400 rows -> the large-M tile path, 65 row-tile
    bands, EPI_HEADS across 16 sequences) and config 5 (the same with e4m3 / MXFP8 GEMM operands) at FULL size, against the
    REFERENCE's fp32 outputs for prompt 0 (tests/golden/dit_full_T1024.npz: `out` at cfg_scale 1, `cfg7` with batched CFG 7):
      * prompt 0 inside the batch of 8, cfg 1 (M = 8200) and CFG 7 (M = 16400), vs the reference;
      * batch invariance: every prompt of the batched evaluation against its own B=1 evaluation.  Not bit-equal by design: the
        key-side shift (b*S)&3 moves the attention tile boundaries per sequence, so P is rounded against a different running
        max; CFG 7 then amplifies the bf16 (e4m3) rounding noise ~7-10x exactly as it does against the oracle.
    Gates = 2x the values measured on MI355X (bf16: 3.7e-3 / CFG-7 batch invariance 7.8e-3; fp8: 6.4e-2 at cfg 1, 1.8e-1 at CFG 7, batch
    invariance 1.6e-1 -- e4m3 has 3 mantissa bits: config 5's stated tolerance is 1.3e-1 per denoiser call without CFG, 3.6e-1 with CFG 7;
    what that does to a trajectory is pinned by test_full_size_trajectory)."""
    import cases
    bf = gemm_dtype == "bf16"
    # per-call gates: (prompt 0 vs reference at cfg 1, at CFG 7, batch invariance at cfg 1, at CFG 7 and the fused denoise entry)
    g_ref1, g_ref7, g_inv1, g_inv7 = {"bf16": (8e-3, 3e-2, 3e-3, 1.6e-2), "fp8-all": (1.3e-1, 3.6e-1, 8e-2, 3.2e-1), "fp8": (9e-3, 2.5e-2, 3e-3, 2e-2),
                                      "fp16": (1e-3, 4e-3, 5e-4, 2.5e-3)}[gemm_dtype]
    x, t, c, g = _batch8_inputs()
    t[0] = cases.dit_inputs(1, 1024, 768, 1536, 1)[1][0]
    gold = cases.load("dit_full_T1024")
    full_dit.set_gemm_dtype(gemm_dtype)
    try:
        xd, td, cd, gd = x.to(dev), t.to(dev), c.to(dev), g.to(dev)
        got1 = full_dit(xd, td, cross_attn_cond=cd, global_embed=gd, cfg_scale=1.0)                 # M = 8200
        e0 = assert_close(f"[{gemm_dtype}] prompt 0 of 8 vs reference (cfg 1)", got1[:1], gold["out"], g_ref1)
        got7 = full_dit(xd, td, cross_attn_cond=cd, global_embed=gd, cfg_scale=7.0)                 # Bf = 16, M = 16400
        assert torch.isfinite(got7).all()
        e7 = assert_close(f"[{gemm_dtype}] prompt 0 of 8 vs reference (CFG 7)", got7[:1], gold["cfg7"], g_ref7)
        w1 = w7 = 0.0
        for i in range(8):
            one1 = full_dit(xd[i:i + 1], td[i:i + 1], cross_attn_cond=cd[i:i + 1], global_embed=gd[i:i + 1], cfg_scale=1.0)
            one7 = full_dit(xd[i:i + 1], td[i:i + 1], cross_attn_cond=cd[i:i + 1], global_embed=gd[i:i + 1], cfg_scale=7.0)
            w1 = max(w1, rel_l2(got1[i:i + 1], one1))
            w7 = max(w7, rel_l2(got7[i:i + 1], one7))
        print(f"\n[config {'5' if gemm_dtype.startswith('fp8') else '3'} full size, B=8, {gemm_dtype}] prompt 0 vs reference: cfg 1 {e0:.2e}, CFG 7 {e7:.2e}; "
              f"batched vs B=1, worst of 8: cfg 1 {w1:.2e}, CFG 7 {w7:.2e}")
        assert w1 <= g_inv1, f"batch of 8 differs from B=1 by {w1:.3e} at cfg 1"
        assert w7 <= g_inv7, f"batch of 8 differs from B=1 by {w7:.3e} at CFG 7"
        # the fused sampler-step entry point at B=8 (what generate_diffusion_cond calls 100 times): same kernels; the input scaling
        # c_in is folded into the input projection, so an fp32 ulp can flip a bf16 rounding -> same noise floor as above
        sigma = 2.5
        full_dit.prepare_generation(cd, gd, 7.0)
        den = full_dit.denoise(xd * sigma, sigma, cfg_scale=7.0)
        from oracle import sampler as osamp
        want_den = osamp.vdenoise(lambda xin, tt: full_dit(xin.to(dev), tt.to(dev), cross_attn_cond=cd, global_embed=gd, cfg_scale=7.0).cpu(),
                                  x * sigma, torch.full((8,), sigma))
        e_d = assert_close(f"[{gemm_dtype}] denoise_cfg B=8 vs forward + VDenoiser scalings", den, want_den, g_inv7)
        print(f"[config {'5' if gemm_dtype.startswith('fp8') else '3'}, {gemm_dtype}] fused denoise_cfg at B=8 vs forward + scalings: {e_d:.2e}")
    finally:
        full_dit.set_gemm_dtype(SUITE.gemm_dtype)


@pytest.mark.parametrize("gemm_dtype", ["bf16", "fp8", "fp16"])
def test_full_size_trajectory(dev, full_dit, gemm_dtype):
    """Multi-step parity at FULL size: 12 steps of DPM-Solver++(3M) SDE, sigma 500 -> 0.3, batched CFG 7, on the
    SA-Open DiT (D = 1536, T = 1024) through the product's own `sample_k`, initial and per-step noise injected, against the CPU
    oracle's trajectory with the SAME rounding points (tests/golden/traj_full.npz, generated by tests/golden/make_traj_golden.py:
    fp32, bf16-matched, e4m3-matched) after 4, 8 and 12 steps.  What it pins that the single-forward tests cannot: the per-step
    rounding noise is fed back through x and amplified by CFG 7 at every step.  Gates = ~2x the values measured on MI355X (printed)."""
    import cases
    import os
    if os.environ.get("SAT_SKIP_SLOW") == "1":
        pytest.skip("SAT_SKIP_SLOW=1")
    path = os.path.join(cases.GOLDEN_DIR, "traj_full.npz")
    if not os.path.exists(path):
        pytest.skip("traj_full.npz not generated")
    from stable_audio_tools.inference.sampling import sample_k
    from stable_audio_tools.models.diffusion import DiTWrapper
    gold = cases.load("traj_full")
    tj = cases.TRAJ
    c, g, noise, step_noise = cases.traj_inputs()
    wrap = DiTWrapper.__new__(DiTWrapper)          # sample_k wants the wrapper type; the full-size DiT of the fixture goes inside
    torch.nn.Module.__init__(wrap)
    wrap.model = full_dit
    snaps = {}

    def cb(info):
        if info["i"] in tj["snapshots"]:           # x at the start of step i = the latents after i steps
            snaps[info["i"]] = info["x"].clone()

    it = iter(step_noise)
    full_dit.set_gemm_dtype(gemm_dtype)
    try:
        x = sample_k(wrap, noise.to(dev), steps=tj["steps"], sampler_type="dpmpp-3m-sde", sigma_min=tj["sigma_min"], sigma_max=tj["sigma_max"],
                     device=str(dev), callback=cb, noise_sampler=lambda s, sn: next(it).to(dev), cfg_scale=tj["cfg_scale"],
                     cross_attn_cond=c.to(dev), global_cond=g.to(dev))
    finally:
        full_dit.set_gemm_dtype(SUITE.gemm_dtype)
    snaps[tj["steps"]] = x
    # tolerances: (vs the matched-rounding oracle, vs the fp32 oracle) per snapshot
    # Measured on MI355X (round 3), rel-L2 after 4 / 8 / 12 steps:
    #   bf16: vs matched oracle 2.9e-4 / 6.5e-3 / 9.4e-3, vs fp32 oracle 3.6e-4 / 5.2e-3 / 8.1e-3 (the matched ORACLE itself is 3.7e-4 / 4.6e-3 /
    #         7.9e-3 away from the fp32 oracle: the trajectory amplifies rounding noise chaotically, two bf16 evaluations with the same
    #         rounding points but different summation order drift apart as fast as either drifts from fp32);
    #   fp8-all (every block GEMM e4m3, round 3's config 5) drifted 8.7e-3 / 1.8e-1 / 3.7e-1 from its matched oracle and 2.9e-1 from fp32 -- the to_out
    #         projections on MXFP8 attention outputs alone cost 3.3e-1 (tools/fp8_budget.py).  Its trajectory case was dropped in round 5: a gate of
    #         7.4e-1 gates nothing; the mode stays covered per forward at full size (test_full_size_batch8_config3_config5[fp8-all]) and
    #         against its matched oracle (test_dit_fp8_gemm_mode[fp8-all], test_fp8_full_width_slice_vs_matched_oracle);
    #   fp8 (round 4's config 5: cross to_q + FF-in + FF-out): vs matched 3.4e-4 / 7.1e-3 / 1.3e-2, vs fp32 4.1e-4 / 5.7e-3 / 1.4e-2 -- bf16-class.
    #   fp16 (round 4): see the printed lines; gates = 2x measured
    tol = {"bf16": {4: (6e-4, 8e-4), 8: (1.3e-2, 1.1e-2), 12: (1.9e-2, 1.7e-2)},
           "fp8": {4: (7e-4, 9e-4), 8: (1.5e-2, 1.2e-2), 12: (2.6e-2, 2.9e-2)},
           "fp16": {4: (2e-4, 2e-4), 8: (3e-3, 3e-3), 12: (5e-3, 5e-3)}}[gemm_dtype]
    msg = []
    tag = gemm_dtype.replace("-", "")          # fixture keys: bf16 / fp16 / fp8 / fp8all
    for i in tj["snapshots"]:
        em = rel_l2(snaps[i], gold[f"{tag}_step{i}"])
        ef = rel_l2(snaps[i], gold[f"fp32_step{i}"])
        om = rel_l2(gold[f"{tag}_step{i}"], gold[f"fp32_step{i}"])
        msg.append(f"after {i:2d} steps: vs matched oracle {em:.2e}, vs fp32 oracle {ef:.2e} (matched oracle vs fp32 oracle {om:.2e})")
        assert torch.isfinite(snaps[i]).all()
    print(f"\n[full-size trajectory, {gemm_dtype}]\n  " + "\n  ".join(msg))
    for i in tj["snapshots"]:
        assert_close(f"[{gemm_dtype}] latents after {i} steps vs matched oracle", snaps[i], gold[f"{tag}_step{i}"], tol[i][0])
        assert_close(f"[{gemm_dtype}] latents after {i} steps vs fp32 oracle", snaps[i], gold[f"fp32_step{i}"], tol[i][1])


# 100-step gates: ~2x the values measured on MI355X (round 5, printed by the test; profiles/r05_traj100_parity.txt), per format:
# {snapshot: latents rel-L2 vs the fp32 oracle}, audio rel-L2 (worst of the two windows)
# measured: fp32x 3.3e-7 / 3.8e-7 / 3.9e-7 / 1.7e-6, audio 7.6e-4 (all of it the fp16 codec: 7.6e-4 on the oracle's own latents);
#           fp16  2.0e-6 / 6.0e-6 / 3.7e-5 / 8.9e-4, audio 8.1e-4;   bf16 1.7e-5 / 4.7e-5 / 3.1e-4 / 4.6e-3, audio 7.9e-3 (bf16 codec alone 7.8e-3);
#           fp8   1.7e-5 / 5.0e-5 / 3.1e-4 / 9.4e-3, audio with the bf16 codec of that mode ~8e-3
TRAJ100_GATES = {
    "fp32x": ({12: 1e-6, 25: 1e-6, 50: 1e-6, 100: 4e-6}, 1.6e-3),
    "fp16": ({12: 5e-6, 25: 1.2e-5, 50: 8e-5, 100: 1.8e-3}, 1.7e-3),
    "bf16": ({12: 3.5e-5, 25: 1e-4, 50: 6.5e-4, 100: 9.5e-3}, 1.6e-2),
    "fp8": ({12: 3.5e-5, 25: 1e-4, 50: 6.5e-4, 100: 1.9e-2}, 2e-2),
}


@pytest.mark.parametrize("gemm_dtype", ["fp32x", "fp16", "bf16", "fp8"])
def test_full_size_trajectory_100_steps(dev, full_dit, gemm_dtype):
    """Parity at the HEADLINE's own length: the metric is a 100-step generation (generate.py:27-32,
    inference/generation.py:95-261).  100 steps of DPM-Solver++(3M) SDE, sigma 500 -> 0.3, batched CFG 7, full-size SA-Open DiT (D = 1536,
    T = 1024), initial and per-step noise injected, through the product's own `sample_k`, then the full-size Oobleck decode -- against the
    CPU oracle's fp32 trajectory and audio (tests/golden/traj100_full.npz, generated by tests/golden/make_traj100_golden.py: latents after
    12 / 25 / 50 / 100 steps, two 65 536-sample windows of the decoded audio).  `fp32x` -- the exact-fp32 verification mode, same plan, other
    summation order -- is the yardstick: the trajectory amplifies any perturbation, so what the 16-bit formats add has to be read against the
    distance between two fp32 evaluations.  Also reported: the codec alone (the ORACLE's final latents through the product's decoder)."""
    import cases
    import os
    if os.environ.get("SAT_SKIP_SLOW") == "1":
        pytest.skip("SAT_SKIP_SLOW=1")
    from stable_audio_tools import synthetic
    from stable_audio_tools.inference.sampling import sample_k
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.autoencoders import OobleckDecoder
    from stable_audio_tools.models.diffusion import DiTWrapper
    gold = cases.load("traj100_full")
    tj = cases.TRAJ100
    c, g, noise, step_noise = cases.traj100_inputs()
    wrap = DiTWrapper.__new__(DiTWrapper)          # sample_k wants the wrapper type; the full-size DiT of the fixture goes inside
    torch.nn.Module.__init__(wrap)
    wrap.model = full_dit
    snaps = {}

    def cb(info):
        if info["i"] in tj["snapshots"]:           # x at the start of step i = the latents after i steps
            snaps[info["i"]] = info["x"].clone()

    sn_dev = [n.to(dev) for n in step_noise]
    it = iter(sn_dev)
    full_dit.set_gemm_dtype(gemm_dtype)
    try:
        x = sample_k(wrap, noise.to(dev), steps=tj["steps"], sampler_type="dpmpp-3m-sde", sigma_min=tj["sigma_min"], sigma_max=tj["sigma_max"],
                     device=str(dev), callback=cb, noise_sampler=lambda s, sn: next(it), cfg_scale=tj["cfg_scale"],
                     cross_attn_cond=c.to(dev), global_cond=g.to(dev))
    finally:
        full_dit.set_gemm_dtype(SUITE.gemm_dtype)
    snaps[tj["steps"]] = x
    assert torch.isfinite(x).all()
    # the codec build that goes with the DiT's format: the ONE rule generate.py and bench.py use (fp16 with fp16 / fp32x, bf16 with bf16 and e4m3)
    from stable_audio_tools import _config
    codec_fmt = _config.codec_gemm_dtype(gemm_dtype)
    with _init.skip_init():
        dec = OobleckDecoder(**cases.vae_kwargs(cases.FULL_VAE, True))
    dec.load_state_dict(synthetic.synth_state_dict(dec.state_dict(), 0))
    dec = dec.to(dev).set_gemm_dtype(codec_fmt)
    audio = dec(x)
    audio_codec_only = dec(gold[f"fp32_step{tj['steps']}"].to(dev))
    assert audio.shape == (1, 2, 1024 * 2048) and torch.isfinite(audio).all()
    lat_gate, audio_gate = TRAJ100_GATES[gemm_dtype]
    msg, e_lat = [], {}
    for i in tj["snapshots"]:
        e_lat[i] = rel_l2(snaps[i], gold[f"fp32_step{i}"])
        msg.append(f"latents after {i:3d} steps vs fp32 oracle {e_lat[i]:.2e}")
    e_audio, e_codec = 0.0, 0.0
    for name, s0 in tj["audio_windows"].items():
        w = slice(s0, s0 + tj["audio_window_len"])
        ea, ec = rel_l2(audio[:, :, w], gold[f"audio_{name}"]), rel_l2(audio_codec_only[:, :, w], gold[f"audio_{name}"])
        msg.append(f"audio window '{name}': generation {ea:.2e}, codec alone ({codec_fmt}, on the oracle's latents) {ec:.2e}")
        e_audio, e_codec = max(e_audio, ea), max(e_codec, ec)
    print(f"\n[100-step full-size trajectory, {gemm_dtype}]\n  " + "\n  ".join(msg))
    for i in tj["snapshots"]:
        assert e_lat[i] <= lat_gate[i], f"[{gemm_dtype}] latents after {i} steps: rel-L2 {e_lat[i]:.3e} > {lat_gate[i]:.1e}"
    assert e_audio <= audio_gate, f"[{gemm_dtype}] decoded audio: rel-L2 {e_audio:.3e} > {audio_gate:.1e}"


# Gate of the claim README / DESIGN section 2 make for the default format -- "<= 1e-3 rel-L2 against the fp32 latents over the 100-step generation
# the metric is quoted on" -- on THREE (prompt, seed) pairs (round 5 had one pair at 8.9e-4 under a 1.8e-3 gate): the gate IS the
# claim.  Measured on MI355X (round 6, printed by the test; profiles/r06_traj100_three_pairs.txt): fp16 9.16e-4 / 5.70e-4 / 6.62e-4, bf16 4.38e-3 / 6.11e-3 /
# 4.37e-3, fp32x 1.66e-6 / 1.64e-6 / 1.68e-6.  The kernels are deterministic: these figures are a property of the build, not of the box.
TRAJ100_3PAIRS_GATES = {"fp16": 1e-3, "bf16": 9.5e-3, "fp32x": 4e-6}


@pytest.mark.parametrize("gemm_dtype", ["fp16", "bf16", "fp32x"])
def test_full_size_trajectory_100_steps_three_pairs(dev, full_dit, gemm_dtype):
    """The 100-step full-size trajectory of `test_full_size_trajectory_100_steps` on three (prompt, seed) pairs -- variant 0 is that test's own
    fixture, variants 1 and 2 have their own conditioning tensors and their own initial / per-step noise (tests/golden/traj100_full_v{1,2}.npz,
    make_traj100_golden.py <threads> <variant>: fp32 oracle latents after 50 / 100 steps).  The gate is the MAX over the three of the rel-L2 of the
    final latents, set at the figure the documentation claims (fp16: 1e-3 = north_star's tolerance)."""
    import cases
    import os
    if os.environ.get("SAT_SKIP_SLOW") == "1":
        pytest.skip("SAT_SKIP_SLOW=1")
    from stable_audio_tools.inference.sampling import sample_k
    from stable_audio_tools.models.diffusion import DiTWrapper
    tj = cases.TRAJ100
    wrap = DiTWrapper.__new__(DiTWrapper)
    torch.nn.Module.__init__(wrap)
    wrap.model = full_dit
    errs, msg = {}, []
    full_dit.set_gemm_dtype(gemm_dtype)
    try:
        for v in cases.TRAJ100_VARIANTS:
            gold = cases.load("traj100_full" if v == 0 else f"traj100_full_v{v}")
            c, g, noise, step_noise = cases.traj100_inputs(v)
            snaps = {}

            def cb(info):
                if info["i"] == 50:
                    snaps[50] = info["x"].clone()

            it = iter([n.to(dev) for n in step_noise])
            x = sample_k(wrap, noise.to(dev), steps=tj["steps"], sampler_type="dpmpp-3m-sde", sigma_min=tj["sigma_min"], sigma_max=tj["sigma_max"],
                         device=str(dev), callback=cb, noise_sampler=lambda s, sn: next(it), cfg_scale=tj["cfg_scale"],
                         cross_attn_cond=c.to(dev), global_cond=g.to(dev))
            assert torch.isfinite(x).all()
            e50, e100 = rel_l2(snaps[50], gold["fp32_step50"]), rel_l2(x, gold["fp32_step100"])
            errs[v] = e100
            msg.append(f"pair {v}: latents after 50 / 100 steps vs fp32 oracle {e50:.2e} / {e100:.2e}")
    finally:
        full_dit.set_gemm_dtype(SUITE.gemm_dtype)
    worst = max(errs.values())
    print(f"\n[100-step full-size trajectory, three (prompt, seed) pairs, {gemm_dtype}]\n  " + "\n  ".join(msg) + f"\n  max over the pairs {worst:.2e} (gate {TRAJ100_3PAIRS_GATES[gemm_dtype]:.1e})")
    assert worst <= TRAJ100_3PAIRS_GATES[gemm_dtype], f"[{gemm_dtype}] final latents, worst of three pairs: rel-L2 {worst:.3e} > {TRAJ100_3PAIRS_GATES[gemm_dtype]:.1e}"


def test_fp32x_mode_vs_reference_golden(dev, full_dit, small_dit):
    """gemm_dtype="fp32x": the fp32 verification mode (csrc/f32_ref.hip: exact fp32 MFMA, fp32 LayerNorm output, fp32 q / k / v / P)
    through the SAME plan, workspace layout, RoPE table, prepend token, null-context skip and CFG batching as the bf16 path.
    north_star's tolerance -- <= 1e-3 rel-L2 vs the reference latents -- against the REFERENCE's own fp32 outputs at full size
    (T = 1024 with and without CFG 7, T = 6144), and <= 1e-4 against the fp32 oracle on the reduced model.  What the bf16 path
    adds on top of this (3.5e-3 at full size) is therefore operand rounding, not indexing."""
    import cases
    from oracle import dit as odit
    cfg, model, sd = small_dit
    dc = cfg["model"]["diffusion"]["config"]
    dsd = _sub(sd, "model.model.")
    dit = model.model.model
    x, c, g = _inputs(2, 77, dc["cond_token_dim"])
    t = torch.tensor([0.31, 0.87])
    dit.set_gemm_dtype("fp32x")
    try:
        for cfg_scale, tol in ((1.0, 1e-4), (7.0, 1e-4)):
            got = model.model(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_cond=g.to(dev), cfg_scale=cfg_scale)
            want = odit.dit_forward(dsd, x, t, c, g, dc["depth"], dc["num_heads"], cfg_scale=cfg_scale)
            e = assert_close(f"fp32x reduced DiT cfg {cfg_scale} vs fp32 oracle", got, want, tol)
            print(f"\n[fp32x reduced, cfg {cfg_scale}] rel-L2 vs fp32 oracle {e:.2e}")
    finally:
        dit.set_gemm_dtype(SUITE.gemm_dtype)
    full_dit.set_gemm_dtype("fp32x")
    try:
        gold = cases.load("dit_full_T1024")
        x, t, c, g = cases.dit_inputs(1, 1024, 768, 1536, 1)
        got = full_dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=1.0)
        e1 = assert_close("fp32x full-size DiT T=1024 vs reference", got, gold["out"], 1e-3)
        got = full_dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=7.0)
        e7 = assert_close("fp32x full-size DiT T=1024 CFG 7 vs reference", got, gold["cfg7"], 1e-3)
        import os
        e6 = float("nan")
        if os.path.exists(os.path.join(cases.GOLDEN_DIR, "dit_full_T6144.npz")):
            x, t, c, g = cases.dit_inputs(1, 6144, 768, 1536, 1)
            got = full_dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=1.0)
            e6 = assert_close("fp32x full-size DiT T=6144 vs reference", got, cases.load("dit_full_T6144")["out"], 1e-3)
        print(f"\n[fp32x full size] rel-L2 vs the reference: T=1024 {e1:.2e}, T=1024 CFG 7 {e7:.2e}, T=6144 {e6:.2e}  (north_star: 1e-3)")
    finally:
        full_dit.set_gemm_dtype(SUITE.gemm_dtype)


@pytest.mark.parametrize("mode", ["fp8", "fp8-all"])
def test_fp8_full_width_slice_vs_matched_oracle(dev, mode):
    """Config 5 at full WIDTH: a 2-layer slice of the SA-Open DiT (D=1536, 24 heads, FF 6144, cond 768) in fp8 mode against the
    oracle that quantises at the same points (oracle.dit.Fp8Rounding): gate 5e-3, as for the reduced model."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from oracle import dit as odit
    from stable_audio_tools import synthetic
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.dit import DiffusionTransformer
    kw = dict(cases.FULL_DIT, depth=2)
    with _init.skip_init():
        dit = DiffusionTransformer(**kw)
    sd = synthetic.synth_state_dict(dit.state_dict(), 5)
    dit.load_state_dict(sd)
    dit = dit.to(dev).eval().set_gemm_dtype(mode)
    x, t, c, g = cases.dit_inputs(2, 200, 768, 1536, 3)
    got = dit(x.to(dev), t.to(dev), cross_attn_cond=c.to(dev), global_embed=g.to(dev), cfg_scale=1.0)
    want_m = odit.dit_forward(sd, x, t, c, g, 2, 24, rnd=odit.Fp8Rounding(odit.FP8_DEFAULT_FAMILIES if mode == "fp8" else odit.FP8_FAMILIES))
    want_f = odit.dit_forward(sd, x, t, c, g, 2, 24)
    e_m = assert_close("fp8 full-width slice vs matched fp8 oracle", got, want_m, 5e-3)
    e_f = assert_close("fp8 full-width slice vs fp32 oracle", got, want_f, 2e-2)
    print(f"\n[{mode} full-width 2-layer slice] rel-L2 vs matched {e_m:.2e}, vs fp32 {e_f:.2e}")


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_full_size_decoder_vs_reference_golden(dev, fmt):
    """BASELINE config 1 shape: full-size Oobleck decoder, z[1,64,43] -> [1,2,88064], vs the reference's fp32 output."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from stable_audio_tools import synthetic
    from stable_audio_tools.models import _init
    from stable_audio_tools.models.autoencoders import OobleckDecoder, OobleckEncoder
    g = cases.load("vae")
    with _init.skip_init():
        dec = OobleckDecoder(**cases.vae_kwargs(cases.FULL_VAE, True))
    dec.load_state_dict(synthetic.synth_state_dict(dec.state_dict(), 0))
    got = dec.to(dev).set_gemm_dtype(fmt)(synthetic.synth_input("z_full", (1, 64, 43), 1).to(dev))
    e = assert_close("full-size decode vs reference", got, g["full_decode_T43"], CODEC[fmt][1])
    with _init.skip_init():
        enc = OobleckEncoder(**cases.vae_kwargs(cases.FULL_VAE, False))
    enc.load_state_dict(synthetic.synth_state_dict(enc.state_dict(), 0))
    got = enc.to(dev).set_gemm_dtype(fmt)(synthetic.synth_input("a_full", (1, 2, 2048 * 16), 2, 0.3).to(dev))
    e2 = assert_close("full-size encode vs reference", got, g["full_encode_T16"], CODEC[fmt][1])
    print(f"\n[full codec, {fmt}] decode rel-L2 {e:.2e}, encode rel-L2 {e2:.2e} vs the reference's fp32 output")


def test_chunked_codec_and_audio_to_audio(dev, small_dit, small_vae):
    """AudioAutoencoder.encode_audio / decode_audio / reconstruct_audio with chunking + Bartlett overlap-add (the
    reconstruct_audios.py path, autoencoders.py:410-645) and generate_diffusion_cond with init_audio (variation path,
    generation.py:170-217) against the oracle, VAE noise injected."""
    from oracle import dit as odit, oobleck as oob, sampler as osamp
    from stable_audio_tools import synthetic
    cfg, vae, sd = small_vae
    ratio = cfg["model"]["downsampling_ratio"]
    strides = cfg["model"]["decoder"]["config"]["strides"]
    dsd, esd = _sub(sd, "decoder."), _sub(sd, "encoder.")
    dec = lambda z: oob.oobleck_decoder(dsd, z, strides=strides, rnd=SUITE.round)
    # chunked decode == oracle chunked decode; and agrees with the un-chunked decode away from the seams
    z = synthetic.synth_input("zc", (2, 64, 23), 51)
    got = vae.decode_audio(z.to(dev), chunked=True, chunk_size=8, overlap=2, max_batch_size=3)
    want = oob.decode_audio_chunked(dec, z, 8, 2, ratio)
    assert got.shape == want.shape == (2, 2, 23 * ratio)
    assert_close("decode_audio chunked", got, want, T(1e-2))
    # chunked encode with injected VAE noise: patch the bottleneck's draw through the `noise=` hook per call
    audio = synthetic.synth_input("ac", (1, 2, 19 * ratio), 52, 0.3)
    noises = [synthetic.synth_input(f"vn{i}", (2, 64, 8), 60 + i) for i in range(4)]
    calls = {"i": 0}
    orig_encode = vae.bottleneck.encode

    def encode_with_noise(x, return_info=False, **kw):
        nz = noises[calls["i"]][: x.shape[0]].to(x.device)
        calls["i"] += 1
        return orig_encode(x, return_info=return_info, noise=nz)

    vae.bottleneck.encode = encode_with_noise
    try:
        got = vae.encode_audio(audio.to(dev), chunked=True, chunk_size=8, overlap=2, max_batch_size=2)
        n_calls = calls["i"]
        calls["i"] = 0
        rec = vae.reconstruct_audio(audio.to(dev), chunked=True, chunk_size=8, overlap=2, max_batch_size=2)
    finally:
        vae.bottleneck.encode = orig_encode
    # oracle: same chunk batching (max_batch_size=2) and the same noise per call
    it = {"i": 0}

    def enc_chunks(chunks):
        outs = []
        for i in range(0, len(chunks), 2):
            grp = torch.cat(chunks[i:i + 2], dim=0)
            ms = oob.oobleck_encoder(esd, grp, strides=cfg["model"]["encoder"]["config"]["strides"], rnd=SUITE.round)
            outs += list(oob.vae_sample(ms, noises[it["i"]][: grp.shape[0]]).split(1, dim=0))
            it["i"] += 1
        return outs

    import math
    import torch.nn.functional as F
    cs, hop = 8 * ratio, 6 * ratio
    n_chunk = int(math.ceil((audio.shape[-1] - cs) / hop)) + 1
    padded = F.pad(audio, (0, cs + hop * (n_chunk - 1) - audio.shape[-1]))
    zs = iter(enc_chunks([padded[..., i * hop: i * hop + cs] for i in range(n_chunk)]))
    want = oob.encode_audio_chunked(lambda c: next(zs), audio, 8, 2, ratio, 64)
    assert n_calls == it["i"]
    assert_close("encode_audio chunked", got, want, T(1.5e-2))
    it["i"] = 0
    n_chunk_r = n_chunk                      # reconstruct pads with hop * n_chunk (reference quirk) but slices n_chunk chunks
    padded = F.pad(audio, (0, cs + hop * n_chunk_r - audio.shape[-1]))
    zs2 = enc_chunks([padded[..., i * hop: i * hop + cs] for i in range(n_chunk_r)])
    outs = iter([dec(zz) for zz in zs2])
    want = oob.reconstruct_audio_chunked(lambda c, i: next(outs), audio, 8, 2, ratio)
    assert rec.shape == audio.shape
    assert_close("reconstruct_audio chunked", rec, want, T(2e-2))

    # audio-to-audio (init_audio, init_noise_level): encode -> x = init + noise*sigma_max' -> sample -> latents
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    cfgd, model, sdd = small_dit
    dc = cfgd["model"]["diffusion"]["config"]
    pr = cfgd["model"]["pretransform"]["config"]
    ratio_d = pr["downsampling_ratio"]
    t_len, steps, b = 16, 4, 2
    init = synthetic.synth_input("init", (2, t_len * ratio_d - 100), 70, 0.3)           # shorter than target: zero-padded
    prompt = synthetic.synth_input("prompt_a2a", (b, 128, dc["cond_token_dim"]), 71)
    cond = model.conditioner([{"seconds_start": 0, "seconds_total": 5}] * b)
    cond["prompt"] = (prompt.to(dev), torch.ones(b, 128, device=dev))
    cond = {k: cond[k] for k in ("prompt", "seconds_start", "seconds_total")}
    noise = synthetic.synth_input("noise_a2a", (b, 64, t_len), 72)
    vnoise = synthetic.synth_input("vn_a2a", (1, 64, t_len), 73)
    step_noise = [synthetic.synth_input(f"sn_a2a{i}", (b, 64, t_len), 80 + i) for i in range(steps)]
    itn = iter(step_noise)
    bn = model.pretransform.model.bottleneck
    orig = bn.encode
    bn.encode = lambda x, return_info=False, **kw: orig(x, return_info=return_info, noise=vnoise.to(x.device))
    try:
        lat = generate_diffusion_cond(model, steps=steps, cfg_scale=7.0, conditioning_tensors=cond, sample_size=t_len * ratio_d, seed=1,
                                      device=str(dev), init_audio=(44100, init), init_noise_level=4.0, return_latents=True,
                                      sampler_type="dpmpp-3m-sde", sigma_min=0.3, sigma_max=500, noise=noise,
                                      noise_sampler=lambda s, sn: next(itn).to(dev))
    finally:
        bn.encode = orig
    padded = F.pad(init, (0, 100)).unsqueeze(0)
    esd2 = _sub(sdd, "pretransform.model.encoder.")
    ms = oob.oobleck_encoder(esd2, padded, strides=pr["encoder"]["config"]["strides"], rnd=SUITE.round)
    z0 = oob.vae_sample(ms, vnoise).repeat(b, 1, 1)
    ci = model.get_conditioning_inputs(cond)
    cac, gc = ci["cross_attn_cond"].cpu().float(), ci["global_cond"].cpu().float()
    sig = osamp.get_sigmas_polyexponential(steps, 0.3, 4.0, 1.0)          # sigma_max <- init_noise_level (generation.py:214-217)
    dsd2 = _sub(sdd, "model.model.")
    fn = lambda xin, tt: odit.dit_forward(dsd2, xin, tt, cac, gc, dc["depth"], dc["num_heads"], cfg_scale=7.0, rnd=_matched())
    want = osamp.sample_dpmpp_3m_sde(lambda x, s: osamp.vdenoise(fn, x, s), z0 + noise * sig[0], sig, lambda i, s, sn: step_noise[i])
    assert_close("audio-to-audio latents", lat, want, T(2e-2))

    # inpainting through the public entry point (generation.py:195-213): cut & paste of the init latents + soft mask; the
    # region the last step's binary mask keeps (mask <= 1: everything) must come out as init + renoise * sigma_last
    from stable_audio_tools.inference.generation import build_mask
    margs = dict(cropfrom=0, pastefrom=25, pasteto=100, maskstart=25, maskend=75, softnessL=10, softnessR=10, marination=0)
    itn2 = iter(step_noise)
    renoise = [synthetic.synth_input(f"rn_a2a{i}", (b, 64, t_len), 90 + i) for i in range(steps)]
    bn.encode = lambda x, return_info=False, **kw: orig(x, return_info=return_info, noise=vnoise.to(x.device))
    try:
        lat_in = generate_diffusion_cond(model, steps=steps, cfg_scale=7.0, conditioning_tensors=cond, sample_size=t_len * ratio_d,
                                         seed=1, device=str(dev), init_audio=(44100, init), mask_args=margs, return_latents=True,
                                         sampler_type="dpmpp-2m-sde", sigma_min=0.3, sigma_max=50, noise=noise,
                                         noise_sampler=lambda s, sn: next(itn2).to(dev), inpaint_noise=lambda i: renoise[i].to(dev))
    finally:
        bn.encode = orig
    cut = torch.zeros_like(z0)
    pf, cl = int(0.25 * t_len), t_len - int(0.25 * t_len)
    cut[:, :, pf:pf + cl] = z0[:, :, :cl]
    sig_in = osamp.get_sigmas_polyexponential(steps, 0.3, 50.0, 1.0)
    x0, cb = osamp.inpainting_start_and_callback(cut, noise * sig_in[0], build_mask(t_len, margs), steps, lambda i: renoise[i])
    want_in = osamp.sample_dpmpp_2m_sde(lambda x, s: osamp.vdenoise(fn, x, s), x0.clone(), sig_in, lambda i, s, sn: step_noise[i], callback=cb)
    assert_close("inpainting latents (generate_diffusion_cond)", lat_in, want_in, T(2e-2))
