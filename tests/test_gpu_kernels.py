"""Per-kernel parity: HIP kernels (through the C ABI) vs plain fp32 torch statements of the same
op / the oracle, on seeded inputs.  Tolerances: bf16-operand kernels are compared with a reference
fed the SAME bf16-rounded operands and fp32 accumulation, so only accumulation order and the final
bf16 store differ: rel-L2 <= 1e-3 for fp32 outputs, <= 4e-3 for bf16 outputs (one rounding, 2^-9)."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import FORMATS, assert_close, bf16_round, rel_l2

pytestmark = pytest.mark.gpu


def _lib():
    from stable_audio_tools import _hip
    return _hip, _hip.lib()


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("m,d", [(2050, 1536), (37, 256), (5, 768)])
def test_layernorm(dev, m, d, fmt):
    _hip, lib = _lib()
    x = _rand((m, d), 1, 2.0) + 0.3
    g = 0.5 + 0.1 * _rand((d,), 2)
    b = 0.05 * _rand((d,), 3)
    want = F.layer_norm(x, (d,), g, b, eps=1e-5)
    xd, gd, bd = x.to(dev), g.to(dev), b.to(dev)
    y = torch.empty((m, d), dtype=fmt.dtype, device=dev)
    _hip.check(fmt.fn(lib, "sat_layernorm_bf16")(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(y), m, d, _hip.stream()))
    assert_close("layernorm", y, want, fmt.tol(4e-3))
    assert_close("layernorm-vs-rounded", y, fmt.round(want), fmt.tol(1e-3))


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
def test_cast_bf16(dev, fmt):
    _hip, lib = _lib()
    x = _rand((1000003,), 4)
    if fmt.f16:       # range policy of the fp16 build: saturate at +-65504, keep subnormals, flush below 2^-25
        x[:8] = torch.tensor([7e4, -7e4, 65504.0, 65520.0, 1e30, 3e-6, 2e-8, -1e-9])
    xd = x.to(dev)
    y = torch.empty(x.shape, dtype=fmt.dtype, device=dev)
    _hip.check(fmt.fn(lib, "sat_cast_bf16")(_hip.ptr(xd), _hip.ptr(y), x.numel(), _hip.stream()))
    assert torch.equal(y.cpu(), x.clamp(-65504.0, 65504.0).to(fmt.dtype) if fmt.f16 else x.to(fmt.dtype))


# the shipped tile configurations (gemm_bf16.hip: launch_epi); 0 = the launcher's own choice
# 80: the 256x256 tile on 8 waves with the 8-phase schedule (gemm_ph8.hip) -- what variant 0 picks whenever it picks that tile
# 80 | 0x10000 (fp32 output only): the same with the remainder round split along K -- slabs in caller-supplied workspace + ph8_reduce_f32_kernel --
# forced here, the launcher only splits long reductions behind a whole round; goes through the *_ws entry points
GEMM_VARIANTS = [1, 5, 15, 16, 22, 30, 80]
SPLIT = 80 | 0x10000
# 44: tile 15 with a 4-stage ring (fp32 output, long K); 49: the 128 x 128 tile on two K-groups of 64 x 64 waves (what variant 0 picks for
# the fp32-output GEMMs that fit one round of workgroups: to_out / FF-out at one prompt)
GEMM_F32_VARIANTS = GEMM_VARIANTS + [SPLIT, 44, 49]


def _split_ws(dev, m, n, k, variant):
    """workspace of the K-split (sat_gemm_f32_workspace_bytes) -> (tensor or None, bytes)"""
    import ctypes
    _hip, lib = _lib()
    need = ctypes.c_size_t()
    _hip.check(lib.sat_gemm_f32_workspace_bytes(m, n, k, variant, ctypes.byref(need)))
    if not need.value:
        return None, 0
    return torch.empty(need.value, dtype=torch.uint8, device=dev), need.value


def _skip_tile(variant, n, k):
    v = variant & 0xff
    if v in (22, 80) and n % 256:
        pytest.skip("256-column tile needs n % 256 == 0")
    if v == 80 and k % 128:
        pytest.skip("the 8-phase tile walks K in pairs of 64-wide tiles")
    if v == 30 and n % 192:
        pytest.skip("192-column tile needs n % 192 == 0")
    if v >= 9 and k < 256:
        pytest.skip("the deep-prefetch variants need K >= stages * BK")
    if v == 49 and k % 128:
        pytest.skip("two K-groups walk K in slices of 2 x 64")


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("variant", GEMM_F32_VARIANTS)
@pytest.mark.parametrize("m,n,k", [(2050, 1536, 1536), (130, 256, 128), (1, 512, 64), (257, 768, 6144)])
def test_gemm_f32(dev, variant, m, n, k, fmt):
    _skip_tile(variant, n, k)
    _hip, lib = _lib()
    a = _rand((m, k), 5).to(fmt.dtype)
    # asymmetric weights so that a transposed / mis-indexed tile cannot pass
    w = (_rand((n, k), 6) * 0.05 + torch.linspace(-0.02, 0.03, n)[:, None]).to(fmt.dtype)
    bias = _rand((n,), 7)
    c0 = _rand((m, n), 8)
    want = a.float() @ w.float().T + bias + c0
    ad, wd, bd, cd = a.to(dev), w.to(dev), bias.to(dev), c0.to(dev)
    if variant & 0x10000:
        ws, ws_bytes = _split_ws(dev, m, n, k, variant)
        assert ws_bytes == 256 * 65536 * 4
        _hip.check(fmt.fn(lib, "sat_gemm_bf16_f32_ws")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(cd), m, n, k, 1, variant, _hip.ptr(ws),
                                                       ws_bytes, _hip.stream()))
        # a forced split without scratch is refused, not silently run whole (a single tile with a single K unit has nothing to split)
        rc = -4 if (m, n, k) == (130, 256, 128) else fmt.fn(lib, "sat_gemm_bf16_f32")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(torch.empty_like(cd)), m, n, k, 1, variant, _hip.stream())
        assert rc == -4, f"forced K-split without workspace returned {rc}, expected SAT_E_WORKSPACE"
    else:
        _hip.check(fmt.fn(lib, "sat_gemm_bf16_f32")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(cd), m, n, k, 1, variant, _hip.stream()))
    assert_close(f"gemm v{variant} {m}x{n}x{k}", cd, want, 1e-3 if not fmt.f16 else 1e-5)      # same operands: fp32 summation order only


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("m", [300, 770])
@pytest.mark.parametrize("variant", [0] + GEMM_VARIANTS)
def test_gemm_swiglu(dev, variant, m, fmt):
    _hip, lib = _lib()
    k, inner = 256, 768
    _skip_tile(variant, 2 * inner, k)
    a = _rand((m, k), 9).to(fmt.dtype)
    w = _rand((2 * inner, k), 10) * 0.08
    bias = _rand((2 * inner,), 11) * 0.1
    h = F.linear(a.float(), fmt.round(w), bias)
    val, gate = h.chunk(2, dim=-1)
    want = val * F.silu(gate)
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    wp = torch.empty((2 * inner, k), dtype=fmt.dtype, device=dev)
    bp = torch.empty((2 * inner,), dtype=torch.float32, device=dev)
    out = torch.empty((m, inner), dtype=fmt.dtype, device=dev)
    _hip.check(fmt.fn(lib, "sat_gemm_swiglu_bf16")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(wp), _hip.ptr(bp), _hip.ptr(out), m,
                                                   2 * inner, k, variant, _hip.stream()))
    assert_close("swiglu", out, want, fmt.tol(4e-3))


def _vt_perm(n):
    """sat_common.h: vt_pos -- V^T stores every aligned group of 16 keys as [0-3, 8-11, 4-7, 12-15] (bits 2 and 3 of the key index
    swapped; an involution), the order in which the second attention MFMA consumes a lane's probabilities."""
    p = torch.arange(n)
    return (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1)


def _pad_heads(x, s_pad, key_side=False):
    # [B,H,S,64] -> zero-padded [B,H,s_pad,64]; key-side tensors (K, V) of sequence b start at row (b*S) & 3
    b, h, s, d = x.shape
    out = torch.zeros((b, h, s_pad, d), dtype=x.dtype)
    for i in range(b):
        ob = (i * s) & 3 if key_side else 0
        out[i, :, ob:ob + s] = x[i]
    return out


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("prescaled", [False, True])
@pytest.mark.parametrize("b,h,kvh,sq,sk", [(2, 4, 4, 1025, 1025), (1, 4, 2, 300, 130), (2, 2, 2, 64, 64), (1, 2, 1, 129, 7), (8, 64, 8, 300, 700),
                                          (2, 24, 24, 1025, 1025)])          # the last: one prompt with CFG at full size = 240 single-range workgroups (round 5's layout rule)
def test_attention(dev, b, h, kvh, sq, sk, prescaled, fmt):
    """prescaled: the layout the DiT plan runs -- Q carries log2(e)/8 (written so by the QKV epilogue); every shape with <= 512 keys or
    >= 1024 workgroups of 256 queries then takes the single-KV-group kernel whose softmax reference rides in the matrix pipe."""
    from oracle import dit as odit
    _hip, lib = _lib()
    if b * h > 16 and not prescaled:
        pytest.skip("the 1024-workgroup grid is there for the pre-scaled single-group kernel on a long key range (11 tiles)")
    q = (_rand((b, h, sq, 64), 12) * 1.5).to(fmt.dtype)
    k = (_rand((b, kvh, sk, 64), 13) * 1.5).to(fmt.dtype)
    v = _rand((b, kvh, sk, 64), 14).to(fmt.dtype)
    # spike one key against one query so that the running max jumps late in the sequence (rescale branch)
    k[0, 0, sk - 1] = q[0, 0, min(5, sq - 1)] * 3
    fn = fmt.fn(lib, "sat_attention_bf16")
    if prescaled:
        c = 0.125 * 1.4426950408889634
        q = (q.float() * c).to(fmt.dtype)               # what the producer stores ...
        fn = fmt.fn(lib, "sat_attention_prescaled_bf16")
        q_eff = q.float() / c                           # ... and the query it stands for
    else:
        q_eff = q.float()
    want = odit._merge(odit.attention_core(q_eff, k.float(), v.float(), rnd=fmt.round))
    sq_pad = (sq + 127) // 128 * 128
    sk_pad = (sk + 3 + 63) // 64 * 64
    qd = _pad_heads(q, sq_pad).to(dev)
    kd = _pad_heads(k, sk_pad, key_side=True).to(dev)
    vtd = _pad_heads(v, sk_pad, key_side=True).transpose(2, 3)[..., _vt_perm(sk_pad)].contiguous().to(dev)
    out = torch.empty((b * sq, h * 64), dtype=fmt.dtype, device=dev)
    _hip.check(fn(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vtd), _hip.ptr(out), b, h, kvh, sq, sk, sq_pad, sk_pad, _hip.stream()))
    assert_close(f"attention {b}x{h}x{sq}x{sk}", out.view(b, sq, h * 64), want, fmt.tol(5e-3))
    exact = odit._merge(odit.attention_core(q_eff, k.float(), v.float()))
    assert rel_l2(out.view(b, sq, h * 64), exact) < fmt.tol(1e-2)


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_all_scores_strongly_negative(dev, prescaled, fmt):
    """a query whose scores are ALL far below zero (log2-domain scores < -126).  The standing softmax reference of the
    single-group pre-scaled kernel starts at 0; the sequence's first block has to SET it to the true maximum instead of clamping it
    at >= 0, or every p flushes to 0 and the row divides by a zero sum.  Queries 0..31 anti-aligned with every key (logits ~ -150),
    the rest ordinary; the second sequence tile holds the row's maximum (the reference has to move up, once)."""
    from oracle import dit as odit
    _hip, lib = _lib()
    b, h, kvh, sq, sk = 1, 2, 2, 96, 200
    base = F.normalize(_rand((64,), 300), dim=0)
    k = (base[None, None, None, :] * 12.0 + 0.03 * _rand((b, kvh, sk, 64), 301)).to(fmt.dtype)
    q = (_rand((b, h, sq, 64), 302) * 1.5)
    q[:, :, :32] = -base * 100.0 + 0.03 * _rand((b, h, 32, 64), 303)           # q . k / 8 ~ -150: log2-domain ~ -216
    q = q.to(fmt.dtype)
    k[0, :, 150] = k[0, :, 150] * 0.62                                          # the negative rows' maximum (~ -93) sits in the third KV tile
    v = _rand((b, kvh, sk, 64), 304).to(fmt.dtype)
    fn = fmt.fn(lib, "sat_attention_bf16")
    q_eff = q.float()
    if prescaled:
        c = 0.125 * 1.4426950408889634
        q = (q.float() * c).to(fmt.dtype)
        fn = fmt.fn(lib, "sat_attention_prescaled_bf16")
        q_eff = q.float() / c
    scores = torch.einsum("bhid,bhjd->bhij", q_eff, k.float()) / 8
    assert scores[:, :, :32].max().item() < -88, "the first 32 queries must have every logit below -87 (2^-126 in the log2 domain)"
    want = odit._merge(odit.attention_core(q_eff, k.float(), v.float(), rnd=fmt.round))
    sq_pad, sk_pad = 128, 256
    qd = _pad_heads(q, sq_pad).to(dev)
    kd = _pad_heads(k, sk_pad, key_side=True).to(dev)
    vtd = _pad_heads(v, sk_pad, key_side=True).transpose(2, 3)[..., _vt_perm(sk_pad)].contiguous().to(dev)
    out = torch.empty((b * sq, h * 64), dtype=fmt.dtype, device=dev)
    _hip.check(fn(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vtd), _hip.ptr(out), b, h, kvh, sq, sk, sq_pad, sk_pad, _hip.stream()))
    assert_close("attention, all-negative rows", out.view(b, sq, h * 64)[:, :32], want[:, :32], 2e-2 if not fmt.f16 else 5e-3)
    assert_close("attention, ordinary rows", out.view(b, sq, h * 64)[:, 32:], want[:, 32:], fmt.tol(5e-3))


def test_fp16_range_policy(dev):
    """gemm_dtype = 3 range policy (include/sat_hip.h): residual-stream rows up to +-3e4 go through the LayerNorm-fold producer and
    consumer at full fp16 relative precision; rows beyond +-65504 SATURATE in the fp16 image (finite output, statistics of the
    saturated row) instead of turning the row into inf / NaN."""
    fmt = FORMATS[1]
    _hip, lib = _lib()
    m, d, inner = 300, 768, 768
    cd, xb, part = _ln_fold_producer(dev, m, d, 256, 0, seed=400, fmt=fmt, x_scale=1.0e4)       # |x| up to ~4e4
    assert cd.abs().max().item() > 3.0e4
    w = _rand((2 * inner, d), 401) * 0.08
    gamma = 0.8 + 0.2 * _rand((d,), 402)
    beta = 0.1 * _rand((d,), 403)
    bias = 0.1 * _rand((2 * inner,), 404)
    wd, gd, bd, bbd = w.to(dev), gamma.to(dev), beta.to(dev), bias.to(dev)
    wp = torch.empty((2 * inner, d), dtype=fmt.dtype, device=dev)
    c12 = torch.empty((4 * inner,), dtype=torch.float32, device=dev)
    out = torch.full((m, inner), float("nan"), dtype=fmt.dtype, device=dev)
    _hip.check(lib.sat_gemm_swiglu_ln_f16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(bbd), _hip.ptr(wp),
                                          _hip.ptr(c12), _hip.ptr(out), m, 2 * inner, d, 0, _hip.stream()))
    val, gate = F.linear(F.layer_norm(cd.cpu(), (d,), gamma, beta, eps=1e-5), w, bias).chunk(2, dim=-1)
    assert_close("fp16 fold on rows of magnitude 3e4 vs fp32 LayerNorm + Linear", out, val * F.silu(gate), 2.5e-3)
    # beyond the range: saturation, not infinity
    x = _rand((64, d), 405) * 3.0e4
    x[:, 7] = 2.0e5
    x[:, 9] = -1.0e6
    y = torch.empty((64, d), dtype=torch.float16, device=dev)
    xd = x.to(dev)
    _hip.check(lib.sat_cast_f16(_hip.ptr(xd), _hip.ptr(y), x.numel(), _hip.stream()))
    assert torch.isfinite(y).all() and y[:, 7].eq(65504).all() and y[:, 9].eq(-65504).all()
    a = torch.full((64, 256), 200.0, dtype=torch.float16)
    wq = torch.full((d, 256), 2.0, dtype=torch.float16)
    c0 = torch.zeros((64, d))
    xb2 = torch.empty((64, d), dtype=torch.float16, device=dev)
    part2 = torch.empty((64, d // 64, 2), dtype=torch.float32, device=dev)
    ad, wqd, c0d = a.to(dev), wq.to(dev), c0.to(dev)
    _hip.check(lib.sat_gemm_resid_ln_f16(_hip.ptr(ad), _hip.ptr(wqd), None, _hip.ptr(c0d), _hip.ptr(xb2), _hip.ptr(part2), 64, d, 256, 0, _hip.stream()))
    assert c0d.eq(102400.0).all(), "fp32 output: exact"
    assert xb2.eq(65504).all(), "the fp16 image of an over-range row saturates"
    assert torch.allclose(part2[..., 0].cpu(), torch.full((64, d // 64), 64 * 65504.0)), "row statistics are those of the saturated image"
    # below the normal range: fp16 subnormal operands (|x| < 2^-14) are multiplied as they are, not flushed, by the fp16 MFMAs
    for variant in (15, 80):
        a = torch.full((256, 256), 2.0 ** -20, dtype=torch.float16)
        a[:, ::2] = 3 * 2.0 ** -24
        wq = torch.full((256, 256), 4.0, dtype=torch.float16)
        cz = torch.zeros((256, 256))
        ad, wqd, czd = a.to(dev), wq.to(dev), cz.to(dev)
        _hip.check(lib.sat_gemm_f16_f32(_hip.ptr(ad), _hip.ptr(wqd), None, _hip.ptr(czd), 256, 256, 256, 0, variant, _hip.stream()))
        assert torch.equal(czd.cpu(), a.float() @ wq.float().T), f"tile {variant}: subnormal fp16 operands were flushed by the MFMA"


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("b,s,d,kvh,sk", [(1, 1025, 256, 2, 130), (3, 300, 256, 4, 7), (2, 129, 256, 1, 189), (1, 50, 1536, 12, 130)])
def test_cross_attention_fused(dev, b, s, d, kvh, sk, fmt):
    """to_q projection + cross-attention core in one launch (the 128 x 64 GEMM tile keeps Q in registers and attends to the context keys
    staged in LDS; transformer.py:430-437 + 496-536) against the oracle on the query the epilogue stands for (pre-scaled by log2(e)/8,
    one bf16 rounding).  s = 300 / 129: tiles and waves whose rows straddle two sequences (second pass on the next sequence's keys)."""
    from oracle import dit as odit
    _hip, lib = _lib()
    h = d // 64
    a = _rand((b * s, d), 31).to(fmt.dtype)
    wq = (_rand((d, d), 32) * (1.5 / d ** 0.5)).to(fmt.dtype)
    k = (_rand((b, kvh, sk, 64), 33) * 1.5).to(fmt.dtype)
    v = _rand((b, kvh, sk, 64), 34).to(fmt.dtype)
    c = 0.125 * 1.4426950408889634
    q = (a.float() @ wq.float().T).view(b, s, h, 64).permute(0, 2, 1, 3)
    q_eff = (q * c).to(fmt.dtype).float() / c
    want = odit._merge(odit.attention_core(q_eff, k.float(), v.float(), rnd=fmt.round))
    sk_pad = (sk + 3 + 63) // 64 * 64
    kd = _pad_heads(k, sk_pad, key_side=True).to(dev)
    vtd = _pad_heads(v, sk_pad, key_side=True).transpose(2, 3)[..., _vt_perm(sk_pad)].contiguous().to(dev)
    out = torch.zeros((b * s, d), dtype=fmt.dtype, device=dev)
    ad, wd = a.to(dev), wq.to(dev)          # (named: a temporary would be freed before the launch reads it)
    _hip.check(fmt.fn(lib, "sat_cross_attention_fused_bf16")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(kd), _hip.ptr(vtd), _hip.ptr(out), b, s, d,
                                                             kvh, sk, sk_pad, _hip.stream()))
    assert_close(f"fused cross-attention {b}x{s}x{d} kv{kvh} sk{sk}", out.view(b, s, d), want, fmt.tol(5e-3))


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("s,s_pad", [(197, 256), (385, 512)])
@pytest.mark.parametrize("variant", [0] + GEMM_VARIANTS)
def test_qkv_rope(dev, variant, s, s_pad, fmt):
    from oracle import dit as odit
    _hip, lib = _lib()
    b, d = 2, 256
    _skip_tile(variant, 3 * d, d)
    h = d // 64
    a = _rand((b * s, d), 15).to(fmt.dtype)
    w = (_rand((3 * d, d), 16) * 0.1).to(fmt.dtype)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    qkv = (a.float() @ w.float().T).view(b, s, 3 * d)
    q, k, v = (odit._heads(t, h) for t in qkv.chunk(3, dim=-1))
    freqs = odit.rotary_freqs(inv_freq, s)
    q, k = odit.apply_rotary(q, freqs), odit.apply_rotary(k, freqs)
    ad, wd, fd = a.to(dev), w.to(dev), inv_freq.to(dev)
    qd = torch.full((b, h, s_pad, 64), float("nan"), dtype=fmt.dtype, device=dev)
    kd = torch.full_like(qd, float("nan"))
    vtd = torch.full((b, h, 64, s_pad), float("nan"), dtype=fmt.dtype, device=dev)
    scratch = torch.empty((2 * s * 16,), dtype=torch.float32, device=dev)
    _hip.check(fmt.fn(lib, "sat_qkv_rope_bf16")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(fd), _hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vtd),
                                                _hip.ptr(scratch), b, s, s_pad, d, variant, _hip.stream()))
    assert_close("rope q", qd[:, :, :s], q, fmt.tol(4e-3))
    vtd = vtd[..., _vt_perm(s_pad).to(vtd.device)]      # undo the key permutation of the V^T layout
    for i in range(b):     # key-side tensors of sequence i start at row/column (i*s) & 3
        ob = (i * s) & 3
        assert_close("rope k", kd[i, :, ob:ob + s], k[i], fmt.tol(4e-3))
        assert_close("v^T", vtd[i, :, :, ob:ob + s], v[i].transpose(1, 2), fmt.tol(4e-3))
        assert (kd[i, :, :ob] == 0).all() and (kd[i, :, ob + s:] == 0).all(), "K pads must be zero"
        assert (vtd[i, :, :, :ob] == 0).all() and (vtd[i, :, :, ob + s:] == 0).all(), "V^T pads must be zero"
    assert (qd[:, :, s:] == 0).all(), "Q pads must be zero"


def _ln_fold_producer(dev, m, d, k, variant, seed=40, fmt=FORMATS[0], x_scale=2.0):
    """x0 + a w^T + bias through the producer entry; returns (c fp32, xb bf16 / fp16, ln_part) on the device after checking them."""
    _hip, lib = _lib()
    a = _rand((m, k), seed).to(fmt.dtype)
    w = (_rand((d, k), seed + 1) * 0.05 + torch.linspace(-0.02, 0.03, d)[:, None]).to(fmt.dtype)
    bias = _rand((d,), seed + 2)
    x0 = _rand((m, d), seed + 3, x_scale) + 0.3       # rows with a mean: the fold has to subtract it
    want = a.float() @ w.float().T + bias + x0
    ad, wd, bd, cd = a.to(dev), w.to(dev), bias.to(dev), x0.to(dev)
    xb = torch.full((m, d), float("nan"), dtype=fmt.dtype, device=dev)
    part = torch.full((m, d // 64, 2), float("nan"), dtype=torch.float32, device=dev)
    if variant & 0x10000:
        ws, ws_bytes = _split_ws(dev, m, d, k, variant)
        _hip.check(fmt.fn(lib, "sat_gemm_resid_ln_bf16_ws")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(cd), _hip.ptr(xb), _hip.ptr(part), m, d, k,
                                                            variant, _hip.ptr(ws), ws_bytes, _hip.stream()))
    else:
        _hip.check(fmt.fn(lib, "sat_gemm_resid_ln_bf16")(_hip.ptr(ad), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(cd), _hip.ptr(xb), _hip.ptr(part), m, d, k,
                                                         variant, _hip.stream()))
    assert_close(f"ln-fold producer v{variant}", cd, want, 1e-3)
    assert torch.equal(xb, cd.clamp(-65504.0, 65504.0).to(fmt.dtype)), "xb must be the 16-bit rounding (fp16: saturating) of the fp32 rows just written"
    blocks = xb.float().view(m, d // 64, 64).double()
    assert_close("ln-fold partial sums", part[..., 0], blocks.sum(-1), 1e-5)
    assert_close("ln-fold partial squares", part[..., 1], (blocks * blocks).sum(-1), 1e-5)
    return cd, xb, part


def _ln_fold_reference(xb, w, gamma, beta, bias, fmt=FORMATS[0]):
    """What the consumer computes, in fp64 on the same rounded operands: rstd (xb (gamma.w)^T - mean c1) + c2."""
    x = xb.double().cpu()
    mean = x.mean(-1, keepdim=True)
    var = (x * x).mean(-1, keepdim=True) - mean * mean
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    wp = fmt.round(gamma * w).double()
    c2 = (w.double() * beta.double()).sum(-1) + (bias.double() if bias is not None else 0.0)
    return (rstd * (x @ wp.T - mean * wp.sum(-1)) + c2).float()


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("m", [300, 770])
@pytest.mark.parametrize("prod,cons", [(15, 22), (16, 30), (22, 15), (30, 16), (44, 22), (49, 15), (0, 0), (80, 80), (15, 80), (80, 30), (SPLIT, 80)])
def test_ln_fold_swiglu(dev, prod, cons, m, fmt):
    """LayerNorm folded into FF-in (sat_dit_cfg.ln_fold): producer epilogue -> bf16 rows + partial sums -> SwiGLU GEMM that finishes
    the normalisation.  Gates: 4e-3 against the same arithmetic in fp64 (one bf16 rounding of the output), 1e-2 against the plain fp32
    LayerNorm -> Linear -> SwiGLU of the reference (transformer.py:700, 222, 232-235; adds the bf16 rounding of the operands)."""
    _hip, lib = _lib()
    d, inner = 768, 768
    cd, xb, part = _ln_fold_producer(dev, m, d, 256, prod, fmt=fmt)
    w = _rand((2 * inner, d), 50) * 0.08
    gamma = 0.8 + 0.2 * _rand((d,), 51)
    beta = 0.1 * _rand((d,), 52)
    bias = 0.1 * _rand((2 * inner,), 53)
    wd, gd, bd, bbd = w.to(dev), gamma.to(dev), beta.to(dev), bias.to(dev)
    wp = torch.empty((2 * inner, d), dtype=fmt.dtype, device=dev)
    c12 = torch.empty((4 * inner,), dtype=torch.float32, device=dev)
    out = torch.full((m, inner), float("nan"), dtype=fmt.dtype, device=dev)
    _hip.check(fmt.fn(lib, "sat_gemm_swiglu_ln_bf16")(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(bbd),
                                                      _hip.ptr(wp), _hip.ptr(c12), _hip.ptr(out), m, 2 * inner, d, cons, _hip.stream()))
    val, gate = _ln_fold_reference(xb, w, gamma, beta, bias, fmt).chunk(2, dim=-1)
    assert_close("ln-fold swiglu vs same arithmetic", out, val * F.silu(gate), fmt.tol(4e-3))
    val, gate = F.linear(F.layer_norm(cd.cpu(), (d,), gamma, beta, eps=1e-5), w, bias).chunk(2, dim=-1)
    assert_close("ln-fold swiglu vs fp32 LayerNorm + Linear", out, val * F.silu(gate), fmt.tol(1e-2))


@pytest.mark.parametrize("row_mean,outlier", [(0.0, 0.0), (8.0, 0.0), (30.0, 0.0), (0.0, 60.0)])
def test_ln_fold_rows_with_common_mode(dev, row_mean, outlier):
    """the fold feeds un-normalised bf16(x) to the MFMA and subtracts mean * c1 afterwards, so its rounding error scales with
    |x| instead of |x - mean|: rows with a common-mode offset (or a few outlier channels -- real checkpoints have both) lose precision
    by ~sqrt(1 + mean^2 / var).  This test measures that on a residual stream with a chosen row mean / outlier channel, for the fold
    AND for the three-kernel plan (`set_layernorm_fusion(False)`: layernorm_kernel -> bf16 -> GEMM), against the fp32 reference
    LayerNorm -> Linear -> SwiGLU, and gates the fold at the predicted noise ratio (2x headroom)."""
    _hip, lib = _lib()
    m, d, inner = 300, 768, 768
    x = _rand((m, d), 90) + row_mean
    if outlier:
        x[:, 5] += outlier
        x[:, 300] -= outlier
    w = _rand((2 * inner, d), 91) * 0.08
    gamma = 0.8 + 0.2 * _rand((d,), 92)
    beta = 0.1 * _rand((d,), 93)
    bias = 0.1 * _rand((2 * inner,), 94)
    val, gate = F.linear(F.layer_norm(x, (d,), gamma, beta, eps=1e-5), w, bias).chunk(2, dim=-1)
    want = val * F.silu(gate)
    xd, wd, gd, bd, bbd = x.to(dev), w.to(dev), gamma.to(dev), beta.to(dev), bias.to(dev)
    # fold: bf16 image + partial statistics (what a producer epilogue writes), then the consumer
    xb = xd.to(torch.bfloat16)
    blocks = xb.float().view(m, d // 64, 64)
    part = torch.stack([blocks.sum(-1), (blocks * blocks).sum(-1)], -1).contiguous()
    wp = torch.empty((2 * inner, d), dtype=torch.bfloat16, device=dev)
    c12 = torch.empty((4 * inner,), dtype=torch.float32, device=dev)
    out_f = torch.full((m, inner), float("nan"), dtype=torch.bfloat16, device=dev)
    _hip.check(lib.sat_gemm_swiglu_ln_bf16(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(bbd), _hip.ptr(wp),
                                           _hip.ptr(c12), _hip.ptr(out_f), m, 2 * inner, d, 0, _hip.stream()))
    # three-kernel plan: LayerNorm kernel (fp32 statistics of the fp32 rows, bf16 output) -> SwiGLU GEMM
    y = torch.empty((m, d), dtype=torch.bfloat16, device=dev)
    _hip.check(lib.sat_layernorm_bf16(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(y), m, d, _hip.stream()))
    wp2 = torch.empty((2 * inner, d), dtype=torch.bfloat16, device=dev)
    bp2 = torch.empty((2 * inner,), dtype=torch.float32, device=dev)
    out_s = torch.full((m, inner), float("nan"), dtype=torch.bfloat16, device=dev)
    _hip.check(lib.sat_gemm_swiglu_bf16(_hip.ptr(y), _hip.ptr(wd), _hip.ptr(bbd), _hip.ptr(wp2), _hip.ptr(bp2), _hip.ptr(out_s), m, 2 * inner, d, 0,
                                        _hip.stream()))
    e_f, e_s = rel_l2(out_f, want), rel_l2(out_s, want)
    var = x.var(-1, unbiased=False).mean().item()
    ratio = (1.0 + (x.mean(-1) ** 2).mean().item() / var) ** 0.5 if not outlier else (x.pow(2).mean().item() / _rand((m, d), 90).var().item()) ** 0.5
    print(f"\n[ln-fold common mode] row mean {row_mean}, outlier {outlier}: fold {e_f:.2e}, three-kernel plan {e_s:.2e}, predicted noise ratio {ratio:.1f}")
    assert e_s <= 1e-2, f"three-kernel plan: {e_s:.3e}"
    assert e_f <= 1e-2 * max(1.0, ratio) * 2, f"fold: {e_f:.3e} exceeds 2 x 1e-2 x the predicted ratio {ratio:.1f}"


@pytest.mark.parametrize("fmt", FORMATS, ids=repr)
@pytest.mark.parametrize("s,s_pad", [(197, 256), (385, 512), (12, 128)])
@pytest.mark.parametrize("prod,cons", [(15, 30), (16, 22), (22, 16), (30, 15), (49, 30), (0, 0), (80, 80), (16, 80), (80, 15), (SPLIT, 80)])
def test_ln_fold_qkv_rope(dev, prod, cons, s, s_pad, fmt):
    """LayerNorm folded into to_qkv + RoPE + head split (transformer.py:692, 314, 430-452): q / k through the transposed epilogue,
    V^T through the un-swapped one -- both have to apply the per-row statistics.  The 8-phase kernel (cons = 80) walks the rotation from row
    block to row block by angle addition and re-reads the table where a lane crosses into the next sequence (s = 197, 385: inside a tile) or
    always (s = 12: sequences shorter than a row block)."""
    if s == 12 and (prod, cons) not in ((80, 80), (0, 0), (16, 80), (15, 30)):
        pytest.skip("short-sequence case: one producer / consumer pair per kernel family is enough")
    from oracle import dit as odit
    _hip, lib = _lib()
    b, d = (2, 768) if s != 12 else (5, 768)
    h = d // 64
    cd, xb, part = _ln_fold_producer(dev, b * s, d, 256, prod, seed=60, fmt=fmt)
    w = _rand((3 * d, d), 70) * 0.06
    gamma = 0.8 + 0.2 * _rand((d,), 71)
    beta = 0.1 * _rand((d,), 72)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    freqs = odit.rotary_freqs(inv_freq, s)

    def split(qkv):
        q, k, v = (odit._heads(t, h) for t in qkv.view(b, s, 3 * d).chunk(3, dim=-1))
        return odit.apply_rotary(q, freqs), odit.apply_rotary(k, freqs), v

    same = split(_ln_fold_reference(xb, w, gamma, beta, None, fmt))
    plain = split(F.linear(F.layer_norm(cd.cpu(), (d,), gamma, beta, eps=1e-5), w))
    wd, gd, bd, fd = w.to(dev), gamma.to(dev), beta.to(dev), inv_freq.to(dev)
    wp = torch.empty((3 * d, d), dtype=fmt.dtype, device=dev)
    c12 = torch.empty((6 * d,), dtype=torch.float32, device=dev)
    qd = torch.full((b, h, s_pad, 64), float("nan"), dtype=fmt.dtype, device=dev)
    kd = torch.full_like(qd, float("nan"))
    vtd = torch.full((b, h, 64, s_pad), float("nan"), dtype=fmt.dtype, device=dev)
    scratch = torch.empty((2 * s * 16,), dtype=torch.float32, device=dev)
    _hip.check(fmt.fn(lib, "sat_qkv_rope_ln_bf16")(_hip.ptr(xb), _hip.ptr(part), _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(wp),
                                                   _hip.ptr(c12), _hip.ptr(fd), _hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vtd), _hip.ptr(scratch),
                                                   b, s, s_pad, d, cons, _hip.stream()))
    vtd = vtd[..., _vt_perm(s_pad).to(vtd.device)]
    for (q, k, v), tol in ((same, fmt.tol(4e-3)), (plain, fmt.tol(1e-2))):
        assert_close("ln-fold q", qd[:, :, :s], q, tol)
        for i in range(b):
            ob = (i * s) & 3
            assert_close("ln-fold k", kd[i, :, ob:ob + s], k[i], tol)
            assert_close("ln-fold v^T", vtd[i, :, :, ob:ob + s], v[i].transpose(1, 2), tol)
    assert (qd[:, :, s:] == 0).all(), "Q pads must be zero"


def test_snake_vae_int16(dev):
    from oracle import oobleck as oob
    _hip, lib = _lib()
    x = _rand((2, 5, 333), 17, 2.0)
    al, be = _rand((5,), 18, 0.3), _rand((5,), 19, 0.3)
    want = oob.snake_beta(x, al, be)
    xd, ad, bd = x.to(dev), al.to(dev), be.to(dev)
    y = torch.empty_like(xd)
    _hip.check(lib.sat_snake_beta(_hip.ptr(xd), _hip.ptr(ad), _hip.ptr(bd), _hip.ptr(y), 2, 5, 333, _hip.stream()))
    assert_close("snake", y, want, 1e-5)

    ms = _rand((2, 8, 50), 20)
    nz = _rand((2, 4, 50), 21)
    want = oob.vae_sample(ms, nz)
    z = torch.empty((2, 4, 50), device=dev)
    msd, nzd = ms.to(dev), nz.to(dev)     # keep the device tensors alive across the call
    _hip.check(lib.sat_vae_sample(_hip.ptr(msd), _hip.ptr(nzd), _hip.ptr(z), 2, 4, 50, _hip.stream()))
    assert_close("vae_sample", z, want, 1e-6)

    from stable_audio_tools.utils.audio_utils import float_to_int16_audio
    for scale in (0.3, 2.5):
        a = _rand((2, 4099), 22, scale)
        for maximize in (False, True):
            div = a.abs().max().item()
            if not maximize:
                div = max(div, 1.0)
            want = a.div(div).mul(32767).to(torch.int16)
            got = float_to_int16_audio(a.to(dev), maximize=maximize)
            assert got.dtype == torch.int16 and got.device.type == "cpu"
            assert torch.equal(got, want), f"int16 quantisation not bit-exact: {(got != want).sum().item()} samples differ"
    # the reference's own outputs (tests/golden/ops.npz, utils/audio_utils.py:21-26 run in the build container): bit-exact
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from stable_audio_tools import synthetic
    gold = cases.load("ops")
    a = synthetic.synth_input("i16", (2, 4099), 108, 0.7)
    for key, x, maximize in (("int16_quiet", a, False), ("int16_loud", a * 4, False), ("int16_max", a, True)):
        got = float_to_int16_audio(x.to(dev), maximize=maximize)
        assert torch.equal(got, gold[key].to(torch.int16)), f"{key}: differs from the reference's int16 output"


def test_number_conditioner_hip(dev):
    """NumberConditioner on the C ABI (sat_number_embed: clamp, normalise, Fourier features, Linear) against the REFERENCE's output
    (tests/golden/ops.npz: number_cond, conditioners.py:64-102 run in the build container) and against the oracle on other values."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from oracle import conditioners as ocond
    from stable_audio_tools import synthetic
    from stable_audio_tools.models.conditioners import NumberConditioner
    nc = NumberConditioner(768, min_val=0, max_val=512)
    sd = synthetic.synth_state_dict(nc.state_dict(), 1)
    nc.load_state_dict(sd)
    nc = nc.to(dev)
    gold = cases.load("ops")
    emb, mask = nc([0.0, 47.5, 600.0, -3.0])
    assert emb.shape == (4, 1, 768) and emb.device.type == "cuda"
    assert_close("number conditioner vs reference", emb, gold["number_cond"], 1e-5)
    assert torch.equal(mask.cpu(), gold["number_mask"])
    vals = [1.25, 511.9, 300.0, 47.0, 0.001]
    assert_close("number conditioner vs oracle", nc(vals)[0], ocond.number_conditioner(sd, "", vals, 0, 512)[0], 1e-5)


def test_cfg_combine_and_sampler_update(dev):
    from oracle import sampler as osamp
    _hip, lib = _lib()
    b, c, t = 2, 8, 77
    mo = _rand((2 * b, c, t), 23)
    cond, uncond = mo[:b], mo[b:]
    for scale_phi in (0.0, 0.6):
        cfg = uncond + (cond - uncond) * 7.0
        if scale_phi:
            cfg = scale_phi * (cfg * (cond.std(dim=1, keepdim=True) / cfg.std(dim=1, keepdim=True))) + (1 - scale_phi) * cfg
        out = torch.empty((b, c, t), device=dev)
        mod = mo.to(dev)
        _hip.check(lib.sat_cfg_combine(_hip.ptr(mod), _hip.ptr(out), b, c, t, 7.0, scale_phi, _hip.stream()))
        assert_close(f"cfg combine phi={scale_phi}", out, cfg, 1e-5)

    # the fused (a,b,c1,c2,cn) update must reproduce the multistep form of the oracle sampler on a linear denoiser
    from stable_audio_tools.inference.sampling import dpmpp3m_coefficients, get_sigmas_polyexponential
    steps = 12
    sig = get_sigmas_polyexponential(steps, 0.3, 500.0, 1.0)
    assert torch.allclose(torch.tensor(sig), osamp.get_sigmas_polyexponential(steps, 0.3, 500.0, 1.0))
    coeffs = dpmpp3m_coefficients(sig)
    x0 = _rand((b, c, t), 24) * sig[0]
    target = _rand((b, c, t), 25)
    noises = [_rand((b, c, t), 100 + i) for i in range(steps)]

    def den(x, sigma):   # a smooth, sigma-dependent "denoiser"
        s = sigma.view(-1, 1, 1)
        return target + (x - target) / (1 + s * s) + 0.1 * torch.tanh(x / (1 + s))

    want = osamp.sample_dpmpp_3m_sde(den, x0.clone(), torch.tensor(sig), lambda i, s, sn: noises[i])
    x = x0.clone().to(dev)
    d, d1, d2 = (torch.empty_like(x) for _ in range(3))
    have = 0
    for i in range(steps):
        d.copy_(den(x.cpu(), torch.full((b,), sig[i])).to(dev))
        a_, b_, c1, c2, cn = coeffs[i]
        nz = noises[i].to(dev) if cn != 0 else None
        _hip.check(lib.sat_dpmpp3m_update(_hip.ptr(x), _hip.ptr(d), _hip.ptr(d1) if have >= 1 else None,
                                          _hip.ptr(d2) if have >= 2 else None, _hip.ptr(nz), a_, b_, c1, c2, cn, x.numel(),
                                          _hip.stream()))
        d, d1, d2 = d2, d, d1
        have = min(have + 1, 2)
    assert_close("dpmpp3m trajectory", x, want, 2e-5)


def _deq8(q_bytes, scale):
    return q_bytes.view(torch.float8_e4m3fn).to(torch.float32) * scale[:, None]


def test_fp8_quant_and_layernorm(dev):
    """sat_quant_rows_fp8 / sat_layernorm_fp8 (BASELINE config 5 building blocks) against torch's e4m3 conversion: the bytes must
    be identical except where x / scale sits on a rounding boundary (kernel multiplies by 1/scale), and never off by more than
    one code."""
    from oracle import dit as odit
    _hip, lib = _lib()
    for rows, k in ((37, 256), (300, 1536)):
        x = _rand((rows, k), 200, 3.0)
        x[1] = 0
        xd = x.to(dev)
        q = torch.empty((rows, k), dtype=torch.uint8, device=dev)
        sc = torch.empty((rows,), dtype=torch.float32, device=dev)
        _hip.check(lib.sat_quant_rows_fp8(_hip.ptr(xd), _hip.ptr(q), _hip.ptr(sc), rows, k, _hip.stream()))
        amax = x.abs().amax(dim=1)
        want_sc = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
        assert torch.allclose(sc.cpu(), want_sc, rtol=1e-6)
        got = _deq8(q.cpu(), sc.cpu())
        assert_close("fp8 rows", got, odit.fp8_rows(x), 2e-3)
        assert (got - x).norm() / x.norm() < 4e-2                      # e4m3: 3 mantissa bits
        assert (q.cpu()[1] == 0).all() and sc.cpu()[1] == 1.0
    m, d = 130, 1536
    x = _rand((m, d), 201, 2.0)
    gm, bt = 1 + 0.1 * _rand((d,), 202), 0.1 * _rand((d,), 203)
    xd, gd, bd = x.to(dev), gm.to(dev), bt.to(dev)
    q = torch.empty((m, d), dtype=torch.uint8, device=dev)
    sc = torch.empty((m,), dtype=torch.float32, device=dev)
    _hip.check(lib.sat_layernorm_fp8(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(q), _hip.ptr(sc), m, d, _hip.stream()))
    want = odit.fp8_rows(F.layer_norm(x, (d,), gm, bt))
    assert_close("layernorm fp8", _deq8(q.cpu(), sc.cpu()), want, 3e-3)


@pytest.mark.parametrize("plain", [0, 256])
@pytest.mark.parametrize("variant", [0, 15, 16, 22, 30])
@pytest.mark.parametrize("m,n,k", [(2050, 1536, 1536), (257, 768, 6144), (130, 256, 384), (1, 768, 256)])
def test_gemm_fp8(dev, variant, m, n, k, plain):
    """e4m3 x e4m3 -> fp32 GEMM with per-row scales on both operands against an fp32 matmul of the de-quantised operands
    (exact products, fp32 accumulation): only the accumulation order differs.  plain=0: v_mfma_scale_f32_32x32x64_f8f6f4 with
    unit block scales (2x MFMA rate, the default); plain=256: v_mfma_f32_32x32x16_fp8_fp8."""
    if (variant & 0xff) in (22,) and n % 256:
        pytest.skip("256-column tile")
    if (variant & 0xff) == 30 and n % 192:
        pytest.skip("192-column tile")
    if (variant & 0xff) in (15, 16) and k < 384:
        pytest.skip("3-stage tiles need K >= 384")
    if (variant & 0xff) in (22, 30) and k < 256:
        pytest.skip("2-stage tiles need K >= 256")
    _hip, lib = _lib()
    a = _rand((m, k), 210, 2.0)
    w = _rand((n, k), 211) * 0.05 + torch.linspace(-0.02, 0.03, n)[:, None]
    bias = _rand((n,), 212)
    c0 = _rand((m, n), 213)
    ad, wd = a.to(dev), w.to(dev)
    a8 = torch.empty((m, k), dtype=torch.uint8, device=dev)
    w8 = torch.empty((n, k), dtype=torch.uint8, device=dev)
    sa = torch.empty((m,), dtype=torch.float32, device=dev)
    sw = torch.empty((n,), dtype=torch.float32, device=dev)
    _hip.check(lib.sat_quant_rows_fp8(_hip.ptr(ad), _hip.ptr(a8), _hip.ptr(sa), m, k, _hip.stream()))
    _hip.check(lib.sat_quant_rows_fp8(_hip.ptr(wd), _hip.ptr(w8), _hip.ptr(sw), n, k, _hip.stream()))
    want = _deq8(a8.cpu(), sa.cpu()) @ _deq8(w8.cpu(), sw.cpu()).T + bias + c0
    cd, bd = c0.to(dev), bias.to(dev)
    _hip.check(lib.sat_gemm_fp8_f32(_hip.ptr(a8), _hip.ptr(sa), _hip.ptr(w8), _hip.ptr(sw), _hip.ptr(bd), _hip.ptr(cd), m, n, k, 1,
                                    variant | plain, _hip.stream()))
    assert_close(f"gemm fp8 v{variant} {m}x{n}x{k}", cd, want, 1e-4)
    full = a @ w.T + bias + c0
    assert rel_l2(cd, full) < 5e-2, "e4m3 operands: a few percent from the fp32 product"


def _deq_mx(q_bytes, e8m0):
    sc = torch.pow(2.0, e8m0.to(torch.float64) - 127.0).float()
    return (q_bytes.view(torch.float8_e4m3fn).to(torch.float32).reshape(q_bytes.shape[0], -1, 32) * sc[:, :, None]).reshape(q_bytes.shape)


@pytest.mark.parametrize("variant", [0, 15, 16, 22, 30])
@pytest.mark.parametrize("m,n,k", [(2050, 1536, 6144), (257, 768, 1536), (130, 256, 384), (1, 768, 256)])
def test_gemm_mxfp8(dev, variant, m, n, k):
    """MXFP8 A operand (e4m3 + one E8M0 scale per 32 k, applied by v_mfma_scale_f32_32x32x64_f8f6f4 itself; the scales ride
    through the LDS ring as one more 4-byte LDS-DMA per row) x per-channel e4m3 W -> fp32: the producer against the oracle's
    block quantiser, the GEMM against an fp32 matmul of the de-quantised operands."""
    from oracle import dit as odit
    if variant == 22 and n % 256:
        pytest.skip("256-column tile")
    if variant == 30 and n % 192:
        pytest.skip("192-column tile")
    if variant in (15, 16) and k < 384:
        pytest.skip("3-stage tiles need K >= 384")
    _hip, lib = _lib()
    a = _rand((m, k), 220, 2.0) * torch.logspace(-2, 1, k)[None, :]      # block amax spans three decades
    a[0, :64] = 0
    w = _rand((n, k), 221) * 0.05 + torch.linspace(-0.02, 0.03, n)[:, None]
    bias, c0 = _rand((n,), 222), _rand((m, n), 223)
    ad, wd = a.to(dev), w.to(dev)
    a8 = torch.empty((m, k), dtype=torch.uint8, device=dev)
    asc = torch.empty((m, k // 32), dtype=torch.uint8, device=dev)
    w8 = torch.empty((n, k), dtype=torch.uint8, device=dev)
    sw = torch.empty((n,), dtype=torch.float32, device=dev)
    _hip.check(lib.sat_quant_mx_rows_fp8(_hip.ptr(ad), _hip.ptr(a8), _hip.ptr(asc), m, k, _hip.stream()))
    _hip.check(lib.sat_quant_rows_fp8(_hip.ptr(wd), _hip.ptr(w8), _hip.ptr(sw), n, k, _hip.stream()))
    a_deq = _deq_mx(a8.cpu(), asc.cpu())
    assert_close("mxfp8 producer vs oracle", a_deq, odit.mxfp8_blocks(a), 1e-3)
    assert (asc.cpu()[0, :2] == 0).all() and (a8.cpu()[0, :64] == 0).all()
    want = a_deq @ _deq8(w8.cpu(), sw.cpu()).T + bias + c0
    cd, bd = c0.to(dev), bias.to(dev)
    _hip.check(lib.sat_gemm_mxfp8_f32(_hip.ptr(a8), _hip.ptr(asc), _hip.ptr(w8), _hip.ptr(sw), _hip.ptr(bd), _hip.ptr(cd), m, n, k, 1,
                                      variant, _hip.stream()))
    assert_close(f"gemm mxfp8 v{variant} {m}x{n}x{k}", cd, want, 1e-4)
