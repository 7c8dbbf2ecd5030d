"""Oracle: DiffusionTransformer / ContinuousTransformer forward (fp32, functional).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function cites the reference
file:line it restates (paths relative to /root/reference/stable_audio_tools/).

``sd`` is a state dict with the reference's key names *relative to* the
``DiffusionTransformer`` module (i.e. ``model.model.`` stripped): ``timestep_features.weight``,
``to_timestep_embed.{0,2}.{weight,bias}``, ``to_cond_embed.{0,2}.weight``,
``to_global_embed.{0,2}.weight``, ``preprocess_conv.weight``, ``postprocess_conv.weight``,
``transformer.project_in.weight``, ``transformer.project_out.weight``,
``transformer.rotary_pos_emb.inv_freq``, ``transformer.layers.N.*``.

``rnd`` is the *matched-rounding* hook: ``None`` = pure fp32 (the reference's CPU path);
``bf16_round`` = round tensors to bf16 at exactly the points where the HIP kernels store
bf16 (GEMM operands, q/k/v, P, attention output, FF hidden).  Accumulation stays fp32.
"""
import math

import torch
import torch.nn.functional as F


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def fp16_round(x):
    """IEEE fp16 rounding of the fp16 build (gemm_dtype = 3): round to nearest even, SATURATING at +-65504 (the kernels run with
    MODE.FP16_OVFL set), subnormals kept."""
    return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


def _r(rnd, x):
    return x if rnd is None else rnd(x)


def fp8_rows(x):
    """Per-row (last dim) OCP e4m3 quantisation as the fp8_gemm kernels do it: scale = amax / 448, value * (1 / scale) rounded
    to nearest even -- returned de-quantised.  (Activations: one scale per token; weights: one per output channel.)"""
    amax = x.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    return (x * (1.0 / scale)).to(torch.float8_e4m3fn).to(torch.float32) * scale


def mxfp8_blocks(x, block=32):
    """MXFP8 as the fp8_gemm SwiGLU epilogue writes it: every ``block`` consecutive elements of the last dim share a power-of-two
    scale 2^ceil(log2(amax / 448)) (E8M0), values are e4m3 -- returned de-quantised."""
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // block, block)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    t = amax * (1.0 / 448.0)
    # ceil(log2(t)) through the float representation, exactly as the kernel does it (no rounding issues of log2 at powers of two)
    tb = t.contiguous().view(torch.int32)
    e_bits = ((tb >> 23) & 0xff) - 127 + ((tb & 0x7fffff) != 0).to(torch.int32)
    e = torch.where(amax > 0, e_bits.clamp(-127, 127).float(), torch.full_like(amax, -127.0))
    scale = torch.pow(2.0, e.double()).float()
    q = (xb * torch.pow(2.0, -e.double()).float()).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * scale).reshape(shp)


FP8_FAMILIES = ("qkv", "cq", "ff1", "ff2", "o")      # to_qkv, cross to_q, FF-in (LayerNorm-fed: per-token scales); FF-out, to_out (MXFP8 A operand)
FP8_DEFAULT_FAMILIES = ("cq", "ff1", "ff2")          # SAT_FP8_DEFAULT: what gemm_dtype "fp8" quantises; FP8_FAMILIES = "fp8-all" (SAT_FP8_ALL)


class Fp8Rounding:
    """Matched-rounding hook of BASELINE config 5 (sat_dit_cfg.gemm_dtype = 1): bf16 everywhere, except that the GEMM families named in
    ``families`` take e4m3 operands -- the LayerNorm outputs and the weights of the GEMMs they feed ("qkv", "cq", "ff1") with per-row
    scales; FF-out ("ff2") takes the SwiGLU output as MXFP8 (block-32 power-of-two scales) and per-channel e4m3 weights; the attention
    outputs feed the to_out projections ("o", self and cross) as MXFP8 too.  Default = the plan's "fp8" (cross to_q, FF-in, FF-out);
    ``families=FP8_FAMILIES`` = "fp8-all": every GEMM of the blocks except the per-generation to_kv of the context.  A family left out
    keeps bf16 operands."""

    def __init__(self, families=FP8_DEFAULT_FAMILIES):
        unknown = set(families) - set(FP8_FAMILIES)
        if unknown:
            raise ValueError(f"unknown fp8 GEMM families {sorted(unknown)}")
        self.families = frozenset(families)

    def __call__(self, x):
        return bf16_round(x)

    def _pick(self, fam, q, x):
        return q(x) if fam in self.families else bf16_round(x)

    def act(self, x, fam="qkv"):               # a LayerNorm output: one scale per token
        return self._pick(fam, fp8_rows, x)

    def weight(self, w, fam="qkv"):            # the weight behind it: one scale per output channel
        return self._pick(fam, fp8_rows, w)

    def hidden(self, x):                       # SwiGLU output = A operand of FF-out (hardware block scales)
        return self._pick("ff2", mxfp8_blocks, x)

    def weight2(self, w):                      # FF-out weight, per output channel
        return self._pick("ff2", fp8_rows, w)

    def attn_out(self, x):                     # attention output = A operand of to_out: MXFP8, one scale per half head
        return self._pick("o", mxfp8_blocks, x)

    def weight_o(self, w):                     # to_out weights (self and cross), per output channel
        return self._pick("o", fp8_rows, w)


class _Folded:
    """Un-normalised rows + the affine parameters of the LayerNorm in front of a Linear (LnFoldRounding only)."""

    def __init__(self, x, gamma, beta):
        self.x, self.gamma, self.beta = x, gamma, beta


class LnFoldRounding:
    """Matched-rounding hook of ``sat_dit_cfg.ln_fold`` (the default plan for bf16 / "prepend" models): bf16 at the same points as
    ``bf16_round``, except that a LayerNorm and the Linear behind it (to_qkv from the second block on, cross to_q, FF-in) are
    evaluated the way the fused kernels do it -- the activation is rounded to bf16 BEFORE the normalisation, the statistics are
    those of the rounded rows, gamma is folded into the bf16 weights, beta and the bias into a per-channel constant:
        LN(x) W^T + b  =  rstd (bf16(x) bf16(gamma . W)^T  -  mean rowsum(bf16(gamma . W)))  +  (W beta + b)."""

    ln_fold = True
    round = staticmethod(bf16_round)          # (tests replace it by the identity to pin the ALGEBRA of the fold against the reference)

    def __call__(self, x):
        return self.round(x)

    def folded_linear(self, h, w, bias=None):
        xb = self.round(h.x)
        mean = xb.mean(dim=-1, keepdim=True)
        var = ((xb * xb).mean(dim=-1, keepdim=True) - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + 1e-5)
        wp = self.round(h.gamma * w)
        c2 = F.linear(h.beta, w) if bias is None else F.linear(h.beta, w) + bias
        return rstd * (F.linear(xb, wp) - mean * wp.sum(dim=-1)) + c2


class LnFoldRoundingF16(LnFoldRounding):
    """The same plan on IEEE fp16 operands (``sat_dit_cfg.gemm_dtype = 3``)."""

    round = staticmethod(fp16_round)


def _norm_for_gemm(rnd, x, gamma, beta, fold=True, fam="qkv"):
    """LayerNorm whose output feeds a GEMM (transformer.py:692, 695, 700), with the operand rounding of ``rnd``; ``fam`` names the
    GEMM family it feeds ("qkv", "cq", "ff1": hooks that treat the families differently, Fp8Rounding)."""
    if fold and getattr(rnd, "ln_fold", False):
        return _Folded(x, gamma, beta)
    return _ra(rnd, layer_norm(x, gamma, beta), fam)


def _lin(rnd, h, w, bias=None, fam="qkv"):
    """The Linear behind a LayerNorm."""
    if isinstance(h, _Folded):
        return rnd.folded_linear(h, w, bias)
    return F.linear(h, _rw(rnd, w, fam), bias)


def _ra(rnd, x, fam="qkv"):      # a LayerNorm output that feeds a GEMM
    return rnd.act(x, fam) if hasattr(rnd, "act") else _r(rnd, x)


def _rw(rnd, w, fam="qkv"):      # the weight of a GEMM fed by a LayerNorm output
    return rnd.weight(w, fam) if hasattr(rnd, "weight") else _r(rnd, w)


# models/transformer.py:188-206 (LayerNorm: F.layer_norm with gamma, beta buffer, eps 1e-5)
def layer_norm(x, gamma, beta):
    return F.layer_norm(x, x.shape[-1:], weight=gamma, bias=beta)


# models/transformer.py:130-148 (RotaryEmbedding.forward_from_seq_len / forward)
def rotary_freqs(inv_freq, seq_len):
    t = torch.arange(seq_len, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq.float())
    return torch.cat((freqs, freqs), dim=-1)  # [S, rot_dim]


# models/transformer.py:158-183 (rotate_half / apply_rotary_pos_emb, partial rotary, fp32)
def apply_rotary(t, freqs):
    rot_dim = freqs.shape[-1]
    seq_len = t.shape[-2]
    freqs = freqs[-seq_len:, :]
    t_rot, t_pass = t[..., :rot_dim], t[..., rot_dim:]
    half = rot_dim // 2
    x1, x2 = t_rot[..., :half], t_rot[..., half:]
    rotated = torch.cat((-x2, x1), dim=-1)
    t_rot = t_rot * freqs.cos() + rotated * freqs.sin()
    return torch.cat((t_rot, t_pass), dim=-1)


# models/transformer.py:525-536 (the einsum/softmax(fp32) fallback branch = CPU path)
def attention_core(q, k, v, rnd=None):
    """q [B,H,Nq,dh]; k,v [B,KVH,Nk,dh]; GQA via repeat_interleave (transformer.py:512-515)."""
    h, kvh = q.shape[1], k.shape[1]
    if h != kvh:
        k = k.repeat_interleave(h // kvh, dim=1)
        v = v.repeat_interleave(h // kvh, dim=1)
    scale = 1.0 / math.sqrt(q.shape[-1])
    dots = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    if rnd is None:
        attn = F.softmax(dots, dim=-1, dtype=torch.float32)
        return torch.einsum("bhij,bhjd->bhid", attn, v)
    # matched rounding: P = exp(s - max) is rounded to bf16 before P.V; the normaliser is
    # the fp32 sum of the un-rounded P (what the flash kernel accumulates).
    m = dots.amax(dim=-1, keepdim=True)
    p = torch.exp(dots - m)
    l = p.sum(dim=-1, keepdim=True)
    return torch.einsum("bhij,bhjd->bhid", rnd(p), v) / l


def _heads(t, h):
    b, n, _ = t.shape
    return t.view(b, n, h, -1).permute(0, 2, 1, 3)


def _merge(t):
    b, h, n, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * d)


# models/transformer.py:407-554 (Attention.forward), self-attention branch (to_qkv)
def self_attention(sd, pfx, x, freqs, num_heads, rnd=None):
    q, k, v = _lin(rnd, x, sd[pfx + "to_qkv.weight"]).chunk(3, dim=-1)
    q, k, v = (_heads(t, num_heads) for t in (q, k, v))
    q = apply_rotary(q.float(), freqs)      # transformer.py:438-452
    k = apply_rotary(k.float(), freqs)
    q, k, v = _r(rnd, q), _r(rnd, k), _r(rnd, v)
    out = _merge(attention_core(q, k, v, rnd))
    out = rnd.attn_out(out) if hasattr(rnd, "attn_out") else _r(rnd, out)
    w_o = rnd.weight_o(sd[pfx + "to_out.weight"]) if hasattr(rnd, "weight_o") else _r(rnd, sd[pfx + "to_out.weight"])
    return F.linear(out, w_o)


# models/transformer.py:407-554, cross-attention branch (to_q / to_kv; no RoPE: :438)
def cross_attention(sd, pfx, x, context, num_heads, dim_heads, rnd=None):
    q = _heads(_lin(rnd, x, sd[pfx + "to_q.weight"], fam="cq"), num_heads)
    kv = F.linear(context, _r(rnd, sd[pfx + "to_kv.weight"]))
    k, v = kv.chunk(2, dim=-1)
    kv_heads = k.shape[-1] // dim_heads
    k, v = _heads(k, kv_heads), _heads(v, kv_heads)
    q, k, v = _r(rnd, q), _r(rnd, k), _r(rnd, v)
    out = _merge(attention_core(q, k, v, rnd))
    out = rnd.attn_out(out) if hasattr(rnd, "attn_out") else _r(rnd, out)
    w_o = rnd.weight_o(sd[pfx + "to_out.weight"]) if hasattr(rnd, "weight_o") else _r(rnd, sd[pfx + "to_out.weight"])
    return F.linear(out, w_o)


# models/transformer.py:211-287 (GLU + FeedForward; value = first half, gate = second half)
def feed_forward(sd, pfx, x, rnd=None):
    h = _lin(rnd, x, sd[pfx + "ff.0.proj.weight"], sd[pfx + "ff.0.proj.bias"], fam="ff1")
    val, gate = h.chunk(2, dim=-1)
    h = rnd.hidden(val * F.silu(gate)) if hasattr(rnd, "hidden") else _r(rnd, val * F.silu(gate))
    w2 = rnd.weight2(sd[pfx + "ff.2.weight"]) if hasattr(rnd, "weight2") else _r(rnd, sd[pfx + "ff.2.weight"])
    return F.linear(h, w2, sd[pfx + "ff.2.bias"])


# models/transformer.py:656-702 (TransformerBlock.forward: adaLN branch :665-689, plain branch :691-700)
def transformer_block(sd, pfx, x, context, freqs, num_heads, dim_heads, rnd=None, global_cond=None, first=False):
    if global_cond is not None and (pfx + "to_scale_shift_gate.1.weight") in sd:
        ssg = F.linear(F.silu(global_cond), sd[pfx + "to_scale_shift_gate.1.weight"]).unsqueeze(1)      # :667
        scale_self, shift_self, gate_self, scale_ff, shift_ff, gate_ff = ssg.chunk(6, dim=-1)
        h = layer_norm(x, sd[pfx + "pre_norm.gamma"], sd[pfx + "pre_norm.beta"])
        h = _ra(rnd, h * (1 + scale_self) + shift_self)                                                   # :671-672
        x = x + self_attention(sd, pfx + "self_attn.", h, freqs, num_heads, rnd) * torch.sigmoid(1 - gate_self)   # :673-675
        if context is not None:                                                                          # :677-678 (un-modulated)
            h = _ra(rnd, layer_norm(x, sd[pfx + "cross_attend_norm.gamma"], sd[pfx + "cross_attend_norm.beta"]), "cq")
            x = x + cross_attention(sd, pfx + "cross_attn.", h, context, num_heads, dim_heads, rnd)
        h = layer_norm(x, sd[pfx + "ff_norm.gamma"], sd[pfx + "ff_norm.beta"])
        h = _ra(rnd, h * (1 + scale_ff) + shift_ff, "ff1")                                                        # :685-686
        x = x + feed_forward(sd, pfx + "ff.", h, rnd) * torch.sigmoid(1 - gate_ff)                        # :687-689
        return x
    # (the first block's pre_norm reads rows that no GEMM epilogue wrote: the ln_fold plan keeps it a standalone LayerNorm)
    h = _norm_for_gemm(rnd, x, sd[pfx + "pre_norm.gamma"], sd[pfx + "pre_norm.beta"], fold=not first)
    x = x + self_attention(sd, pfx + "self_attn.", h, freqs, num_heads, rnd)
    if context is not None:
        h = _norm_for_gemm(rnd, x, sd[pfx + "cross_attend_norm.gamma"], sd[pfx + "cross_attend_norm.beta"], fam="cq")
        x = x + cross_attention(sd, pfx + "cross_attn.", h, context, num_heads, dim_heads, rnd)
    h = _norm_for_gemm(rnd, x, sd[pfx + "ff_norm.gamma"], sd[pfx + "ff_norm.beta"], fam="ff1")
    x = x + feed_forward(sd, pfx + "ff.", h, rnd)
    return x


# models/transformer.py:764-809 (ContinuousTransformer.forward)
def continuous_transformer(sd, x, prepend_embeds, context, depth, num_heads, rnd=None, return_hidden=False, global_cond=None):
    pfx = "transformer."
    x = F.linear(x, sd[pfx + "project_in.weight"])
    if prepend_embeds is not None:
        x = torch.cat((prepend_embeds, x), dim=-2)
    dim_heads = x.shape[-1] // num_heads
    freqs = rotary_freqs(sd[pfx + "rotary_pos_emb.inv_freq"], x.shape[1])
    hidden = []
    for i in range(depth):
        x = transformer_block(sd, f"{pfx}layers.{i}.", x, context, freqs, num_heads, dim_heads, rnd, global_cond, first=i == 0)
        if return_hidden:
            hidden.append(x)
    out = F.linear(x, sd[pfx + "project_out.weight"])
    return (out, hidden) if return_hidden else out


# models/blocks.py:88-97 (FourierFeatures)
def fourier_features(weight, inp):
    f = 2 * math.pi * inp @ weight.T
    return torch.cat([f.cos(), f.sin()], dim=-1)


def _mlp(sd, pfx, x, bias):
    x = F.linear(x, sd[pfx + "0.weight"], sd[pfx + "0.bias"] if bias else None)
    x = F.silu(x)
    return F.linear(x, sd[pfx + "2.weight"], sd[pfx + "2.bias"] if bias else None)


# models/dit.py:135-226 (DiffusionTransformer._forward; global_cond_type "prepend" :186-197 or "adaLN" :205-206)
def dit_inner_forward(sd, x, t, cross_attn_cond, global_embed, depth, num_heads, rnd=None, return_hidden=False, adaln=False):
    """x [B,C,T] fp32, t [B], cross_attn_cond [B,Lc,Dc] or None, global_embed [B,Dg] or None."""
    context = None
    if cross_attn_cond is not None:
        context = _r(rnd, _mlp(sd, "to_cond_embed.", cross_attn_cond, bias=False))   # dit.py:150
    if global_embed is not None:
        global_embed = _mlp(sd, "to_global_embed.", global_embed, bias=False)          # dit.py:154
    timestep_embed = _mlp(sd, "to_timestep_embed.", fourier_features(sd["timestep_features.weight"], t[:, None]), bias=True)
    global_embed = timestep_embed if global_embed is None else global_embed + timestep_embed  # dit.py:179-182
    prepend = None if adaln else global_embed.unsqueeze(1)                               # dit.py:185-195
    x = F.conv1d(x, sd["preprocess_conv.weight"]) + x                                    # dit.py:197
    x = x.transpose(1, 2)                                                                # dit.py:199
    out = continuous_transformer(sd, x, prepend, context, depth, num_heads, rnd, return_hidden,
                                 global_cond=global_embed if adaln else None)
    if return_hidden:
        out, hidden = out
    out = out.transpose(1, 2)[:, :, (0 if adaln else 1):]                                # dit.py:219
    out = F.conv1d(out, sd["postprocess_conv.weight"]) + out                             # dit.py:224
    return (out, hidden) if return_hidden else out


# models/dit.py:228-364 (DiffusionTransformer.forward: batched CFG :270-349)
def dit_forward(sd, x, t, cross_attn_cond, global_embed, depth, num_heads, cfg_scale=1.0, scale_phi=0.0,
                negative_cross_attn_cond=None, rnd=None, adaln=False, negative_cross_attn_mask=None):
    if cfg_scale != 1.0 and cross_attn_cond is not None:
        bx = torch.cat([x, x], dim=0)
        bt = torch.cat([t, t], dim=0)
        bg = None if global_embed is None else torch.cat([global_embed, global_embed], dim=0)
        null = torch.zeros_like(cross_attn_cond)
        if negative_cross_attn_cond is not None:                                         # dit.py:294-300
            if negative_cross_attn_mask is not None:
                negative_cross_attn_cond = torch.where(negative_cross_attn_mask.to(torch.bool).unsqueeze(2), negative_cross_attn_cond, null)
            null = negative_cross_attn_cond
        bc = torch.cat([cross_attn_cond, null], dim=0)
        out = dit_inner_forward(sd, bx, bt, bc, bg, depth, num_heads, rnd, adaln=adaln)
        cond, uncond = torch.chunk(out, 2, dim=0)
        cfg = uncond + (cond - uncond) * cfg_scale                                       # dit.py:338-339
        if scale_phi != 0.0:                                                             # dit.py:342-345
            cond_std = cond.std(dim=1, keepdim=True)
            cfg_std = cfg.std(dim=1, keepdim=True)
            return scale_phi * (cfg * (cond_std / cfg_std)) + (1 - scale_phi) * cfg
        return cfg
    return dit_inner_forward(sd, x, t, cross_attn_cond, global_embed, depth, num_heads, rnd, adaln=adaln)
