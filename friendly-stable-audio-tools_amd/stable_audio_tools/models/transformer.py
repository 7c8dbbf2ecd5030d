"""``ContinuousTransformer`` and its parts -- state-dict compatible parameter containers.

Mirror of the module tree of the reference's ``models/transformer.py`` (``LayerNorm`` :188-206,
``GLU``/``FeedForward`` :211-287, ``Attention`` :290-554, ``TransformerBlock`` :594-702,
``RotaryEmbedding`` :99-155, ``ContinuousTransformer`` :705-809) restricted to the options
the shipped DiT configs use.  These classes only *hold* parameters under the reference's
names; the forward pass of the whole stack is one C-ABI call (``sat_dit_forward`` /
``sat_dit_denoise_cfg``) issued by ``models/dit.py``.  Unsupported options raise.
"""
import torch
from torch import nn

from . import _init


class LayerNorm(nn.Module):
    def __init__(self, dim, bias=False, fix_scale=False):
        super().__init__()
        if fix_scale:
            self.register_buffer("gamma", torch.ones(dim))
        else:
            self.gamma = nn.Parameter(torch.ones(dim))
        if bias:
            self.beta = nn.Parameter(torch.zeros(dim))
        else:
            self.register_buffer("beta", torch.zeros(dim))


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, base=10000):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)


class GLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = _init.linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, zero_init_output=True, **unsupported):
        super().__init__()
        if unsupported:
            raise NotImplementedError(f"FeedForward options not supported by the HIP path: {sorted(unsupported)}")
        inner_dim = int(dim * mult)
        linear_out = _init.linear(inner_dim, dim, zero=zero_init_output)
        self.ff = nn.Sequential(GLU(dim, inner_dim), nn.Identity(), linear_out, nn.Identity())


class Attention(nn.Module):
    def __init__(self, dim, dim_heads=64, dim_context=None, causal=False, zero_init_output=True, **unsupported):
        super().__init__()
        if causal or unsupported:
            raise NotImplementedError(f"Attention options not supported by the HIP path: causal={causal} {sorted(unsupported)}")
        if dim_heads != 64:
            raise NotImplementedError("the HIP attention kernel is built for dim_heads == 64")
        self.dim, self.dim_heads = dim, dim_heads
        dim_kv = dim_context if dim_context else dim
        self.num_heads = dim // dim_heads
        self.kv_heads = dim_kv // dim_heads
        if dim_context:
            self.to_q = _init.linear(dim, dim, bias=False)
            self.to_kv = _init.linear(dim_kv, dim_kv * 2, bias=False)
        else:
            self.to_qkv = _init.linear(dim, dim * 3, bias=False)
        self.to_out = _init.linear(dim, dim, bias=False, zero=zero_init_output)


class TransformerBlock(nn.Module):
    def __init__(self, dim, dim_heads=64, cross_attend=False, dim_context=None, global_cond_dim=None, causal=False,
                 zero_init_branch_outputs=True, conformer=False, layer_ix=-1, remove_norms=False, attn_kwargs={}, ff_kwargs={},
                 norm_kwargs={}):
        super().__init__()
        if conformer or remove_norms:
            raise NotImplementedError("conformer / remove_norms are not supported by the HIP path")
        self.dim, self.dim_heads, self.cross_attend, self.dim_context = dim, dim_heads, cross_attend, dim_context
        self.layer_ix = layer_ix
        self.pre_norm = LayerNorm(dim, **norm_kwargs)
        self.self_attn = Attention(dim, dim_heads=dim_heads, causal=causal, zero_init_output=zero_init_branch_outputs, **attn_kwargs)
        if cross_attend:
            self.cross_attend_norm = LayerNorm(dim, **norm_kwargs)
            self.cross_attn = Attention(dim, dim_heads=dim_heads, dim_context=dim_context, causal=causal,
                                        zero_init_output=zero_init_branch_outputs, **attn_kwargs)
        self.ff_norm = LayerNorm(dim, **norm_kwargs)
        self.ff = FeedForward(dim, zero_init_output=zero_init_branch_outputs, **ff_kwargs)
        self.global_cond_dim = global_cond_dim
        if global_cond_dim:
            # adaLN (reference transformer.py:650-656): SiLU -> Linear(global_cond_dim, 6*dim, no bias), zero-initialised;
            # the HIP plan stacks the 24 weights and evaluates them in one launch per forward
            if global_cond_dim != dim:
                raise NotImplementedError("adaLN: the HIP plan expects global_cond_dim == dim (DiffusionTransformer always passes embed_dim)")
            self.to_scale_shift_gate = nn.Sequential(nn.SiLU(), _init.linear(global_cond_dim, dim * 6, bias=False, zero=True))


class ContinuousTransformer(nn.Module):
    def __init__(self, dim, depth, *, dim_in=None, dim_out=None, dim_heads=64, cross_attend=False, cond_token_dim=None,
                 global_cond_dim=None, causal=False, rotary_pos_emb=True, zero_init_branch_outputs=True, conformer=False,
                 use_sinusoidal_emb=False, use_abs_pos_emb=False, abs_pos_emb_max_length=10000, **kwargs):
        super().__init__()
        if not rotary_pos_emb or use_sinusoidal_emb or use_abs_pos_emb:
            raise NotImplementedError("only rotary positional embedding is supported by the HIP path")
        self.dim, self.depth, self.causal = dim, depth, causal
        self.project_in = _init.linear(dim_in, dim, bias=False) if dim_in else nn.Identity()
        self.project_out = _init.linear(dim, dim_out, bias=False) if dim_out else nn.Identity()
        self.rotary_pos_emb = RotaryEmbedding(max(dim_heads // 2, 32))
        self.layers = nn.ModuleList([
            TransformerBlock(dim, dim_heads=dim_heads, cross_attend=cross_attend, dim_context=cond_token_dim,
                             global_cond_dim=global_cond_dim, causal=causal, zero_init_branch_outputs=zero_init_branch_outputs,
                             conformer=conformer, layer_ix=i, **kwargs)
            for i in range(depth)
        ])

    def forward(self, *args, **kwargs):
        raise RuntimeError("ContinuousTransformer has no standalone forward in this build: the stack runs inside "
                           "DiffusionTransformer (one sat_dit_forward call); there is no CPU/eager path")
