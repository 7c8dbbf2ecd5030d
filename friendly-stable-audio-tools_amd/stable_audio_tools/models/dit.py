"""``DiffusionTransformer`` (reference ``models/dit.py:14-364``) on the HIP C ABI.

The module holds the reference's parameters (same names/shapes, so reference checkpoints
load) and owns a ``sat_dit_plan``: bf16 re-packed weights, RoPE table and the per-generation
cross-attention K/V cache.  ``forward`` keeps the reference semantics
(``x, t, cross_attn_cond, global_embed, cfg_scale, scale_phi ...`` -> model output with
batched CFG, dit.py:228-364); ``denoise`` is the fused k-diffusion ``VDenoiser`` evaluation
the sampler loop uses (one ``sat_dit_denoise_cfg`` call per step).
"""
import ctypes
import typing as tp

import torch
from torch import nn

from .. import _config, _hip
from . import _init
from .blocks import FourierFeatures
from .transformer import ContinuousTransformer


GEMM_DTYPES = {"bf16": 0, "fp8": 1, "fp8-all": 1, "fp32x": 2, "fp16": 3}      # sat_dit_cfg.gemm_dtype (include/sat_hip.h)
# sat_dit_cfg.fp8_families (SAT_FP8_*): "fp8" = cross to_q + FF-in + FF-out in e4m3 (the subset whose quantisation the 12-step trajectory
# tolerates: tools/fp8_budget.py), "fp8-all" = every GEMM of the blocks (round 3's mode: to_qkv and the to_out projections as well)
FP8_FAMILIES = {"fp8": 2 | 4 | 8, "fp8-all": 31}


class DiffusionTransformer(nn.Module):
    def __init__(self, io_channels: int = 32, patch_size: int = 1, embed_dim: int = 768, cond_token_dim: int = 0,
                 project_cond_tokens: bool = True, global_cond_dim: int = 0, project_global_cond: bool = True,
                 input_concat_dim: int = 0, prepend_cond_dim: int = 0, depth: int = 12, num_heads: int = 8,
                 transformer_type: str = "x-transformers", global_cond_type: str = "prepend", max_seq_len: int = 8192, **kwargs):
        super().__init__()
        if transformer_type != "continuous_transformer":
            raise NotImplementedError("only transformer_type='continuous_transformer' is implemented (the shipped DiT configs)")
        if global_cond_type not in ("prepend", "adaLN"):
            raise ValueError(f"Unknown global_cond_type: {global_cond_type}")
        if patch_size != 1 or input_concat_dim != 0 or prepend_cond_dim != 0:
            raise NotImplementedError("patch_size>1 / input_concat / prepend_cond are outside the supported hot path")
        if not project_global_cond and global_cond_dim > 0:
            raise NotImplementedError("project_global_cond=False is not supported")
        self.patch_size = patch_size
        self.io_channels = io_channels
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.cond_token_dim = cond_token_dim
        self.global_cond_dim = global_cond_dim
        self.input_concat_dim = input_concat_dim
        self.max_seq_len = max_seq_len
        self.transformer_type = transformer_type
        self.global_cond_type = global_cond_type

        self.timestep_features = FourierFeatures(1, 256)
        self.to_timestep_embed = nn.Sequential(_init.linear(256, embed_dim), nn.SiLU(), _init.linear(embed_dim, embed_dim))
        if cond_token_dim > 0:
            cond_embed_dim = cond_token_dim if not project_cond_tokens else embed_dim
            self.to_cond_embed = nn.Sequential(_init.linear(cond_token_dim, cond_embed_dim, bias=False), nn.SiLU(),
                                               _init.linear(cond_embed_dim, cond_embed_dim, bias=False))
        else:
            cond_embed_dim = 0
        self.cond_embed_dim = cond_embed_dim
        if global_cond_dim > 0:
            self.to_global_embed = nn.Sequential(_init.linear(global_cond_dim, embed_dim, bias=False), nn.SiLU(),
                                                 _init.linear(embed_dim, embed_dim, bias=False))
        self.transformer = ContinuousTransformer(dim=embed_dim, depth=depth, dim_heads=embed_dim // num_heads,
                                                 dim_in=io_channels, dim_out=io_channels, cross_attend=cond_token_dim > 0,
                                                 cond_token_dim=cond_embed_dim,
                                                 global_cond_dim=embed_dim if global_cond_type == "adaLN" else None, **kwargs)
        self.preprocess_conv = _init.conv1d(io_channels, io_channels, 1, bias=False, zero=True)
        self.postprocess_conv = _init.conv1d(io_channels, io_channels, 1, bias=False, zero=True)

        self._plan = None
        self._plan_version = None
        self._ws = None
        self._ctx_key = None
        self.gemm_dtype = _config.default_gemm_dtype()
        self.layernorm_fusion = True
        self.cross_attention_fusion = True
        self.tile_policy = 0

    def residual_stream_report(self, enable: bool = True):
        """Build extension (``sat_dit_debug``), for checkpoints this build was never run on.  ``residual_stream_report(True)`` switches the
        diagnostics on; every forward / ``denoise`` then leaves, per block and residual update (self-attention to_out, cross-attention to_out,
        FF-out), the statistics of the fp32 residual rows that update wrote: ``max_abs``, ``common_mode`` = max over rows of |mean| / std (the
        LayerNorm fold rounds the UN-normalised row to 16 bits: its error grows with this ratio -- ``set_layernorm_fusion(False)`` from ~8 on),
        ``saturated`` = elements beyond +-65504 (what the fp16 image clamps -- ``set_gemm_dtype("bf16")`` if not 0) and ``crest`` = max |x| / rms.
        ``residual_stream_report(False)`` returns the table of the LAST forward as a list of dicts and switches the diagnostics off."""
        import ctypes as _ct
        lib = _hip.lib()
        plan = self._ensure_plan()
        if enable:
            _hip.check(lib.sat_dit_debug(plan, 1))
            return None
        n = self.depth * 3 * 4
        buf = (_ct.c_float * n)()
        _hip.check(lib.sat_dit_debug_read(plan, buf, n, _hip.stream()))
        _hip.check(lib.sat_dit_debug(plan, 0))
        names = ("self_attn.to_out", "cross_attn.to_out", "ff.out")
        return [dict(layer=l, update=names[j], max_abs=buf[(l * 3 + j) * 4], common_mode=buf[(l * 3 + j) * 4 + 1],
                     saturated=int(buf[(l * 3 + j) * 4 + 2]), crest=buf[(l * 3 + j) * 4 + 3]) for l in range(self.depth) for j in range(3)]

    def set_cross_attention_fusion(self, on: bool):
        """Build extension, A/B switch: the to_q projection + cross-attention core as ONE launch where it applies (one prompt; the default) or
        always as two kernels (``sat_dit_cfg.cross_attention``).  Per model; rebuilds the plan on next use."""
        if bool(on) != self.cross_attention_fusion:
            self.cross_attention_fusion = bool(on)
            self._plan_version = None
        return self

    def set_tile_policy(self, policy: int):
        """Build extension, A/B measurement switch (``sat_dit_cfg.tile_policy``): 0 / 80 the measured tile choice, 22 the 16-wave 256 x 256 tile
        of rounds 1-2, 81 the 8-phase kernel for every fp32-output GEMM, 82 no two-K-group 128 x 128 tile.  Per model; rebuilds the plan."""
        if policy not in (0, 22, 80, 81, 82):
            raise ValueError("tile_policy must be 0, 22, 80, 81 or 82")
        if policy != self.tile_policy:
            self.tile_policy = policy
            self._plan_version = None
        return self

    def set_layernorm_fusion(self, on: bool):
        """Build extension: run the LayerNorms of the blocks (transformer.py:692-700) inside the epilogues of the GEMMs either side of
        them (``sat_dit_cfg.ln_fold``; bf16 / "prepend" models) or as standalone kernels.  Rebuilds the plan on next use."""
        if bool(on) != self.layernorm_fusion:
            self.layernorm_fusion = bool(on)
            self._plan_version = None
        return self

    def set_gemm_dtype(self, dtype: str):
        """Build extension: operand format of the block GEMMs and attention kernels (accumulation is fp32 in all of them).
        "fp16" (the package default, stable_audio_tools/_config.py) -- IEEE fp16 operands on the fp16 build of the kernels: the same MFMA rate on gfx950, three more
        significand bits, and the arithmetic the reference itself uses on a GPU (``torch.cuda.amp.autocast`` in
        ``inference/sampling.py:210``, fp16 flash attention in ``models/transformer.py:496-504``); conversions saturate at +-65504;
        "bf16" -- the bf16 build: 3-4 % faster, 8x the operand rounding error;
        "fp8" (BASELINE config 5: OCP e4m3 / MXFP8 operands for cross to_q, FF-in and FF-out -- 56 % of the block's FLOPs, the families
        whose quantisation the sampler trajectory tolerates; the rest bf16), "fp8-all" (every GEMM of the blocks in e4m3: round 3's mode,
        3e-1 off after 12 steps -- the to_out projections on MXFP8 attention outputs are what breaks it); "fp32x" -- the fp32 verification mode (exact fp32
        MFMA, fp32 q / k / v / P; ~20x slower): the same plan and data flow with no operand rounding.  Rebuilds the plan on next use."""
        if dtype not in GEMM_DTYPES:
            raise ValueError(f"gemm_dtype must be one of {sorted(GEMM_DTYPES)}")
        if dtype != self.gemm_dtype:
            self.gemm_dtype = dtype
            self._plan_version = None
        return self

    # ------------------------------------------------------------------ plan management
    def __del__(self):
        try:
            if self._plan is not None:
                _hip.lib().sat_dit_plan_destroy(self._plan)
        except Exception:
            pass

    def _ensure_plan(self):
        ver = _init.params_version(self)
        if self._plan is not None and ver == self._plan_version:
            return self._plan
        lib = _hip.lib()
        dev = self.timestep_features.weight.device
        if dev.type != "cuda":
            raise _hip.SatError("DiffusionTransformer must be on a HIP device (model.to('cuda')); there is no CPU path")
        if self._plan is not None:
            lib.sat_dit_plan_destroy(self._plan)
            self._plan = None
        cfg = _hip.SatDitCfg(self.io_channels, self.embed_dim, self.depth, self.num_heads, self.cond_token_dim,
                             self.cond_embed_dim, self.global_cond_dim, self.max_seq_len,
                             1 if self.global_cond_type == "adaLN" else 0, GEMM_DTYPES[self.gemm_dtype], FP8_FAMILIES.get(self.gemm_dtype, 0),
                             1 if self.layernorm_fusion else 0, 0 if self.cross_attention_fusion else 1, self.tile_policy)
        plan = ctypes.c_void_p()
        _hip.check(lib.sat_dit_plan_create_sized(ctypes.byref(cfg), ctypes.sizeof(cfg), ctypes.byref(plan)))
        keep = []
        for name, t in self.state_dict().items():
            t32 = t.detach().to(torch.float32).contiguous()
            keep.append(t32)
            _hip.check(lib.sat_dit_plan_set_tensor(plan, name.encode(), _hip.ptr(t32), t32.numel()))
        _hip.check(lib.sat_dit_plan_finalize(plan, _hip.stream()))
        del keep
        self._plan = plan
        self._plan_version = ver
        self._ctx_key = None
        return plan

    def _workspace(self, bf, t_len):
        need = ctypes.c_size_t()
        _hip.check(_hip.lib().sat_dit_workspace_bytes(self._plan, bf, t_len, ctypes.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != self.timestep_features.weight.device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.timestep_features.weight.device)
        return self._ws

    def prepare_context(self, cross_attn_cond, global_embed, null_from=-1):
        """Per-generation constants (cond/global MLPs + per-layer cross K/V).  ``cross_attn_cond``
        [bf, Lc, cond_token_dim] and ``global_embed`` [bf, global_cond_dim] already hold the
        CFG-doubled batch (cond half first, null/negative half second).  ``null_from``: index of the
        first sequence whose context is the all-zero null embed (its cross-attention is exactly 0)."""
        plan = self._ensure_plan()
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) if t is not None else None for t in (cross_attn_cond, global_embed))
        key = key + (null_from,)
        if key == self._ctx_key:
            return
        c = None if cross_attn_cond is None else cross_attn_cond.detach().float().contiguous()
        g = None if global_embed is None else global_embed.detach().float().contiguous()
        bf = c.shape[0] if c is not None else (g.shape[0] if g is not None else 0)
        lc = c.shape[1] if c is not None else 0
        if c is not None and c.shape[2] != self.cond_token_dim:
            raise ValueError(f"cross_attn_cond has {c.shape[2]} channels, model expects {self.cond_token_dim}")
        _hip.check(_hip.lib().sat_dit_prepare_context(plan, _hip.ptr(c), bf, lc, _hip.ptr(g), _hip.stream()))
        if null_from >= 0 and c is not None:
            _hip.check(_hip.lib().sat_dit_set_null_context_from(plan, int(null_from)))
        self._ctx_key = key
        self._ctx_keep = (cross_attn_cond, global_embed)

    def prepare_context_bf(self, bf):
        """Context for a model without cross-attention / global conditioning."""
        plan = self._ensure_plan()
        _hip.check(_hip.lib().sat_dit_prepare_context(plan, None, bf, 0, None, _hip.stream()))
        self._ctx_key = ("bf", bf)

    # ------------------------------------------------------------------ reference-semantics forward
    @torch.no_grad()
    def _forward(self, x, t, cross_attn_cond=None, global_embed=None, null_from=-1, **ignored):
        """dit.py:135-226 on the given batch (no CFG logic)."""
        self._ensure_plan()
        x = x.detach().float().contiguous()
        t = t.detach().float().contiguous()
        bf, _, t_len = x.shape
        if cross_attn_cond is None and global_embed is None:
            self.prepare_context_bf(bf)
        else:
            self.prepare_context(cross_attn_cond, global_embed, null_from)
        ws = self._workspace(bf, t_len)
        out = torch.empty_like(x)
        _hip.check(_hip.lib().sat_dit_forward(self._plan, _hip.ptr(x), _hip.ptr(t), _hip.ptr(out), bf, t_len, _hip.ptr(ws),
                                              ws.numel(), _hip.stream()))
        return out

    @torch.no_grad()
    def forward(self, x, t, cross_attn_cond=None, cross_attn_cond_mask=None, negative_cross_attn_cond=None,
                negative_cross_attn_mask=None, input_concat_cond=None, global_embed=None, prepend_cond=None,
                prepend_cond_mask=None, cfg_scale=1.0, cfg_dropout_prob=0.0, causal=False, scale_phi=0.0, mask=None,
                return_info=False, **kwargs):
        assert not causal, "Causal mode is not supported for DiffusionTransformer"
        if input_concat_cond is not None or prepend_cond is not None or return_info:
            raise NotImplementedError("input_concat_cond / prepend_cond / return_info are outside the supported hot path")
        # masks are discarded exactly as the reference does at inference (dit.py:250-252)
        if cfg_scale != 1.0 and cross_attn_cond is not None:
            b = x.shape[0]
            bc, bg = self._cfg_batch(cross_attn_cond, global_embed, negative_cross_attn_cond, negative_cross_attn_mask)
            out = self._forward(torch.cat([x, x], dim=0), torch.cat([t, t], dim=0), bc, bg,
                                null_from=b if negative_cross_attn_cond is None else -1)
            res = torch.empty_like(out[:b])
            # CFG combine (+ optional std rescale), dit.py:336-345, as a HIP kernel: denoise form with c_out=1, c_skip=0
            _hip.check(_hip.lib().sat_cfg_combine(_hip.ptr(out), _hip.ptr(res), b, out.shape[1], out.shape[2], float(cfg_scale),
                                                  float(scale_phi), _hip.stream()))
            return res
        return self._forward(x, t, cross_attn_cond, global_embed)

    @staticmethod
    def _cfg_batch(cross_attn_cond, global_embed, negative_cross_attn_cond=None, negative_cross_attn_mask=None):
        """cat([cond, null]) / cat([global, global]) (dit.py:273-300)."""
        null = torch.zeros_like(cross_attn_cond)
        if negative_cross_attn_cond is not None:
            if negative_cross_attn_mask is not None:
                m = negative_cross_attn_mask.to(torch.bool).unsqueeze(2)
                negative_cross_attn_cond = torch.where(m, negative_cross_attn_cond, null)
            null = negative_cross_attn_cond
        bc = torch.cat([cross_attn_cond, null], dim=0)
        bg = None if global_embed is None else torch.cat([global_embed, global_embed], dim=0)
        return bc, bg

    # ------------------------------------------------------------------ fused sampler-step path
    @torch.no_grad()
    def prepare_generation(self, cross_attn_cond, global_embed, cfg_scale, negative_cross_attn_cond=None,
                           negative_cross_attn_mask=None):
        """Once per ``generate_diffusion_cond`` call: everything that is constant over the steps."""
        self._ensure_plan()
        use_cfg = cfg_scale != 1.0 and cross_attn_cond is not None
        if use_cfg:
            bc, bg = self._cfg_batch(cross_attn_cond, global_embed, negative_cross_attn_cond, negative_cross_attn_mask)
        else:
            bc, bg = cross_attn_cond, global_embed
        if bc is None and bg is None:
            raise ValueError("prepare_generation needs conditioning tensors")
        null_from = cross_attn_cond.shape[0] if (use_cfg and negative_cross_attn_cond is None) else -1
        self.prepare_context(bc, bg, null_from)

    @torch.no_grad()
    def denoise(self, x, sigma: float, cfg_scale: float = 1.0, scale_phi: float = 0.0, out=None):
        """k-diffusion VDenoiser(DiT with batched CFG)(x, sigma) -- ``sat_dit_denoise_cfg``."""
        b, _, t_len = x.shape
        use_cfg = cfg_scale != 1.0 and self.cond_token_dim > 0
        ws = self._workspace(2 * b if use_cfg else b, t_len)
        if out is None:
            out = torch.empty_like(x)
        _hip.check(_hip.lib().sat_dit_denoise_cfg(self._plan, _hip.ptr(x), float(sigma), float(cfg_scale), float(scale_phi),
                                                  _hip.ptr(out), b, t_len, _hip.ptr(ws), ws.numel(), _hip.stream()))
        return out
