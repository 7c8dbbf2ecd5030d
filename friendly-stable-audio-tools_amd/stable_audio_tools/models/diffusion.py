"""``DiTWrapper`` / ``ConditionedDiffusionModelWrapper`` / ``create_diffusion_cond_from_config``
(reference ``models/diffusion.py:34-209, 482-529, 585-655``)."""
import typing as tp

import torch
from torch import nn

from .conditioners import MultiConditioner, create_multi_conditioner_from_conditioning_config
from .dit import DiffusionTransformer
from .factory import create_pretransform_from_config
from .pretransforms import Pretransform


class ConditionedDiffusionModel(nn.Module):
    def __init__(self, *args, supports_cross_attention: bool = False, supports_input_concat: bool = False,
                 supports_global_cond: bool = False, supports_prepend_cond: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.supports_cross_attention = supports_cross_attention
        self.supports_input_concat = supports_input_concat
        self.supports_global_cond = supports_global_cond
        self.supports_prepend_cond = supports_prepend_cond


class DiTWrapper(ConditionedDiffusionModel):
    """reference models/diffusion.py:482-529.  Parameters are multiplied by 0.5 after
    construction, as the reference does (:487-489), unless built under ``skip_init``."""

    def __init__(self, *args, **kwargs):
        super().__init__(supports_cross_attention=True, supports_global_cond=False, supports_input_concat=False)
        self.model = DiffusionTransformer(*args, **kwargs)
        from . import _init
        if not _init._SKIP:
            with torch.no_grad():
                for param in self.model.parameters():
                    param *= 0.5

    def forward(self, x, t, cross_attn_cond=None, cross_attn_mask=None, negative_cross_attn_cond=None,
                negative_cross_attn_mask=None, input_concat_cond=None, negative_input_concat_cond=None, global_cond=None,
                negative_global_cond=None, prepend_cond=None, prepend_cond_mask=None, cfg_scale=1.0,
                cfg_dropout_prob: float = 0.0, batch_cfg: bool = True, rescale_cfg: bool = False, scale_phi: float = 0.0,
                **kwargs):
        assert batch_cfg, "batch_cfg must be True for DiTWrapper"
        # rescale_cfg is accepted and ignored, exactly like the reference
        return self.model(x, t, cross_attn_cond=cross_attn_cond, cross_attn_cond_mask=cross_attn_mask,
                          negative_cross_attn_cond=negative_cross_attn_cond, negative_cross_attn_mask=negative_cross_attn_mask,
                          input_concat_cond=input_concat_cond, prepend_cond=prepend_cond, prepend_cond_mask=prepend_cond_mask,
                          cfg_scale=cfg_scale, cfg_dropout_prob=cfg_dropout_prob, scale_phi=scale_phi, global_embed=global_cond,
                          **kwargs)


class ConditionedDiffusionModelWrapper(nn.Module):
    """A diffusion model that takes in conditioning (reference models/diffusion.py:90-209)."""

    def __init__(self, model: ConditionedDiffusionModel, conditioner: MultiConditioner, io_channels, sample_rate,
                 min_input_length: int, diffusion_objective: str = "v", pretransform: tp.Optional[Pretransform] = None,
                 cross_attn_cond_ids: tp.List[str] = [], global_cond_ids: tp.List[str] = [], input_concat_ids: tp.List[str] = [],
                 prepend_cond_ids: tp.List[str] = []):
        super().__init__()
        self.model = model
        self.conditioner = conditioner
        self.io_channels = io_channels
        self.sample_rate = sample_rate
        self.diffusion_objective = diffusion_objective
        self.pretransform = pretransform
        self.cross_attn_cond_ids = cross_attn_cond_ids
        self.global_cond_ids = global_cond_ids
        self.input_concat_ids = input_concat_ids
        self.prepend_cond_ids = prepend_cond_ids
        self.min_input_length = min_input_length

    def get_conditioning_inputs(self, conditioning_tensors: tp.Dict[str, tp.Any], negative=False):
        """Assemble the denoiser's keyword arguments from the per-id ``(tensor, mask)`` pairs (counterpart of the reference's
        models/diffusion.py:123-203; host-side concatenation only): cross-attention ids are joined along the token axis
        (per-item vectors [B, C] count as one token), global ids along the channel axis (a singleton token axis is dropped),
        input-concat ids along channels, prepend ids along tokens."""

        def pairs(ids):
            return [conditioning_tensors[i] for i in ids]

        def as_tokens(tensor, mask):
            return (tensor[:, None], mask[:, None]) if tensor.dim() == 2 else (tensor, mask)

        cross = cross_mask = global_cond = concat = prepend = prepend_mask = None
        if self.cross_attn_cond_ids:
            tokens = [as_tokens(t, m) for t, m in pairs(self.cross_attn_cond_ids)]
            cross = torch.cat([t for t, _ in tokens], dim=1)
            cross_mask = torch.cat([m for _, m in tokens], dim=1)
        if self.global_cond_ids:
            global_cond = torch.cat([t for t, _ in pairs(self.global_cond_ids)], dim=-1)
            if global_cond.dim() == 3:
                global_cond = global_cond.squeeze(1)
        if self.input_concat_ids:
            concat = torch.cat([t for t, _ in pairs(self.input_concat_ids)], dim=1)
        if self.prepend_cond_ids:
            prepend = torch.cat([t for t, _ in pairs(self.prepend_cond_ids)], dim=1)
            prepend_mask = torch.cat([m for _, m in pairs(self.prepend_cond_ids)], dim=1)
        if negative:       # the negative set has no prepend entries in the reference either (:186-191)
            return {"negative_cross_attn_cond": cross, "negative_cross_attn_mask": cross_mask, "negative_global_cond": global_cond,
                    "negative_input_concat_cond": concat}
        return {"cross_attn_cond": cross, "cross_attn_mask": cross_mask, "global_cond": global_cond, "input_concat_cond": concat,
                "prepend_cond": prepend, "prepend_cond_mask": prepend_mask}

    def forward(self, x: torch.Tensor, t: torch.Tensor, cond: tp.Dict[str, tp.Any], **kwargs):
        return self.model(x, t, **self.get_conditioning_inputs(cond), **kwargs)

    def generate(self, *args, **kwargs):
        from ..inference.generation import generate_diffusion_cond
        return generate_diffusion_cond(self, *args, **kwargs)


def create_diffusion_cond_from_config(config: tp.Dict[str, tp.Any]):
    """``model_type`` "diffusion_cond" / "diffusion_cond_inpaint" with a DiT denoiser (counterpart of the reference's
    models/diffusion.py:585-655): denoiser, conditioner set and (optional) autoencoder pretransform from one config dict."""
    if config["model_type"] not in ("diffusion_cond", "diffusion_cond_inpaint"):
        raise NotImplementedError(f"model_type '{config['model_type']}' is outside this build's hot path")
    model_cfg = config["model"]
    diffusion_cfg = model_cfg["diffusion"]
    if diffusion_cfg["type"] != "dit":
        raise NotImplementedError(f"diffusion model type '{diffusion_cfg['type']}' is outside this build's hot path (DiT only)")
    denoiser = DiTWrapper(**diffusion_cfg["config"])
    conditioning = model_cfg.get("conditioning")
    conditioner = create_multi_conditioner_from_conditioning_config(conditioning) if conditioning else None
    pretransform = model_cfg.get("pretransform")
    if pretransform:
        pretransform = create_pretransform_from_config(pretransform, config["sample_rate"])
    # shortest input the model accepts: one latent frame (x patch size)
    min_input_length = (pretransform.downsampling_ratio if pretransform else 1) * denoiser.model.patch_size
    id_lists = {name: diffusion_cfg.get(key, []) for name, key in (("cross_attn_cond_ids", "cross_attention_cond_ids"),
                                                                    ("global_cond_ids", "global_cond_ids"),
                                                                    ("input_concat_ids", "input_concat_ids"),
                                                                    ("prepend_cond_ids", "prepend_cond_ids"))}
    return ConditionedDiffusionModelWrapper(denoiser, conditioner, io_channels=model_cfg["io_channels"], sample_rate=config["sample_rate"],
                                            min_input_length=min_input_length, pretransform=pretransform,
                                            diffusion_objective=diffusion_cfg.get("diffusion_objective", "v"), **id_lists)
