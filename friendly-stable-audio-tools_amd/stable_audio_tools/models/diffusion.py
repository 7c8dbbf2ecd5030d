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
        # reference models/diffusion.py:123-203 (tensor bookkeeping only: cat on seq / channel dims)
        cross_attention_input = cross_attention_masks = global_cond = input_concat_cond = None
        prepend_cond = prepend_cond_mask = None
        if len(self.cross_attn_cond_ids) > 0:
            ins, masks = [], []
            for key in self.cross_attn_cond_ids:
                cross_attn_in, cross_attn_mask = conditioning_tensors[key]
                if len(cross_attn_in.shape) == 2:
                    cross_attn_in = cross_attn_in.unsqueeze(1)
                    cross_attn_mask = cross_attn_mask.unsqueeze(1)
                ins.append(cross_attn_in)
                masks.append(cross_attn_mask)
            cross_attention_input = torch.cat(ins, dim=1)
            cross_attention_masks = torch.cat(masks, dim=1)
        if len(self.global_cond_ids) > 0:
            global_cond = torch.cat([conditioning_tensors[key][0] for key in self.global_cond_ids], dim=-1)
            if len(global_cond.shape) == 3:
                global_cond = global_cond.squeeze(1)
        if len(self.input_concat_ids) > 0:
            input_concat_cond = torch.cat([conditioning_tensors[key][0] for key in self.input_concat_ids], dim=1)
        if len(self.prepend_cond_ids) > 0:
            conds, cmasks = [], []
            for key in self.prepend_cond_ids:
                c, m = conditioning_tensors[key]
                conds.append(c)
                cmasks.append(m)
            prepend_cond = torch.cat(conds, dim=1)
            prepend_cond_mask = torch.cat(cmasks, dim=1)
        if negative:
            return {"negative_cross_attn_cond": cross_attention_input, "negative_cross_attn_mask": cross_attention_masks,
                    "negative_global_cond": global_cond, "negative_input_concat_cond": input_concat_cond}
        return {"cross_attn_cond": cross_attention_input, "cross_attn_mask": cross_attention_masks, "global_cond": global_cond,
                "input_concat_cond": input_concat_cond, "prepend_cond": prepend_cond, "prepend_cond_mask": prepend_cond_mask}

    def forward(self, x: torch.Tensor, t: torch.Tensor, cond: tp.Dict[str, tp.Any], **kwargs):
        return self.model(x, t, **self.get_conditioning_inputs(cond), **kwargs)

    def generate(self, *args, **kwargs):
        from ..inference.generation import generate_diffusion_cond
        return generate_diffusion_cond(self, *args, **kwargs)


def create_diffusion_cond_from_config(config: tp.Dict[str, tp.Any]):
    # reference models/diffusion.py:585-655
    model_config = config["model"]
    model_type = config["model_type"]
    diffusion_config = model_config["diffusion"]
    diffusion_model_type = diffusion_config["type"]
    diffusion_model_config = diffusion_config["config"]
    if diffusion_model_type != "dit":
        raise NotImplementedError(f"diffusion model type '{diffusion_model_type}' is outside this build's hot path (DiT only)")
    diffusion_model = DiTWrapper(**diffusion_model_config)

    io_channels = model_config["io_channels"]
    sample_rate = config["sample_rate"]
    diffusion_objective = diffusion_config.get("diffusion_objective", "v")
    conditioning_config = model_config.get("conditioning", None)
    conditioner = None
    if conditioning_config:
        conditioner = create_multi_conditioner_from_conditioning_config(conditioning_config)
    cross_attn_cond_ids = diffusion_config.get("cross_attention_cond_ids", [])
    global_cond_ids = diffusion_config.get("global_cond_ids", [])
    input_concat_ids = diffusion_config.get("input_concat_ids", [])
    prepend_cond_ids = diffusion_config.get("prepend_cond_ids", [])
    pretransform = model_config.get("pretransform", None)
    if pretransform:
        pretransform = create_pretransform_from_config(pretransform, sample_rate)
        min_input_length = pretransform.downsampling_ratio
    else:
        min_input_length = 1
    min_input_length *= diffusion_model.model.patch_size
    if model_type not in ("diffusion_cond", "diffusion_cond_inpaint"):
        raise NotImplementedError(f"model_type '{model_type}' is outside this build's hot path")
    return ConditionedDiffusionModelWrapper(diffusion_model, conditioner, min_input_length=min_input_length,
                                            sample_rate=sample_rate, cross_attn_cond_ids=cross_attn_cond_ids,
                                            global_cond_ids=global_cond_ids, input_concat_ids=input_concat_ids,
                                            prepend_cond_ids=prepend_cond_ids, pretransform=pretransform, io_channels=io_channels,
                                            diffusion_objective=diffusion_objective)
