"""The autoencoder pretransform of a latent diffusion model: audio <-> latents through the HIP Oobleck codec.

Counterpart of the reference's ``models/pretransforms.py:6-91`` (``Pretransform`` / ``AutoencoderPretransform``): same class
names, constructor arguments and attributes, because ``create_pretransform_from_config``, ``generate_diffusion_cond`` and
reference checkpoints (key prefix ``pretransform.model.``) rely on them.  ``model_half`` (pretransforms.py:39-59: the codec in fp16)
selects the fp16 build of the codec kernels; parameters, inputs and outputs stay fp32 as the reference's ``.float()`` leaves them.
"""
from torch import nn


class Pretransform(nn.Module):
    """What a diffusion wrapper needs to know about its pretransform (filled in by the subclass)."""

    encoded_channels = None      # latent channels
    downsampling_ratio = None    # audio samples per latent frame

    def __init__(self, enable_grad: bool, io_channels: int, is_discrete: bool):
        super().__init__()
        self.enable_grad, self.io_channels, self.is_discrete = enable_grad, io_channels, is_discrete

    def encode(self, x):
        raise NotImplementedError(f"{type(self).__name__} does not encode")

    def decode(self, z):
        raise NotImplementedError(f"{type(self).__name__} does not decode")


class AutoencoderPretransform(Pretransform):
    """Wraps an ``AudioAutoencoder``.  ``scale`` divides the latents on the way in and multiplies them on the way out
    (pretransforms.py:62, :65); ``chunked`` / ``iterate_batch`` are forwarded to the codec's ``encode_audio`` / ``decode_audio``.

    Deviation from the reference, stated: there ``model_half=False`` runs the codec in fp32 (pretransforms.py:39-59).  This build has no fp32
    codec kernels -- the convolutions always take 16-bit operands with fp32 accumulation (0.7e-3 of the reference's fp32 output in fp16, 6-8e-3
    in bf16, tests/test_gpu_models.py) -- so ``model_half`` only PINS fp16; with ``model_half=False`` the codec runs in the package default
    (fp16 unless SAT_GEMM_DTYPE / set_default_gemm_dtype / set_gemm_dtype say bf16).  fp16 conversions saturate at +-65504 instead of
    producing inf as the reference's fp16 path would: with a checkpoint whose activations leave that range use ``set_gemm_dtype("bf16")``."""

    def __init__(self, model, scale=1.0, model_half=False, iterate_batch=False, chunked=False):
        bottleneck = model.bottleneck
        super().__init__(enable_grad=False, io_channels=model.io_channels, is_discrete=bool(bottleneck is not None and bottleneck.is_discrete))
        self.model = model.requires_grad_(False).eval()
        self.scale, self.model_half, self.iterate_batch, self.chunked = scale, bool(model_half), iterate_batch, chunked
        if self.model_half:
            self.model.set_gemm_dtype("fp16")
        for attr, value in (("downsampling_ratio", model.downsampling_ratio), ("sample_rate", model.sample_rate),
                            ("encoded_channels", model.latent_dim), ("num_quantizers", None), ("codebook_size", None)):
            setattr(self, attr, value)

    def _codec_options(self, extra):
        return dict(extra, chunked=self.chunked, iterate_batch=self.iterate_batch)

    def encode(self, x, **kwargs):
        return self.model.encode_audio(x, **self._codec_options(kwargs)) / self.scale

    def decode(self, z, **kwargs):
        return self.model.decode_audio(z * self.scale, **self._codec_options(kwargs))

    def load_state_dict(self, state_dict, strict=True):
        # the checkpoint of a pretransform is the checkpoint of its autoencoder (pretransforms.py:90-91)
        return self.model.load_state_dict(state_dict, strict=strict)
