"""``AutoencoderPretransform`` (reference ``models/pretransforms.py:6-91``)."""
from torch import nn


class Pretransform(nn.Module):
    def __init__(self, enable_grad: bool, io_channels: int, is_discrete: bool):
        super().__init__()
        self.is_discrete = is_discrete
        self.io_channels = io_channels
        self.encoded_channels = None
        self.downsampling_ratio = None
        self.enable_grad = enable_grad

    def encode(self, x):
        raise NotImplementedError

    def decode(self, z):
        raise NotImplementedError


class AutoencoderPretransform(Pretransform):
    def __init__(self, model, scale=1.0, model_half=False, iterate_batch=False, chunked=False):
        super().__init__(enable_grad=False, io_channels=model.io_channels,
                         is_discrete=model.bottleneck is not None and model.bottleneck.is_discrete)
        if model_half:
            raise NotImplementedError("model_half: the HIP codec already stores activations in bf16; fp16 mode is not provided")
        self.model = model
        self.model.requires_grad_(False).eval()
        self.scale = scale
        self.downsampling_ratio = model.downsampling_ratio
        self.io_channels = model.io_channels
        self.sample_rate = model.sample_rate
        self.model_half = model_half
        self.iterate_batch = iterate_batch
        self.encoded_channels = model.latent_dim
        self.chunked = chunked
        self.num_quantizers = None
        self.codebook_size = None

    def encode(self, x, **kwargs):
        # pretransforms.py:51-62
        encoded = self.model.encode_audio(x, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)
        return encoded / self.scale

    def decode(self, z, **kwargs):
        # pretransforms.py:64-76
        z = z * self.scale
        return self.model.decode_audio(z, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)

    def load_state_dict(self, state_dict, strict=True):
        self.model.load_state_dict(state_dict, strict=strict)
