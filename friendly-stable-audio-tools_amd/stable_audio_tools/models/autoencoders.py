"""Oobleck encoder/decoder + ``AudioAutoencoder`` on the HIP C ABI.

Module tree and state-dict keys follow the reference's ``models/autoencoders.py``
(``ResidualUnit`` :45-68, ``EncoderBlock`` :71-85, ``DecoderBlock`` :88-116, ``OobleckEncoder``
:119-153, ``OobleckDecoder`` :156-194, ``AudioAutoencoder`` :234-645, factories :695-787) and
``dac.nn.layers.WNConv1d`` (``weight_g``/``weight_v``/``bias``).  The classes hold parameters; a
whole encoder/decoder forward is one ``sat_oobleck_encode`` / ``sat_oobleck_decode`` call.
"""
import ctypes
import math
import typing as tp

import torch
from torch import nn
from torch.nn import functional as F

from .. import _config, _hip
from . import _init
from .blocks import SnakeBeta
from .bottleneck import Bottleneck
from .factory import create_bottleneck_from_config, create_pretransform_from_config


class _WNBase(nn.Module):
    """Weight-normed conv parameters exactly as ``torch.nn.utils.weight_norm`` (dim=0) stores them."""

    def _register(self, conv, bias):
        v = conv.weight.detach()
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))).clone())
        self.weight_v = nn.Parameter(v.clone())
        if bias:
            self.bias = nn.Parameter(conv.bias.detach().clone())
        else:
            self.register_parameter("bias", None)


class WNConv1d(_WNBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self._register(_init.conv1d(in_channels, out_channels, kernel_size, bias=bias), bias)


class WNConvTranspose1d(_WNBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        self._register(_init.conv_transpose1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding), True)


def _act(use_snake, channels):
    if not use_snake:
        raise NotImplementedError("only use_snake=True (the Stable Audio VAE configs) is supported by the HIP codec")
    return SnakeBeta(channels)


class ResidualUnit(nn.Module):
    def __init__(self, in_channels, out_channels, dilation, use_snake=False, antialias_activation=False):
        super().__init__()
        if antialias_activation:
            raise NotImplementedError("antialias_activation is not supported by the HIP codec")
        self.dilation = dilation
        self.layers = nn.Sequential(
            _act(use_snake, out_channels),
            WNConv1d(in_channels, out_channels, 7, dilation=dilation, padding=(dilation * 6) // 2),
            _act(use_snake, out_channels),
            WNConv1d(out_channels, out_channels, 1))


class EncoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride, use_snake=False, antialias_activation=False):
        super().__init__()
        self.layers = nn.Sequential(
            ResidualUnit(in_channels, in_channels, 1, use_snake=use_snake),
            ResidualUnit(in_channels, in_channels, 3, use_snake=use_snake),
            ResidualUnit(in_channels, in_channels, 9, use_snake=use_snake),
            _act(use_snake, in_channels),
            WNConv1d(in_channels, out_channels, 2 * stride, stride=stride, padding=math.ceil(stride / 2)))


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride, use_snake=False, antialias_activation=False, use_nearest_upsample=False):
        super().__init__()
        if use_nearest_upsample:
            raise NotImplementedError("use_nearest_upsample is not supported by the HIP codec")
        self.layers = nn.Sequential(
            _act(use_snake, in_channels),
            WNConvTranspose1d(in_channels, out_channels, 2 * stride, stride=stride, padding=math.ceil(stride / 2)),
            ResidualUnit(out_channels, out_channels, 1, use_snake=use_snake),
            ResidualUnit(out_channels, out_channels, 3, use_snake=use_snake),
            ResidualUnit(out_channels, out_channels, 9, use_snake=use_snake))


class _OobleckHip(nn.Module):
    """Shared plan handling of encoder and decoder."""
    _is_decoder = False

    def _init_plan_state(self):
        self.gemm_dtype = _config.default_gemm_dtype()
        self._plan = None
        self._plan_version = None
        self._ws = None

    def set_gemm_dtype(self, dtype: str):
        """Build extension: 16-bit format of the activations and weights inside the convolution kernels -- "fp16" (the package default,
        stable_audio_tools/_config.py: IEEE fp16 on the fp16 build of the kernels, what the reference's ``model_half`` runs,
        ``models/pretransforms.py:39-59``) or "bf16" (8x the rounding error, a few per cent faster).  Parameters stay fp32 in the module; rebuilds the plan on next use."""
        if dtype not in ("bf16", "fp16"):
            raise ValueError("the codec kernels take 'bf16' or 'fp16' operands")
        if dtype != self.gemm_dtype:
            self.gemm_dtype = dtype
            self._plan_version = None
        return self

    def __del__(self):
        try:
            if self._plan is not None:
                _hip.lib().sat_oobleck_plan_destroy(self._plan)
        except Exception:
            pass

    def _ensure_plan(self):
        ver = _init.params_version(self)
        if self._plan is not None and ver == self._plan_version:
            return self._plan
        lib = _hip.lib()
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _hip.SatError("Oobleck modules must be on a HIP device (model.to('cuda')); there is no CPU path")
        if self._plan is not None:
            lib.sat_oobleck_plan_destroy(self._plan)
            self._plan = None
        cfg = _hip.SatOobleckCfg()
        cfg.is_decoder = 1 if self._is_decoder else 0
        cfg.io_channels = self.io_channels
        cfg.channels = self.channels
        cfg.latent_dim = self.latent_dim
        cfg.n_blocks = len(self.strides)
        for i, (c, s) in enumerate(zip(self.c_mults, self.strides)):
            cfg.c_mults[i] = c
            cfg.strides[i] = s
        cfg.gemm_dtype = 3 if self.gemm_dtype == "fp16" else 0          # SAT_GEMM_FP16 / SAT_GEMM_BF16 (include/sat_hip.h)
        plan = ctypes.c_void_p()
        _hip.check(lib.sat_oobleck_plan_create(ctypes.byref(cfg), ctypes.byref(plan)))
        keep = []
        for name, t in self.state_dict().items():
            t32 = t.detach().to(torch.float32).contiguous()
            keep.append(t32)
            _hip.check(lib.sat_oobleck_plan_set_tensor(plan, name.encode(), _hip.ptr(t32), t32.numel()))
        _hip.check(lib.sat_oobleck_plan_finalize(plan, _hip.stream()))
        del keep
        self._plan, self._plan_version = plan, ver
        return plan

    def _workspace(self, b, t_len):
        need = ctypes.c_size_t()
        _hip.check(_hip.lib().sat_oobleck_workspace_bytes(self._plan, b, t_len, ctypes.byref(need)))
        dev = next(self.parameters()).device
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        return self._ws


class OobleckEncoder(_OobleckHip):
    def __init__(self, in_channels=2, channels=128, latent_dim=32, c_mults=[1, 2, 4, 8], strides=[2, 4, 8, 8], use_snake=False,
                 antialias_activation=False):
        super().__init__()
        self.io_channels, self.channels, self.latent_dim = in_channels, channels, latent_dim
        self.c_mults, self.strides = list(c_mults), list(strides)
        self.ratio = int(math.prod(strides))
        cm = [1] + list(c_mults)
        self.depth = len(cm)
        layers = [WNConv1d(in_channels, cm[0] * channels, 7, padding=3)]
        for i in range(self.depth - 1):
            layers.append(EncoderBlock(cm[i] * channels, cm[i + 1] * channels, strides[i], use_snake=use_snake))
        layers += [_act(use_snake, cm[-1] * channels), WNConv1d(cm[-1] * channels, latent_dim, 3, padding=1)]
        self.layers = nn.Sequential(*layers)
        self._init_plan_state()

    @torch.no_grad()
    def forward(self, x):
        plan = self._ensure_plan()
        x = x.detach().float().contiguous()
        b, c, length = x.shape
        assert c == self.io_channels, f"encoder expects {self.io_channels} channels, got {c}"
        assert length % self.ratio == 0, "audio length must be a multiple of the downsampling ratio"
        t_len = length // self.ratio
        ws = self._workspace(b, t_len)
        out = torch.empty((b, self.latent_dim, t_len), device=x.device, dtype=torch.float32)
        _hip.check(_hip.lib().sat_oobleck_encode(plan, _hip.ptr(x), _hip.ptr(out), b, t_len, _hip.ptr(ws), ws.numel(), _hip.stream()))
        return out


class OobleckDecoder(_OobleckHip):
    _is_decoder = True

    def __init__(self, out_channels=2, channels=128, latent_dim=32, c_mults=[1, 2, 4, 8], strides=[2, 4, 8, 8], use_snake=False,
                 antialias_activation=False, use_nearest_upsample=False, final_tanh=True):
        super().__init__()
        if final_tanh:
            raise NotImplementedError("final_tanh=True is not supported by the HIP codec (Stable Audio VAEs use final_tanh=False)")
        self.io_channels, self.channels, self.latent_dim = out_channels, channels, latent_dim
        self.c_mults, self.strides = list(c_mults), list(strides)
        self.ratio = int(math.prod(strides))
        cm = [1] + list(c_mults)
        self.depth = len(cm)
        layers = [WNConv1d(latent_dim, cm[-1] * channels, 7, padding=3)]
        for i in range(self.depth - 1, 0, -1):
            layers.append(DecoderBlock(cm[i] * channels, cm[i - 1] * channels, strides[i - 1], use_snake=use_snake,
                                       antialias_activation=antialias_activation, use_nearest_upsample=use_nearest_upsample))
        layers += [_act(use_snake, cm[0] * channels), WNConv1d(cm[0] * channels, out_channels, 7, padding=3, bias=False), nn.Identity()]
        self.layers = nn.Sequential(*layers)
        self._init_plan_state()

    @torch.no_grad()
    def forward(self, z):
        plan = self._ensure_plan()
        z = z.detach().float().contiguous()
        b, c, t_len = z.shape
        assert c == self.latent_dim, f"decoder expects {self.latent_dim} latent channels, got {c}"
        ws = self._workspace(b, t_len)
        out = torch.empty((b, self.io_channels, t_len * self.ratio), device=z.device, dtype=torch.float32)
        _hip.check(_hip.lib().sat_oobleck_decode(plan, _hip.ptr(z), _hip.ptr(out), b, t_len, _hip.ptr(ws), ws.numel(), _hip.stream()))
        return out


def _batched(fn, x, iterate_batch):
    if not iterate_batch:
        return fn(x)
    max_bs = int(iterate_batch)
    return torch.cat([fn(x[i:i + max_bs]) for i in range(0, x.shape[0], max_bs)], dim=0)


class AudioAutoencoder(nn.Module):
    def __init__(self, encoder, decoder, latent_dim, downsampling_ratio, sample_rate, io_channels=2, bottleneck: Bottleneck = None,
                 pretransform=None, in_channels=None, out_channels=None, soft_clip=False):
        super().__init__()
        if pretransform is not None or soft_clip:
            raise NotImplementedError("autoencoder pretransform / soft_clip are outside the supported hot path")
        self.downsampling_ratio = downsampling_ratio
        self.min_length = self.downsampling_ratio
        self.sample_rate = sample_rate
        self.latent_dim = latent_dim
        self.io_channels = io_channels
        self.in_channels = io_channels if in_channels is None else in_channels
        self.out_channels = io_channels if out_channels is None else out_channels
        self.encoder = encoder
        self.decoder = decoder
        self.bottleneck = bottleneck
        self.pretransform = pretransform
        self.soft_clip = soft_clip
        self.is_discrete = self.bottleneck and self.bottleneck.is_discrete

    def set_gemm_dtype(self, dtype: str):
        """"bf16" | "fp16" for the encoder and decoder kernels (see ``_OobleckHip.set_gemm_dtype``)."""
        for part in (self.encoder, self.decoder):
            if isinstance(part, _OobleckHip):
                part.set_gemm_dtype(dtype)
        return self

    # autoencoders.py:268-304
    def encode(self, audio, return_info=False, skip_pretransform=False, iterate_batch=False, **kwargs):
        latents = _batched(self.encoder, audio, iterate_batch) if self.encoder else audio
        info = {}
        if self.bottleneck:
            latents, bottleneck_info = self.bottleneck.encode(latents, return_info=True, **kwargs)
            info.update(bottleneck_info)
        return (latents, info) if return_info else latents

    # autoencoders.py:306-343
    def decode(self, latents, iterate_batch=False, **kwargs):
        if self.bottleneck:
            latents = self.bottleneck.decode(latents)
        return _batched(self.decoder, latents, iterate_batch)

    # autoencoders.py:356-408
    def preprocess_audio_for_encoder(self, audio, in_sr):
        return self.preprocess_audio_list_for_encoder([audio], [in_sr])

    def preprocess_audio_list_for_encoder(self, audio_list, in_sr_list):
        from ..inference.utils import prepare_audio
        batch_size = len(audio_list)
        if isinstance(in_sr_list, int):
            in_sr_list = [in_sr_list] * batch_size
        assert len(in_sr_list) == batch_size, "list of sample rates must be the same length of audio_list"
        new_audio, max_length = [], 0
        for audio, in_sr in zip(audio_list, in_sr_list):
            if audio.dim() == 3 and audio.shape[0] == 1:
                audio = audio.squeeze(0)
            elif audio.dim() == 1:
                audio = audio.unsqueeze(0)
            assert audio.dim() == 2, "Audio should be shape (Channels x Length) with no batch dimension"
            if in_sr != self.sample_rate:           # autoencoders.py:394-397 of the reference (torchaudio Resample): HIP polyphase kernel
                from ..inference.resample import resample
                audio = resample(audio, in_sr, self.sample_rate)
            new_audio.append(audio)
            max_length = max(max_length, audio.shape[-1])
        padded = max_length + (self.min_length - (max_length % self.min_length)) % self.min_length
        out = [prepare_audio(a, in_sr=self.sample_rate, target_sr=self.sample_rate, target_length=padded,
                             target_channels=self.in_channels, device=a.device).squeeze(0) for a in new_audio]
        return torch.stack(out)

    # ------------------------------------------------------------------ chunked paths (autoencoders.py:410-645)
    # The reference spells chunking + cross-fade out three times (encode_audio, decode_audio, reconstruct_audio).  Here they share
    # one driver: cut the (padded) input into overlapping chunks, run the codec on batches of at most `max_batch_size` chunks,
    # overlap-add the results with a Bartlett cross-fade (one HIP kernel, sat_overlap_add) and crop.
    def _run_chunked(self, x, fn, in_chunk, in_hop, n_chunk, out_hop, out_overlap, max_batch_size, keep):
        bs, channels = x.shape[0], x.shape[1]
        chunks = torch.stack([x[..., i * in_hop: i * in_hop + in_chunk] for i in range(n_chunk)], dim=1)
        chunks = chunks.reshape(bs * n_chunk, channels, in_chunk)
        pieces = torch.cat([fn(chunks[i: i + max_batch_size]) for i in range(0, chunks.shape[0], max_batch_size)], dim=0)
        out_ch, out_chunk = pieces.shape[1], pieces.shape[2]
        pieces = pieces.reshape(bs, n_chunk, out_ch, out_chunk).float().contiguous()
        total = out_hop * (n_chunk - 1) + out_chunk
        window = torch.bartlett_window(max(out_overlap, 1) * 2, device=x.device).float().contiguous()     # (unused when overlap == 0)
        out = torch.empty((bs, out_ch, total), dtype=torch.float32, device=x.device)
        _hip.check(_hip.lib().sat_overlap_add(_hip.ptr(pieces), _hip.ptr(window), _hip.ptr(out), bs, n_chunk, out_ch, out_chunk, out_hop,
                                              out_overlap, total, _hip.stream()))
        return out[..., :keep]

    @staticmethod
    def _n_chunks(length, chunk, hop):
        return int(math.ceil((length - chunk) / hop)) + 1

    def encode_audio(self, audio, chunked=False, chunk_size=128, overlap=4, max_batch_size=1, **kwargs):
        """Audio [B, C, L] (L a multiple of the compression ratio) -> latents; ``chunked``: windows of ``chunk_size`` latent frames
        overlapping by ``overlap`` frames, zero-padded at the end, cross-faded in the latent domain."""
        _, n_ch, length = audio.shape
        ratio = self.downsampling_ratio
        assert n_ch == self.in_channels
        assert length % ratio == 0, "The audio length must be a multiple of compression ratio."
        if not chunked:
            return self.encode(audio, **kwargs)
        chunk_s, hop_s = chunk_size * ratio, (chunk_size - overlap) * ratio
        n_chunk = self._n_chunks(length, chunk_s, hop_s)
        audio = F.pad(audio, (0, chunk_s + hop_s * (n_chunk - 1) - length))
        return self._run_chunked(audio, self.encode, chunk_s, hop_s, n_chunk, chunk_size - overlap, overlap, max_batch_size, length // ratio)

    def decode_audio(self, latents, chunked=False, chunk_size=128, overlap=4, max_batch_size=1, **kwargs):
        """Latents [B, latent_dim, T] -> audio; ``chunked``: the latents are reflect-padded to a whole number of hops and the decoded
        windows cross-faded over ``overlap`` frames' worth of samples."""
        _, latent_dim, t_len = latents.shape
        ratio = self.downsampling_ratio
        assert latent_dim == self.latent_dim
        if not chunked:
            return self.decode(latents, **kwargs)
        hop = chunk_size - overlap
        n_chunk = self._n_chunks(t_len, chunk_size, hop)
        latents = F.pad(latents, (0, chunk_size + hop * (n_chunk - 1) - t_len), mode="reflect")
        return self._run_chunked(latents, self.decode, chunk_size, hop, n_chunk, hop * ratio, overlap * ratio, max_batch_size, t_len * ratio)

    @torch.no_grad()
    def reconstruct_audio(self, audio, chunked=True, chunk_size=128, overlap=4, max_batch_size=1, **kwargs):
        """encode -> decode per window, cross-faded in the audio domain.  The reference pads with one hop more than the windows
        it then takes (``hop * n_chunk``, autoencoders.py:604); the crop below makes that invisible and it is kept as is."""
        _, n_ch, length = audio.shape
        ratio = self.downsampling_ratio
        assert n_ch == self.in_channels
        if not chunked:
            return self.decode(self.encode(audio, **kwargs), **kwargs)
        chunk_s, overlap_s = chunk_size * ratio, overlap * ratio
        hop_s = chunk_s - overlap_s
        n_chunk = self._n_chunks(length, chunk_s, hop_s)
        audio = F.pad(audio, (0, chunk_s + hop_s * n_chunk - length))
        codec = lambda chunk: self.decode(self.encode(chunk, **kwargs))
        return self._run_chunked(audio, codec, chunk_s, hop_s, n_chunk, hop_s, overlap_s, max_batch_size, length)


# ---------------------------------------------------------------------------------- factories (autoencoders.py:695-787)
def create_encoder_from_config(encoder_config: tp.Dict[str, tp.Any]):
    if encoder_config["type"] != "oobleck":
        raise NotImplementedError(f"encoder type '{encoder_config['type']}' is outside this build's hot path (oobleck only)")
    encoder = OobleckEncoder(**encoder_config["config"])
    if not encoder_config.get("requires_grad", True):
        for p in encoder.parameters():
            p.requires_grad = False
    return encoder


def create_decoder_from_config(decoder_config: tp.Dict[str, tp.Any]):
    if decoder_config["type"] != "oobleck":
        raise NotImplementedError(f"decoder type '{decoder_config['type']}' is outside this build's hot path (oobleck only)")
    decoder = OobleckDecoder(**decoder_config["config"])
    if not decoder_config.get("requires_grad", True):
        for p in decoder.parameters():
            p.requires_grad = False
    return decoder


def create_autoencoder_from_config(config: tp.Dict[str, tp.Any]):
    ae_config = config["model"]
    encoder = create_encoder_from_config(ae_config["encoder"])
    decoder = create_decoder_from_config(ae_config["decoder"])
    bottleneck = ae_config.get("bottleneck", None)
    pretransform = ae_config.get("pretransform", None)
    if pretransform:
        pretransform = create_pretransform_from_config(pretransform, config["sample_rate"])
    if bottleneck:
        bottleneck = create_bottleneck_from_config(bottleneck)
    return AudioAutoencoder(encoder, decoder, io_channels=ae_config["io_channels"], latent_dim=ae_config["latent_dim"],
                            downsampling_ratio=ae_config["downsampling_ratio"], sample_rate=config["sample_rate"],
                            bottleneck=bottleneck, pretransform=pretransform, in_channels=ae_config.get("in_channels", None),
                            out_channels=ae_config.get("out_channels", None), soft_clip=ae_config["decoder"].get("soft_clip", False))
