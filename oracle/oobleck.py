"""Oracle: Oobleck 1-D conv VAE (encoder / decoder / VAE bottleneck / chunked OLA), fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Citations relative to
/root/reference/stable_audio_tools/.

``sd`` holds the reference's keys relative to ``OobleckEncoder`` / ``OobleckDecoder``
(``layers.N....{weight_g,weight_v,bias,alpha,beta}``).  ``rnd`` is the matched-rounding
hook (see oracle/dit.py): with ``bf16_round`` the folded conv weights, every conv *input*
(after Snake) and every stored activation are rounded to bf16, as the HIP kernels do.
"""
import math

import torch
import torch.nn.functional as F


def _r(rnd, x):
    return x if rnd is None else rnd(x)


# dac.nn.layers.WNConv1d == torch.nn.utils.weight_norm(nn.Conv1d) (dim=0):
# w = g * v / ||v||, norm over all dims except 0.  (autoencoders.py:11)
def fold_weight_norm(g, v):
    norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


# models/blocks.py:318-319, 350-358 (SnakeBeta, alpha_logscale=True)
def snake_beta(x, alpha, beta):
    a = torch.exp(alpha).view(1, -1, 1)
    b = torch.exp(beta).view(1, -1, 1)
    return x + (1.0 / (b + 0.000000001)) * torch.pow(torch.sin(x * a), 2)


def _wconv(sd, pfx, rnd):
    return _r(rnd, fold_weight_norm(sd[pfx + "weight_g"], sd[pfx + "weight_v"]))


# models/autoencoders.py:45-68 (ResidualUnit).  Matched-rounding convention (rnd != None): tensors
# travel as their fp32 PRE-ROUND value v; a consumer sees rnd(snake(v)) as conv input and rnd(v)
# as residual -- exactly what the HIP epilogues store (Snake is applied to the fp32 accumulator).
def residual_unit(sd, pfx, v, dilation, rnd=None):
    h = _r(rnd, snake_beta(v, sd[pfx + "layers.0.alpha"], sd[pfx + "layers.0.beta"]))
    y = F.conv1d(h, _wconv(sd, pfx + "layers.1.", rnd), sd[pfx + "layers.1.bias"],
                 dilation=dilation, padding=(dilation * 6) // 2)
    h = _r(rnd, snake_beta(y, sd[pfx + "layers.2.alpha"], sd[pfx + "layers.2.beta"]))
    h = F.conv1d(h, _wconv(sd, pfx + "layers.3.", rnd), sd[pfx + "layers.3.bias"])
    return h + _r(rnd, v)


# models/autoencoders.py:88-116 (DecoderBlock) + :156-194 (OobleckDecoder)
def oobleck_decoder(sd, z, strides=(2, 4, 4, 8, 8), rnd=None, return_stages=False):
    """z [B,latent,T] -> [B,out_channels,T*prod(strides)]."""
    stages = []
    v = F.conv1d(_r(rnd, z), _wconv(sd, "layers.0.", rnd), sd["layers.0.bias"], padding=3)
    stages.append(v)
    depth = len(strides)
    for bi in range(depth):
        stride = strides[depth - 1 - bi]
        pfx = f"layers.{bi + 1}."
        h = _r(rnd, snake_beta(v, sd[pfx + "layers.0.alpha"], sd[pfx + "layers.0.beta"]))
        v = F.conv_transpose1d(h, _wconv(sd, pfx + "layers.1.", rnd), sd[pfx + "layers.1.bias"],
                               stride=stride, padding=math.ceil(stride / 2))
        for ri, dil in enumerate((1, 3, 9)):
            v = residual_unit(sd, f"{pfx}layers.{2 + ri}.", v, dil, rnd)
        stages.append(v)
    k = depth + 1
    h = _r(rnd, snake_beta(v, sd[f"layers.{k}.alpha"], sd[f"layers.{k}.beta"]))
    out = F.conv1d(h, _wconv(sd, f"layers.{k + 1}.", rnd), None, padding=3)   # bias=False, final_tanh=False
    return (out, stages) if return_stages else out


# models/autoencoders.py:71-85 (EncoderBlock) + :119-153 (OobleckEncoder)
def oobleck_encoder(sd, audio, strides=(2, 4, 4, 8, 8), rnd=None):
    """audio [B,in_ch,L] -> [B,latent_dim(=2*latent for VAE),L/prod(strides)].  The first conv reads
    the fp32 audio with fp32 (folded) weights in the HIP path as well, hence no rounding there."""
    v = F.conv1d(audio, fold_weight_norm(sd["layers.0.weight_g"], sd["layers.0.weight_v"]), sd["layers.0.bias"], padding=3)
    depth = len(strides)
    for bi in range(depth):
        stride = strides[bi]
        pfx = f"layers.{bi + 1}."
        for ri, dil in enumerate((1, 3, 9)):
            v = residual_unit(sd, f"{pfx}layers.{ri}.", v, dil, rnd)
        h = _r(rnd, snake_beta(v, sd[pfx + "layers.3.alpha"], sd[pfx + "layers.3.beta"]))
        v = F.conv1d(h, _wconv(sd, pfx + "layers.4.", rnd), sd[pfx + "layers.4.bias"],
                     stride=stride, padding=math.ceil(stride / 2))
    k = depth + 1
    h = _r(rnd, snake_beta(v, sd[f"layers.{k}.alpha"], sd[f"layers.{k}.beta"]))
    return F.conv1d(h, _wconv(sd, f"layers.{k + 1}.", rnd), sd[f"layers.{k + 1}.bias"], padding=1)


# models/bottleneck.py:46-52 (vae_sample) with the noise INJECTED (reference: randn_like)
def vae_sample(mean_scale, noise):
    mean, scale = mean_scale.chunk(2, dim=1)
    stdev = F.softplus(scale) + 1e-4
    return noise * stdev + mean


def split_sd(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# models/autoencoders.py:499-571 (decode_audio, chunked branch; reflect pad, Bartlett OLA)
def decode_audio_chunked(decode_fn, latents, chunk_size, overlap, downsampling_ratio):
    bs, _, latent_length = latents.shape
    hop = chunk_size - overlap
    cs_s, ov_s, hop_s = chunk_size * downsampling_ratio, overlap * downsampling_ratio, hop * downsampling_ratio
    win = torch.bartlett_window(ov_s * 2)
    n_chunk = int(math.ceil((latent_length - chunk_size) / hop)) + 1
    pad_len = chunk_size + hop * (n_chunk - 1) - latent_length
    latents = F.pad(latents, (0, pad_len), mode="reflect")
    xs = [decode_fn(latents[..., i * hop: i * hop + chunk_size]) for i in range(n_chunk)]
    audios = torch.zeros((bs, xs[0].shape[1], latents.shape[-1] * downsampling_ratio))
    for i, x_ in enumerate(xs):
        x_ = x_.clone()
        if i != 0:
            x_[:, :, :ov_s] *= win[None, None, :ov_s]
        if i != n_chunk - 1:
            x_[:, :, -ov_s:] *= win[None, None, -ov_s:]
        audios[..., i * hop_s: i * hop_s + cs_s] += x_
    return audios[..., :latent_length * downsampling_ratio]


# models/autoencoders.py:410-497 (encode_audio, chunked branch; zero pad, latent-domain OLA)
def encode_audio_chunked(encode_fn, audio, chunk_size, overlap, downsampling_ratio, latent_dim):
    bs, _, sample_length = audio.shape
    latent_length = sample_length // downsampling_ratio
    cs_l, ov_l, hop_l = chunk_size, overlap, chunk_size - overlap
    win = torch.bartlett_window(overlap * 2)
    cs = chunk_size * downsampling_ratio
    hop = (chunk_size - overlap) * downsampling_ratio
    n_chunk = int(math.ceil((sample_length - cs) / hop)) + 1
    pad_len = cs + hop * (n_chunk - 1) - sample_length
    audio = F.pad(audio, (0, pad_len))
    zs = [encode_fn(audio[..., i * hop: i * hop + cs]) for i in range(n_chunk)]
    latents = torch.zeros((bs, latent_dim, audio.shape[-1] // downsampling_ratio))
    for i, z_ in enumerate(zs):
        z_ = z_.clone()
        if i != 0:
            z_[:, :, :ov_l] *= win[None, None, :ov_l]
        if i != n_chunk - 1:
            z_[:, :, -ov_l:] *= win[None, None, -ov_l:]
        latents[..., i * hop_l: i * hop_l + cs_l] += z_
    return latents[..., :latent_length]


# models/autoencoders.py:573-645 (reconstruct_audio, chunked; NOTE pad uses n_chunk, not
# n_chunk-1 -- a reference quirk that is kept: :604)
def reconstruct_audio_chunked(recon_fn, audio, chunk_size, overlap, downsampling_ratio):
    bs, _, sample_length = audio.shape
    ov_s = overlap * downsampling_ratio
    win = torch.bartlett_window(ov_s * 2)
    cs = chunk_size * downsampling_ratio
    hop = cs - ov_s
    n_chunk = int(math.ceil((sample_length - cs) / hop)) + 1
    pad_len = cs + hop * n_chunk - sample_length
    audio = F.pad(audio, (0, pad_len))
    xs = [recon_fn(audio[..., i * hop: i * hop + cs], i) for i in range(n_chunk)]
    out = torch.zeros((bs, xs[0].shape[1], audio.shape[-1]))
    for i, x_ in enumerate(xs):
        x_ = x_.clone()
        if i != 0:
            x_[:, :, :ov_s] *= win[None, None, :ov_s]
        if i != n_chunk - 1:
            x_[:, :, -ov_s:] *= win[None, None, -ov_s:]
        out[:, :, i * hop: i * hop + cs] += x_
    return out[..., :sample_length]
