"""Counterpart of the reference's ``reconstruct_audios.py`` (:40-149): every WAV under --audio-dir is encoded and
decoded chunk-wise by the autoencoder (``AudioAutoencoder.reconstruct_audio``: chunking + Bartlett cross-fade) and
written to --output-dir; files are sharded ``files[rank::world]``.  WAV I/O uses the stdlib (no torchaudio here), so
inputs must be PCM WAV at the model's sample rate."""
import argparse
import json
import os
from pathlib import Path

import torch

from stable_audio_tools import create_model_from_config, model_configs
from stable_audio_tools.data.dataset import get_audio_filenames
from stable_audio_tools.data.modification import Mono, Stereo
from stable_audio_tools.models.utils import load_ckpt_state_dict
from stable_audio_tools.utils.torch_common import copy_state_dict, count_parameters, get_rank, get_world_size
from stable_audio_tools.utils.wav_io import load_wav, save_wav_float


def get_args():
    p = argparse.ArgumentParser()
    p.add_argument("--model-config", type=str, default=None, help="autoencoder config json; default: built-in Stable Audio VAE shape")
    p.add_argument("--ckpt-path", type=str, default=None)
    p.add_argument("--synthetic-weights", type=int, default=None, metavar="SEED")
    p.add_argument("--audio-dir", type=str, required=True)
    p.add_argument("--output-dir", type=str, required=True)
    p.add_argument("--frame-duration", type=float, default=1.0)
    p.add_argument("--overlap-rate", type=float, default=0.01)
    p.add_argument("--batch-size", type=int, default=20)
    return p.parse_args()


def main():
    args = get_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=device)
    rank, world = get_rank(), get_world_size()

    cfg = json.load(open(args.model_config)) if args.model_config else model_configs.stable_audio_vae()
    model = create_model_from_config(cfg)
    if args.ckpt_path:
        copy_state_dict(model, load_ckpt_state_dict(args.ckpt_path))
    elif args.synthetic_weights is not None:
        from stable_audio_tools import synthetic
        model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), args.synthetic_weights))
    model = model.to(device).eval()
    sr, ratio = model.sample_rate, model.downsampling_ratio
    chunk_size = int((args.frame_duration * sr) / ratio)                              # reconstruct_audios.py:86-87
    overlap = max(int((args.frame_duration * sr * args.overlap_rate) / ratio), 1)
    if rank == 0:
        print(f"=== autoencoder: {count_parameters(model) / 1e6:.2f} M params, sr {sr}, ratio {ratio}, latent {model.latent_dim}, "
              f"chunk {chunk_size} / overlap {overlap} latents")
    files = sorted(f for f in get_audio_filenames(args.audio_dir) if f.lower().endswith(".wav"))[rank::world]
    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    (out_dir.parent / "original").mkdir(parents=True, exist_ok=True)
    fix_channels = Mono() if model.in_channels == 1 else Stereo()
    for f in files:
        audio, in_sr = load_wav(f)
        audio = audio.to(device)
        if in_sr != sr:                             # reconstruct_audios.py:33-35 of the reference (torchaudio Resample): HIP polyphase kernel
            from stable_audio_tools.inference.resample import resample
            audio = resample(audio, in_sr, sr)
        audio = fix_channels(audio).unsqueeze(0)
        rec = model.reconstruct_audio(audio, chunked=True, chunk_size=chunk_size, overlap=overlap, max_batch_size=args.batch_size)
        name = os.path.basename(f)
        save_wav_float(out_dir / name, rec.squeeze(0), sr)
        save_wav_float(out_dir.parent / "original" / name, audio.squeeze(0), sr)
        print(f"rank {rank} : {f}")
    print(f"[Finished : rank-{rank}]")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
