"""Counterpart of the reference's ``generate.py`` (:23-153): YAML condition tree -> prompts (x n_sample_per_cond)
-> rank-strided shards -> ``generate_diffusion_cond`` -> int16 WAV per item.

Same flags as the reference plus what an offline MI355X box needs:
  --model-config / --ckpt-path   instead of the Hugging Face download of --model-name
  --synthetic-weights SEED       random-init weights of the configured architecture (benchmarks)
  --text-embeds FILE | random    T5/CLAP encoders are out of scope for this build: the "prompt" entry of the conditioning comes
                                 from a file of precomputed embeddings; "random" (seed-from-text Gaussian [128, cond_dim]) is
                                 accepted only with --synthetic-weights, never with a real checkpoint
Launch with ``python -m torch.distributed.run --nproc-per-node N generate.py ...`` for N GPUs: one process per GPU,
full replica each, prompts ``items[rank::world]``, every rank writes its own files (as the reference does).
"""
import argparse
import hashlib
import json
import math
import os
from pathlib import Path

import torch
import yaml

from stable_audio_tools import create_model_from_config, get_pretrained_model, model_configs
from stable_audio_tools.inference.generation import generate_diffusion_cond
from stable_audio_tools.models.utils import load_ckpt_state_dict
from stable_audio_tools.utils.audio_utils import float_to_int16_audio
from stable_audio_tools.utils.torch_common import copy_state_dict, count_parameters, get_rank, get_world_size
from stable_audio_tools.utils.wav_io import save_wav_int16


def get_args():
    p = argparse.ArgumentParser()
    p.add_argument("--output-dir", type=str, required=True)
    p.add_argument("--cond-yaml-path", type=str, required=True)
    p.add_argument("--model-name", type=str, default=None, help="HF model name (needs network), e.g. stabilityai/stable-audio-open-1.0")
    p.add_argument("--model-config", type=str, default=None, help="model config json; default: the built-in SA-Open-1.0 shape")
    p.add_argument("--ckpt-path", type=str, default=None)
    p.add_argument("--synthetic-weights", type=int, default=None, metavar="SEED")
    p.add_argument("--text-embeds", type=str, default=None,
                   help="source of the text-encoder ('prompt') conditioning, which this build does not compute: a .pt / .safetensors file "
                        "mapping each prompt string (or condition path) to a precomputed [tokens, cond_dim] embedding, or 'random' = "
                        "seed-from-text Gaussian embeddings (benchmarks; only with --synthetic-weights)")
    p.add_argument("--sampler-type", type=str, default="dpmpp-3m-sde")
    p.add_argument("--sample-steps", type=int, default=100)
    p.add_argument("--cfg-scale", type=float, default=7.0)
    p.add_argument("--n-sample-per-cond", type=int, default=1)
    p.add_argument("--batch-size", type=int, default=10)
    p.add_argument("--clip-length", action="store_true")
    p.add_argument("--seed", type=int, default=-1)
    p.add_argument("--gemm-dtype", choices=["fp16", "bf16", "fp8", "fp8-all"], default=None,
                   help="build extension: operand format of the transformer GEMMs / attention (and, for fp16 / bf16, the codec).  Default: the "
                        "package default (fp16, the reference's own GPU arithmetic); bf16 = 3-4 %% faster, 8x the operand rounding; fp8 = OCP e4m3 "
                        "/ MXFP8 operands for the block GEMMs (BASELINE config 5)")
    return p.parse_args()


def flatten_conditions(tree, parent="", sep="/"):
    """Nested {group: {...: {name: {prompt:..., seconds_start:..., seconds_total:...}}}} -> {"group/.../name": cond}
    (reference generate.py:38-50: inner-most dicts are the conditions; at least two levels)."""
    out = {}
    for key, val in tree.items():
        assert isinstance(val, dict), "the condition file is a tree of dicts"
        path = f"{parent}{sep}{key}" if parent else key
        if all(not isinstance(v, dict) for v in val.values()):
            assert parent, "conditions must sit at least one level below the root"
            out[path] = dict(val)
        else:
            assert all(isinstance(v, dict) for v in val.values()), "a level mixes conditions and sub-trees"
            out.update(flatten_conditions(val, path, sep))
    return out


def text_embedding(prompt: str, dim: int, tokens: int = 128):
    seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:7], "little")
    g = torch.Generator().manual_seed(seed)
    return torch.randn(tokens, dim, generator=g)


def text_embed_source(args, cond_dim):
    """Returns ``fn(path, cond) -> [tokens, cond_dim]``.  Random embeddings carry no meaning: they are allowed only next to
    random weights; with a real checkpoint the embeddings must come from a file (computed by the T5 / CLAP encoder elsewhere)."""
    src = args.text_embeds
    if src is None:
        src = "random" if args.synthetic_weights is not None else None
    if src == "random":
        if args.synthetic_weights is None:
            raise SystemExit("--text-embeds random produces audio unrelated to the prompt text: it is only accepted together with "
                             "--synthetic-weights.  With real weights pass --text-embeds FILE (precomputed text-encoder outputs).")
        return lambda path, cond: text_embedding(str(cond["prompt"]), cond_dim)
    if src is None:
        raise SystemExit("this build has no text encoder (T5 / CLAP need a download): pass --text-embeds FILE with the precomputed "
                         "embedding of every prompt, or --synthetic-weights SEED for a random-weights benchmark run")
    if src.endswith(".safetensors"):
        from safetensors.torch import load_file
        table = load_file(src)
    else:
        table = torch.load(src, map_location="cpu", weights_only=True)

    def lookup(path, cond):
        for key in (str(cond["prompt"]), path):
            if key in table:
                emb = table[key].float()
                if emb.ndim != 2 or emb.shape[1] != cond_dim:
                    raise SystemExit(f"--text-embeds: entry '{key}' has shape {tuple(emb.shape)}, expected [tokens, {cond_dim}]")
                return emb
        raise SystemExit(f"--text-embeds: no entry for prompt '{cond['prompt']}' (nor for '{path}') in {src}")
    return lookup


def main():
    args = get_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=device)      # RCCL; only rank/world bookkeeping is used
    rank, world = get_rank(), get_world_size()

    if args.model_name:
        model, model_config = get_pretrained_model(args.model_name)
    else:
        model_config = json.load(open(args.model_config)) if args.model_config else model_configs.stable_audio_open_1_0()
        if args.model_config:     # text encoders are supplied as embeddings
            cc = model_config["model"]["conditioning"]["configs"]
            model_config["model"]["conditioning"]["configs"] = [c for c in cc if c["type"] in ("number", "int")]
        model = create_model_from_config(model_config)
        if args.ckpt_path:
            copy_state_dict(model, load_ckpt_state_dict(args.ckpt_path))
        elif args.synthetic_weights is not None:
            from stable_audio_tools import synthetic
            model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), args.synthetic_weights))
    sample_rate, sample_size = model_config["sample_rate"], model_config["sample_size"]
    model = model.to(device).eval()
    if args.gemm_dtype is not None:
        from stable_audio_tools import _config
        model.model.model.set_gemm_dtype(args.gemm_dtype)
        if model.pretransform is not None:          # the codec follows by ONE rule (same in bench.py): fp16 with fp16, bf16 with bf16 and the e4m3 modes
            model.pretransform.model.set_gemm_dtype(_config.codec_gemm_dtype(args.gemm_dtype))
    cond_dim = model_config["model"]["conditioning"]["cond_dim"]
    if model.conditioner is not None:
        model.conditioner.set_device(str(device))      # what generate_diffusion_cond does first (generation.py:125); needed by encoders here
    # ids the diffusion model consumes but the conditioner cannot produce here (text / audio encoders): fail before any work is done
    have = model.conditioner.conditioners if model.conditioner is not None else {}
    missing = [k for k in model.cross_attn_cond_ids if k not in have]
    if any(k != "prompt" for k in missing):
        raise SystemExit(f"generate.py: the model consumes cross-attention conditioning {missing} that no registered conditioner produces; "
                         f"--text-embeds only supplies the 'prompt' id (other encoders, e.g. CLAP, are outside this build)")
    needs_text = bool(missing)
    embed_of = text_embed_source(args, cond_dim) if needs_text else None

    conds = flatten_conditions(yaml.safe_load(open(args.cond_yaml_path)))
    paths, items = [], []
    for p, c in conds.items():
        for i in range(args.n_sample_per_cond):
            paths.append(f"{p}_item-{i + 1}")
            items.append(c)
    if rank == 0:
        print(f"=== model: diffusion {count_parameters(model.model) / 1e6:.3f} M params, sample size {sample_size} "
              f"({sample_size / sample_rate:.3f} s), {len(conds)} prompts x {args.n_sample_per_cond}, {world} rank(s)")
    paths, items = paths[rank::world], items[rank::world]                   # reference generate.py:119-120
    batch = max(args.batch_size // 2, 1) if args.cfg_scale != 1.0 else args.batch_size   # generate.py:75

    for i in range(int(math.ceil(len(items) / batch))):
        p_i, c_i = paths[i * batch:(i + 1) * batch], items[i * batch:(i + 1) * batch]
        cond = model.conditioner(c_i)
        if "prompt" not in cond:
            embs = [embed_of(p, c) for p, c in zip(p_i, c_i)]
            width = max(e.shape[0] for e in embs)
            emb = torch.zeros(len(embs), width, cond_dim)
            mask = torch.zeros(len(embs), width)
            for n, e in enumerate(embs):                                  # right-pad like the tokenizer (conditioners.py:322-331)
                emb[n, : e.shape[0]] = e
                mask[n, : e.shape[0]] = 1
            cond["prompt"] = (emb.to(device), mask.to(device))
        # --seed (build extension; the reference always draws a fresh seed, generate.py:129-141): every call of every rank gets its own
        # stream, so copies of one prompt on different ranks / batches never repeat the noise
        call_seed = -1 if args.seed < 0 else args.seed + rank + world * i
        order = model.cross_attn_cond_ids + [k for k in cond if k not in model.cross_attn_cond_ids]
        cond = {k: cond[k] for k in order if k in cond}
        audio = generate_diffusion_cond(model, steps=args.sample_steps, cfg_scale=args.cfg_scale, conditioning_tensors=cond,
                                        sample_size=sample_size, sigma_min=0.3, sigma_max=500, sampler_type=args.sampler_type,
                                        device=str(device), seed=call_seed)
        for n in range(audio.shape[0]):
            pcm = float_to_int16_audio(audio[n])
            if args.clip_length:
                pcm = pcm[:, : int(c_i[n]["seconds_total"] * sample_rate)]
            out = Path(args.output_dir) / f"{p_i[n]}.wav"
            out.parent.mkdir(parents=True, exist_ok=True)
            save_wav_int16(out, pcm, sample_rate)
    print(f"->->-> Rank-{rank}: Finished.")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
