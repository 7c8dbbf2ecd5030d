"""The conditioner front-end of a conditioned diffusion model, as far as this build computes it itself.

Counterparts of ``Conditioner`` / ``NumberConditioner`` / ``MultiConditioner`` / ``create_multi_conditioner_from_conditioning_config``
(reference ``models/conditioners.py:18-102, 506-599``) and of the ``NumberEmbedder`` they embed numbers with
(``models/adp.py:680-701, 1495-1514``): same module tree and parameter names, so reference checkpoints load.

``NumberConditioner`` runs on the HIP C ABI (``sat_number_embed``: clamp, normalise, Fourier features and the Linear in one
launch per call); like the rest of the package it has no CPU path.  ``T5Conditioner`` (conditioners.py:261-346) runs the T5 encoder
stack on the C ABI as well (``sat_t5_*``, csrc/t5_encoder.hip); its weights and tokenizer are not part of a stable-audio checkpoint
-- the reference downloads them -- so it is registered only when they are in the local Hugging Face cache (or handed over with
``load_encoder``).  The other encoder-backed types (CLAP, phonemes, ...) and a T5 without weights are recorded in
``MultiConditioner.external_ids``: the caller supplies those entries through ``conditioning_tensors=`` -- the "random T5 embeds"
configuration of BASELINE.json.
"""
import ctypes
import typing as tp

import torch
from torch import nn

from .. import _hip

# conditioner types whose tensors must come from outside (encoders this build does not ship)
_EXTERNAL_TYPES = ("clap_text", "clap_audio", "phoneme", "lut", "pretransform", "int")


class Conditioner(nn.Module):
    def __init__(self, dim: int, output_dim: int, project_out: bool = False):
        super().__init__()
        self.dim, self.output_dim = dim, output_dim
        self.proj_out = nn.Linear(dim, output_dim) if (project_out or dim != output_dim) else nn.Identity()

    def set_device(self, device: tp.Any) -> None:
        raise NotImplementedError()


class LearnedPositionalEmbedding(nn.Module):
    """Parameter holder of adp.py:680-694 (``weights`` [dim / 2]); evaluated inside ``sat_number_embed``."""

    def __init__(self, dim: int):
        super().__init__()
        if dim % 2:
            raise ValueError("LearnedPositionalEmbedding needs an even dim")
        self.weights = nn.Parameter(torch.randn(dim // 2))


class NumberEmbedder(nn.Module):
    """Parameter holder of adp.py:1495-1514: ``embedding.0`` = LearnedPositionalEmbedding(dim), ``embedding.1`` = Linear(dim + 1, features)."""

    def __init__(self, features: int, dim: int = 256):
        super().__init__()
        self.features = features
        self.embedding = nn.Sequential(LearnedPositionalEmbedding(dim), nn.Linear(dim + 1, features))

    def forward(self, values: torch.Tensor, min_val: float = 0.0, max_val: float = 1.0) -> torch.Tensor:
        """values [B] (raw numbers, clamped to [min_val, max_val] and normalised by the kernel) -> [B, features]."""
        pos, lin = self.embedding[0].weights, self.embedding[1]
        values = values.to(pos.device, torch.float32).contiguous()
        out = torch.empty((values.numel(), self.features), dtype=torch.float32, device=pos.device)
        _hip.check(_hip.lib().sat_number_embed(_hip.ptr(values), values.numel(), float(min_val), float(max_val),
                                               _hip.ptr(pos.detach().float().contiguous()), pos.numel(),
                                               _hip.ptr(lin.weight.detach().float().contiguous()), _hip.ptr(lin.bias.detach().float().contiguous()),
                                               self.features, _hip.ptr(out), _hip.stream()))
        return out


class NumberConditioner(Conditioner):
    """A list of numbers -> ``[B, 1, output_dim]`` embeddings and an all-ones ``[B, 1]`` mask (conditioners.py:64-102)."""

    def __init__(self, output_dim: int, min_val: float = 0, max_val: float = 1):
        super().__init__(output_dim, output_dim)
        self.min_val, self.max_val = min_val, max_val
        self.embedder = NumberEmbedder(features=output_dim)

    @property
    def device(self):
        return next(self.embedder.parameters()).device

    def set_device(self, device):
        self.to(device)

    @torch.no_grad()
    def forward(self, floats: tp.List[float]) -> tp.Any:
        values = torch.tensor([float(v) for v in floats], dtype=torch.float32)
        embeds = self.embedder(values, self.min_val, self.max_val).unsqueeze(1)
        return [embeds, torch.ones(embeds.shape[0], 1, device=embeds.device)]


class T5Conditioner(Conditioner):
    """Prompts -> ``[B, max_length, output_dim]`` T5 encoder states (zero at padding) and the boolean attention mask
    (conditioners.py:261-346).  The encoder (``transformers.T5EncoderModel`` in the reference, under fp16 autocast) runs in fp32 on
    ``sat_t5_encode``; ``proj_out`` and the final mask multiply happen in the same call.  As in the reference the encoder weights
    live outside the module's state dict."""

    T5_MODEL_DIMS = {"t5-small": 512, "t5-base": 768, "t5-large": 1024, "t5-3b": 1024, "t5-11b": 1024,
                     "google/flan-t5-small": 512, "google/flan-t5-base": 768, "google/flan-t5-large": 1024,
                     "google/flan-t5-xl": 2048, "google/flan-t5-xxl": 4096}

    def __init__(self, output_dim: int, t5_model_name: str = "t5-base", max_length: int = 128, enable_grad: bool = False,
                 project_out: bool = False):
        if t5_model_name not in self.T5_MODEL_DIMS:
            raise ValueError(f"Unknown T5 model name: {t5_model_name}")
        if enable_grad:
            raise NotImplementedError("T5Conditioner: the HIP encoder is inference-only (enable_grad=True is a training option)")
        super().__init__(self.T5_MODEL_DIMS[t5_model_name], output_dim, project_out=project_out)
        self.t5_model_name, self.max_length = t5_model_name, max_length
        self.tokenizer = None
        self.__dict__["_enc"] = None          # (plan handle, device, params_version of proj_out) -- not a submodule, not in state_dict
        self.__dict__["_enc_src"] = None      # (state dict on the host, config) kept to rebuild the plan after .to(device) / a weight load
        self.__dict__["_ws"] = None
        self._device = "cpu"

    # ---------------------------------------------------------------- encoder weights
    @classmethod
    def cached_locally(cls, t5_model_name: str) -> bool:
        """True when tokenizer and encoder weights of ``t5_model_name`` can be loaded without a download."""
        try:
            from huggingface_hub import try_to_load_from_cache
            from transformers import AutoConfig, AutoTokenizer
            AutoConfig.from_pretrained(t5_model_name, local_files_only=True)
            AutoTokenizer.from_pretrained(t5_model_name, local_files_only=True)
            # a cache that holds only config.json would register the conditioner and then fail at the first forward: probe the weights
            return any(isinstance(try_to_load_from_cache(t5_model_name, f), str)
                       for f in ("model.safetensors", "pytorch_model.bin", "model.safetensors.index.json", "pytorch_model.bin.index.json"))
        except Exception:
            return False

    def load_encoder(self, state_dict: tp.Dict[str, torch.Tensor], config: tp.Any, tokenizer: tp.Any = None) -> "T5Conditioner":
        """Hand over a Hugging Face T5 encoder: ``state_dict`` with the checkpoint's key names (``T5EncoderModel.state_dict()``),
        ``config`` with the ``T5Config`` attributes, ``tokenizer`` a callable with the ``transformers`` tokenizer interface."""
        if config.d_model != self.dim:
            raise ValueError(f"T5 config d_model {config.d_model} != {self.dim} expected for {self.t5_model_name}")
        act = getattr(config, "feed_forward_proj", "relu")
        if act not in ("relu", "gated-gelu"):
            raise NotImplementedError(f"T5 feed_forward_proj {act!r} (supported: 'relu', 'gated-gelu')")
        keep = {k: v.detach().to("cpu", torch.float32) for k, v in state_dict.items() if k.startswith("encoder.") or k == "shared.weight"}
        self.__dict__["_enc_src"] = (keep, config)
        self._drop_plan()
        if tokenizer is not None:
            self.tokenizer = tokenizer
        return self

    def _load_from_cache(self):
        from transformers import AutoTokenizer, T5EncoderModel
        try:
            tokenizer = AutoTokenizer.from_pretrained(self.t5_model_name, local_files_only=True)
            model = T5EncoderModel.from_pretrained(self.t5_model_name, local_files_only=True)
        except Exception as e:
            raise _hip.SatError(f"T5Conditioner: '{self.t5_model_name}' is not in the local Hugging Face cache ({type(e).__name__}); "
                                "call load_encoder(state_dict, config, tokenizer) or pass the embeddings through conditioning_tensors=") from e
        self.load_encoder(model.state_dict(), model.config, tokenizer)

    def _drop_plan(self):
        enc = self.__dict__.get("_enc")
        if enc is not None:
            _hip.lib().sat_t5_plan_destroy(enc[0])
        self.__dict__["_enc"] = None

    def __del__(self):
        try:
            self._drop_plan()
        except Exception:
            pass

    def _plan(self):
        from . import _init
        dev = torch.device(self._device)
        if dev.type != "cuda":
            raise _hip.SatError("T5Conditioner must be on a HIP device (set_device('cuda')); there is no CPU path")
        ver = _init.params_version(self)
        enc = self.__dict__["_enc"]
        if enc is not None and enc[1] == dev and enc[2] == ver:
            return enc[0]
        self._drop_plan()
        if self.__dict__["_enc_src"] is None:
            self._load_from_cache()
        sd, config = self.__dict__["_enc_src"]
        lib = _hip.lib()
        has_proj = isinstance(self.proj_out, nn.Linear)
        cfg = _hip.SatT5Cfg(config.vocab_size, config.d_model, config.d_kv, config.d_ff, config.num_layers, config.num_heads,
                            config.relative_attention_num_buckets, getattr(config, "relative_attention_max_distance", 128),
                            1 if getattr(config, "feed_forward_proj", "relu") == "gated-gelu" else 0,
                            self.output_dim if has_proj else 0, float(config.layer_norm_epsilon))
        plan = ctypes.c_void_p()
        _hip.check(lib.sat_t5_plan_create(ctypes.byref(cfg), ctypes.byref(plan)))
        keep = []
        tensors = dict(sd)
        if has_proj:
            tensors["proj_out.weight"], tensors["proj_out.bias"] = self.proj_out.weight.detach(), self.proj_out.bias.detach()
        try:
            for name, t in tensors.items():
                td = t.to(dev, torch.float32).contiguous()
                keep.append(td)
                _hip.check(lib.sat_t5_plan_set_tensor(plan, name.encode(), _hip.ptr(td), td.numel()))
            _hip.check(lib.sat_t5_plan_finalize(plan, _hip.stream()))
        except Exception:
            lib.sat_t5_plan_destroy(plan)
            raise
        torch.cuda.current_stream().synchronize()      # the plan copied from `keep`
        self.__dict__["_enc"] = (plan, dev, ver)
        return plan

    # ---------------------------------------------------------------- Conditioner interface
    def set_device(self, device):
        self.to(device)
        self._device = str(device)

    @torch.no_grad()
    def encode_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        """Tokenised prompts ``[B, L]`` -> (``[B, L, output_dim]`` fp32, boolean mask), the tail of conditioners.py:326-343."""
        plan = self._plan()
        dev = torch.device(self._device)
        ids = input_ids.to(dev, torch.int32).contiguous()
        mask = attention_mask.to(dev, torch.int32).contiguous()
        b, l = ids.shape
        lib = _hip.lib()
        need = ctypes.c_size_t()
        _hip.check(lib.sat_t5_workspace_bytes(plan, b, l, ctypes.byref(need)))
        ws = self.__dict__["_ws"]
        if dev.index is None:                      # torch.device("cuda") != tensor.device ("cuda:0"): compare resolved devices
            dev = torch.device(dev.type, torch.cuda.current_device())
        if ws is None or ws.numel() < need.value or ws.device != dev:
            ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
            self.__dict__["_ws"] = ws
        out = torch.empty((b, l, self.output_dim), dtype=torch.float32, device=dev)
        _hip.check(lib.sat_t5_encode(plan, _hip.ptr(ids), _hip.ptr(mask), _hip.ptr(out), b, l, 1, _hip.ptr(ws), ws.numel(), _hip.stream()))
        return out, mask.to(torch.bool)

    def forward(self, texts: tp.List[str]) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        if self.tokenizer is None:
            self._load_from_cache()
        encoded = self.tokenizer(texts, truncation=True, max_length=self.max_length, padding="max_length", return_tensors="pt")
        return self.encode_ids(encoded["input_ids"], encoded["attention_mask"])


class MultiConditioner(nn.Module):
    """Applies each conditioner to its entry of the per-item metadata dicts (conditioners.py:506-549)."""

    def __init__(self, conditioners: tp.Dict[str, Conditioner], default_keys: tp.Dict[str, str] = {}):
        super().__init__()
        self.conditioners = nn.ModuleDict(conditioners)
        self.default_keys = default_keys
        self.external_ids: tp.List[str] = []

    def set_device(self, device):
        for conditioner in self.conditioners.values():
            conditioner.set_device(device)

    def forward(self, batch_metadata: tp.List[tp.Dict[str, tp.Any]]) -> tp.Dict[str, tp.Any]:
        out = {}
        for key, conditioner in self.conditioners.items():
            lookup, inputs = key, []
            for item in batch_metadata:
                if lookup not in item:
                    if lookup not in self.default_keys:
                        raise ValueError(f"Conditioner key {lookup} not found in batch metadata")
                    lookup = self.default_keys[lookup]        # sticks for the rest of the batch, as in the reference (:533-538)
                value = item[lookup]
                # the reference unwraps ANY list, but a tuple only when it has exactly one element (operator precedence, :541)
                if isinstance(value, list) or (isinstance(value, tuple) and len(value) == 1):
                    value = value[0]
                inputs.append(value)
            out[key] = conditioner(inputs)
        return out


def create_multi_conditioner_from_conditioning_config(config: tp.Dict[str, tp.Any]) -> MultiConditioner:
    """conditioners.py:552-599 for the conditioner types this build evaluates: "number", and "t5" when its weights are in the
    local Hugging Face cache; the other encoder-backed types are listed in ``external_ids`` instead of being instantiated."""
    built, external = {}, []
    for entry in config["configs"]:
        kind = entry["type"]
        if kind == "number":
            built[entry["id"]] = NumberConditioner(**{"output_dim": config["cond_dim"], **entry["config"]})
        elif kind == "t5" and T5Conditioner.cached_locally(entry["config"].get("t5_model_name", "t5-base")):
            built[entry["id"]] = T5Conditioner(**{"output_dim": config["cond_dim"], **entry["config"]})
        elif kind == "t5" or kind in _EXTERNAL_TYPES:
            external.append(entry["id"])
        else:
            raise ValueError(f"Unknown conditioner type: {kind}")
    multi = MultiConditioner(built, default_keys=config.get("default_keys", {}))
    multi.external_ids = external
    return multi
