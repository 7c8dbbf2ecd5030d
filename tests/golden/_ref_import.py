"""Import the *reference* (``/root/reference``) in THIS container to generate golden vectors.

TEST INFRASTRUCTURE ONLY.  Never imported by the product, by ``-m gpu`` tests, by
``smoke()`` or by ``bench.py`` -- ``/root/reference`` does not exist on the GPU box.

The reference imports several third-party packages at module top level that are not
installed here.  We install minimal ``sys.modules`` placeholders
for them *before* importing the reference:

* ``dac.nn.layers.WNConv1d / WNConvTranspose1d`` -- descript-audio-codec==1.0.0
  (reference ``setup.py:13``): published definition is
  ``torch.nn.utils.weight_norm(nn.Conv1d(...))`` (old-style ``weight_g``/``weight_v``).
* ``k_diffusion`` -- k-diffusion==0.1.1 (reference ``setup.py:21``): import-only
  placeholder; the sampler arithmetic is restated by this build (oracle/sampler.py) and
  is "parity unpinned" (SURVEY F10).
* ``x_transformers``, ``alias_free_torch``, ``torchaudio``, ``einops_exts``,
  ``vector_quantize_pytorch``: import-only placeholders (unused by the hot path).
"""
import importlib
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package so that submodule imports resolve
    sys.modules[name] = m
    return m


def install_stubs():
    if "dac" in sys.modules and getattr(sys.modules["dac"], "_sat_stub", False):
        return
    # --- dac -----------------------------------------------------------------
    dac = _mod("dac")
    dac._sat_stub = True
    dac_nn = _mod("dac.nn")
    dac_layers = _mod("dac.nn.layers")
    dac_quant = _mod("dac.nn.quantize")
    dac.nn = dac_nn
    dac_nn.layers = dac_layers
    dac_nn.quantize = dac_quant

    def WNConv1d(*args, **kwargs):
        return torch.nn.utils.weight_norm(nn.Conv1d(*args, **kwargs))

    def WNConvTranspose1d(*args, **kwargs):
        return torch.nn.utils.weight_norm(nn.ConvTranspose1d(*args, **kwargs))

    class Snake1d(nn.Module):  # unused on the path
        def __init__(self, channels):
            super().__init__()

    class ResidualVectorQuantize(nn.Module):  # import-only
        pass

    dac_layers.WNConv1d = WNConv1d
    dac_layers.WNConvTranspose1d = WNConvTranspose1d
    dac_layers.Snake1d = Snake1d
    dac_quant.ResidualVectorQuantize = ResidualVectorQuantize

    # --- k_diffusion: import-only ---------------------------------------------
    k = _mod("k_diffusion")
    k.external = _mod("k_diffusion.external")
    k.sampling = _mod("k_diffusion.sampling")
    k.utils = _mod("k_diffusion.utils")

    # --- x_transformers ---------------------------------------------------------
    xt = _mod("x_transformers")
    xt.ContinuousTransformerWrapper = object
    xt.Encoder = object

    # --- alias_free_torch ---------------------------------------------------------
    af = _mod("alias_free_torch")
    af.Activation1d = object

    # --- torchaudio -----------------------------------------------------------------
    ta = _mod("torchaudio")
    ta_t = _mod("torchaudio.transforms")
    ta.transforms = ta_t

    class Resample(nn.Module):
        def __init__(self, a, b):
            super().__init__()
            assert a == b, "stub Resample is identity only"

        def forward(self, x):
            return x

    ta_t.Resample = Resample

    # --- einops_exts / vector_quantize_pytorch ------------------------------------
    ee = _mod("einops_exts")
    ee.rearrange_many = lambda *a, **k: None
    vq = _mod("vector_quantize_pytorch")
    vq.ResidualVQ = object
    vq.FSQ = object


def import_reference():
    """Returns the reference's top-level ``stable_audio_tools`` package (module object).

    The build's own drop-in package has the same import name, so the reference is loaded
    under its real name with ``/root/reference`` first on ``sys.path`` and then all
    ``stable_audio_tools*`` entries are moved to ``ref_stable_audio_tools*`` aliases so
    that both can coexist in one process.
    """
    if "ref_stable_audio_tools" in sys.modules:
        return sys.modules["ref_stable_audio_tools"]
    install_stubs()
    saved = {k: v for k, v in sys.modules.items() if k == "stable_audio_tools" or k.startswith("stable_audio_tools.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        importlib.import_module("stable_audio_tools")
        for sub in ("models.factory", "models.dit", "models.transformer", "models.autoencoders", "models.diffusion",
                    "models.conditioners", "models.bottleneck", "models.pretransforms", "models.blocks",
                    "inference.generation", "inference.sampling", "inference.utils",
                    "utils.audio_utils", "utils.torch_common", "data.modification"):
            importlib.import_module("stable_audio_tools." + sub)
    finally:
        sys.path.remove(REFERENCE_ROOT)
    ref = {k: v for k, v in sys.modules.items() if k == "stable_audio_tools" or k.startswith("stable_audio_tools.")}
    for k, v in ref.items():
        sys.modules["ref_" + k] = v
        del sys.modules[k]
        # The reference imports lazily inside functions (``from .diffusion import create_diffusion_cond_from_config`` in
        # models/factory.py:14 ...).  A relative import resolves through ``__package__`` -> ``sys.modules``; with the original
        # package name it would silently pick up the BUILD's same-named module once the names are handed back.  Rename.
        v.__name__ = "ref_" + v.__name__
        if getattr(v, "__package__", None):
            v.__package__ = "ref_" + v.__package__
        if getattr(v, "__spec__", None) is not None:
            v.__spec__.name = "ref_" + v.__spec__.name
    sys.modules.update(saved)
    return sys.modules["ref_stable_audio_tools"]


def ref(sub):
    import_reference()
    return sys.modules["ref_stable_audio_tools." + sub]
