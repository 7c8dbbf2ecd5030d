"""MI355X-native drop-in for the diffusion-sampling hot path of ``stable_audio_tools``.

Same import surface as the reference package for the path (reference
``stable_audio_tools/__init__.py:1-2``): ``create_model_from_config``,
``create_model_from_config_path``, ``get_pretrained_model``.  The arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI of ``include/sat_hip.h``
(``lib/libsat_hip.so``); there is no CPU fallback.
"""
from ._config import default_gemm_dtype, set_default_gemm_dtype
from .models.factory import create_model_from_config, create_model_from_config_path
from .models.pretrained import get_pretrained_model

__all__ = ["create_model_from_config", "create_model_from_config_path", "get_pretrained_model", "default_gemm_dtype", "set_default_gemm_dtype"]
