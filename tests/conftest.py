import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "friendly-stable-audio-tools_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The suite runs on the build a user gets: the package default operand format (fp16 since round 4).  SAT_TEST_DTYPE=bf16 runs the same suite
# on the bf16 build (tests/util.py: SUITE, gates / 4 under fp16); tests parametrised over the format cover both either way.
from stable_audio_tools import _config  # noqa: E402
from util import SUITE  # noqa: E402

PACKAGE_DEFAULT_GEMM_DTYPE = _config.set_default_gemm_dtype(SUITE.gemm_dtype)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
