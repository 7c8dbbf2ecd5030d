"""The C-ABI library: loads, exports every symbol declared in include/sat_hip.h, and the product refuses to
run without a HIP device (no CPU fallback, no route into the oracle).  No compute calls here."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "friendly-stable-audio-tools_amd")


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sat_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from stable_audio_tools import _hip
    lib = _hip.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libsat_hip.so does not export {name}"
    assert sorted(_hip.exported_symbols()) == declared, "ctypes signature table and header disagree"
    assert lib.sat_version() == _hip.ABI_VERSION == 6


def test_library_has_no_torch_dependency():
    from stable_audio_tools import _hip
    out = subprocess.run(["ldd", _hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "libtorch" not in out and "libc10" not in out, "the C ABI must not link against torch"
    assert "libamdhip64" in out


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments with SAT_E_* codes and a message instead of crashing."""
    import ctypes
    from stable_audio_tools import _hip
    lib = _hip.lib()
    plan = ctypes.c_void_p()
    bad = _hip.SatDitCfg(64, 1000, 2, 10, 0, 0, 0, 128)        # dim_heads != 64
    assert lib.sat_dit_plan_create(ctypes.byref(bad), ctypes.byref(plan)) == -2
    assert b"dim_heads" in lib.sat_last_error()
    ok = _hip.SatDitCfg(64, 256, 2, 4, 128, 128, 64, 128)
    assert lib.sat_dit_plan_create(ctypes.byref(ok), ctypes.byref(plan)) == 0
    need = ctypes.c_size_t()
    assert lib.sat_dit_workspace_bytes(plan, 2, 64, ctypes.byref(need)) == -5      # not finalized
    assert lib.sat_dit_plan_set_tensor(plan, b"x", None, 4) == -1
    lib.sat_dit_plan_destroy(plan)
    ocfg = _hip.SatOobleckCfg()
    ocfg.is_decoder, ocfg.io_channels, ocfg.channels, ocfg.latent_dim, ocfg.n_blocks = 1, 2, 100, 64, 2
    assert lib.sat_oobleck_plan_create(ctypes.byref(ocfg), ctypes.byref(plan)) == -2
    assert lib.sat_dpmpp3m_update(None, None, None, None, None, 0.0, 1.0, 0.0, 0.0, 0.0, 10, None) == -1
    # the A/B switches are per plan (sat_dit_cfg, ABI version 5) and take their documented values only
    for tile, want in ((22, 0), (81, 0), (82, 0), (80, 0), (0, 0), (7, -1)):
        cfg = _hip.SatDitCfg(64, 256, 2, 4, 128, 128, 64, 128, 0, 0, 0, 1, 0, tile)
        rc = lib.sat_dit_plan_create(ctypes.byref(cfg), ctypes.byref(plan))
        assert rc == want, (tile, rc, lib.sat_last_error())
        if rc == 0:
            lib.sat_dit_plan_destroy(plan)
    assert b"tile_policy" in lib.sat_last_error()
    cfg = _hip.SatDitCfg(64, 256, 2, 4, 128, 128, 64, 128, 0, 0, 0, 1, 2, 0)
    assert lib.sat_dit_plan_create(ctypes.byref(cfg), ctypes.byref(plan)) == -1 and b"cross_attention" in lib.sat_last_error()
    # a struct laid out for an older header (fp8_families where ln_fold used to be) fails loudly instead of silently mis-reading it
    cfg = _hip.SatDitCfg(64, 256, 2, 4, 128, 128, 64, 128, 0, 0, 1, 0)
    assert lib.sat_dit_plan_create(ctypes.byref(cfg), ctypes.byref(plan)) == -1 and b"fp8_families" in lib.sat_last_error()
    # ABI version 6: the caller's struct size travels with the call -- a struct of another version is rejected by its size, not read
    # past its end; the constructor Python uses carries sizeof(its struct)
    ok = _hip.SatDitCfg(64, 256, 2, 4, 128, 128, 64, 128)
    assert ctypes.sizeof(ok) == 56
    for size, want in ((56, 0), (52, -1), (60, -1), (0, -1)):
        rc = lib.sat_dit_plan_create_sized(ctypes.byref(ok), size, ctypes.byref(plan))
        assert rc == want, (size, rc, lib.sat_last_error())
        if rc == 0:
            lib.sat_dit_plan_destroy(plan)
    assert b"bytes" in lib.sat_last_error()


def test_library_exports_nothing_but_the_header():
    """built with -fvisibility=hidden, the dynamic symbol table holds the sat_* functions of include/sat_hip.h and nothing
    else of ours -- no mangled launchers, kernel stubs or template instantiations (106 of them in round 4)."""
    from stable_audio_tools import _hip
    out = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    ours = sorted(n for n in names if n.startswith("sat_"))
    assert ours == _declared_symbols(), sorted(set(ours) ^ set(_declared_symbols()))
    other = [n for n in names if not n.startswith("sat_") and n not in ("_init", "_fini", "__bss_start", "_edata", "_end")]
    assert not other, f"{len(other)} non-sat_ dynamic symbols exported, e.g. {other[:5]}"


def test_product_has_no_cpu_path():
    from stable_audio_tools import _hip
    from stable_audio_tools import model_configs as MC
    import stable_audio_tools as S
    model = S.create_model_from_config(MC.reduced(MC.stable_audio_vae()))
    with pytest.raises(_hip.SatError):
        model.decode(torch.zeros(1, 64, 4))          # CPU tensors / CPU module -> loud failure
    with pytest.raises(_hip.SatError):
        model.bottleneck.encode(torch.zeros(1, 128, 4))
    from stable_audio_tools.utils.audio_utils import float_to_int16_audio
    with pytest.raises(_hip.SatError):
        float_to_int16_audio(torch.zeros(2, 8))
    dit = S.create_model_from_config(MC.reduced(MC.stable_audio_open_1_0()))
    with pytest.raises(_hip.SatError):
        dit.model(torch.zeros(1, 64, 8), torch.zeros(1), cross_attn_cond=torch.zeros(1, 4, 128), global_cond=torch.zeros(1, 256))


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "/root/reference" in src:
                    bad.append(os.path.join(base, f))
    assert not bad, f"product files reference the oracle / the reference tree: {bad}"
    for f in ("bench.py",):
        src = open(os.path.join(ROOT, f)).read()
        assert "/root/reference" not in src


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from stable_audio_tools import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_hip.SatError, match="not built"):
        _hip.lib()
