"""Build-time guard (CPU, hipcc cross-compiles): the register budgets the kernel designs rest on.

The 8-phase GEMM runs 8 waves of 256 registers with NO scratch -- one hoisted address pair is enough to push a handful of values
into private memory around the main loop (it happened in round 4 when the schedule moved into the kernel arguments); the attention
kernel's two-KV-group layout needs <= 128 VGPRs for its four waves per SIMD.  Both operand builds are checked."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "friendly-stable-audio-tools_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _kernels(src, defines, tmp_path):
    out = os.path.join(tmp_path, "k.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out] + defines,
                   check=True, capture_output=True)
    text = open(out).read()
    meta = {}
    for blk in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
    return meta


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("defines", [[], ["-DSAT_OPERAND_F16"]], ids=["bf16", "f16"])
def test_ph8_kernels_use_no_scratch(defines, tmp_path):
    meta = _kernels("gemm_ph8.hip", defines, str(tmp_path))
    ph8 = {k: v for k, v in meta.items() if "gemm_ph8_kernel" in k}
    assert len(ph8) >= 3, sorted(meta)
    for name, m in ph8.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, f"{name}: {m}"
        assert m["vgpr_count"] <= 256, f"{name}: {m}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("defines", [[], ["-DSAT_OPERAND_F16"]], ids=["bf16", "f16"])
def test_attention_kernels_fit_four_waves_per_simd(defines, tmp_path):
    meta = _kernels("attention.hip", defines, str(tmp_path))
    att = {k: v for k, v in meta.items() if "attention_kernel" in k}
    assert att, sorted(meta)
    for name, m in att.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, f"{name}: {m}"
        assert m["vgpr_count"] <= 128, f"{name}: {m}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("defines", [[], ["-DSAT_OPERAND_F16"]], ids=["bf16", "f16"])
def test_one_prompt_gemm_tiles_keep_their_occupancy(defines, tmp_path):
    """The one-prompt tiles of gemm_bf16.hip and the register budgets their occupancy rests on: the 12-wave 256 x 192 heads tile (to_qkv; three
    waves per SIMD = 168 VGPRs -- round 5 fetches the rotation table of both row blocks up front, which took it from 142 to 158) and the
    two-K-group 128 x 128 tile (FF-out / to_out; 8 waves of 256 registers), both without scratch."""
    meta = _kernels("gemm_bf16.hip", defines, str(tmp_path))
    qkv = {k: v for k, v in meta.items() if "gemm_pipe_kernelILi256ELi192ELi64ELi4ELi3ELi2ELi3ELi0ELi1ELb0E" in k}
    kgroup = {k: v for k, v in meta.items() if "gemm_pipe_kernelILi128ELi128ELi64ELi2ELi2ELi2ELi0ELi0ELi2ELb0E" in k}
    assert len(qkv) == 1 and len(kgroup) == 1, sorted(meta)[:5]
    for name, m in list(qkv.items()) + list(kgroup.items()):
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, f"{name}: {m}"
    assert next(iter(qkv.values()))["vgpr_count"] <= 168
    assert next(iter(kgroup.values()))["vgpr_count"] <= 256


def _regs(tok):
    """VGPR numbers named by one operand token: v12 -> {12}, v[4:7] -> {4..7}"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(a), int(b) + 1))
    out |= {int(a) for a in re.findall(r"\bv(\d+)\b", tok)}
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("defines", [[], ["-DSAT_OPERAND_F16"]], ids=["bf16", "f16"])
def test_ph8_asm_loads_are_not_read_before_their_wait(defines, tmp_path):
    """gemm_ph8.hip issues loads from inline assembly with plain "=v" outputs (rope_load: global_load_dwordx4; lds_ld32 / 64 /
    128: ds_read) and waits for them LATER with its own s_waitcnt (rope_wait / lds_wait) -- the compiler believes the destination registers are
    valid from the asm statement on, so a copy, live-range split or spill it inserted in that window would read them before the data lands.  The
    no-scratch test above rules out spills; this one reads the ISA of every 8-phase kernel: between an asm load and the first s_waitcnt on its
    counter (vmcnt for global loads, lgkmcnt for LDS reads), no instruction may name one of its destination registers.  Verified with the
    hipcc of ROCm 7.2.0 (the image this repository builds in); it is a property of the allocation, so it is re-checked at every build."""
    out = os.path.join(str(tmp_path), "k.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(CSRC, "gemm_ph8.hip"), "-o", out] + defines,
                   check=True, capture_output=True)
    lines = open(out).read().split("\n")
    checked = {"global_load": 0, "ds_read": 0}
    in_asm = False
    pending = []          # (counter, destination registers, line number, text) of asm loads still in flight
    for no, raw in enumerate(lines):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.endswith(":") or line.startswith("."):
            if line.startswith(".Lfunc_end") or line.startswith(".end_amdhsa_kernel"):
                pending = []
            continue
        op = line.split()[0]
        if op == "s_waitcnt":
            if "vmcnt" in line:
                pending = [p_ for p_ in pending if p_[0] != "vmcnt"]
            if "lgkmcnt" in line:
                pending = [p_ for p_ in pending if p_[0] != "lgkmcnt"]
            continue
        operands = line[len(op):]
        used = _regs(operands)
        for counter, dest, at, text in pending:
            assert not (used & dest), f"line {no + 1}: `{line}` names v{sorted(used & dest)} before the wait of the asm load at line {at + 1}: `{text}`"
        if in_asm and (op.startswith("global_load") or op.startswith("ds_read")):
            dest = _regs(operands.split(",")[0])
            kind = "global_load" if op.startswith("global_load") else "ds_read"
            checked[kind] += 1
            pending.append(("vmcnt" if kind == "global_load" else "lgkmcnt", dest, no, line))
    assert checked["global_load"] >= 8 and checked["ds_read"] >= 30, checked
