#!/bin/bash
# LayerNorm fold: kernel + model parity, then interleaved A/B of the headline bench
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q -rfP --no-header -p no:cacheprovider -k "ln_fold or layernorm_fusion or test_dit_forward or test_gemm_f32 or test_gemm_swiglu or test_qkv_rope or full_size_dit" > gpurun_out/r2_pytest_16.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest_16.log
grep -E "passed|failed|rc=" gpurun_out/r2_pytest_16.log | tail -3; grep -E "^\[|Error|^FAILED|^E " gpurun_out/r2_pytest_16.log | tail -40
for i in 1 2; do
  for ln in fused standalone; do
    timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --layernorm $ln > gpurun_out/r2_bench_16_${ln}_$i.json 2> gpurun_out/r2_bench_16_${ln}_$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2_bench_16_${ln}_$i.json"))
print("$ln", $i, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
  done
done
