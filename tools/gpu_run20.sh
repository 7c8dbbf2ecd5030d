#!/bin/bash
# ResidualUnit fusion: codec parity + A/B of the headline and of the audio-to-audio workload (codec-heavy)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q -rfP --no-header -p no:cacheprovider -k "oobleck or codec or decoder or vae or generate_diffusion or reconstruct or reference_generate" > gpurun_out/r2_pytest_20.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest_20.log
grep -E "passed|failed|rc=" gpurun_out/r2_pytest_20.log | tail -3; grep -E "^\[|Error|^FAILED|^E " gpurun_out/r2_pytest_20.log | tail -30
for i in 1 2; do
  for mode in fused unfused; do
    if [ $mode = unfused ]; then export SAT_HIP_EXP=1 SAT_OOBLECK_UNFUSED=1; else unset SAT_OOBLECK_UNFUSED SAT_HIP_EXP; fi
    timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_20_${mode}_$i.json 2> gpurun_out/r2_bench_20_${mode}_$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2_bench_20_${mode}_$i.json"))
print("sa_open $mode", $i, round(d["value"],2), round(d["ms_per_step"],1))
PY
  done
done
for mode in fused unfused; do
  if [ $mode = unfused ]; then export SAT_HIP_EXP=1 SAT_OOBLECK_UNFUSED=1; else unset SAT_OOBLECK_UNFUSED SAT_HIP_EXP; fi
  timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload sa2_a2a > gpurun_out/r2_bench_20_sa2_${mode}.json 2> gpurun_out/r2_bench_20_sa2_${mode}.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_bench_20_sa2_${mode}.json"))
print("sa2_a2a $mode", round(d["value"],2), round(d["ms_per_step"],1))
PY
done
