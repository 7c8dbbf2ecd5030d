#!/bin/bash
# interleaved A/B of the headline bench: fused vs standalone LayerNorm plan (TAG=.. names the outputs; REPS rounds)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
TAG=${TAG:-ab}
for i in $(seq 1 ${REPS:-2}); do
  for ln in fused standalone; do
    timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --layernorm $ln $BENCH_ARGS > gpurun_out/r2_bench_${TAG}_${ln}_$i.json 2> gpurun_out/r2_bench_${TAG}_${ln}_$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2_bench_${TAG}_${ln}_$i.json"))
print("$ln", $i, round(d["value"],2), round(d["ms_per_step"],1), round(d["roofline"]["avg_launch_us"],2))
PY
  done
done
