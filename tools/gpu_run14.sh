#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4 | tee gpurun_out/r2_pytest_14.log
for args in "--steps 2 --warmup 1 --no-cpu-baseline" "--steps 2 --warmup 1 --no-cpu-baseline --dtype fp8" "--steps 1 --warmup 1 --no-cpu-baseline --batch 8 --dtype fp8"; do
  timeout 600 python bench.py $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', round(d['value'],2), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), round(d['roofline']['avg_launch_us'],1))"
done
SAT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | head -c 300
