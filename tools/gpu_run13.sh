#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench_final_b1.json 2> gpurun_out/r2_bench_final_b1.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype fp8 > gpurun_out/r2_bench_fp8.json 2>/dev/null
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 8 > gpurun_out/r2_bench_b8.json 2>/dev/null
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 8 --dtype fp8 > gpurun_out/r2_bench_b8_fp8.json 2>/dev/null
timeout 900 python bench.py --steps 1 --warmup 1 --workload sa2_a2a > gpurun_out/r2_bench_sa2.json 2>/dev/null
SAT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_forcedist.json 2> gpurun_out/r2_bench_forcedist.err
for f in gpurun_out/r2_bench_final_b1.json gpurun_out/r2_bench_fp8.json gpurun_out/r2_bench_b8.json gpurun_out/r2_bench_b8_fp8.json gpurun_out/r2_bench_sa2.json gpurun_out/r2_bench_forcedist.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"],2), round(d["ms_per_step"],1), round(d["roofline"]["achieved"],1), round(d["roofline"]["avg_launch_us"],1), d.get("rccl_ranks"), d.get("cpu_baseline",{}).get("value"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -3 gpurun_out/r2_bench_forcedist.err
