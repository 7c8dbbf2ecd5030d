#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -6 | tee gpurun_out/r2_pytest_10.log
timeout 300 python tools/gpu_probe.py b8tiles 2>&1 | grep -v amdgpu | tee gpurun_out/r2_b8tiles.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r2_bench_10.json
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 8 2>/dev/null | cut -c1-330 | tee gpurun_out/r2_bench_10_b8.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof10 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_prof10.log 2>&1)
find gpurun_out/r2_prof10 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_kernel_stats10.csv \;
rm -rf gpurun_out/r2_prof10
head -16 gpurun_out/r2_kernel_stats10.csv | cut -c1-150
