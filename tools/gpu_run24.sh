#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q -rfP --no-header -p no:cacheprovider -k "test_dit_forward or full_size_dit or adaln or fp32x or layernorm_fusion or generate_diffusion" > gpurun_out/r2_pytest_24.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest_24.log
grep -E "passed|failed|rc=" gpurun_out/r2_pytest_24.log | tail -3; grep -E "^\[|Error|^FAILED|^E " gpurun_out/r2_pytest_24.log | tail -20
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_24.json 2> gpurun_out/r2_bench_24.err; cut -c1-300 gpurun_out/r2_bench_24.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof24
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof24 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r2_prof24.err
f=$(find /tmp/prof24 -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/r2_kernel_stats_24.csv
head -24 "$f" | cut -c1-140
