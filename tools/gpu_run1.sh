#!/bin/bash
# GPU-box session 1 of round 2: parity suite, epilogue A/B, bench, kernel stats.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 1500 python -m pytest tests -m gpu -q -rfP --no-header -p no:cacheprovider > gpurun_out/r2_pytest1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest1.log
timeout 600 python tools/gpu_probe.py epi > gpurun_out/r2_epi1.log 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_prof1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_prof1.log 2>&1)
find gpurun_out/r2_prof1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_kernel_stats1.csv \;
find gpurun_out/r2_prof1 -type f ! -name "*stats*" -delete
tail -5 gpurun_out/r2_pytest1.log; cat gpurun_out/r2_epi1.log | tail -20; cat gpurun_out/r2_bench1.json
