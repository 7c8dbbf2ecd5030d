#!/bin/bash
# GPU-box session 2: where the GEMM time goes (ablation modes of the experiments build) + kernel stats of one generation.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
SAT_HIP_EXP=1 timeout 900 python tools/gpu_probe.py ablate > gpurun_out/r2_ablate1.log 2>&1
timeout 300 python tools/gpu_probe.py attn ln > gpurun_out/r2_attn_ln1.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_prof2.log 2>&1)
find gpurun_out/r2_prof2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_kernel_stats2.csv \;
find gpurun_out/r2_prof2 -type f ! -name "*stats*" -delete
cat gpurun_out/r2_ablate1.log | tail -20; cat gpurun_out/r2_attn_ln1.log | tail; head -30 gpurun_out/r2_kernel_stats2.csv | cut -c1-200
