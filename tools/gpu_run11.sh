#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4 | tee gpurun_out/r2_pytest_11.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof11 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_prof11.log 2>&1)
find gpurun_out/r2_prof11 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_kernel_stats11.csv \;
rm -rf gpurun_out/r2_prof11
grep -E "proj_kernel|layernorm|attention" gpurun_out/r2_kernel_stats11.csv | cut -c1-60,150-260
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r2_bench_11.json
