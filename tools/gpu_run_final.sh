#!/bin/bash
# Round-end evidence: benches of the four workloads, rocprofv3 kernel stats of the headline command, PMC passes.
# STAGE=bench|stats|pmc selects a part (each fits one gpurun call).
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
STAGE=${STAGE:-bench}
if [ $STAGE = bench ]; then
  timeout 500 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; cut -c1-250 gpurun_out/final_bench.json
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8 > gpurun_out/final_bench_b8.json 2> gpurun_out/final_bench_b8.err; cut -c1-250 gpurun_out/final_bench_b8.json
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype fp8 > gpurun_out/final_bench_fp8.json 2> gpurun_out/final_bench_fp8.err; cut -c1-250 gpurun_out/final_bench_fp8.json
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype fp8 --batch 8 > gpurun_out/final_bench_b8_fp8.json 2> gpurun_out/final_bench_b8_fp8.err; cut -c1-250 gpurun_out/final_bench_b8_fp8.json
  timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload sa2_a2a > gpurun_out/final_bench_sa2.json 2> gpurun_out/final_bench_sa2.err; cut -c1-250 gpurun_out/final_bench_sa2.json
  SAT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final_bench_dist1.json 2> gpurun_out/final_bench_dist1.err; cut -c1-250 gpurun_out/final_bench_dist1.json
fi
if [ $STAGE = stats ]; then
  cd /tmp && export TMPDIR=/tmp
  for wl in sa_open sa2_a2a; do
    rm -rf /tmp/prof_$wl
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload $wl > /dev/null 2> $R/gpurun_out/final_prof_$wl.err
    f=$(find /tmp/prof_$wl -name '*kernel_stats.csv' | head -1)
    cp "$f" $R/gpurun_out/final_kernel_stats_$wl.csv
    head -8 "$f" | cut -c1-150
  done
fi
if [ $STAGE = pmc ]; then
  cd /tmp; export TMPDIR=/tmp
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_$tag -- python $R/tools/gpu_probe.py full > $R/gpurun_out/r2_pmc_$tag.log 2>&1
    f=$(find $R/gpurun_out/r2_pmc_$tag -name "*counter_collection.csv" | head -1)
    python $R/tools/pmc_summarize.py $f > $R/gpurun_out/r2_pmc_${tag}_per_kernel.csv
    rm -rf $R/gpurun_out/r2_pmc_$tag
  done
  head -4 $R/gpurun_out/r2_pmc_FETCH_SIZE_per_kernel.csv | cut -c1-200
fi
