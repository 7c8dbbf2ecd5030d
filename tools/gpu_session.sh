#!/bin/bash
# One parametrised GPU-box session (run through gpurun from the repo root):   tools/gpu_session.sh <stage> [args...]
#   pytest [-k expr]        the -m gpu parity suite                          -> gpurun_out/${TAG}_pytest.log
#   bench [bench.py args]   one bench line                                   -> gpurun_out/${TAG}_bench.json
#   stats [bench.py args]   rocprofv3 --kernel-trace --stats of bench.py     -> gpurun_out/${TAG}_kernel_stats.csv + _summary.md
#   pmc                     FETCH_SIZE / WRITE_SIZE / SQ counter passes of tools/gpu_probe.py full (separate passes, no traces)
#   probe <sections...>     tools/ph8_probe.py sections with the experiments build
#   ab                      same-box interleaved A/B of the DiT step against the round-5 library (tools/ab_r05.py; needs tools/ab/libsat_hip_r05.so)
#   power [bench.py args]   rocm-smi power / clock samples while bench.py runs (tools/power_probe.py)
#   both                    the -m gpu suite under BOTH operand formats (SAT_TEST_DTYPE=fp16 = the package default, then bf16), prints shown (-s)
# TAG (default r06) names the outputs.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
TAG=${TAG:-r06}
stage=$1; shift
case $stage in
  pytest)
    timeout 2400 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider "$@" > gpurun_out/${TAG}_pytest.log 2>&1
    echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -5 gpurun_out/${TAG}_pytest.log ;;
  both)
    for fmt in fp16 bf16; do
      SAT_TEST_DTYPE=$fmt timeout 2400 python -m pytest tests -m gpu -q -s -rf --no-header -p no:cacheprovider "$@" > gpurun_out/${TAG}_pytest_$fmt.log 2>&1
      echo "pytest[$fmt] rc=$?" >> gpurun_out/${TAG}_pytest_$fmt.log; tail -3 gpurun_out/${TAG}_pytest_$fmt.log
    done ;;
  bench)
    timeout 900 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench.json ;;
  stats)
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG
    timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_stats_bench.json 2> $R/gpurun_out/${TAG}_stats.err
    f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
    t=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1); python $R/tools/trace_blocks.py "$t" > $R/gpurun_out/${TAG}_block_trace.txt 2>&1
    python $R/tools/stats_summary.py $R/gpurun_out/${TAG}_kernel_stats.csv 2 "$TAG: bench.py $*" > $R/gpurun_out/${TAG}_kernel_stats_summary.md; head -16 $R/gpurun_out/${TAG}_kernel_stats_summary.md | cut -c1-180 ;;
  pmc)
    cd /tmp; export TMPDIR=/tmp
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
      tag=$(echo $pass | cut -d' ' -f1)
      timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$tag -- python $R/tools/gpu_probe.py full > $R/gpurun_out/${TAG}_pmc_$tag.log 2>&1
      f=$(find $R/gpurun_out/${TAG}_pmc_$tag -name "*counter_collection.csv" | head -1)
      python $R/tools/pmc_summarize.py $f > $R/gpurun_out/${TAG}_pmc_${tag}_per_kernel.csv
      rm -rf $R/gpurun_out/${TAG}_pmc_$tag
    done
    head -4 $R/gpurun_out/${TAG}_pmc_FETCH_SIZE_per_kernel.csv | cut -c1-200 ;;
  clock)
    # effective clock per kernel: GRBM_GUI_ACTIVE / wall (one counter, one pass, kernel trace only) at PROBE_B=$1 (default 1)
    cd /tmp; export TMPDIR=/tmp; B=${1:-1}; rm -rf /tmp/clk_$TAG
    PROBE_B=$B timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk_$TAG -- python $R/tools/gpu_probe.py full > $R/gpurun_out/${TAG}_clock_b$B.log 2>&1
    c=$(find /tmp/clk_$TAG -name "*counter_collection.csv" | head -1); t=$(find /tmp/clk_$TAG -name "*kernel_trace.csv" | head -1)
    head -2 $c > $R/gpurun_out/${TAG}_clock_b${B}_csv_head.txt
    python $R/tools/pmc_clock.py $c $t > $R/gpurun_out/${TAG}_clock_b$B.md 2>&1; head -14 $R/gpurun_out/${TAG}_clock_b$B.md | cut -c1-220 ;;
  ab)
    timeout 1200 python tools/ab_r05.py "$@" > gpurun_out/${TAG}_ab_r05.log 2>&1; tail -6 gpurun_out/${TAG}_ab_r05.log ;;
  ab4)
    timeout 900 python tools/ab_r04.py "$@" > gpurun_out/${TAG}_ab_r04.log 2>&1; tail -6 gpurun_out/${TAG}_ab_r04.log ;;
  power)
    timeout 900 python tools/power_probe.py "bench.py $*" -- python bench.py --no-cpu-baseline "$@" > gpurun_out/${TAG}_power.txt 2>&1; grep "Socket\|sclk clock speed" gpurun_out/${TAG}_power.txt ;;
  probe)
    SAT_HIP_EXP=1 timeout 900 python tools/ph8_probe.py "$@" > gpurun_out/${TAG}_probe.log 2>&1; tail -40 gpurun_out/${TAG}_probe.log ;;
  *) echo "unknown stage $stage"; exit 2 ;;
esac
