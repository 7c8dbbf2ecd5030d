#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
for mode in fused unfused fused unfused; do
  if [ $mode = unfused ]; then export SAT_HIP_EXP=1 SAT_OOBLECK_UNFUSED=1; else unset SAT_OOBLECK_UNFUSED SAT_HIP_EXP; fi
  echo "== $mode"; timeout 300 python tools/codec_only.py 2>&1 | tail -2
done
cd /tmp && export TMPDIR=/tmp
for mode in fused unfused; do
  if [ $mode = unfused ]; then export SAT_HIP_EXP=1 SAT_OOBLECK_UNFUSED=1; else unset SAT_OOBLECK_UNFUSED SAT_HIP_EXP; fi
  rm -rf /tmp/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python $R/tools/codec_only.py > /dev/null 2> $R/gpurun_out/r2_prof21_$mode.err
  f=$(find /tmp/prof_$mode -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/r2_codec_kernel_stats_$mode.csv
  echo "== $mode"; head -9 "$f" | cut -c1-150
done
