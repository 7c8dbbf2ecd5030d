#!/bin/bash
# Same-box, in-model, per-kernel A/B: rocprofv3 --kernel-trace of `tools/gpu_probe.py full` (12 CFG denoiser steps of the full-size model + decodes)
# once with the round-4 library and once with this tree's, back to back on one box -> per-launch block traces of both (tools/trace_blocks.py).
# usage (on the GPU box, from the repo root): tools/ab_trace.sh [PROBE_B]      outputs: gpurun_out/r05_ab_trace_{r04,r05}[_bN].txt
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
B=${1:-1}
sfx=""; [ "$B" != "1" ] && sfx="_b$B"
cd /tmp && export TMPDIR=/tmp
for which in r04 r05 r04 r05; do
  rm -rf /tmp/abt_$which
  if [ $which = r04 ]; then export SAT_HIP_LIB=tools/ab/libsat_hip_r04.so SAT_HIP_ABI=4; else unset SAT_HIP_LIB SAT_HIP_ABI; fi
  PROBE_B=$B timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/abt_$which -- python $R/tools/gpu_probe.py full > $R/gpurun_out/r05_ab_trace_${which}${sfx}.log 2>&1
  t=$(find /tmp/abt_$which -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_blocks.py "$t" >> $R/gpurun_out/r05_ab_trace_${which}${sfx}.txt 2>&1
done
for which in r04 r05; do echo "== $which"; grep "block total\|avg" $R/gpurun_out/r05_ab_trace_${which}${sfx}.txt | cut -c1-110; done
