#!/bin/bash
# Same-box, in-model, per-launch A/B of one per-plan switch (round 6): rocprofv3 --kernel-trace of `tools/gpu_probe.py full` (CFG denoiser steps of the
# full-size model) under each value of an environment switch, alternating, back to back on one box -> per-launch block traces (tools/trace_blocks.py).
# usage (on the GPU box, from the repo root): tools/ab_trace6.sh <PROBE_B> <ENVVAR> <value> <value> ...     e.g.  tools/ab_trace6.sh 1 SAT_PREFETCH 1 0
# outputs: gpurun_out/${TAG}_trace_<ENVVAR>_<value>_b<B>.txt
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
TAG=${TAG:-r06}
B=$1; VAR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for val in "$@"; do
    rm -rf /tmp/abt6
    env $VAR=$val PROBE_B=$B timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/abt6 -- python $R/tools/gpu_probe.py full > $R/gpurun_out/${TAG}_trace_${VAR}_${val}_b${B}.log 2>&1
    t=$(find /tmp/abt6 -name '*kernel_trace.csv' | head -1)
    python $R/tools/trace_blocks.py "$t" >> $R/gpurun_out/${TAG}_trace_${VAR}_${val}_b${B}.txt 2>&1
  done
done
for val in "$@"; do echo "== $VAR=$val"; grep "block total\|avg" $R/gpurun_out/${TAG}_trace_${VAR}_${val}_b${B}.txt | cut -c1-120; done
