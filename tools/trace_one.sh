#!/bin/bash
# rocprofv3 --kernel-trace of tools/gpu_probe.py full under one environment setting -> block trace + one block's timeline
# usage (GPU box, repo root): tools/trace_one.sh <PROBE_B> <ENVVAR>=<value> <tag>
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
B=$1; SET=$2; TAGX=$3
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr1
env $SET PROBE_B=$B timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python $R/tools/gpu_probe.py full > $R/gpurun_out/${TAGX}.log 2>&1
t=$(find /tmp/tr1 -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_timeline.py "$t" > $R/gpurun_out/${TAGX}_timeline.txt 2>&1
python $R/tools/trace_blocks.py "$t" > $R/gpurun_out/${TAGX}_blocks.txt 2>&1
grep "DiT CFG step" $R/gpurun_out/${TAGX}.log; cat $R/gpurun_out/${TAGX}_timeline.txt
