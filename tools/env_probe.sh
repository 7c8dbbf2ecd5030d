#!/bin/bash
# Round 6 probe: ROCclr environment knobs of the launch path against the DiT step (tools/gpu_probe.py full, one prompt), sequential runs on one box,
# each under its own timeout (one of these settings hangs the process).  Output: gpurun_out/r06_env_probe.txt
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
mkdir -p gpurun_out; out=gpurun_out/r06_env_probe.txt; : > $out
run() { r=$(env $1 timeout 100 python tools/gpu_probe.py full 2>/dev/null | grep 'DiT CFG step' | cut -c1-40); echo "$1: ${r:-TIMEOUT / failed}" | tee -a $out; }
run X=0
for kv in AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 ROC_SYSTEM_SCOPE_SIGNAL=0 GPU_FLUSH_ON_EXECUTION=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 GPU_MAX_HW_QUEUES=1 ROC_SKIP_KERNEL_ARG_COPY=1 AMD_DIRECT_DISPATCH=0; do run $kv; done
run X=0
