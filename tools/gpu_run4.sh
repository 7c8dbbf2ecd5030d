#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
PROBE_QUANT=1 timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep -v amdgpu | tee gpurun_out/r2_quant.log
