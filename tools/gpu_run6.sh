#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -15 | tee gpurun_out/r2_pytest_6.log
timeout 300 python tools/gpu_probe.py attn 2>&1 | grep -v amdgpu | tee gpurun_out/r2_attn6.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r2_bench_6.json
