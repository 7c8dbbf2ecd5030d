#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -8 | tee gpurun_out/r2_pytest_5.log
timeout 300 python tools/gpu_probe.py f32epi 2>&1 | grep -v amdgpu | tee gpurun_out/r2_f32epi.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r2_bench_5.json
