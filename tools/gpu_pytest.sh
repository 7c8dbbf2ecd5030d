#!/bin/bash
# GPU-box session: the -m gpu parity suite (optionally a -k filter as $1) + one headline bench line.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
TAG=${TAG:-x}
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 1800 python -m pytest tests -m gpu -q -rfP --no-header -p no:cacheprovider "${K[@]}" > gpurun_out/r2_pytest_$TAG.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest_$TAG.log
if [ -z "$NOBENCH" ]; then timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_$TAG.json 2> gpurun_out/r2_bench_$TAG.err; fi
grep -E "passed|failed|rc=" gpurun_out/r2_pytest_$TAG.log | tail -3; grep -E "^\[|Error" gpurun_out/r2_pytest_$TAG.log | tail -40; cat gpurun_out/r2_bench_$TAG.json 2>/dev/null | cut -c1-400
