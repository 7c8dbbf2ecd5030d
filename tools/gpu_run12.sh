#!/bin/bash
# PMC passes (counters only, separate passes as MI355X_MICROARCH.md prescribes) over 12 CFG denoiser steps of the full model
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
cd /tmp; export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_$tag -- python $R/tools/gpu_probe.py full > $R/gpurun_out/r2_pmc_$tag.log 2>&1
  f=$(find $R/gpurun_out/r2_pmc_$tag -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summarize.py $f > $R/gpurun_out/r2_pmc_${tag}_per_kernel.csv
  rm -rf $R/gpurun_out/r2_pmc_$tag
done
head -8 $R/gpurun_out/r2_pmc_FETCH_SIZE_per_kernel.csv | cut -c1-200; head -8 $R/gpurun_out/r2_pmc_WRITE_SIZE_per_kernel.csv | cut -c1-200; head -12 $R/gpurun_out/r2_pmc_SQ_VALU_MFMA_BUSY_CYCLES_per_kernel.csv | cut -c1-260
