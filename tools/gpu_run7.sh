#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
SAT_HIP_EXP=1 timeout 600 python tools/gpu_probe.py attn_ablate 2>&1 | grep -v amdgpu | tee gpurun_out/r2_attn_ablate.log
# SQ counters of the attention kernel alone
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_attn -- python $R/tools/gpu_probe.py attn > $R/gpurun_out/r2_pmc_attn.log 2>&1)
find gpurun_out/r2_pmc_attn -name "*counter_collection.csv" -exec cp {} gpurun_out/r2_pmc_attn_counters.csv \;
rm -rf gpurun_out/r2_pmc_attn
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r2_pmc_attn_counters.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = (r["Kernel_Name"][:40], r.get("Grid_Size", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in agg.items():
    n = max(cnt[k], 1)
    print(k, n, {c: round(x / n) for c, x in v.items()})
PY
