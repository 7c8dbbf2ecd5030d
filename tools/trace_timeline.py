"""One transformer block of a rocprofv3 --kernel-trace CSV as a timeline (developer tool): every kernel that STARTS between two consecutive FF-in SwiGLU
launches in the middle of the trace, with start / end relative to the first, queue id and grid -- shows what runs beside what (side-stream launches).
usage: python tools/trace_timeline.py <kernel_trace.csv> [block index from the middle, default 0] [anchor substring]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")),
                     r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
rows.sort()
anchor = sys.argv[3] if len(sys.argv) > 3 else "gemm_ph8_kernel<2"
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
k = len(idx) // 2 + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
a, b = idx[k], idx[k + 1]
t0 = rows[a][0]
print(f"block {k} of {len(idx)}: {(rows[b][0] - t0) / 1e3:.1f} us from FF-in to FF-in")
for r in rows[a:b]:
    name = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1", "", r[2])[:60]
    print(f"  q{r[3]:>3} {(r[0] - t0) / 1e3:8.1f} -> {(r[1] - t0) / 1e3:8.1f}  ({(r[1] - r[0]) / 1e3:6.1f} us)  grid {r[4]:>8} wg {r[5]:>4}  {name}")
