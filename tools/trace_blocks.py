"""Per-position kernel durations of one transformer block from a rocprofv3 --kernel-trace CSV (developer tool): the launches between two
consecutive FF-in SwiGLU GEMMs are one block; positions are averaged over all blocks with the same launch count.
usage: python tools/trace_blocks.py <kernel_trace.csv> [anchor-substring]"""
import csv
import re
import statistics
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "gemm_ph8_kernel<2"
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(idx) < 3:
    anchor = "gemm_pipe_kernel<128, 128, 64, 4, 2, 4"
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
blocks = {}
for a, b in zip(idx[:-1], idx[1:]):
    blocks.setdefault(b - a, []).append(rows[a:b])
n, blks = max(blocks.items(), key=lambda kv: len(kv[1]))
print(f"{len(blks)} blocks of {n} launches (anchor {anchor!r})")
tot = 0.0
for p in range(n):
    d = [(blk[p][1] - blk[p][0]) / 1e3 for blk in blks]
    gap = [(blk[p][0] - blk[p - 1][1]) / 1e3 for blk in blks] if p else [0.0]
    name = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1", "", blks[0][p][2])[:70]
    tot += statistics.mean(d)
    print(f"  {p:2d} avg {statistics.mean(d):8.1f} us (min {min(d):7.1f} max {max(d):7.1f}) gap {statistics.mean(gap):5.1f}  grid {blks[0][p][3]:>8} wg {blks[0][p][4]:>4}  {name}")
print(f"  block total {tot:.1f} us")
