"""Effective shader clock per kernel (MI355X_MICROARCH.md "DVFS give-back"): GRBM_GUI_ACTIVE / wall time of the dispatch, from
ONE rocprofv3 pass `--pmc GRBM_GUI_ACTIVE --kernel-trace` (the counter CSV carries each dispatch's start / end timestamps).  Per kernel name: launches,
average wall time, average effective clock, and the MFMA roofline fraction re-priced at that clock for the kernels whose FLOPs are given.
usage: python tools/pmc_clock.py <counter_collection.csv> [<kernel_trace.csv>]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
cols = rows[0].keys() if rows else []
ts = {}
if "Start_Timestamp" not in cols and len(sys.argv) > 2:          # older layout: timestamps only in the kernel trace, joined on the dispatch id
    for r in csv.DictReader(open(sys.argv[2])):
        ts[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])          # launches, sum ns, sum counter, max counter share
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    if "Start_Timestamp" in cols:
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    else:
        t0, t1 = ts.get(r.get("Dispatch_Id") or r.get("Correlation_Id"), (0, 0))
    if t1 <= t0:
        continue
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += t1 - t0
    a[2] += float(r["Counter_Value"])
print("| kernel | launches | wall us | GRBM_GUI_ACTIVE per launch | cycles / ns (raw) | effective clock GHz (raw / 8 XCDs) |")
print("|---|---|---|---|---|---|")
for k, (n, ns, c, _) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns / n < 3000:
        continue
    name = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1", "", k)[:90]
    print(f"| `{name}` | {n} | {ns / n / 1e3:.1f} | {c / n:.0f} | {c / ns:.3f} | {c / ns / 8:.3f} |")
