"""rocprofv3 --pmc counter_collection.csv -> per-kernel averages (one row per kernel name).  usage: pmc_summarize.py <csv> > out.csv"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches"] + names)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    n = max(cnt[k].values())
    w.writerow([k[:160], n] + [f"{v[c] / max(cnt[k][c], 1):.1f}" if c in v else "" for c in names])
