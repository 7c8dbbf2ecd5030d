"""rocprofv3 --stats kernel_stats.csv -> markdown table per generation.  usage: stats_summary.py <csv> <generations> "<title>" > out.md"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
gens = int(sys.argv[2])
total = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / gens
print(f"# {sys.argv[3]}\n")
print("| ms / generation | % | launches / generation | avg us | kernel |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    ms = float(r["TotalDurationNs"]) / 1e6 / gens
    if ms < 0.3:
        continue
    print(f"| {ms:.2f} | {100 * ms / total:.2f} | {int(r['Calls']) // gens} | {float(r['AverageNs']) / 1e3:.1f} | `{r['Name'][:150]}` |")
print(f"\nGPU kernel time per generation: {total:.1f} ms")
