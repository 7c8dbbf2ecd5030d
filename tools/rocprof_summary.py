"""rocprofv3 --kernel-trace --stats (csv) -> markdown table, normalised per generation.
usage: python tools/rocprof_summary.py <kernel_stats.csv> <generations> "<command that was profiled>" > profiles/rNN_kernel_stats_summary.md"""
import csv
import sys

path, gens, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats of `{cmd}` ({gens} generations)\n")
print("| ms / generation | % | launches / generation | avg us | kernel |")
print("|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t = float(r["TotalDurationNs"])
    if t / tot < 5e-4:
        continue
    name = r["Name"].replace("|", "/")
    print(f"| {t / gens / 1e6:.2f} | {100 * t / tot:.2f} | {int(r['Calls']) // gens} | {float(r['AverageNs']) / 1e3:.1f} | `{name[:150]}` |")
print(f"\nGPU kernel time per generation: {tot / gens / 1e6:.1f} ms")
