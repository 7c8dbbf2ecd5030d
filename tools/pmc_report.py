"""Per-kernel PMC CSVs (tools/pmc_summarize.py output of the three passes of `tools/gpu_session.sh pmc`) -> the committed artefacts:
   profiles/<tag>_pmc_fetch_write_per_kernel.csv, profiles/<tag>_pmc_sq_summary.md, profiles/<tag>_ffn_traffic.json.
usage: python tools/pmc_report.py <tag> <dir with <tag>_pmc_*_per_kernel.csv>"""
import csv
import json
import os
import sys

tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return {r["kernel"]: r for r in csv.DictReader(open(os.path.join(src, f"{tag}_pmc_{name}_per_kernel.csv")))}


fetch, write, sq = load("FETCH_SIZE"), load("WRITE_SIZE"), load("SQ_VALU_MFMA_BUSY_CYCLES")
with open(os.path.join(root, "profiles", f"{tag}_pmc_fetch_write_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KB_raw_avg", "WRITE_SIZE_KB_raw_avg"])
    for k, r in sorted(fetch.items(), key=lambda kv: -float(kv[1]["FETCH_SIZE"]) * int(kv[1]["launches"])):
        w.writerow([k, r["launches"], r["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", "")])

lines = [f"# {tag} -- SQ / GRBM counters per kernel: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -- python tools/gpu_probe.py full` (12 CFG denoiser steps + "
         "4 decodes, 1 prompt, the package default operand format (fp16 since round 4), LayerNorm fold on; tools/gpu_session.sh pmc)", "",
         "Derived per launch: busy cycles per shader engine = SQ_BUSY_CYCLES / 32; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x busy "
         "cycles per SE); wave-time split = SQ_WAIT_ANY (parked at s_waitcnt / s_barrier), SQ_WAIT_INST_ANY (ready, pipe busy / dependency), "
         "SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.", "",
         "| kernel | launches | busy cycles / SE | MfmaUtil % | wait (waitcnt/barrier) % | wait (issue) % | active % |", "|---|---|---|---|---|---|---|"]
for k, r in sorted(sq.items(), key=lambda kv: -float(kv[1]["SQ_BUSY_CYCLES"]) * int(kv[1]["launches"])):
    busy = float(r["SQ_BUSY_CYCLES"]) / 32.0
    wc = float(r["SQ_WAVE_CYCLES"]) or 1.0
    if busy <= 0:
        continue
    lines.append(f"| `{k[:110]}` | {r['launches']} | {busy:.0f} | {100 * float(r['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024 * busy):.1f} | "
                 f"{100 * float(r['SQ_WAIT_ANY']) / wc:.1f} | {100 * float(r['SQ_WAIT_INST_ANY']) / wc:.1f} | {100 * float(r['SQ_ACTIVE_INST_ANY']) / wc:.1f} |")
open(os.path.join(root, "profiles", f"{tag}_pmc_sq_summary.md"), "w").write("\n".join(lines) + "\n")

ffn = [k for k in fetch if "gemm_ph8_kernel<2, 0" in k] or [k for k in fetch if "gemm_pipe_kernel<256, 256, 64, 4, 4, 2, 2" in k]
if ffn:
    k = ffn[0]
    fk, wk = float(fetch[k]["FETCH_SIZE"]), float(write[k]["WRITE_SIZE"])
    json.dump({"kernel": k, "launches_sampled": int(fetch[k]["launches"]), "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB_raw": wk,
               "hbm_bytes_per_launch": (2 * fk + wk) * 1024, "algorithmic_bytes_per_launch": 2 * (2050 * 1536 + 12288 * 1536 + 2050 * 6144),
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (counters only, with --kernel-trace) over `tools/gpu_probe.py "
                       "full` (12 CFG denoise steps, 1 prompt, M=2050 N=12288 K=1536, LayerNorm fold on: A = the 16-bit image of the residual stream); "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM (gfx950 reports half of a wide coalesced read stream); WRITE_SIZE as "
                       "reported (= the 25.2 MB 16-bit output).  Fabric-side bytes: the excess over the algorithmic 69.2 MB is what 8 private L2s cost -- "
                       "round 4 runs this launch as two balanced rounds of 216 workgroups (27 per XCD: 8 row tiles x 3.4 column tiles): every XCD "
                       "pulls the 8 A panels (6.3 MB) + its W panels through its own L2 in both rounds; the 6.3 MB A operand does not fit a 4 MiB "
                       "L2 next to the W stream, so it is re-fetched (from the 256 MiB Infinity Cache).  At ~2.5 TB/s over the launch this is not "
                       "the bound (MFMA / LDS issue is, DESIGN.md section 4)."},
              open(os.path.join(root, "profiles", f"{tag}_ffn_traffic.json"), "w"), indent=1)
    print("FF-in traffic per launch: %.1f MB" % ((2 * fk + wk) * 1024 / 1e6))
