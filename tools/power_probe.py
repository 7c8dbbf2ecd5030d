"""Is the generation power-limited?  Samples rocm-smi (socket power, sclk / mclk, temperature, the power cap) twice a second while a
command runs on the GPU box and prints the distribution.  Developer tool (round 5: the same-box A/B of profiles/r05_ab_vs_r04_same_box.txt
showed that removing 1-4 us of stalls per K-range moves the denoiser step by < 1 % -- the question is whether the chip sits at its power cap
with the clock pulled down, so that time = energy / cap).      usage: python tools/power_probe.py <label> -- <command ...>"""
import json
import statistics
import subprocess
import sys
import threading
import time

label = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]
samples, stop = [], threading.Event()


def poll():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "-P", "-c", "-t", "-M", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            samples.append((time.time(), card))
        except Exception as e:          # noqa: BLE001
            samples.append((time.time(), {"error": repr(e)}))
        stop.wait(0.5)


th = threading.Thread(target=poll, daemon=True)
th.start()
t0 = time.time()
rc = subprocess.run(cmd).returncode
t1 = time.time()
stop.set()
th.join(timeout=3)


def num(v):
    try:
        return float(str(v).strip("()").replace("Mhz", "").replace("MHz", "").replace("W", "").replace("C", ""))
    except ValueError:
        return None


keys = {}
for _, card in samples:
    for k, v in card.items():
        x = num(v)
        if x is not None:
            keys.setdefault(k, []).append(x)
print(f"[{label}] rc={rc}, {t1 - t0:.0f} s, {len(samples)} rocm-smi samples")
if samples:
    print("  first sample keys:", {k: v for k, v in list(samples[len(samples) // 2][1].items())[:16]})
for k, v in sorted(keys.items()):
    if len(v) >= 3:
        v2 = sorted(v)
        print(f"  {k:60s} median {statistics.median(v):9.1f}  p10 {v2[len(v2) // 10]:9.1f}  p90 {v2[(9 * len(v2)) // 10]:9.1f}  max {v2[-1]:9.1f}")
