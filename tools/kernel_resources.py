"""Developer helper: compile one .hip with -Rpass-analysis=kernel-resource-usage and print a table
(kernel, VGPRs, AGPRs, scratch bytes, occupancy, LDS).  Usage: python tools/kernel_resources.py csrc/gemm_bf16.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/ru/out.o",
       "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[3:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
if len(sys.argv) > 2 and sys.argv[2] == "--raw":
    print(out)
    sys.exit(0)
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        name = re.sub(r"\(anonymous namespace\)::", "", k)[:110]
        print(f"{name:110s} vgpr {v.get('VGPRs', -1):4d} agpr {v.get('AGPRs', -1):4d} scratch {v.get('ScratchSize [bytes/lane]', -1):5d} "
              f"occ {v.get('Occupancy [waves/SIMD]', -1):2d} lds {v.get('LDS Size [bytes/block]', -1)}")
