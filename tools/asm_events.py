"""Order of spills / MFMAs / barriers / branches in the gfx950 assembly of selected kernels: tells a hot-loop spill from an epilogue one.
usage: python tools/asm_events.py file.s substring [substring...]   (file.s from `hipcc -S --cuda-device-only`)"""
import re
import sys

text = open(sys.argv[1]).read()
for chunk in re.split(r"\n(?=_Z\S*: )", text):
    m = re.match(r"(_Z\S*):", chunk)
    if not m or not any(k in m.group(1) for k in sys.argv[2:]):
        continue
    out, last, cnt = [], None, 0
    for line in chunk.split("\n"):
        t = line.strip().split()
        if not t or not re.match(r"scratch_|v_mfma|s_barrier|s_cbranch|s_endpgm", t[0]):
            continue
        op = t[0][:14]
        if op == last:
            cnt += 1
        else:
            if last:
                out.append(f"{last}x{cnt}")
            last, cnt = op, 1
    out.append(f"{last}x{cnt}")
    print(m.group(1)[:110])
    print("  " + " ".join(out))
