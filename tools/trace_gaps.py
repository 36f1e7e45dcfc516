#!/usr/bin/env python3
"""Per-kernel time and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace csv.

    python tools/trace_gaps.py <..._kernel_trace.csv> [first_row [rows]]
"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
rows = rows[lo:lo + n]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-40:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None:
        gap[name].append(s - prev_end)
    prev_end = e
for k in dur:
    d, g = dur[k], gap[k] or [0]
    d2 = sorted(d)
    print(f"{k:42s} n={len(d):5d} dur avg {sum(d)/len(d)/1e3:7.2f} med {d2[len(d2)//2]/1e3:7.2f} us   gap-before avg {sum(g)/len(g)/1e3:6.2f} med {sorted(g)[len(g)//2]/1e3:6.2f} us")
print(f"span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp']))/1e6:.3f} ms over {len(rows)} kernels")
