#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps from a rocprofv3 kernel_trace.csv (steady state:
the last N dispatches of the rvc kernels)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rvc::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -70:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void rvc::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append((e - s) / 1e3)
    if prev is not None:
        gap[name].append((s - prev) / 1e3)
    prev = e
tot = 0
for k in dur:
    d = sum(dur[k]) / len(dur[k]); g = sum(gap[k]) / max(len(gap[k]), 1)
    tot += d + g
    print(f"{k:34s} n={len(dur[k]):3d} dur {d:7.2f} us   gap-before {g:6.2f} us")
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"span {span:.1f} us over {len(rows)} dispatches; sum(dur+gap) per step ~ {tot:.1f} us")
