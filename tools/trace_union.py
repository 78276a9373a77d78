#!/usr/bin/env python3
"""Per kernel: dispatches, mean duration, and the UNION of the dispatch intervals of a rocprofv3 --kernel-trace CSV.
The engine serves a many-channel set by two child sets on their own streams: launches of one kernel family overlap in
time, so the time the family keeps the device busy is the union of its intervals, not dispatches x mean duration.
   python tools/trace_union.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

iv = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    iv[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("%-90s %9s %12s %12s %12s %6s" % ("kernel", "dispatch", "mean_us", "sum_ms", "union_ms", "conc"))
for k, v in sorted(iv.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
    v.sort()
    tot = sum(b - a for a, b in v)
    busy, end = 0, -1
    for a, b in v:
        if b > end:
            busy += b - max(a, end)
            end = b
    print("%-90s %9d %12.2f %12.3f %12.3f %6.2f" % (k[:90], len(v), tot / len(v) / 1e3, tot / 1e6, busy / 1e6, tot / max(busy, 1)))
