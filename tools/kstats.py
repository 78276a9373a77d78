#!/usr/bin/env python3
"""Print name / calls / avg / min / max (us) of the kernels matching a substring from rocprofv3 *kernel_stats.csv files."""
import csv, glob, sys
pat = sys.argv[1]
for path in sys.argv[2:]:
    for f in glob.glob(path, recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Name"]:
                print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} min {int(r["MinNs"])/1e3:9.1f} max {int(r["MaxNs"])/1e3:9.1f} us')
