#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files (two separate
passes of `python bench.py`, as MI355X_MICROARCH.md section HBM prescribes) into per-launch HBM
traffic per kernel family:

    bytes = 2 * FETCH_SIZE[KiB] * 1024   (gfx950: FETCH_SIZE counts half of a streaming read --
                                          confirmed here with a 1 GiB copy, profiles/*/calib_*)
          + WRITE_SIZE[KiB] * 1024       (exact on the same calibration)

usage: pmc_summarize.py FETCH.csv WRITE.csv --head-log 9 --tail-log 13 -o profiles/rNN_traffic.json
"""
import argparse
import collections
import csv
import json
import re


def family(name: str, head_log: int, tail_log: int):
    m = re.search(r"k_fir(?:_lds|_row)?<(?:\d+, )?(\d)>", name)
    if m:
        return "fir_head" if m.group(1) == "0" else "fir_tail"   # <1> tail stage, <2> whole-IR line (timed as fir_tail)
    m = re.search(r"k_fft8?_(fwd|inv)<(\d+), float>", name)
    if m:
        lg = int(m.group(2))
        st = "head" if lg == head_log else ("tail" if lg == tail_log else None)
        return f"fft_{m.group(1)}_{st}" if st else None
    if "k_ingest" in name:
        return "ingest"
    return None


def per_family(path, head_log, tail_log):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        f = family(r["Kernel_Name"], head_log, tail_log)
        if f:
            acc[f].append(float(r["Counter_Value"]))
    # drop the first (warm-up / cold cache) launch of each family when there are several
    return {k: sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0] for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--head-log", type=int, default=9)
    ap.add_argument("--tail-log", type=int, default=13)
    ap.add_argument("--command", default="python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --stream-calls 0")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    fe = per_family(a.fetch_csv, a.head_log, a.tail_log)
    wr = per_family(a.write_csv, a.head_log, a.tail_log)
    out = {"command": a.command, "unit": "bytes per launch",
           "correction": "2*FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 FETCH_SIZE = 1/2 of streamed bytes; calibrated)",
           "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        f = 2.0 * fe.get(k, 0.0) * 1024.0
        w = wr.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "traffic_bytes": f + w}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
