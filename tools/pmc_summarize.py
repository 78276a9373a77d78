#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files (two separate
passes of `python bench.py`, as MI355X_MICROARCH.md section HBM prescribes) into per-launch HBM
traffic per kernel family:

    bytes = f_fetch * FETCH_SIZE[KiB] * 1024 + f_write * WRITE_SIZE[KiB] * 1024

The correction factors come from a calibration pass over a 1 GiB device copy (tools/pmc_calib.py,
--calib-fetch / --calib-write): on gfx950 FETCH_SIZE counts half of a streamed read (f_fetch = 2),
WRITE_SIZE is exact (f_write = 1). Without calibration files those two values are assumed.

usage: pmc_summarize.py FETCH.csv WRITE.csv [--calib-fetch C.csv --calib-write C.csv] -o out.json
"""
import argparse
import collections
import csv
import json
import re

CALIB_BYTES = 4 * (1 << 28)      # tools/pmc_calib.py copies 1 GiB per dispatch: this many bytes read AND as many written




PATCH0_FAMILY = "premultiply"     # k_fdl_patch<0, ..>: the zero-latency stage's patch; "fir_head" for sets whose per-block call takes
                                  # the general path (many channels with a large head block: BASELINE config 5's geometry)


def family(name: str, head_log: int, tail_log: int):
    if "k_fused_block" in name or "k_block_step" in name:
        return "fused_block"
    m = re.search(r"k_fdl_sweep<(\d+), (\d+), (\d), \d+, \d+, \d+, (true|false)(?:, (?:true|false))?>", name)   # <K, SPLIT, STAGE, LW, D, LB, NT[, NTH]>
    if m:
        st = "head" if m.group(3) == "0" else "tail"
        # second-level sweeps are exactly the own-tile K = 8 instantiation with ordinary loads (rvc_sweep.hip launch_stage)
        second = m.group(1) == "8" and m.group(2) == "1" and m.group(4) == "false"
        if m.group(1) == "4":                # round 6: third-level sweeps (four blocks half way through a group of 8)
            return "sweep3_" + st
        return ("sweep2_" if second else "sweep_") + st
    m = re.search(r"k_fdl_sweep_lds<\d+, \d+, \d+, (\d), (?:true|false), \d+(?:, (?:true|false))?>", name)   # <KW, NKW, A, STAGE, NT, LB, M3>: first level only
    if m:
        return "sweep_head" if m.group(1) == "0" else "sweep_tail"
    m = re.search(r"k_fft8_(fwd|inv)_loop<(\d+)>", name)
    if m:
        return f"fft_{m.group(1)}_tail"
    if "k_fft8_inv_dif2<" in name:           # the 8192-bin double / 16384-bin float inverse as two half-size sub-transforms (lock-step sets' tail)
        return "fft_inv_tail"
    if "k_fft8_fwd_dif2<" in name:           # the 16384-bin float forward the same way (round 6)
        return "fft_fwd_tail"
    m = re.search(r"k_fir_row<(\d)>", name)
    if m:
        return "premultiply" if m.group(1) == "0" else "fir_tail"
    m = re.search(r"k_fdl_patch(?:_groups)?<(\d)", name)     # (_groups, round 6: the patches of all phase groups of a tail stage in one launch)
    if m:
        return PATCH0_FAMILY if m.group(1) == "0" else "fir_tail"
    m = re.search(r"k_fir(?:_lds)?<(?:\d+, )?(\d)>", name)
    if m:
        return "fir_head" if m.group(1) == "0" else "fir_tail"   # <1> tail stage, <2> whole-IR line (timed as fir_tail)
    # (double: only the inverse -- the lock-step sets' tail inverse since round 5; the double FORWARD launches are the IR spectra at init)
    m = re.search(r"k_fft8?_(fwd|inv)<(\d+), float(?:, (?:true|false))*>", name) or re.search(r"k_fft8?_(inv)<(\d+), double(?:, (?:true|false))*>", name)
    if m:
        lg = int(m.group(2))
        st = "head" if lg == head_log else ("tail" if lg >= tail_log else None)
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
    # steady state: the second half of the dispatches of each family (the first ones run on a delay line
    # that is still filling: rows before time 0 are not fetched)
    return {k: (sum(v[len(v) // 2:]) / len(v[len(v) // 2:]), len(v)) for k, v in acc.items()}


def calib_factor(path, default):
    if not path:
        return default, None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "copyBuffer" in r["Kernel_Name"]]
    vals = [v for v in vals if v > 1024.0]
    if not vals:
        return default, None
    kib = sum(vals) / len(vals)
    return CALIB_BYTES / (kib * 1024.0), kib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--calib-fetch")
    ap.add_argument("--calib-write")
    ap.add_argument("--head-log", type=int, default=9)
    ap.add_argument("--tail-log", type=int, default=13)
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--time-tiling", type=int, default=1)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--k1-head", type=int, default=16, help="(recorded only)")
    ap.add_argument("--k1-tail", type=int, default=16, help="(recorded only)")
    ap.add_argument("--subsets", type=int, default=1, help="child sets (bench.py --child-sets 1): every launch covers channels / subsets")
    ap.add_argument("--patch0-family", default="premultiply", help="family of k_fdl_patch<0,..>: premultiply, or fir_head (general per-block path)")
    ap.add_argument("--key", default="", help="key of the entry in the output (default config<C>)")
    ap.add_argument("--command", default="python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --side 0")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    global PATCH0_FAMILY
    PATCH0_FAMILY = a.patch0_family
    fe = per_family(a.fetch_csv, a.head_log, a.tail_log)
    wr = per_family(a.write_csv, a.head_log, a.tail_log)
    ff, fk = calib_factor(a.calib_fetch, 2.0)
    fw, wk = calib_factor(a.calib_write, 1.0)
    out = {"command": a.command, "channels": a.channels, "channels_per_launch": a.channels // max(1, a.subsets),
           "config": a.config, "time_tiling": a.time_tiling,
           "unit": "bytes per launch (mean over the second half of each family's dispatches)",
           "correction": {"fetch_factor": round(ff, 4), "write_factor": round(fw, 4),
                          "calibration": "1 GiB device copy (tools/pmc_calib.py): FETCH_SIZE %s KiB, WRITE_SIZE %s KiB per dispatch "
                                         "for 1048576 KiB read + 1048576 KiB written" % (fk, wk)},
           "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        f = ff * fe.get(k, (0.0, 0))[0] * 1024.0
        w = fw * wr.get(k, (0.0, 0))[0] * 1024.0
        out["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "traffic_bytes": f + w,
                             "dispatches_seen": max(fe.get(k, (0, 0))[1], wr.get(k, (0, 0))[1])}
    json.dump({a.key or "config%d" % a.config: out}, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
