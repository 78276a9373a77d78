#!/bin/bash
# rocprofv3 evidence for the default bench.py workload (run on the GPU box, from the repo root):
#   1. --kernel-trace --stats      -> <out>/kernel_stats.csv    (per-kernel average duration)
#   2. --pmc FETCH_SIZE            -> <out>/FETCH_SIZE.csv      (separate pass, MI355X_MICROARCH.md HBM section)
#   3. --pmc WRITE_SIZE            -> <out>/WRITE_SIZE.csv
#   4. the same two counters over a 1 GiB copy (tools/pmc_calib.py) -> <out>/calib_*.csv
# then tools/pmc_summarize.py turns 2-4 into <out>/traffic.json.   usage: tools/profile_bench.sh <out dir> [bench args]
set -u
OUT=$(realpath -m "${1:-gpurun_out/prof}"); shift || true
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --side 0 --distinct 64 $*"
cd /tmp
# (kernel trace: the driver's step counts, so that the timed steps outweigh the untimed pre-roll -- a delay line that is still
#  filling moves fewer bytes per launch -- and the one event-instrumented step of bench.py's own per-kernel measurement)
KT_ARGS="--steps 20 --warmup 5 --cpu-seconds 0 --side 0 --distinct 64 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python "$ROOT/bench.py" $KT_ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
find "$OUT/kt" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
# launches of the two child sets overlap: per family, the UNION of the dispatch intervals of the trace (tools/trace_union.py)
find "$OUT/kt" -name '*kernel_trace.csv' -exec python "$ROOT/tools/trace_union.py" {} \; > "$OUT/kernel_union.txt" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_$c.err"
  find "$OUT/pmc_$c" -name '*counter_collection.csv' -exec cp {} "$OUT/$c.csv" \;
  rocprofv3 --pmc $c --output-format csv -d "$OUT/calib_$c" -o pmc -- python "$ROOT/tools/pmc_calib.py" > /dev/null 2> "$OUT/calib_$c.err"
  find "$OUT/calib_$c" -name '*counter_collection.csv' -exec cp {} "$OUT/calib_$c.csv" \;
done
cd "$ROOT"
# PMC_ARGS: geometry of the profiled configuration for tools/pmc_summarize.py (default: config 2, two child sets), e.g.
#   PMC_ARGS="--config 3 --head-log 8 --k1-head 16 --k1-tail 32 --channels 2048" tools/profile_bench.sh out --config 3
python tools/pmc_summarize.py "$OUT/FETCH_SIZE.csv" "$OUT/WRITE_SIZE.csv" --calib-fetch "$OUT/calib_FETCH_SIZE.csv" \
  --calib-write "$OUT/calib_WRITE_SIZE.csv" --command "python bench.py $ARGS" ${PMC_ARGS:-} -o "$OUT/traffic.json" > "$OUT/traffic.txt" 2>&1
# the raw per-dispatch traces are large: keep the summaries only
rm -rf "$OUT/kt" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/calib_FETCH_SIZE "$OUT"/calib_WRITE_SIZE
python - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
# per-kernel mean of the per-dispatch counters (compact form of the two big CSVs), then drop the raw files
for c in ("FETCH_SIZE", "WRITE_SIZE", "calib_FETCH_SIZE", "calib_WRITE_SIZE"):
    p = os.path.join(out, c + ".csv")
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(out, c + "_per_kernel.csv"), "w") as f:
        f.write("Kernel_Name,Dispatches,Mean_KiB,Min_KiB,Max_KiB\n")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.3f,%.3f,%.3f\n' % (k, len(v), sum(v) / len(v), min(v), max(v)))
    os.remove(p)
PY
ls -la "$OUT"
