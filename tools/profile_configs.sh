#!/bin/bash
# Rounds 4-6: the rocprofv3 evidence behind bench.py's `roofline*.traffic` and the per-kernel durations, for BASELINE configs
# 2 (headline), 1, 3 and 5 in the lock-step regime AT THE BENCH'S OWN LAUNCH SIZES (the engine's default: child sets).
#   per config:  --kernel-trace --stats   -> <out>/config<C>/kernel_stats.csv, kernel_union.txt (union of dispatch intervals)
#                --pmc FETCH_SIZE         -> FETCH_SIZE_per_kernel.csv   (own pass, MI355X_MICROARCH.md HBM section)
#                --pmc WRITE_SIZE         -> WRITE_SIZE_per_kernel.csv   (own pass)
#   once:        the same two counters over a 1 GiB copy (tools/pmc_calib.py) -> calib_*_per_kernel.csv
# then tools/pmc_summarize.py -> <out>/config<C>/traffic.json, merged into <out>/traffic.json (= profiles/r6_traffic.json).
#   usage (GPU box, repo root): tools/profile_configs.sh <out dir> [configs, default "2 1 3 5"]
set -u
OUT=$(realpath -m "${1:-gpurun_out/prof}")
CONFIGS="${2:-2 1 3 5}"
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
compact() {   # per-kernel mean of a counter_collection.csv -> small CSV
python - "$1" "$2" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("Kernel_Name,Dispatches,Mean_KiB,Min_KiB,Max_KiB\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%.3f,%.3f,%.3f\n' % (k, len(v), sum(v) / len(v), min(v), max(v)))
PY
}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/calib_$c" -o pmc -- python "$ROOT/tools/pmc_calib.py" > /dev/null 2> "$OUT/calib_$c.err"
  find "$OUT/calib_$c" -name '*counter_collection.csv' -exec cp {} "$OUT/calib_$c.csv" \;
  compact "$OUT/calib_$c.csv" "$OUT/calib_${c}_per_kernel.csv"
  rm -rf "$OUT/calib_$c"
done
for C in $CONFIGS; do
  D="$OUT/config$C"; mkdir -p "$D"
  case $C in
    1) GEO="--channels 8192 --subsets 4 --head-log 9 --tail-log 13 --k1-head 32 --k1-tail 0" ;;
    # (configs 2 / 5: the tail stage one block late over IR[T,..), the zero-latency stage half as long; config 3: the tail at block 16384)
    2) GEO="--channels 4096 --subsets 2 --head-log 9 --tail-log 13 --k1-head 8 --k1-tail 32" ;;
    3) GEO="--channels 2048 --subsets 2 --head-log 8 --tail-log 14 --k1-head 32 --k1-tail 32" ;;
    5) GEO="--channels 4096 --subsets 2 --head-log 12 --tail-log 13 --k1-head 0 --k1-tail 16 --patch0-family fir_head" ;;
  esac
  LS=""; [ "$C" = "5" ] && LS="--lockstep 1"     # (config 5's geometry in the lock-step regime: the entry `config5` of the default line)
  STEPS=4; [ "$C" = "3" ] && STEPS=8          # (whole first-level tiles of the tail stage: config 3: 32 blocks of 16384 = 8 steps, config 2: 32 blocks of 8192 = 2 steps)
  [ "$C" = "3" ] && KSTEPS=24 || KSTEPS=20
  ARGS="--config $C $LS --steps $STEPS --warmup 1 --cpu-seconds 0 --side 0 --distinct 64"
  KT_ARGS="--config $C $LS --steps $KSTEPS --warmup 5 --cpu-seconds 0 --side 0 --distinct 64"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D/kt" -o kt -- python "$ROOT/bench.py" $KT_ARGS > "$D/bench_under_rocprof.json" 2> "$D/kt.err"
  find "$D/kt" -name '*kernel_stats.csv' -exec cp {} "$D/kernel_stats.csv" \;
  find "$D/kt" -name '*kernel_trace.csv' -exec python "$ROOT/tools/trace_union.py" {} \; > "$D/kernel_union.txt" 2>&1
  rm -rf "$D/kt"
  # the same with the set on ONE queue (every launch has the device to itself): the per-launch averages bench.py's roofline.frac
  # of the dominant kernel is checked against (rocprof_cross_check)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D/kt1" -o kt -- python "$ROOT/bench.py" $KT_ARGS --child-sets 0 > "$D/bench_under_rocprof_one_queue.json" 2> "$D/kt1.err"
  find "$D/kt1" -name '*kernel_stats.csv' -exec cp {} "$D/kernel_stats_one_queue.csv" \;
  rm -rf "$D/kt1"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$D/pmc_$c" -o pmc -- python "$ROOT/bench.py" $ARGS > /dev/null 2> "$D/pmc_$c.err"
    find "$D/pmc_$c" -name '*counter_collection.csv' -exec cp {} "$D/$c.csv" \;
    rm -rf "$D/pmc_$c"
  done
  python "$ROOT/tools/pmc_summarize.py" "$D/FETCH_SIZE.csv" "$D/WRITE_SIZE.csv" --calib-fetch "$OUT/calib_FETCH_SIZE.csv" \
    --calib-write "$OUT/calib_WRITE_SIZE.csv" --command "python bench.py $ARGS" --config $C $GEO -o "$D/traffic.json" > "$D/traffic.txt" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do compact "$D/$c.csv" "$D/${c}_per_kernel.csv"; rm -f "$D/$c.csv"; done
  # the same two counter passes with the set on ONE queue (--child-sets 0: every launch covers all the channels): the traffic
  # behind the `one_queue` entries of the bench line
  GEO1=$(echo "$GEO" | sed 's/--subsets [0-9]*/--subsets 1/')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$D/pmc1_$c" -o pmc -- python "$ROOT/bench.py" $ARGS --child-sets 0 > /dev/null 2> "$D/pmc1_$c.err"
    find "$D/pmc1_$c" -name '*counter_collection.csv' -exec cp {} "$D/one_queue_$c.csv" \;
    rm -rf "$D/pmc1_$c"
  done
  python "$ROOT/tools/pmc_summarize.py" "$D/one_queue_FETCH_SIZE.csv" "$D/one_queue_WRITE_SIZE.csv" --calib-fetch "$OUT/calib_FETCH_SIZE.csv" \
    --calib-write "$OUT/calib_WRITE_SIZE.csv" --command "python bench.py $ARGS --child-sets 0" --config $C $GEO1 --key "config${C}_one_queue" \
    -o "$D/traffic_one_queue.json" > "$D/traffic_one_queue.txt" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do compact "$D/one_queue_$c.csv" "$D/one_queue_${c}_per_kernel.csv"; rm -f "$D/one_queue_$c.csv"; done
done
rm -f "$OUT"/calib_FETCH_SIZE.csv "$OUT"/calib_WRITE_SIZE.csv
python - "$OUT" $CONFIGS <<'PY'
import json, os, sys
out, cfgs = sys.argv[1], sys.argv[2:]
merged = {}
for c in cfgs:
    for name in ("traffic.json", "traffic_one_queue.json"):
        p = os.path.join(out, "config" + c, name)
        if os.path.exists(p):
            merged.update(json.load(open(p)))
json.dump(merged, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print({k: v["channels_per_launch"] for k, v in merged.items()})
PY
cd "$ROOT"
du -sh "$OUT"
