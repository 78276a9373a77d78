"""Run the seeded fuzz tests over many more seeds than the suite does (GPU box):
    python tests/stress_fuzz.py [first] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = runs = 0
for seed in range(first, first + count):
    jobs = [(T.test_fuzz_geometry_and_call_pattern, (seed, "default")), (T.test_fuzz_geometry_and_call_pattern, (seed, "force")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, False)), (T.test_fuzz_block_synchronous_time_tiling, (seed, True)),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "force")), (T.test_fuzz_single_stage_sets, (seed,)),
            (T.test_fuzz_geometry_and_call_pattern, (seed, "force2")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "force2")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "force2_k32")), (T.test_guard_bands_stay_intact_and_outputs_finite, (seed,)),
            (T.test_fuzz_child_sets_call_patterns, (seed,)),
            # the delay-1 tail stage of many-channel sets, forced on these small ones: the tail at twice the block / half the zero-latency stage
            (T.test_fuzz_geometry_and_call_pattern, (seed, "widen")), (T.test_fuzz_geometry_and_call_pattern, (seed, "shrink")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "widen")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "shrink_force2")),
            # round 6: the tail tiles in channel groups out of phase (what sets of >= 256 channels run), and spread sweeps
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "phases_force2_k32")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "phases_shrink_force2")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "phases_widen")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "phases_force")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "spread3_force2")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "spread1_force2_k32")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "phspread_force2_k32")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "phspread_shrink_force2")),
            # round 6, late: third-level sweeps in both stages on top of those
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "third_force")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "third_force2_k32")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "third_phases_force2_k32")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "third_phases_shrink_force2")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "third_phases_widen")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "third_spread3_force2")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "third_phspread_force2_k32")),
            # ... and the general per-block path (every transform in double), time-tiled
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "f64_force")), (T.test_fuzz_block_synchronous_time_tiling, (seed, "f64_third_force2_k32")),
            (T.test_fuzz_block_synchronous_time_tiling, (seed, "f64_third_phases_force2_k32"))]
    for fn, a in jobs:
        runs += 1
        if os.environ.get("STRESS_LOG"):
            print("RUN", fn.__name__, a, flush=True)
        try:
            fn(*a)
        except Exception as e:
            bad += 1
            print("FAIL", fn.__name__, a, str(e)[:300], flush=True)
print("seeds", first, "..", first + count - 1, "runs", runs, "failures", bad)
