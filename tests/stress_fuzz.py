"""Run the seeded fuzz tests over many more seeds than the suite does (GPU box):
    python tests/stress_fuzz.py [first] [count]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    for fn in (T.test_fuzz_geometry_and_call_pattern, T.test_fuzz_single_stage_sets):
        try:
            fn(seed)
        except Exception as e:
            bad += 1
            print("FAIL", fn.__name__, seed, str(e)[:300])
print("seeds", first, "..", first + count - 1, "failures", bad)
