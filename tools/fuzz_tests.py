#!/usr/bin/env python3
"""Run the two differential fuzz tests of tests/test_gpu_fuzz.py on case numbers beyond the
ones the suite pins.  usage: python tools/fuzz_tests.py pipeline|entry first count"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as T
fn = T.test_fuzz_pipeline if sys.argv[1] == "pipeline" else T.test_fuzz_entry_points
first, count = int(sys.argv[2]), int(sys.argv[3])
bad = 0
for case in range(first, first + count):
    try:
        fn(case)
    except AssertionError as e:
        bad += 1
        print("FAIL case", case, str(e)[:900], flush=True)
        if bad >= 5:
            break
    except Exception:
        bad += 1
        print("ERROR case", case, traceback.format_exc()[-900:], flush=True)
        if bad >= 5:
            break
    if (case - first) % 500 == 0:
        print("case", case, flush=True)
print("done:", bad, "failures in", count, flush=True)
sys.exit(1 if bad else 0)
