"""Parity soak (not part of the suite): many more seeds of the randomized engine-vs-oracle comparisons than the tests run.
python tools/soak_parity.py [first_seed] [count]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fz          # noqa: E402
import test_gpu_strict as st        # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for s in range(first, first + count):
    try:
        fz._case(s, fz.MS, False)
    except Exception:                # noqa: BLE001
        bad += 1
        print("default-mode seed", s, "FAILED")
        traceback.print_exc(limit=2)
for s in range(14):
    pass
print("default-mode soak: %d seeds, %d failures" % (count, bad))
