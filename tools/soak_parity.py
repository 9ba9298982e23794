"""Parity soak (not part of the suite): many more seeds of the randomized engine-vs-oracle comparisons than the tests run.
python tools/soak_parity.py [first_seed] [count] [strict_count]"""
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
print("default-mode soak: %d seeds, %d failures" % (count, bad))
nstrict = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # strict q(f) mode: seeds beyond the suite's 14
sbad = 0
for s in range(first, first + nstrict):
    try:
        st._strict_case(s)
    except Exception:                # noqa: BLE001
        sbad += 1
        print("strict-mode seed", s, "FAILED")
        traceback.print_exc(limit=2)
if nstrict:
    print("strict-mode soak: %d seeds, %d failures" % (nstrict, sbad))
