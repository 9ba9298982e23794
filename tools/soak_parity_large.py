import os, sys, traceback
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fz
bad = 0
first, count = int(sys.argv[1]), int(sys.argv[2])
for s in range(first, first + count):
    try:
        fz._case(s, fz.MS_LARGE, True)
    except Exception:
        bad += 1
        print("large-M seed", s, "FAILED")
        traceback.print_exc(limit=2)
print("large-M soak (M in %s): %d seeds, %d failures" % (fz.MS_LARGE, count, bad))
