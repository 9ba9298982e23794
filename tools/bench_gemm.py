#!/usr/bin/env python
"""Micro-benchmark of the two FP64-MFMA row-pass contractions (random operands in HBM):  python tools/bench_gemm.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hetmogp_amd._lib import lib, check  # noqa: E402


def run(role, n, M, iters=10):
    ms = C.c_double()
    check(lib.hmogp_bench_contraction(0, role, n, M, iters, C.byref(ms)))
    flops = (1.0 if role == 2 else 2.0) * n * M * M
    return ms.value, flops / ms.value / 1e9


if __name__ == "__main__":
    if len(sys.argv) >= 4:                       # single shape: role[,role...] n M [iters]
        n, M = int(sys.argv[2]), int(sys.argv[3])
        for role in [int(r) for r in sys.argv[1].split(",")]:      # one role or a comma-separated list (one process)
            ms, tf = run(role, n, M, int(sys.argv[4]) if len(sys.argv) > 4 else 10)
            if role == 5:        # K_uf construction: 3 latents x 8 n M bytes written per launch
                print("role=5 n=%d M=%d : %.3f ms %.0f GB/s written" % (n, M, ms, 3 * 8.0 * n * M / ms / 1e6))
            else:
                print("role=%d n=%d M=%d : %.3f ms %.1f TFLOP/s" % (role, n, M, ms, tf))
        sys.exit(0)
    for role, name in ((1, "forward P~=K^C      (2nM^2)"), (3, "forward + fused stats   "), (4, "fwd + stats, no P~ store"),
                       (2, "gram H+=K^T b K^ (nM^2) ")):
        for n, M in ((131072, 1024), (68928, 1024), (131072, 512), (8192, 1024), (65536, 2048)):
            ms, tf = run(role, n, M)
            print("%s n=%6d M=%4d : %8.3f ms  %6.1f TFLOP/s (algorithmic)  %4.1f%% of 78.6" % (name, n, M, ms, tf, 100 * tf / 78.6))
