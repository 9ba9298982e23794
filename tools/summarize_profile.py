#!/usr/bin/env python
"""Summarise a tools/profile_round.sh run (gpurun_out/prof_rNN) into profiles/:
  rNN_kernel_stats.csv      rocprofv3 --kernel-trace --stats per-kernel table (verbatim)
  rNN_pmc_hbm.csv           per kernel and grid: launches, raw FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports), and the
                            HBM bytes per launch corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
                            hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
                            (FETCH_SIZE counts 128-B fabric read requests at 64 B; calibrated here on rowstats/colstats,
                            which read K^ and P~ exactly once: raw FETCH is 0.50x the known byte count)
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "01"
    src = os.path.join(ROOT, "gpurun_out", "prof_r%s" % rnd)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "r%s_kernel_stats.csv" % rnd))
    agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for which, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = os.path.join(src, "pmc_%s" % which, "pmc_counter_collection.csv")
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == ctr:
                agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][ctr].append(float(r["Counter_Value"]))
    rows = []
    for (name, grid), v in agg.items():
        f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        rows.append(dict(kernel=name, grid_threads=grid, launches=max(len(f), len(w)), fetch_kb_raw_avg=round(fa, 1),
                         write_kb_raw_avg=round(wa, 1), hbm_bytes_per_launch=int((2.0 * fa + wa) * 1024)))
    rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
    with open(os.path.join(dst, "r%s_pmc_hbm.csv" % rnd), "w", newline="") as fh:
        wri = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        wri.writeheader()
        wri.writerows(rows)
    for r in rows[:12]:
        print("%-40s grid=%-9d n=%-3d fetch_raw=%10.0f KB write=%10.0f KB  hbm/launch=%.3f GB" %
              (r["kernel"][:40], r["grid_threads"], r["launches"], r["fetch_kb_raw_avg"], r["write_kb_raw_avg"],
               r["hbm_bytes_per_launch"] / 1e9))
    # the forward contraction of the headline workload's full chunk: the dominant kernel's traffic for bench.py
    fwd = [r for r in rows if r["kernel"].startswith("gemm_f64_kernel<false, true, 1>")]
    fwd.sort(key=lambda r: -r["grid_threads"])
    if fwd:
        json.dump({"kernel": fwd[0]["kernel"], "grid_threads": fwd[0]["grid_threads"], "launches": fwd[0]["launches"],
                   "hbm_bytes_per_launch": fwd[0]["hbm_bytes_per_launch"],
                   "note": "forward contraction of ONE launch = one task chunk (200000 rows) x all Q=3 latents of the headline workload; "
                           "(2*FETCH_SIZE + WRITE_SIZE)*1024 from separate --pmc passes, round %s" % rnd},
                  open(os.path.join(dst, "pmc_forward_gemm.json"), "w"))


if __name__ == "__main__":
    main()
