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
    rnd = sys.argv[1] if len(sys.argv) > 1 else "02"
    if len(sys.argv) > 2:          # a workload of tools/profile_configs.sh: gpurun_out/prof_r<round>_<W> -> profiles/r<round>_<W>_*
        rnd = "%s_%s" % (rnd, sys.argv[2])
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
    # MFMA utilisation per kernel (third PMC pass): busy cycles of the matrix pipes / (active cycles per XCD x 1024 SIMDs).
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs here (9.85e6 x 8 for a 4.27 ms kernel at 2.3 GHz).
    mpath = os.path.join(src, "pmc_mfma", "pmc_counter_collection.csv")
    if os.path.exists(mpath):
        m = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(mpath)):
            m[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        mrows = []
        for (name, grid), c in m.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
                continue
            avg = {k: sum(v) / len(v) for k, v in c.items()}
            if avg["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
                continue
            act = avg["GRBM_GUI_ACTIVE"] / 8.0
            mrows.append(dict(kernel=name, grid_threads=grid, launches=len(c["GRBM_GUI_ACTIVE"]),
                              mfma_busy_cycles=int(avg["SQ_VALU_MFMA_BUSY_CYCLES"]), active_cycles_per_xcd=int(act),
                              mfma_util=round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024.0), 4),
                              wait_any_frac=round(avg.get("SQ_WAIT_ANY", 0.0) / max(avg.get("SQ_WAVE_CYCLES", 1.0), 1.0), 4),
                              lds_conflict_frac=round(avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(avg.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 4)))
        mrows.sort(key=lambda r: -r["mfma_busy_cycles"] * r["launches"])
        if mrows:
            with open(os.path.join(dst, "r%s_pmc_mfma.csv" % rnd), "w", newline="") as fh:
                wri = csv.DictWriter(fh, fieldnames=list(mrows[0].keys()))
                wri.writeheader()
                wri.writerows(mrows)
            for r in mrows[:4]:
                print("MFMA util %-36s grid=%-9d util=%.3f wait_any=%.3f lds_conflict=%.3f" %
                      (r["kernel"][:36], r["grid_threads"], r["mfma_util"], r["wait_any_frac"], r["lds_conflict_frac"]))
    # the forward contraction of the headline workload's full chunk: the dominant kernel's traffic for bench.py
    fwd = [r for r in rows if r["kernel"].startswith("rowpass_gemm_kernel<1>") or r["kernel"].startswith("gemm_f64_kernel<false, true, 1>")]
    fwd.sort(key=lambda r: -r["grid_threads"])
    if fwd and len(sys.argv) <= 2:
        json.dump({"kernel": fwd[0]["kernel"], "grid_threads": fwd[0]["grid_threads"], "launches": fwd[0]["launches"],
                   "hbm_bytes_per_launch": fwd[0]["hbm_bytes_per_launch"], "source": "profiles/r%s_pmc_hbm.csv" % rnd,
                   "note": "forward contraction of ONE launch = all 800000 rows x all Q=3 latents of the headline workload; "
                           "(2*FETCH_SIZE + WRITE_SIZE)*1024 from separate --pmc passes, round %s" % rnd},
                  open(os.path.join(dst, "pmc_forward_gemm.json"), "w"))


if __name__ == "__main__":
    main()
