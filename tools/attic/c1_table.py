"""Prints the per-kernel table of gpurun_out/prof_c1 (tools/c1_profile.sh): average duration and launches per evaluation."""
import csv, glob, sys
f = glob.glob("gpurun_out/prof_c1/**/trace_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
evals = 2500
tot = 0.0
for r in rows:
    if int(r["Calls"]) < evals: continue
    print(r["Name"][:72].ljust(72), "%5.2f / eval" % (int(r["Calls"]) / evals), "%8.2f us" % (float(r["AverageNs"]) / 1e3))
    tot += float(r["TotalDurationNs"])
print("kernel time per evaluation %.1f us" % (tot / evals / 1e3))
