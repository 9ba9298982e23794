#!/bin/bash
# rocprofv3 kernel TRACE of the headline workload in the strict q(f) mode: per-launch durations of the triangular-solve kernels by
# column block (gpurun_out/strict_panel_trace.txt) + the usual stats summary (gpurun_out/strict_stats.txt)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/strict_trace -o tr --output-format csv -- python tools/run_config.py ${1:-200000} ${2:-1024} 3 2 7 0 1 > gpurun_out/strict_run.log 2>&1
f=$(find gpurun_out/strict_trace -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-200 > gpurun_out/strict_stats.txt
tail -1 gpurun_out/strict_run.log >> gpurun_out/strict_stats.txt
t=$(find gpurun_out/strict_trace -name "*kernel_trace.csv" | head -1)
python - "$t" > gpurun_out/strict_panel_trace.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "trsm_panel_kernel" in r["Kernel_Name"]]
last = sel[-16:] if len(sel) >= 16 else sel
for r in last:
    print("%-60s %8.3f ms" % (r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
print("sum of the last step's panel launches: %.3f ms" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6))
PY
rm -rf gpurun_out/strict_trace
cat gpurun_out/strict_stats.txt gpurun_out/strict_panel_trace.txt
