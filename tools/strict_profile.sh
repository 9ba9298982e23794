#!/bin/bash
# rocprofv3 kernel stats of the headline workload in the strict q(f) mode: gpurun_out/strict_stats.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/strict_trace -o tr --output-format csv -- python tools/run_config.py ${1:-200000} 1024 3 2 7 0 1 > gpurun_out/strict_run.log 2>&1
f=$(find gpurun_out/strict_trace -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-260 > gpurun_out/strict_stats.txt
tail -1 gpurun_out/strict_run.log >> gpurun_out/strict_stats.txt
rm -rf gpurun_out/strict_trace
cat gpurun_out/strict_stats.txt
