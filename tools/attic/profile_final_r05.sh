#!/bin/bash
# Last session of round 5: profiles of the workloads whose kernels changed (fold-pair forward: C3E, HE; strict mode: HS; C1's
# per-set quadrature kernel) -- tools/profile_configs.sh passes + C1's kernel table.  Summaries: tools/summarize_profile.py 05 <W>.
export TMPDIR=/tmp
for W in C3E HE HS; do bash tools/profile_configs.sh 05 $W 2 > gpurun_out/prof05_$W.log 2>&1; done
mkdir -p gpurun_out/prof_r05_C1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05_C1/trace -o trace --output-format csv -- python tools/c1_step.py > gpurun_out/prof_r05_C1/trace.log 2>&1
find gpurun_out/prof_r05_C1 -name "*_trace.csv" -size +2M -delete
find gpurun_out -name "*agent_info.csv" -delete
grep "C1:" gpurun_out/prof_r05_C1/trace.log
tail -2 gpurun_out/prof05_*.log | cut -c1-300
du -sh gpurun_out/prof_r05*
