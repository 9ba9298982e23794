#!/bin/bash
# Round-5 evidence: per-configuration rocprofv3 passes (tools/profile_configs.sh), C1's kernel table, and the Gram's FETCH with the
# column statistics uncapped (HMOGP_COLSTATS_CAP=0: the guest then runs beside the Gram for ~9 ms instead of ~38).
export TMPDIR=/tmp
for W in C2 C3 C3E C5 HD; do bash tools/profile_configs.sh 05 $W 3 > gpurun_out/prof05_$W.log 2>&1; done
mkdir -p gpurun_out/prof_r05_C1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05_C1/trace -o trace --output-format csv -- python tools/c1_step.py > gpurun_out/prof_r05_C1/trace.log 2>&1
rocprofv3 --hip-trace --stats -d gpurun_out/prof_r05_C1/hip -o hip --output-format csv -- python tools/c1_step.py > gpurun_out/prof_r05_C1/hip.log 2>&1
find gpurun_out/prof_r05_C1 -name "*_trace.csv" -size +2M -delete
HMOGP_COLSTATS_CAP=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_r05_uncapped/pmc_fetch -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact-zero-pass --no-other-configs > gpurun_out/prof_r05_uncapped.log 2>&1
find gpurun_out -name "*agent_info.csv" -delete
find gpurun_out/prof_r05_uncapped -name "*kernel_trace.csv" -size +4M -delete
grep "C1:" gpurun_out/prof_r05_C1/trace.log
du -sh gpurun_out/prof_r05*
