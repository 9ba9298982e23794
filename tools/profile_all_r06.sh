#!/bin/bash
# Round-6 evidence (run on the GPU box via gpurun, from the repo root): the headline bench under rocprofv3 (kernel stats + the three
# PMC passes), and the per-workload passes of the kernels that changed this round -- the strict q(f) mode's one-launch-per-block
# triangular solves (HSL: two-solve form in the ladder regime, HS / HSE: one-solve form), colstats P >= 2 through SGPRs (C5), plus C2 / C3 for the record.
# Summaries: python tools/summarize_profile.py 06 [W]  ->  profiles/r06_*.csv
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/profile_round.sh 06 > gpurun_out/prof06_H.log 2>&1
for W in HS HSL HSE C5 C2 C3; do bash tools/profile_configs.sh 06 $W 2 > gpurun_out/prof06_$W.log 2>&1; done
find gpurun_out -name "*agent_info.csv" -delete
find gpurun_out/prof_r06 -name "*kernel_trace.csv" -size +8M -delete
du -sh gpurun_out/prof_r06*
tail -1 gpurun_out/prof06_*.log | cut -c1-200
