#!/bin/bash
# C1 (BASELINE config 1) step under rocprofv3: per-kernel table of the small-model path -> gpurun_out/prof_c1/
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_c1
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_c1 -o trace --output-format csv -- python /root/repo/tools/c1_step.py 2>&1 | grep "C1:"
