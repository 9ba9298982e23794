#!/bin/bash
# rocprofv3 kernel trace of the facade's device-resident SVI loops (Adadelta, then natural gradient) -> tools/svi_gaps.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/svi_trace -o tr --output-format csv -- python tools/svi_loop_small.py ${1:-adadelta} > gpurun_out/svi_trace.log 2>&1
f=$(find gpurun_out/svi_trace -name "*kernel_trace.csv" | head -1)
python tools/svi_gaps.py $f > gpurun_out/svi_gaps_${1:-adadelta}.txt 2>&1
tail -3 gpurun_out/svi_trace.log >> gpurun_out/svi_gaps_${1:-adadelta}.txt
rm -rf gpurun_out/svi_trace
cat gpurun_out/svi_gaps_${1:-adadelta}.txt
