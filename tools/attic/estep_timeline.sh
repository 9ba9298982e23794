cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/run_config.py 8192 1024 3 100 1 1 2>&1 | tail -1 > gpurun_out/estep.txt
rocprofv3 --kernel-trace -d gpurun_out/estep_trace -o tr --output-format csv -- python tools/run_config.py 8192 1024 3 5 1 1 > /dev/null 2>&1
f=$(find gpurun_out/estep_trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 5 unpack_tril_kernel >> gpurun_out/estep.txt 2>&1
rm -rf gpurun_out/estep_trace
cat gpurun_out/estep.txt
