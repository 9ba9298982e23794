export TMPDIR=/tmp
mkdir -p gpurun_out
for W in C3 C3E; do bash tools/profile_configs.sh 05 $W 2 > gpurun_out/prof05_$W.log 2>&1; done
find gpurun_out -name "*agent_info.csv" -delete
( time python bench.py ) > gpurun_out/bench_r05d.log 2>&1
tail -c 400 gpurun_out/bench_r05d.log
