#!/bin/bash
# A/B on one GPU box: tools/ab_run.sh <baseline .so> -- alone contractions + short bench steps with both builds (gpurun_out/ab.txt)
BASE=${1:-.ab/lib_r3.so}
OUT=gpurun_out/ab.txt
mkdir -p gpurun_out
{
for lib in "$BASE" ""; do
  echo "=== lib: ${lib:-current}"
  export HMOGP_LIB_PATH=$lib
  [ -z "$lib" ] && unset HMOGP_LIB_PATH
  python tools/bench_gemm.py 3,2,6 800000 1024 2>&1 | grep role
  python tools/bench_gemm.py 3,2 800000 512 2>&1 | grep role
  python tools/run_config.py 200000 1024 3 5 2>&1 | tail -1
  python tools/run_config.py 200000 512 3 5 2>&1 | tail -1
  python tools/run_config.py 8192 1024 3 10 2>&1 | tail -1
  python tools/run_config.py 25000 1024 3 10 2>&1 | tail -1
done
} > $OUT 2>&1
cat $OUT
