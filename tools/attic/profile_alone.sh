#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: the "alone" numbers the README quotes, with the rocprofv3 kernel
# statistics of the same commands.  Output: gpurun_out/prof_alone/ -> copy the two summaries into profiles/.
OUT=gpurun_out/prof_alone
export TMPDIR=/tmp
mkdir -p $OUT
{
  echo "# python tools/bench_gemm.py <role> 800000 1024   (HIP events around 10 launches; one latent)"
  for r in 1 3 4 2 6 5; do python tools/bench_gemm.py $r 800000 1024 2>&1 | grep role; done
  echo "# python tools/bench_gemm.py <role> 200000 1024   (the headline's per-segment launch shape, one latent)"
  for r in 1 3 2 5; do python tools/bench_gemm.py $r 200000 1024 2>&1 | grep role; done
  echo "# stage-first bare loops: tools/probes/build/probe_gemm16"
  ./tools/probes/build/probe_gemm16
} > $OUT/alone.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o alone --output-format csv -- python tools/bench_gemm.py 3,2,6,5 800000 1024 > $OUT/rocprof.log 2>&1
cat $OUT/alone.txt
