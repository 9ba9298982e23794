#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats + two PMC passes of the bench workload.
# Output: gpurun_out/prof_r$1/{trace,pmc_fetch,pmc_write}; summarise with tools/summarize_profile.py.
R=${1:-01}
OUT=gpurun_out/prof_r$R
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-zero-pass --no-other-configs --no-wakeup > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact-zero-pass --no-other-configs --no-wakeup > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact-zero-pass --no-other-configs --no-wakeup > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_mfma -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact-zero-pass --no-other-configs --no-wakeup > $OUT/bench_pmc_mfma.log 2>&1
find $OUT -type f | head -50
du -sh $OUT
tail -2 $OUT/bench_trace.log | cut -c1-400
