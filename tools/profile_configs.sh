#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats + the PMC passes (FETCH_SIZE, WRITE_SIZE,
# MFMA busy) of ONE workload of tools/run_workload.py.   tools/profile_configs.sh <round> <workload> [steps]
# Output: gpurun_out/prof_r<round>_<workload>/{trace,pmc_fetch,pmc_write,pmc_mfma}; summarise with
#   python tools/summarize_profile.py <round> <workload>      (-> profiles/r<round>_<workload>_*.csv)
# PMC passes are separate runs WITHOUT --stats / sys-trace (MI355X_MICROARCH.md, HBM section; gpurun refuses the combination).
R=${1:-04}
W=${2:-C2}
S=${3:-3}
OUT=gpurun_out/prof_r${R}_$W
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python tools/run_workload.py $W $S 1 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- python tools/run_workload.py $W 1 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc --output-format csv -- python tools/run_workload.py $W 1 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_mfma -o pmc --output-format csv -- python tools/run_workload.py $W 1 0 > $OUT/pmc_mfma.log 2>&1
# keep only what the summary needs (the merged gpurun_out is capped at 64 MiB)
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -1 $OUT/trace.log | cut -c1-300
du -sh $OUT
