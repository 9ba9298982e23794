#!/bin/bash
# PMC counters of one contraction launch shape: tools/pmc_gemm.sh <role> <n> <M> <tag>
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$4
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $OUT -o pmc --output-format csv -- python tools/bench_gemm.py $1 $2 $3 3 > $OUT/log.txt 2>&1
python - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/pmc_counter_collection.csv")):
    if "gemm_f64_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-45:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in v.items(): print("   %-28s %14.0f (n=%d)"%(c, sum(vals)/len(vals), len(vals)))
PY
tail -1 $OUT/log.txt
