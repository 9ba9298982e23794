export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
for cap in 128 160 192 224 256 320; do
  HMOGP_COLSTATS_CAP=$cap python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-zero-pass --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('M=1024 cap=$cap', 'ms/step %.2f'%d['ms_per_step'], {x:round(k[x],2) for x in ('forward_gemm','gram_gemm','colstats_reduce','rbf_cross_cov','mxm_algebra')})"
done > $O/sweep_cap2.txt 2>&1
for cap in 0 192 384 576 768; do
  HMOGP_COLSTATS_CAP=$cap python bench.py --inducing 512 --steps 10 --warmup 3 --no-cpu-baseline --no-exact-zero-pass --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('M=512 cap=$cap', 'ms/step %.2f'%d['ms_per_step'], {x:round(k[x],2) for x in ('forward_gemm','gram_gemm','colstats_reduce','rbf_cross_cov','mxm_algebra')})"
done >> $O/sweep_cap2.txt 2>&1
for cap in 0 96 192 384; do
  HMOGP_COLSTATS_CAP=$cap python tools/run_config.py 8192 1024 3 10 7 2>/dev/null | sed "s/^/cap=$cap /"
  HMOGP_COLSTATS_CAP=$cap python tools/run_config.py 25000 1024 3 10 7 2>/dev/null | sed "s/^/cap=$cap /"
done >> $O/sweep_cap2.txt 2>&1
python tools/bench_gemm.py > $O/bench_gemm.txt 2>&1
rocprofv3 --kernel-trace -d $O/trace_c3 -o t --output-format csv -- python tools/run_config.py 8192 1024 3 5 7 > $O/trace_c3.log 2>&1
f=$(find $O/trace_c3 -name '*kernel_trace.csv' | head -1)
python tools/step_trace.py $f > $O/c3_step_trace.txt 2>&1
python tools/step_timeline.py $f > $O/c3_step_timeline.txt 2>&1
rocprofv3 --kernel-trace -d $O/trace_c5 -o t --output-format csv -- python tools/run_config.py 25000 2048 2 3 7 > $O/trace_c5.log 2>&1
f=$(find $O/trace_c5 -name '*kernel_trace.csv' | head -1)
python tools/step_trace.py $f > $O/c5_step_trace.txt 2>&1
rm -rf $O/trace_c3 $O/trace_c5
cat $O/sweep_cap2.txt
