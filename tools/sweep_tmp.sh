O=gpurun_out/r3
mkdir -p $O
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_engine.py tests/test_facade_gpu.py -q -m gpu > $O/t6.log 2>&1; echo "rc=$?" >> $O/t6.log
grep -E "^E |FAILED|passed|failed|rc=" $O/t6.log | head -40
