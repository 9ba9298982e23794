export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
for d in 0 1; do
  echo "== HMOGP_KUF_DEFER=$d"
  HMOGP_KUF_DEFER=$d python tools/run_config.py 8192 1024 3 10 7
  HMOGP_KUF_DEFER=$d python tools/run_config.py 25000 1024 3 10 7
  HMOGP_KUF_DEFER=$d python tools/run_config.py 50000 1024 3 10 7
  HMOGP_KUF_DEFER=$d python tools/run_config.py 25000 2048 2 5 7
  HMOGP_KUF_DEFER=$d python tools/run_config.py 200000 512 3 5 7
  HMOGP_KUF_DEFER=$d python tools/run_config.py 200000 1024 3 5 7
done > $O/defer.txt 2>&1
python tools/run_config.py 8192 1024 3 10 1 1 >> $O/defer.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/t5.log 2>&1; echo "rc=$?" >> $O/t5.log
grep -v amdgpu.ids $O/defer.txt; tail -3 $O/t5.log
