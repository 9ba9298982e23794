#!/bin/bash
# u_small_kernel's phases by early exit of block (q, 0) (HMOGP_USMALL_STOP = 1: K_uu built, 2: factorised + inverted, 3: K_uu^-1 and a
# formed; 0: complete).  With STOP=1 the kernel's duration is block (q, 1)'s (the q(u) chain).  Results are wrong for STOP != 0.
cd /tmp && export TMPDIR=/tmp
for s in ${STOPS:-1 2 3 0}; do
  rm -rf /tmp/pu$s
  HMOGP_USMALL_STOP=$s timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pu$s -o trace --output-format csv -- python /root/repo/tools/c1_step.py > /dev/null 2>&1
  echo "STOP=$s $(grep u_small /tmp/pu$s/trace_kernel_stats.csv | cut -d, -f1,4 | head -1)"
done
