#!/bin/bash
# colstats cap sweep at the headline size (and the C2 shape): step, Gram and colstats spans per cap
for cfg in "200000 1024 3" "200000 512 3"; do
  set -- $cfg
  for cap in ${CAPS:-192 256 320 384 512}; do
    HMOGP_COLSTATS_CAP=$cap python bench.py --rows $1 --inducing $2 --latents $3 --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline --no-exact-zero-pass 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read())
k=l['kernel_ms_per_step']
print('rows $1 M $2 cap $cap: step %.2f fwd %.2f gram %.2f colstats %.2f repl %.2f' % (l['ms_per_step'], k['forward_gemm'], k['gram_gemm'], k['colstats_reduce'], k['mxm_algebra']))"
  done
done
