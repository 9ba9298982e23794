for s in 20066 20135 20407 20665; do for v in 1 0; do echo "== seed $s scalar=$v"; HMOGP_COLSTATS_SCALAR=$v python tools/fuzz_seed.py $s 2>&1 | grep -v amdgpu | head -5 | cut -c1-600; done; done
