run() { python tools/run_config.py $1 $2 3 6 7 2>&1 | grep -E "cat 2 |cat 1 |ms/step" | tail -3 | tr '\n' ' ' | sed 's/\[hmogp timeline\]//g' | cut -c1-260; echo; }
for fr in 1 2 3 4 6 8; do for ch in 50000 100000 200000; do
echo "== free $fr layout 1 chunk $ch"; HMOGP_ST2_FREE=$fr HMOGP_ST2_LAYOUT=1 HMOGP_KUF_CHUNK=$ch HMOGP_DEBUG_TIMELINE=1 run 200000 1024
done; done
echo "== M=512 default"; HMOGP_DEBUG_TIMELINE=1 run 200000 512
echo "== M=512 free 4 chunk 200000"; HMOGP_ST2_FREE=4 HMOGP_ST2_LAYOUT=1 HMOGP_KUF_CHUNK=200000 HMOGP_DEBUG_TIMELINE=1 run 200000 512
echo "== 25000 default"; HMOGP_DEBUG_TIMELINE=1 run 25000 1024
echo "== 25000 free 4 chunk 200000"; HMOGP_ST2_FREE=4 HMOGP_ST2_LAYOUT=1 HMOGP_KUF_CHUNK=200000 HMOGP_DEBUG_TIMELINE=1 run 25000 1024
echo "== 8192 default"; HMOGP_DEBUG_TIMELINE=1 run 8192 1024
echo "== 8192 free 4 chunk 200000"; HMOGP_ST2_FREE=4 HMOGP_ST2_LAYOUT=1 HMOGP_KUF_CHUNK=200000 HMOGP_DEBUG_TIMELINE=1 run 8192 1024
