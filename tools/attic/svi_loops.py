"""Times the three device-resident SVI loops of configuration C3 (bench.py's svi_config) on their own."""
import argparse, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
args = argparse.Namespace(other_steps=a.iters)
out = bench.svi_config(args, a.iters)
print(json.dumps({k: v for k, v in out.items() if "svi" in k or k in ("name", "ms_per_step")}))
