"""A/B of the evaluator's MD route (persistent lists with a skin) against the rebuild-everything route: 50 000-atom ternary
frame on a +-0.01 A walk, and the 128-atom host-array call.  UF3_LIB_PATH picks the library build."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("50k", "both"):
    r = bench.extra_eval_50k(torch, dev, cpu=False, steps=int(os.environ.get("STEPS", 200)), warmup=20)
    print("eval_50k MD", r["value"], r["ms_per_step"], "ms; plain", r["md"]["rebuild_everything_route"], "builds", r["md"]["list_builds_in_timed_steps"],
          "kernels", r["roofline"]["eval_kernels_ms_per_step"], "nbr", r["roofline"]["neighbor_ms_per_step"])
if which in ("128", "both"):
    r = bench.extra_eval_128()
    print("eval_128 MD", r["value"], "us; plain", r["md"]["rebuild_everything_route"]["value"], "builds", r["md"]["list_builds_in_timed_steps"])
