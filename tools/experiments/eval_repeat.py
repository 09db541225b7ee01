"""ON THE GPU BOX: bench.py's evaluation leg repeated in one process, per number of forward streams (looks for outliers)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mucon_amd.mucon import evaluators
dev = torch.device("cuda:0")
for ns in (0, 2, 4):
    evaluators.MuConEvaluator.forward_streams = ns
    print(ns, "streams:", [bench.eval_bench(dev, n_videos=32)["ms_per_video"] for _ in range(8)])
