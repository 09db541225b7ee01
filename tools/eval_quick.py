"""ON THE GPU BOX: bench.py's evaluation leg alone, batched and one video at a time."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mucon_amd.mucon import evaluators
dev = torch.device("cuda:0")
for batched in (True, False, True):
    evaluators.MuConEvaluator.batched = batched
    for n in (16, 64):
        r = bench.eval_bench(dev, n_videos=n)
        print("batched" if batched else "per-video", n, "videos:", r["ms_per_video"], "ms per video")
