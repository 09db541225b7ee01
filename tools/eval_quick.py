"""ON THE GPU BOX: bench.py's evaluation leg alone: batched with the chunk's forwards on 0 / 2 / 4 / 8 streams, and one video at a time."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mucon_amd.mucon import evaluators
dev = torch.device("cuda:0")
for batched, ns in ((True, 4), (True, 0), (True, 2), (True, 4), (True, 8), (False, 0)):
    evaluators.MuConEvaluator.batched = batched
    evaluators.MuConEvaluator.forward_streams = ns
    for n in (32, 64):
        r = bench.eval_bench(dev, n_videos=n)
        print("batched" if batched else "per-video", f"{ns} streams", n, "videos:", r["ms_per_video"], "ms per video")
