"""cProfile of a warm MuConEvaluator.evaluate() (bench.py's evaluation leg: the second pass over the videos)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, cProfile, pstats
from mucon_amd.mucon import evaluators
dev = torch.device("cuda:0")
holder = {}
orig = evaluators.MuConEvaluator.evaluate
calls = {"n": 0}
def wrapped(self, *a, **k):
    calls["n"] += 1
    if calls["n"] == 2:
        pr = cProfile.Profile(); pr.enable()
        r = orig(self, *a, **k); torch.cuda.synchronize(); pr.disable(); holder["pr"] = pr
        return r
    return orig(self, *a, **k)
evaluators.MuConEvaluator.evaluate = wrapped
print(bench.eval_bench(dev, n_videos=int(sys.argv[1]) if len(sys.argv) > 1 else 16))
pstats.Stats(holder["pr"]).sort_stats("cumtime").print_stats(38)
