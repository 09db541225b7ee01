"""cProfile of MuConEvaluator.evaluate() (bench.py's evaluation leg)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, cProfile, pstats
pr = cProfile.Profile()
orig = bench.time.perf_counter
res = None
def run():
    global res
    res = bench.eval_bench(torch.device("cuda:0"), n_videos=24)
pr.enable(); run(); pr.disable()
print(res)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
