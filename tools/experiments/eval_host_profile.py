"""Where the host time of one MuConEvaluator.evaluate() pass goes: cProfile over the pass bench.py's `evaluation` leg times
(32 videos of T = 2000, Viterbi on), top functions by cumulative and by own time.  Usage: python tools/eval_host_profile.py [n]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    holder = {}
    real = bench.time.perf_counter
    # eval_bench builds the evaluator and runs two passes; profile a third one through the same objects
    from mucon_amd.mucon import evaluators
    orig = evaluators.MuConEvaluator.evaluate

    def spy(self, *a, **k):
        holder["ev"] = self
        return orig(self, *a, **k)

    evaluators.MuConEvaluator.evaluate = spy
    print(bench.eval_bench(dev))
    evaluators.MuConEvaluator.evaluate = orig
    ev = holder["ev"]
    n = len(ev.test_db)
    torch.cuda.synchronize()
    t0 = real()
    ev.evaluate()
    torch.cuda.synchronize()
    print(f"plain pass: {(real() - t0) / n * 1e3:.3f} ms per video")
    pr = cProfile.Profile()
    pr.enable()
    ev.evaluate()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr).stats          # {(file, line, name): (cc, nc, tottime, cumtime, callers)}
    rows = [(f"{os.path.basename(k[0])}:{k[1]}({k[2]})", v[1], v[2], v[3]) for k, v in st.items()]
    for title, col in (("cumulative", 3), ("own", 2)):
        print(f"-- top 50 by {title} time, us per video (under cProfile: every Python call costs ~1 us extra)")
        for name, nc, tt, ct in sorted(rows, key=lambda r: -r[col])[:50]:
            print(f"{nc:7d} calls  own {tt / n * 1e6:8.1f}  cum {ct / n * 1e6:8.1f}  {name}")


if __name__ == "__main__":
    main()
