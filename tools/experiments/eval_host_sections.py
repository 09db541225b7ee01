"""Host time of the pieces of the batched evaluation (bench.py's evaluation leg: MuConEvaluator.evaluate over 32 videos of T = 2000), measured WITHOUT a profiler:
the functions _evaluate_chunk_on calls are wrapped in perf_counter stamps.  Usage: python tools/experiments/eval_host_sections.py"""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from mucon_amd.mucon import evaluators as E  # noqa: E402

acc = defaultdict(float)
cnt = defaultdict(int)


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[label or name] += time.perf_counter() - t0
        cnt[label or name] += 1
        return r
    setattr(obj, name, g)


for n in ("mean_lengths_from_s_head", "poisson_params_for_many", "create_segmentation_from_segments", "make_same_size_interpolate"):
    if hasattr(E, n):
        wrap(E, n)
orig_init = E.MuConEvaluator.__init__


def init(self, *a, **k):
    orig_init(self, *a, **k)
    wrap(self.model, "forward_deferred")
    wrap(self, "_evaluate_chunk_on", "chunk total")


E.MuConEvaluator.__init__ = init
from mucon_amd.core.metrics import device as MD  # noqa: E402
wrap(MD, "segmental_counters")
from mucon_amd.core.viterbi import viterbi as V  # noqa: E402
wrap(V.Viterbi, "decode_batch")
orig_cat = torch.cat
dev = torch.device("cuda:0")
for rep in range(3):
    acc.clear()
    cnt.clear()
    t0 = time.perf_counter()
    out = bench.eval_bench(dev)          # (runs evaluate() twice: one warm-up pass + the timed one -- the sums below cover both, 64 videos)
    dt = time.perf_counter() - t0
print(out["ms_per_video"], "ms per video (timed pass)")
nv = 64
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:36s} {v / nv * 1e6:8.1f} us per video   ({cnt[k]} calls)")
