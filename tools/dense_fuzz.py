"""Fuzz of the dense path (encoder + y-head, forward and backward) against the float64 oracle at random sizes: tests/test_gpu_dense.py's
test_forward_matches_oracle_f64 / test_backward_matches_oracle_f64 called with random (B, T, config overrides) -- sizes on both sides of every
kernel-selection threshold (rows per level 4,096 / 6,144 / 8,192 / 16,384 / 32,768), odd lengths, the shortest tapes the poolings allow.
Usage: python tools/dense_fuzz.py [seed] [cases]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_dense as td  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
overs = [{}, {}, {}, {"pooling_type": "sum"}, {"leaky_relu": True}, {"last_gn": False}, {"last_relu": False}, {"last_gn_num_groups": 16}]
bad = 0
t0 = time.time()
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    B = int(rng.choice([1, 1, 2, 3, 4, 8]))
    rows = int(rng.choice([rng.integers(16, 400), rng.integers(400, 4200), rng.integers(4000, 8300), rng.integers(8000, 17000), rng.integers(16000, 34000)]))
    T = max(16, rows // B + int(rng.integers(0, 3)))
    over = overs[int(rng.integers(0, len(overs)))]
    for name, fn in (("forward", td.test_forward_matches_oracle_f64), ("backward", td.test_backward_matches_oracle_f64)):
        try:
            fn(B, T, over)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", name, "B", B, "T", T, over, str(e)[:300])
    print(f"case {i}: B={B} T={T} {over} ok ({time.time() - t0:.0f} s)", flush=True)
print("bad", bad)
