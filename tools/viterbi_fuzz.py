"""Fuzz of the Viterbi decode WITHOUT a beam (csrc/viterbi.hip through Viterbi.decode_batch: the one-launch / pair kernels for calls of up to 8
videos, the throughput kernels above) against the literal oracle: random small cases -- frame_sampling 1 .. 30, 2 .. 66 length slots, 1 .. 20 states,
fewer columns than states, integer-valued / constant / zero / Gaussian emissions and length scores (ties everywhere), -inf entries, 1 .. 12 videos per
call.  Usage: python tools/viterbi_fuzz.py [seed] [iterations]"""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import oracle  # noqa: E402
from helpers import f64_bits  # noqa: E402
from mucon_amd.core.viterbi import Viterbi  # noqa: E402
from mucon_amd.core.viterbi.viterbi import NoHypothesisError, ShortSequenceError  # noqa: E402


class Table:
    def __init__(self, P, max_len):
        self.P, self.max_len = P, max_len

    def max_length(self):
        return self.max_len

    def rows_for(self, transcript, fs):
        return self.P


rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
C = 48
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1000):
    fs = int(rng.choice([1, 2, 3, 7, 30]))
    J = int(rng.integers(2, 67))
    max_len = J * fs + int(rng.integers(0, fs))
    lps, trs, lms, wants = [], [], [], []
    for v in range(int(rng.choice([1, 1, 2, 3, 8, 9, 12]))):
        N = int(rng.integers(1, 21))
        K = int(rng.integers(1, min(J * N, 400) + 1))
        T = K * fs + int(rng.integers(0, fs))
        tr = rng.integers(0, C, N).astype(np.int32)
        mode = int(rng.integers(0, 4))
        lp = (rng.integers(-3, 1, (T, C)).astype(np.float32) if mode == 0 else rng.standard_normal((T, C)).astype(np.float32) if mode == 1
              else np.full((T, C), -1.0, np.float32) if mode == 2 else np.zeros((T, C), np.float32))
        P = rng.integers(-2, 1, (J, N)).astype(np.float64) if mode != 1 else rng.standard_normal((J, N))
        if mode == 3:
            P = np.zeros((J, N))
        P[rng.random((J, N)) < 0.05] = -np.inf
        try:
            w = oracle.viterbi_decode_table(lp, tr, P, fs, max_len)
        except oracle.OracleDecodeError as e:
            w = e.status
        lps.append(torch.from_numpy(lp).cuda())
        trs.append([int(x) for x in tr])
        lms.append(Table(P, max_len))
        wants.append(w)
    got = Viterbi(None, None, frame_sampling=fs).decode_batch(lps, trs, lms, return_exceptions=True)
    for g, w, tr, lp in zip(got, wants, trs, lps):
        n += 1
        if isinstance(w, int):
            ok = isinstance(g, NoHypothesisError if w == oracle.ST_NO_HYPOTHESIS else ShortSequenceError)
        else:
            ok = (not isinstance(g, Exception) and f64_bits(g[0]) == f64_bits(w[0]) and np.array_equal(np.asarray(g[1]), w[1])
                  and [s.length for s in g[2]] == w[3].tolist() and [s.label for s in g[2]] == w[2].tolist())
        if not ok:
            bad += 1
            print("MISMATCH", it, "fs", fs, "J", J, "N", len(tr), "T", tuple(lp.shape), g if isinstance(g, Exception) else (g[0], [s.length for s in g[2]]),
                  w if isinstance(w, int) else (w[0], w[3].tolist()))
print("cases", n, "bad", bad)
