"""Viterbi decode under the reference's beam (csrc/viterbi_beam.hip): ms per video for a few (T, N, max_hypotheses), next to the decode without a
beam and to the C oracle's beam on one host core."""
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import oracle  # noqa: E402
from mucon_amd import ops, synth  # noqa: E402

C, FS, MAXLEN = 48, 30, 2000
for T, N, beams in ((2000, 6, (10, 100)), (9741, 30, (100, 1000)), (16384, 64, (100, 1000, 4000))):
    tr = synth.transcript(1, N, C)
    lp = synth.emissions(2, T, C, labels=synth.segment_labels(3, T, tr))
    P = oracle.length_rows(oracle.poisson_table(np.full(C, float(T) / N), MAXLEN), tr, FS, MAXLEN)
    d = torch.from_numpy(lp).cuda()

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    base = timed(lambda: ops.viterbi_decode_batch([d], [tr], [P], FS, MAXLEN), 20)
    print(f"T={T} N={N}: no beam {base:.3f} ms")
    for mh in beams:
        g = timed(lambda: ops.viterbi_decode_beam([d], [tr], [P], FS, MAXLEN, mh))
        t0 = time.perf_counter()
        oracle.viterbi_decode_table(lp, tr, P, FS, MAXLEN, max_hypotheses=mh)
        o = (time.perf_counter() - t0) * 1e3
        g8 = timed(lambda: ops.viterbi_decode_beam([d] * 8, [tr] * 8, [P] * 8, FS, MAXLEN, mh), 3) / 8
        print(f"   max_hypotheses={mh}: device {g:.3f} ms (8 per call: {g8:.3f} ms per video); C oracle, one core {o:.1f} ms")
