"""Fuzz of the beam decode (csrc/viterbi_beam.hip) against the literal oracle: random small cases (frame_sampling 1 .. 30, 1 .. 11 states, beams 1 .. 150, integer-valued / constant / zero /
Gaussian emissions and length scores, -inf entries, 1 .. 5 videos per call).  Usage: python tools/beam_fuzz.py [seed] [iterations]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import oracle
from mucon_amd import ops, _lib
from helpers import f64_bits
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
C = 48
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1500):
    fs = int(rng.choice([1, 2, 3, 7, 30])); J = int(rng.integers(2, 40)); max_len = J * fs + int(rng.integers(0, fs))
    lps, trs, Ps, wants = [], [], [], []
    mh = int(rng.choice([1, 2, 3, 5, 9, 20, 50, 150]))
    for v in range(int(rng.integers(1, 6))):
        N = int(rng.integers(1, 12)); K = int(rng.integers(1, J * N + 2)); T = K * fs + int(rng.integers(0, fs))
        tr = rng.integers(0, C, N).astype(np.int32)
        mode = int(rng.integers(0, 4))
        lp = (rng.integers(-3, 1, (T, C)).astype(np.float32) if mode == 0 else rng.standard_normal((T, C)).astype(np.float32) if mode == 1
              else np.full((T, C), -1.0, np.float32) if mode == 2 else np.zeros((T, C), np.float32))
        P = rng.integers(-2, 1, (J, N)).astype(np.float64) if mode != 1 else rng.standard_normal((J, N))
        if mode == 3: P = np.zeros((J, N))
        P[rng.random((J, N)) < 0.1] = -np.inf
        try:
            w = oracle.viterbi_decode_table(lp, tr, P, fs, max_len, max_hypotheses=mh)
        except oracle.OracleDecodeError as e:
            w = e.status
        lps.append(torch.from_numpy(lp).cuda()); trs.append(tr); Ps.append(P); wants.append(w)
    res = ops.viterbi_decode_beam(lps, trs, Ps, fs, max_len, mh)
    for r, w, tr, lp in zip(res, wants, trs, lps):
        n += 1
        if isinstance(w, int):
            ok = (w == oracle.ST_NO_HYPOTHESIS and r.status == _lib.VIT_NO_HYPOTHESIS) or (w == oracle.ST_INDEX_ERROR and r.status == _lib.VIT_INDEX_ERROR)
        else:
            ok = r.status in (_lib.VIT_OK, _lib.VIT_TRUNCATED) and f64_bits(r.score) == f64_bits(w[0]) and np.array_equal(r.seg_len, w[3]) and np.array_equal(r.labels, w[1])
        if not ok:
            bad += 1
            print("MISMATCH", it, fs, J, mh, len(tr), lp.shape, r, w if isinstance(w, int) else (w[0], w[3]))
print("cases", n, "bad", bad)
