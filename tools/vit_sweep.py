import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import _lib, ops
from mucon_amd.core.viterbi import PoissonModel
C, dev, fs, max_len = 48, "cuda", 30, 2000
for (T, N) in ((2000, 12), (4000, 20), (6000, 30), (16384, 64), (16384, 100)):
    g = torch.Generator().manual_seed(7)
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C); mu[np.unique(tr)] = T / N
    P = PoissonModel(mu).rows_for(tr, fs)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for _ in range(5): ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len)
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len); ts.append(time.perf_counter() - t0)
    # batch of 64
    lps = [lp] * 64; trs = [tr] * 64; Ps = [P] * 64
    for _ in range(2): ops.viterbi_decode_batch(lps, trs, Ps, fs, max_len)
    tb = []
    for _ in range(10):
        t0 = time.perf_counter(); ops.viterbi_decode_batch(lps, trs, Ps, fs, max_len); tb.append(time.perf_counter() - t0)
    print(f"T={T} N={N}: single {sorted(ts)[50]*1e6:.1f} us | batch of 64: {sorted(tb)[5]*1e6/64:.1f} us per video")
