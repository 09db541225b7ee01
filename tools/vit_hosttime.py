"""Where the host time of ONE single-video Viterbi decode goes (T = 2000, N = 6): cProfile over 2,000 calls of
ops.viterbi_decode_batch and of Viterbi.decode (the reference-shaped entry point)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import ops
from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi

C, T, N = 48, 2000, 6
g = torch.Generator().manual_seed(7)
tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
mu = np.ones(C); mu[np.unique(tr)] = T / N
lm = PoissonModel(mu)
P = lm.rows_for(tr, 30)
lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).cuda()
v = Viterbi(SingleTranscriptGrammar([int(x) for x in tr], C), lm, frame_sampling=30)
for _ in range(20):
    ops.viterbi_decode_batch([lp], [tr], [P], 30, 2000); v.decode(lp)
for name, fn in (("ops.viterbi_decode_batch", lambda: ops.viterbi_decode_batch([lp], [tr], [P], 30, 2000)), ("Viterbi.decode", lambda: v.decode(lp))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): fn()
    torch.cuda.synchronize(); print(name, "us/call", (time.perf_counter() - t0) / 2000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): v.decode(lp)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
