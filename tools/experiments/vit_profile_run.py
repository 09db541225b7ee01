"""ON THE GPU BOX, under `rocprofv3 --kernel-trace --stats`: single-video decodes (T=2000/N=6: one-launch kernel; T=16384/N=64: pair kernel)
and batches of 256 (two launches) -- the per-kernel durations behind bench.py's `viterbi` numbers."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import ops
from mucon_amd.core.viterbi import PoissonModel
C, dev, fs, max_len = 48, "cuda", 30, 2000
for (T, N, reps, nb) in ((2000, 6, 200, 256), (16384, 64, 50, 256)):
    g = torch.Generator().manual_seed(7)
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C); mu[np.unique(tr)] = T / N
    P = PoissonModel(mu).rows_for(tr, fs)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for _ in range(reps): ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len)
    lps = [lp.clone() for _ in range(nb)]
    for _ in range(5): ops.viterbi_decode_batch(lps, [tr] * nb, [P] * nb, fs, max_len)
torch.cuda.synchronize()
