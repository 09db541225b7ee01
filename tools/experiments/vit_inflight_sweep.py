"""ON THE GPU BOX under `rocprofv3 --kernel-trace`: config-5 decodes (T = 16,384, N = 64) with 9 .. 512 videos per call (the two-launch
throughput schedule) -- how long the frame-score and the DP launch last as the chip fills."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import ops
from mucon_amd.core.viterbi import PoissonModel
C, dev, fs, max_len, T, N = 48, "cuda", 30, 2000, 16384, 64
g = torch.Generator().manual_seed(7)
tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
mu = np.ones(C); mu[np.unique(tr)] = T / N
P = PoissonModel(mu).rows_for(tr, fs)
base = [torch.log_softmax(3 * torch.randn(T, C, device=dev), dim=1) for _ in range(64)]
for nv in (9, 16, 32, 64, 128, 256, 512):
    lps = (base * 8)[:nv]
    for _ in range(3):
        ops.viterbi_decode_batch(lps, [tr] * nv, [P] * nv, fs, max_len)
torch.cuda.synchronize()
