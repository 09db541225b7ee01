"""ON THE GPU BOX: where the time of an ops.viterbi_decode_batch call goes -- Python before the C call, the C call's own phases
(mucon_test_vit_host_phases: staging set-up incl. the memcpy of the length tables, launches, wait, copy-out), Python after it --
for the bench's shapes, with ONE shared (transcript, table) for all videos (what bench.py passes) and with DISTINCT ones."""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import _lib, ops
from mucon_amd.core.viterbi import PoissonModel
lib = _lib.load()
C, dev, fs, max_len = 48, "cuda", 30, 2000
ph = (ctypes.c_double * 4)()
for (T, N, nv) in ((2000, 6, 1), (2000, 6, 256), (16384, 64, 1), (16384, 64, 64), (16384, 64, 256)):
    g = torch.Generator().manual_seed(7)
    trs, Ps = [], []
    for v in range(nv):
        tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
        mu = np.ones(C); mu[np.unique(tr)] = T / N
        trs.append(tr); Ps.append(PoissonModel(mu).rows_for(tr, fs))
    lps = [torch.log_softmax(3 * torch.randn(T, C, device=dev), dim=1) for _ in range(min(nv, 64))] * max(1, nv // 64)
    for shared in (True, False):
        a_tr, a_P = ([trs[0]] * nv, [Ps[0]] * nv) if shared else (trs, Ps)
        for fmt in ("lazy", "uint8", "int32"):
            for _ in range(3): ops.viterbi_decode_batch(lps, a_tr, a_P, fs, max_len, labels=fmt)
            tot, phs = [], []
            for _ in range(7):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = ops.viterbi_decode_batch(lps, a_tr, a_P, fs, max_len, labels=fmt)
                tot.append((time.perf_counter() - t0) * 1e6)
                lib.mucon_test_vit_host_phases(ph); phs.append(list(ph))
            i = int(np.argsort(tot)[len(tot) // 2])
            p = phs[i]
            t0 = time.perf_counter(); _ = [x.labels for x in r]; t_exp = (time.perf_counter() - t0) * 1e6
            print(f"T={T} N={N} nv={nv} shared={int(shared)} {fmt:5s}: call {tot[i]:8.1f} us = {tot[i]/nv:7.2f}/video | C: stage {p[0]:7.1f} launch {p[1]:6.1f} wait {p[2]:7.1f} out {p[3]:6.1f} | python {tot[i]-sum(p):7.1f} | .labels of all {t_exp:7.1f}")
