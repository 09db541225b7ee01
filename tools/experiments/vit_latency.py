"""ON THE GPU BOX: where the latency of a single short decode goes: the whole ops.viterbi_decode_batch call, the C entry point
alone (arguments prepared once), and the kernel alone (HIP events around the C call, which is synchronous)."""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import _lib, ops
from mucon_amd.core.viterbi import PoissonModel

C, dev, fs, max_len = 48, "cuda", 30, 2000
lib = _lib.load()
for (T, N) in ((2000, 6), (4096, 8), (1000, 3), (16384, 64)):
    g = torch.Generator().manual_seed(7)
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C); mu[np.unique(tr)] = T / N
    P = PoissonModel(mu).rows_for(tr, fs)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for _ in range(5): ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len)
    def med(fn, n=200):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[n // 2] * 1e6
    full = med(lambda: ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len))
    vids = (_lib.ViterbiVideo * 1)()
    q = vids[0]; q.lp, q.transcript, q.table, q.T, q.N, q.force_n, q.force_j = lp.data_ptr(), tr.ctypes.data, P.ctypes.data, T, N, -1, -1
    score, n_seg, status = np.empty(1), np.empty(1, np.int32), np.empty(1, np.int32)
    labels, seg = np.empty(T, np.int32), np.empty(N, np.int32)
    st = _lib.current_stream_ptr()
    c_only = med(lambda: lib.mucon_viterbi_decode_host(1, vids, C, fs, max_len, score.ctypes.data, n_seg.ctypes.data, status.ctypes.data,
                                                        labels.ctypes.data, seg.ctypes.data, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ks = []
    for _ in range(50):
        e0.record()
        lib.mucon_viterbi_decode_host(1, vids, C, fs, max_len, score.ctypes.data, n_seg.ctypes.data, status.ctypes.data, labels.ctypes.data,
                                      seg.ctypes.data, st)
        e1.record(); e1.synchronize(); ks.append(e0.elapsed_time(e1) * 1e3)
    print(f"T={T} N={N}: whole call {full:.1f} us | C entry point {c_only:.1f} us | between HIP events {sorted(ks)[25]:.1f} us")
