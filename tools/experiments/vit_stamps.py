"""ON THE GPU BOX, timing build (MUCON_HIPCC_FLAGS=-DVIT_STAMP=1 python -m mucon_amd.build --force): s_memtime (core clock cycles)
of thread 0 / video 0 at the phase edges of the Viterbi kernels."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import _lib, ops
from mucon_amd.core.viterbi import PoissonModel

C, dev, fs, max_len = 48, "cuda", 30, 2000
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else lib
for (T, N) in ((2000, 6), (2000, 12), (4096, 8), (16384, 64), (16384, 30)):
    g = torch.Generator().manual_seed(7)
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C); mu[np.unique(tr)] = T / N
    P = PoissonModel(mu).rows_for(tr, fs)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for _ in range(5): ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len)
    st = (ctypes.c_longlong * 16)()
    assert raw.mucon_test_vit_stamps(st) == 0
    s = list(st)
    K = T // fs
    base = s[0] if N <= 16 else s[8]
    print(f"   chain wave since kernel start: body entry {s[12]-base}, zero fill done {s[13]-base}, first chunk staged {s[14]-base}")
    print(f"   chain wave: adds {s[10]} cycles ({s[10]/T:.1f} per row), barrier waits {s[11]}; before the chain's first chunk: {(s[1]-s[0] if N <= 16 else s[9]-s[8]) - s[10] - s[11]}")
    if N <= 16:
        print(f"T={T} N={N} one launch: phases 1 + 2 overlapped {s[1]-s[0]} ({(s[1]-s[0])/K:.0f} per column) | finalize {s[5]-s[1]} | labels {s[6]-s[5]} | fence {s[7]-s[6]}   [core cycles]")
    else:
        print(f"T={T} N={N} two launches: chain {s[9]-s[8]} | gap {s[2]-s[9]} | dp setup {s[3]-s[2]} | columns {s[4]-s[3]} ({(s[4]-s[3])/max(K-1,1):.1f} per column) | finalize {s[5]-s[4]} | labels {s[6]-s[5]}   [core cycles]")
