"""ON THE GPU BOX, with a timing build of the library (python -m mucon_amd.build --force under MUCON_HIPCC_FLAGS=-DFS_STAMP=1):
cycles per phase of fs_kernel's block 0 (gemm_fused_split.hpp), one line per kernel variant and wave, for the hot-path step."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mucon_amd import _lib, ops
dev = torch.device("cuda", 0)
spec = ops.EncoderSpec(); C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
tape = torch.randn(B, T, 2048, device=dev); dlogp = torch.randn(B, T, C, device=dev) / (B * T)
for i in range(6):
    for p in params: p.grad = None
    enc = ops.encoder_forward(tape, enc_params, spec, training=True, seed=i)
    _, logp = ops.head_forward(enc, wc, bc, T, want_logits=False)
    logp.backward(dlogp)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (64 * 8 * 8))()
_lib.check(_lib.load().mucon_test_read_stamps(buf, 64 * 8 * 8), "read_stamps")
names = ["prologue", "step0 x6", "step1 x6", "barrier x6", "epilogue1", "stage2", "epilogue2"]
for v in range(64):
    rows = [[buf[(v * 8 + w) * 8 + k] for k in range(8)] for w in range(8)]
    if not any(any(r) for r in rows):
        continue
    bwd, pool, one, nw8 = v // 32, (v // 4) % 8, (v // 2) % 2, v % 2
    print(f"fs_kernel<BWD={bwd}, POOL={pool}, ONE={one}, NW={8 if nw8 else 4}>: cycles of block 0 per wave (total | " + " | ".join(names) + ")")
    for w, r in enumerate(rows):
        if any(r):
            print(f"   wave {w}: {sum(r):7d} | " + " | ".join(f"{x:6d}" for x in r[:7]))
