"""ON THE GPU BOX, with a timing build of the library (MUCON_HIPCC_FLAGS=-DCS_STAMP=1 python3 -m mucon_amd.build --force): the phases of every
cs_kernel launch of ONE hot-path step at the bench shape (B = 8 x T = 4096), from in-kernel s_memtime stamps of the first workgroup and of a
workgroup in the middle of the grid (gemm_coarse_split.hpp).  Cycles are shader cycles; ns from the s_memrealtime pair of the same wave."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from mucon_amd import _lib, ops

dev = torch.device("cuda", 0)
lib = _lib.load()
spec = ops.EncoderSpec()
C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
tape = torch.randn(B, T, 2048, device=dev)
dlogp = torch.randn(B, T, C, device=dev) / (B * T)
stamps = (ctypes.c_longlong * (64 * 2 * 4 * 12))()
info = (ctypes.c_int32 * (64 * 8))()


def one_step(i):
    for p in params:
        p.grad = None
    enc = ops.encoder_forward(tape, enc_params, spec, training=True, seed=i)
    _, logp = ops.head_forward(enc, wc, bc, T, want_logits=False)
    logp.backward(dlogp)


for i in range(30):
    one_step(i)
torch.cuda.synchronize()
lib.mucon_test_read_cs_stamps(stamps, info, 64)      # (resets the slot counter)
one_step(99)
n = lib.mucon_test_read_cs_stamps(stamps, info, 64)
if n <= 0:
    raise SystemExit("not a timing build (MUCON_HIPCC_FLAGS=-DCS_STAMP=1)")
PH = ["loads issued", "rows in + split", "stage-1 MFMAs", "exchange 1", "epilogue 1 + split", "stage-2 MFMAs", "exchange 2", "epilogue 2"]
print(f"{n} cs_kernel launches of one step (B = {B} x T = {T}); cycles per phase, per wave; workgroup 'first' = (0, 0), 'middle' = (gx / 2, gy / 2)")
tot = {}
for s in range(n):
    bwd, pool, taps, one, rb, gx, gy, rows = [info[s * 8 + k] for k in range(8)]
    print(f"\nlaunch {s}: cs_kernel<BWD={bwd}, POOL={pool}, TAPS={taps}, ONE={one}, RB={rb}>  grid {gx} x {gy} = {gx * gy} workgroups, {rows} rows per video")
    for which, wname in ((0, "first "), (1, "middle")):
        for w in range(4):
            o = [stamps[((s * 2 + which) * 4 + w) * 12 + k] for k in range(12)]
            if o[0] == 0:
                continue
            d = [o[k + 1] - o[k] for k in range(8)]
            total = o[8] - o[0]
            ns = (o[10] - o[9]) * 10.0
            ghz = total / ns if ns > 0 else 0.0
            print(f"   {wname} wave {w}: total {total:6d} cyc = {ns / 1e3:5.2f} us ({ghz:.2f} GHz) | " + " | ".join(f"{PH[k]} {d[k]:5d}" for k in range(8)))
            if which == 1 and w == 0:
                key = (bwd, pool, taps, one, rb, gx * gy)
                tot.setdefault(key, []).append((total, ns, d))
print("\nsummary (middle workgroup, wave 0), by kernel variant and grid:")
for key, v in sorted(tot.items()):
    m = len(v)
    avg = [sum(x[2][k] for x in v) / m for k in range(8)]
    print(f"  BWD={key[0]} POOL={key[1]} TAPS={key[2]} ONE={key[3]} RB={key[4]} {key[5]:4d} WGs x{m}: {sum(x[0] for x in v) / m:7.0f} cyc {sum(x[1] for x in v) / m / 1e3:5.2f} us | "
          + " | ".join(f"{PH[k]} {avg[k]:5.0f}" for k in range(8)))
