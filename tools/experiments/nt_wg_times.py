"""ON THE GPU BOX, CLK_STAMP build (MUCON_HIPCC_FLAGS=-DCLK_STAMP=1): main-loop time of every workgroup of first_conv's forward launch (slot 0 of
mucon_test_read_clock: 256 workgroups at the bench shape, one per CU) -- is the launch waiting for stragglers?  Distribution, per-XCD means (workgroup b runs
on XCD b % 8), in-kernel clock."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import bench
from mucon_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda", 0)
spec = ops.EncoderSpec()
C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
tapes = [torch.randn(B, T, 2048, device=dev) for _ in range(4)]
dlogp = torch.randn(B, T, C, device=dev) / (B * T)
for i in range(40):
    for p in params:
        p.grad = None
    enc = ops.encoder_forward(tapes[i % 4], enc_params, spec, training=True, seed=i)
    _, logp = ops.head_forward(enc, wc, bc, T, want_logits=False)
    logp.backward(dlogp)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (2 * 4096))()
n = lib.mucon_test_read_clock(0, buf, 2 * 4096)
if n <= 0:
    raise SystemExit("not a CLK_STAMP build")
us = sorted(buf[2 * i + 1] / 100.0 for i in range(n))
print(f"{n} workgroups; main-loop time (us): min {us[0]:.1f}  p10 {us[n // 10]:.1f}  median {us[n // 2]:.1f}  p90 {us[n * 9 // 10]:.1f}  max {us[-1]:.1f}")
for x in range(8):
    v = [buf[2 * i + 1] / 100.0 for i in range(n) if i % 8 == x]
    g = [buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(n) if i % 8 == x and buf[2 * i + 1]]
    print(f"  XCD {x}: mean {sum(v) / len(v):6.1f}  max {max(v):6.1f}  clock {sum(g) / len(g):.2f} GHz")
for b in range(8):
    v = [buf[2 * i + 1] / 100.0 for i in range(n) if i // 32 == b]
    print(f"  video {b}: mean {sum(v) / len(v):6.1f}  max {max(v):6.1f}")
