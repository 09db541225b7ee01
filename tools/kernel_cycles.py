"""ON THE GPU BOX: box-independent figures of the hot-path step -- shader CYCLES (s_memtime) counted inside the four kernel families, read from the
stamped library variant (mucon_amd/build.py: build_stamp; loaded here through MUCON_LIB_VARIANT=stamp, never by the product path) in one extra
step at the bench shape (B = 8 x T = 4096).  Boxes of the pool differ by 8 % in sustained clock; a cycle count only moves with the code.
Prints ONE JSON object (bench.py runs this as a child and puts it in its line as `kernel_cycles`):
  ts_runs_share       median over the 256 shares of the static-runs weight-gradient launch: cycles from a share's entry to its exit
  nt_split16_loop     median over the workgroups of first_conv forward: cycles of the k-tile loop
  fs_kernel_fwd       fs_kernel<BWD=0, POOL=0> (layer 0 forward): cycles of block 0, wave 0 (sum of its phases)
  cs_kernel_T8_fwd    cs_kernel<BWD=0, POOL=0, TAPS=3, RB=1> (a T/8-level forward layer): cycles of the middle workgroup, wave 0"""
import ctypes
import json
import os
import sys

os.environ["MUCON_LIB_VARIANT"] = "stamp"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from mucon_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda", 0)
spec = ops.EncoderSpec()
C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
tape = torch.randn(B, T, 2048, device=dev)
dlogp = torch.randn(B, T, C, device=dev) / (B * T)


def one_step(i):
    for p in params:
        p.grad = None
    enc = ops.encoder_forward(tape, enc_params, spec, training=True, seed=i)
    _, logp = ops.head_forward(enc, wc, bc, T, want_logits=False)
    logp.backward(dlogp)


def med(xs):
    xs = sorted(xs)
    return int(xs[len(xs) // 2]) if xs else None


for i in range(20):
    one_step(i)
torch.cuda.synchronize()
cs_st = (ctypes.c_longlong * (64 * 2 * 4 * 12))()
cs_info = (ctypes.c_int32 * (64 * 8))()
lib.mucon_test_read_cs_stamps(cs_st, cs_info, 64)      # (resets the slot counter)
one_step(99)
torch.cuda.synchronize()
out = {"shape": f"B={B} x T={T}", "unit": "shader cycles (s_memtime)"}
buf = (ctypes.c_longlong * (2 * 4096))()
n = lib.mucon_test_read_clock(3, buf, 2 * 4096)
out["ts_runs_share"] = med([buf[2 * i] for i in range(max(n, 0)) if buf[2 * i + 1] > 0])
n = lib.mucon_test_read_clock(0, buf, 2 * 4096)
out["nt_split16_loop"] = med([buf[2 * i] for i in range(max(n, 0)) if buf[2 * i + 1] > 0])
fs = (ctypes.c_longlong * (64 * 8 * 8))()
if lib.mucon_test_read_stamps(fs, 64 * 8 * 8) == 0:
    v = 1          # variant index: BWD=0, POOL=0, ONE=0, NW=8 (tools/fs_stamps.py: v = bwd * 32 + pool * 4 + one * 2 + nw8)
    out["fs_kernel_fwd"] = int(sum(fs[(v * 8 + 0) * 8 + k] for k in range(8))) or None
ncs = lib.mucon_test_read_cs_stamps(cs_st, cs_info, 64)
t8 = []
for s in range(max(ncs, 0)):
    bwd, pool, taps, one, rb, gx, gy, rows = [cs_info[s * 8 + k] for k in range(8)]
    if (bwd, pool, taps, one, rb) == (0, 0, 3, 0, 1) and rows == T // 8:
        o = [cs_st[((s * 2 + 1) * 4 + 0) * 12 + k] for k in range(12)]
        t8.append(o[8] - o[0])
out["cs_kernel_T8_fwd"] = med(t8)
print(json.dumps(out))
