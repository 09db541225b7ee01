"""ON THE GPU BOX, CLK_STAMP build (MUCON_HIPCC_FLAGS=-DCLK_STAMP=1): the life of every persistent workgroup of the static-runs weight-gradient
launch (ts_runs_kernel, csrc/gemm_tn_split.hpp) at the bench shape -- s_memrealtime at entry and exit (10-ns ticks) -- in groups of 16 workgroups
along the line of work (first_conv's columns first, then the residual layers from the fine levels down): where the static shares are too long.
    python3 tools/ts_runs_times.py [KNOB=VALUE ...]        e.g. MUCON_TS_COSTS=66,77,99,128"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from mucon_amd import _lib, ops

lib = _lib.load()
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    _lib.set_knob(k, v)
dev = torch.device("cuda", 0)
spec = ops.EncoderSpec()
C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
tape = torch.randn(B, T, 2048, device=dev)
dlogp = torch.randn(B, T, C, device=dev) / (B * T)
for i in range(40):
    for p in params:
        p.grad = None
    enc = ops.encoder_forward(tape, enc_params, spec, training=True, seed=i)
    _, logp = ops.head_forward(enc, wc, bc, T, want_logits=False)
    logp.backward(dlogp)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (2 * 4096))()
n = lib.mucon_test_read_clock(2, buf, 2 * 4096)
if n <= 0:
    raise SystemExit("not a CLK_STAMP build")
t0 = min(buf[2 * i] for i in range(n) if buf[2 * i] > 0)
ent = [(buf[2 * i] - t0) / 100.0 for i in range(n)]
ext = [(buf[2 * i + 1] - t0) / 100.0 for i in range(n)]
life = [b - a for a, b in zip(ent, ext)]
print(f"{n} workgroups ({' '.join(sys.argv[1:])}); life per workgroup (us) in groups of 16: mean / min / max   [latest exit of the group]")
for g0 in range(0, n, 16):
    g = life[g0:g0 + 16]
    print(f"  workgroups {g0:4d}-{g0 + len(g) - 1:4d}: {sum(g) / len(g):7.1f} / {min(g):7.1f} / {max(g):7.1f}   [{max(ext[g0:g0 + 16]):7.1f}]")
print(f"first entry -> last exit {max(ext):.1f} us; latest entry {max(ent):.1f}; sum of lives / {n} = {sum(life) / n:.1f} us; longest {max(life):.1f}, shortest {min(life):.1f}")

# ---- what each share holds (the schedule of csrc/gemm_tn_split.hpp::ts_make_schedule, restated for the encoder's jobs) and a least-squares fit of
# ---- life = a * staggered tiles + b * lock-step tiles + c * two-image tiles + d * runs + e * columns + f: the cost units the schedule should use
import numpy as np

cost = [69, 74, 95, 109]
for kv in sys.argv[1:]:
    if kv.startswith("MUCON_TS_COSTS="):
        cost = [int(x) for x in kv.split("=", 1)[1].split(",")]
Tl = [T]
for i in range(len(spec.stages)):
    Tl.append(Tl[-1] // 2 if (spec.pooling and i in spec.pooling_layers) else Tl[-1])
jobs = [("first_conv", T, 2048 // 128, False, False)] + [(f"layer {i}", Tl[i], 4, True, False) for i in range(len(spec.stages))] + [("last_conv", Tl[-1], 1, False, True)]
cols = []   # (kind, nvid, tv, tcost, vcost, pos0)
pos = 0
for name, rows, nkc, dual, x0_act in jobs:
    flat = not dual
    nvid, tv = (1, (B * rows + 31) // 32) if flat else (B, (rows + 31) // 32)
    for k in range((nkc + 1) // 2):
        two = dual and 2 * k + 1 == nkc - 1
        kind = 2 if (two or x0_act) else (0 if (flat and tv >= 8) else 1)
        tc = cost[kind]
        cols.append((kind, nvid, tv, tc, cost[3] + tv * tc, pos, name))
        pos += nvid * (cost[3] + tv * tc)
W, G = pos, n
S = (W + G - 1) // G


def qmap(c, off):
    kind, nvid, tv, tc, vc, p0, _ = c
    b, rem = divmod(off, vc)
    t = 0 if rem <= cost[3] else (rem - cost[3]) // tc
    return b * tv + min(t, tv)


A = np.zeros((G, 6))
for w in range(G):
    lo, hi = min(w * S, W), min(w * S + S, W)
    for ci, c in enumerate(cols):
        kind, nvid, tv, tc, vc, p0, _ = c
        end = cols[ci + 1][5] if ci + 1 < len(cols) else W
        if p0 >= hi or end <= lo:
            continue
        q0 = qmap(c, lo - p0) if lo > p0 else 0
        q1 = nvid * tv if hi >= end else qmap(c, hi - p0)
        A[w, 4] += 1
        q = q0
        while q < q1:
            b, t0 = divmod(q, tv)
            t1 = min(tv, t0 + (q1 - q))
            A[w, kind] += t1 - t0
            A[w, 3] += 1
            q += t1 - t0
A[:, 5] = 1
y = np.array(life)
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print("fit (us): staggered tile %.3f  lock-step tile %.3f  two-image tile %.3f  per run %.2f  per column %.2f  constant %.2f   (x 32 = cost units: %s)"
      % (*coef, " ".join(f"{c * 32:.0f}" for c in coef[:5])))
print("residuals (us): rms %.2f, worst %.1f at workgroup %d" % (float(np.sqrt(np.mean((pred - y) ** 2))), float(np.abs(pred - y).max()), int(np.abs(pred - y).argmax())))
print("last 12 shares: " + "  ".join(f"[{int(A[w, 0])}/{int(A[w, 1])}/{int(A[w, 2])} tiles, {int(A[w, 3])} runs, {int(A[w, 4])} cols: {life[w]:.0f}]" for w in range(G - 12, G)))
