"""ON THE GPU BOX, CLK_STAMP build (MUCON_HIPCC_FLAGS=-DCLK_STAMP=1): the life of every share of the static-runs weight-gradient
launch (ts_runs_kernel, csrc/gemm_tn_split.hpp) at the bench shape -- s_memrealtime at entry and exit (10-ns ticks) -- in groups of 16 workgroups
along the line of work (first_conv's columns first, then the residual layers from the fine levels down): where the static shares are too long.
    python3 tools/ts_runs_times.py        (prints the least-squares fit of the cost units to put into csrc/mucon_hip.hip: g_ts_cost)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from mucon_amd import _lib, ops

lib = _lib.load()
for kv in sys.argv[1:]:           # (run-time knobs of the library, e.g. MUCON_TS_RUNS=...; the cost units are compile-time constants since r6)
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

# ---- what each share holds (the line of csrc/gemm_tn_split.hpp::ts_make_schedule, restated for the encoder's jobs) and a least-squares fit of
# ---- life = a * staggered tiles + b * lock-step tiles + c * two-image tiles + d * runs + e * units + f: the cost units the schedule should use
import numpy as np

cost = [69, 74, 95, 109]
group_rows = 1 << 30
for kv in sys.argv[1:]:
    if kv.startswith("MUCON_TS_COSTS="):
        cost = [int(x) for x in kv.split("=", 1)[1].split(",")]
    if kv.startswith("MUCON_TS_GROUP_ROWS="):
        group_rows = int(kv.split("=", 1)[1])
ovh = cost[3]
Tl = [T]
for i in range(len(spec.stages)):
    Tl.append(Tl[-1] // 2 if (spec.pooling and i in spec.pooling_layers) else Tl[-1])
jobs = [("first_conv", T, 2048 // 128, False, False)] + [(f"layer {i}", Tl[i], 4, True, False) for i in range(len(spec.stages))] + [("last_conv", Tl[-1], 1, False, True)]
G = n
L = []
tiles = work = 0
for name, rows, nkc, dual, x0_act in jobs:
    flat = not dual
    ncols = (nkc + 1) // 2
    tv = ((B if flat else 1) * rows + 31) // 32
    stag = flat and tv >= 8 and not x0_act
    tc = cost[2] if x0_act else (cost[0] if stag else cost[1])
    tcl = cost[2] if dual else tc
    nv = 1 if flat else B
    tiles += nv * tv * ncols
    work += nv * tv * ((ncols - 1) * tc + tcl)
    L.append(dict(name=name, flat=flat, ncols=ncols, tiles=tv, tc=tc, tcl=tcl, kind=(2 if x0_act else 0 if stag else 1), rows=rows))
aligned_shares = aligned_work = 0
J0 = L[0]
J0["aligned"] = False
if J0["flat"] and J0["ncols"] > 1 and G >= 64:
    w_i = J0["tiles"] * J0["ncols"] * J0["tc"]
    panels = (w_i * G // max(work, 1) + J0["ncols"] // 2) // J0["ncols"]
    panels = min(max(panels, 1), (G - 1) // J0["ncols"])
    if panels >= 1 and J0["tiles"] // panels >= 8:
        J0.update(aligned=True, ngroups=panels, gt=(J0["tiles"] + panels - 1) // panels, vg=1)
        aligned_shares, aligned_work = panels * J0["ncols"], w_i
s_est = max(1, (work - aligned_work) // max(1, G - aligned_shares))
rest = 0
for J in L:
    if not J.get("aligned"):
        gt = max(8, min(J["tiles"], (s_est - ovh) // J["tc"] if s_est > ovh else 8)) if J["flat"] else J["tiles"]
        vg = 1 if J["flat"] else (1 if (J["rows"] >= group_rows or B == 1) else B)
        ng = (J["tiles"] + gt - 1) // gt if J["flat"] else B // vg
        rest += ng * ((J["ncols"] - 1) * vg * (ovh + gt * J["tc"]) + vg * (ovh + gt * J["tcl"]))
S_al = max(1, (rest + (G - aligned_shares) - 1) // (G - aligned_shares)) if aligned_shares else 0
pos = 0
for J in L:
    if J.get("aligned"):
        J["ucost"] = J["ucost_last"] = S_al
    elif J["flat"]:
        J["gt"] = max(8, min(J["tiles"], (s_est - ovh) // J["tc"] if s_est > ovh else 8))
        J["ngroups"] = (J["tiles"] + J["gt"] - 1) // J["gt"]
        J["vg"] = 1
        J["ucost"], J["ucost_last"] = ovh + J["gt"] * J["tc"], ovh + J["gt"] * J["tcl"]
    else:
        J["gt"] = J["tiles"]
        J["vg"] = 1 if (J["rows"] >= group_rows or B == 1) else B
        J["ngroups"] = B // J["vg"]
        J["ucost"], J["ucost_last"] = J["vg"] * (ovh + J["gt"] * J["tc"]), J["vg"] * (ovh + J["gt"] * J["tcl"])
    J["gcost"] = (J["ncols"] - 1) * J["ucost"] + J["ucost_last"]
    J["pos0"] = pos
    pos += J["ngroups"] * J["gcost"]
W = pos
S = S_al if aligned_shares else (W + G - 1) // G


def qmap(vcost, tc, tv, off):
    b, rem = divmod(off, vcost)
    t = 0 if rem <= ovh else (rem - ovh) // tc
    return b * tv + min(t, tv)


A = np.zeros((G, 6))
for sh in range(G):
    lo, hi = min(sh * S, W), min(sh * S + S, W)
    for J in L:
        for g in range(J["ngroups"]):
            for c in range(J["ncols"]):
                u0 = J["pos0"] + g * J["gcost"] + c * J["ucost"]
                last = c + 1 == J["ncols"]
                u1 = u0 + (J["ucost_last"] if last else J["ucost"])
                if u0 >= hi or u1 <= lo:
                    continue
                tc = J["tcl"] if last else J["tc"]
                vcost = ovh + J["gt"] * tc
                nvid = 1 if J["flat"] else J["vg"]
                q0 = qmap(vcost, tc, J["gt"], lo - u0) if (lo > u0 and not J.get("aligned")) else 0
                q1 = nvid * J["gt"] if (hi >= u1 or J.get("aligned")) else qmap(vcost, tc, J["gt"], hi - u0)
                kind = 2 if (last and J["tcl"] != J["tc"]) else J["kind"]
                A[sh, 4] += 1
                if J["flat"]:
                    t0, t1 = g * J["gt"] + q0, min(g * J["gt"] + q1, J["tiles"])
                    if t1 > t0:
                        A[sh, kind] += t1 - t0
                        A[sh, 3] += 1
                else:
                    q = q0
                    while q < q1:
                        b, t0 = divmod(q, J["gt"])
                        t1 = min(J["gt"], t0 + (q1 - q))
                        A[sh, kind] += t1 - t0
                        A[sh, 3] += 1
                        q += t1 - t0
A[:, 5] = 1
y = np.array(life)
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print("line: W %d, S %d, %d shares; units per job: %s" % (W, S, G, " ".join(f"{J['name']}:{J['ngroups']}x{J['ncols']}" for J in L)))
print("fit (us): staggered tile %.3f  lock-step tile %.3f  two-image tile %.3f  per run %.2f  per unit %.2f  constant %.2f   (x 32 = cost units: %s)"
      % (*coef, " ".join(f"{c * 32:.0f}" for c in coef[:5])))
print("residuals (us): rms %.2f, worst %.1f at share %d" % (float(np.sqrt(np.mean((pred - y) ** 2))), float(np.abs(pred - y).max()), int(np.abs(pred - y).argmax())))
print("slabs written (visits): %d" % int(A[:, 4].sum()))
print("last 12 shares: " + "  ".join(f"[{int(A[w, 0])}/{int(A[w, 1])}/{int(A[w, 2])} tiles, {int(A[w, 3])} runs, {int(A[w, 4])} units: {life[w]:.0f}]" for w in range(G - 12, G)))
