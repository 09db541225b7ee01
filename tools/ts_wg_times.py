"""ON THE GPU BOX, CLK_STAMP build (MUCON_HIPCC_FLAGS=-DCLK_STAMP=1): how long every workgroup of the batched weight-gradient launch spends in its
tile loop (s_memrealtime ticks of 10 ns), listed per run of 16 consecutive blocks -- the launch lays its jobs out longest first: first_conv
(128 workgroups at the bench shape), then the residual layers from the fine levels down (32 workgroups each: even blocks the plain half, odd
blocks the half that stages two gradient images and replays the dropout mask).   python3 tools/ts_wg_times.py [MUCON_TS_STAGGER value]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from mucon_amd import _lib, ops

lib = _lib.load()
if len(sys.argv) > 1:
    _lib.set_knob("MUCON_TS_STAGGER", sys.argv[1])
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
n = lib.mucon_test_read_clock(1, buf, 2 * 4096)
if n <= 0:
    raise SystemExit("not a CLK_STAMP build")
us = [buf[2 * i + 1] / 100.0 for i in range(n)]
ghz = [buf[2 * i] / buf[2 * i + 1] * 0.1 if buf[2 * i + 1] else 0.0 for i in range(n)]
print(f"{n} workgroups; tile-loop time per workgroup (us), groups of 16 blocks: even blocks / odd blocks of the group (mean), max, clock")
for g0 in range(0, n, 16):
    grp = [(i, us[i]) for i in range(g0, min(g0 + 16, n)) if us[i] > 0]
    if not grp:
        continue
    ev = [u for i, u in grp if i % 2 == 0]
    od = [u for i, u in grp if i % 2 == 1]
    cl = [ghz[i] for i, _ in grp]
    print(f"  blocks {g0:4d}-{g0 + 15:4d}: even {sum(ev) / max(len(ev), 1):7.1f}  odd {sum(od) / max(len(od), 1):7.1f}  max {max(u for _, u in grp):7.1f}  clock {sum(cl) / len(cl):.2f} GHz")
print(f"longest workgroup {max(us):.1f} us; sum over workgroups / 256 CUs = {sum(us) / 256:.1f} us")
buf2 = (ctypes.c_longlong * (2 * 4096))()
if lib.mucon_test_read_clock(2, buf2, 2 * 4096) > 0:
    t0 = min(buf2[2 * i] for i in range(n) if buf2[2 * i] > 0)
    life = [(buf2[2 * i + 1] - buf2[2 * i]) / 100.0 for i in range(n)]
    ovh = [life[i] - us[i] for i in range(n) if us[i] > 0]
    end = max(buf2[2 * i + 1] for i in range(n))
    print(f"workgroup life minus tile loop (prologue + slab epilogue): mean {sum(ovh) / len(ovh):.2f} us, min {min(ovh):.2f}, max {max(ovh):.2f}; "
          f"first entry to last exit {(end - t0) / 100.0:.1f} us; sum of lives / 256 CUs = {sum(life) / 256:.1f} us")
    starts = sorted((buf2[2 * i] - t0) / 100.0 for i in range(n))
    print("entry times (us after the first): " + " ".join(f"{starts[k]:.0f}" for k in range(0, n, max(n // 24, 1))))

# (r5) where a workgroup's life outside its tile loop goes: job lookup, prologue (first loads, image of tile 0), epilogue (slab write-out)
b3 = (ctypes.c_longlong * (2 * 4096))()
b4 = (ctypes.c_longlong * (2 * 4096))()
if lib.mucon_test_read_clock(3, b3, 2 * 4096) > 0 and lib.mucon_test_read_clock(4, b4, 2 * 4096) > 0:
    print("per 16 blocks (us): job lookup | rest of the prologue | epilogue   [entry time of the group after the launch's first]")
    for g0 in range(0, n, 16):
        rows = [i for i in range(g0, min(g0 + 16, n)) if us[i] > 0 and b3[2 * i] > 0]
        if not rows:
            continue
        look = [(b3[2 * i] - buf2[2 * i]) / 100.0 for i in rows]
        pro = [(b3[2 * i + 1] - b3[2 * i]) / 100.0 for i in rows]
        epi = [(buf2[2 * i + 1] - b4[2 * i]) / 100.0 for i in rows]
        ent = [(buf2[2 * i] - t0) / 100.0 for i in rows]
        print(f"  blocks {g0:4d}-{g0 + 15:4d}: {sum(look) / len(look):5.2f} | {sum(pro) / len(pro):5.2f} | {sum(epi) / len(epi):5.2f} (max {max(epi):5.2f})   [{min(ent):6.1f} .. {max(ent):6.1f}]")
    # how long a freed CU waits for its next workgroup: the k-th workgroup that enters late follows the k-th exit
    exits = sorted((buf2[2 * i + 1] - t0) / 100.0 for i in range(n) if buf2[2 * i + 1] > 0)
    late = sorted((buf2[2 * i] - t0) / 100.0 for i in range(n) if buf2[2 * i] - t0 > 100)
    gaps = [e - x for e, x in zip(late, exits)]
    if gaps:
        gs = sorted(gaps)
        print(f"exit of a workgroup -> entry of the next one on the freed CU (k-th late entry minus k-th exit), {len(gaps)} pairs: "
              f"median {gs[len(gs) // 2]:.2f} us, p10 {gs[len(gs) // 10]:.2f}, p90 {gs[len(gs) * 9 // 10]:.2f}, min {gs[0]:.2f}, max {gs[-1]:.2f}")
