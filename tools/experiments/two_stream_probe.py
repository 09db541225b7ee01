"""ON THE GPU BOX: does running the batch as two half-batches on two streams (videos are independent) shorten the hot-path
step?  Same work as bench.py's step (fwd + bwd + SGD, B=8 x T=4096); timing only."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mucon_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
spec = ops.EncoderSpec()
C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
g = torch.Generator(device=dev).manual_seed(1000)
tape = torch.randn(B, T, spec.in_dim, device=dev, generator=g)
dlogp = torch.randn(B, T, C, device=dev, generator=g) / (B * T)
sgd = ops.FusedClipSGD([params], None, types.SimpleNamespace(param_groups=[{"lr": 0.01, "weight_decay": 0.005, "momentum": 0.0}]))
wc2 = wc.reshape(wc.shape[0], wc.shape[1]) if wc.dim() == 3 else wc


def fwd(tp, i):
    enc, c_enc = ops.run_forward(ops._EncoderFn, tp, spec, True, int(i), *enc_params)
    (_, logp), c_head = ops.run_forward(ops._HeadFn, enc, wc2, bc, int(T), False, True)
    return c_enc, c_head


def bwd(c_enc, c_head, dl):
    d_enc, d_w, d_b = ops.run_backward(ops._HeadFn, c_head, None, dl)[:3]
    g_enc = ops.run_backward(ops._EncoderFn, c_enc, d_enc)[4:]
    return d_w, d_b, g_enc


def step_one(i):
    c = fwd(tape, i)
    d_w, d_b, g_enc = bwd(*c, dlogp)
    wc.grad, bc.grad = d_w.view_as(wc), d_b
    for p_, g_ in zip(enc_params, g_enc):
        p_.grad = g_
    sgd.step()


def make_split(nparts):
    streams = [torch.cuda.Stream() for _ in range(nparts)]
    n = B // nparts
    tapes = [tape[k * n:(k + 1) * n] for k in range(nparts)]
    dls = [dlogp[k * n:(k + 1) * n].contiguous() for k in range(nparts)]

    def step(i):
        main = torch.cuda.current_stream()
        cs, outs = [], []
        for s in streams:
            s.wait_stream(main)
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                cs.append(fwd(tapes[k], i))
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs.append(bwd(*cs[k], dls[k]))
        for s in streams:
            main.wait_stream(s)
        d_w = sum(o[0] for o in outs)
        d_b = sum(o[1] for o in outs)
        flats = [ops_flat(o[2]) for o in outs]
        tot = flats[0]
        for f in flats[1:]:
            tot = tot + f
        wc.grad, bc.grad = d_w.view_as(wc), d_b
        off = 0
        for p_, g_ in zip(enc_params, outs[0][2]):
            p_.grad = g_
        outs[0][2][0].untyped_storage()   # grads of part 0 are views of flats[0]; overwrite it with the total
        flats[0].copy_(tot)
        sgd.step()
    return step


def ops_flat(grads):
    st = grads[0].untyped_storage()
    return torch.empty(0, dtype=torch.float32, device=dev).set_(st)


def timeit(fn, n=200, w=30):
    for i in range(w):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(w + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    print("one stream, B=8          : %.4f ms" % timeit(step_one))
    print("two streams, 2 x 4 videos: %.4f ms" % timeit(make_split(2)))
    print("four streams, 4 x 2      : %.4f ms" % timeit(make_split(4)))
