"""What SGD on synthetic noise does to the hot path's operands (bench.py --keep-drift): the share of exact zeros in the saved ReLU outputs and the
RMS of the activations / gradients of the random-init network and after 200 / 800 steps of the bench's own step (lr 0.01, weight decay 0.005, a fixed
random dL/dlogp)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mucon_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
spec = ops.EncoderSpec()
C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
g = torch.Generator(device=dev).manual_seed(1000)
tape = torch.randn(B, T, spec.in_dim, device=dev, generator=g)
dlogp = torch.randn(B, T, C, device=dev, generator=g) / (B * T)
wc2 = wc.reshape(wc.shape[0], wc.shape[1])


def report(tag):
    with torch.enable_grad():
        ps = [p.detach().requires_grad_(True) for p in enc_params]
        enc = ops.encoder_forward(tape, ps, spec, training=True, seed=1)
        zeros, rms = [], []
        for l in range(len(spec.stages)):
            h = ops.encoder_saved(enc, "h", l)
            zeros.append(float((h == 0).float().mean()))
            rms.append(float(h.pow(2).mean().sqrt()))
        x0 = ops.encoder_saved(enc, "x", 0)
    print(f"{tag}: zeros in the layers' ReLU outputs {' '.join(f'{z:.2f}' for z in zeros)} | their RMS {' '.join(f'{r:.2g}' for r in rms)} | first_conv output zeros {float((x0 == 0).float().mean()):.2f} rms {float(x0.pow(2).mean().sqrt()):.2g} | encoder output rms {float(enc.detach().pow(2).mean().sqrt()):.2g}")


import types
sgd = ops.FusedClipSGD([params], None, types.SimpleNamespace(param_groups=[{"lr": 0.01, "weight_decay": 0.005, "momentum": 0.0}]))


def step(i):
    enc, c_enc = ops.run_forward(ops._EncoderFn, tape, spec, True, int(i), *enc_params)
    (_, logp), c_head = ops.run_forward(ops._HeadFn, enc, wc2, bc, int(T), False, True)
    d_enc, d_w, d_b = ops.run_backward(ops._HeadFn, c_head, None, dlogp)[:3]
    wc.grad, bc.grad = d_w.view_as(wc), d_b
    g_enc = ops.run_backward(ops._EncoderFn, c_enc, d_enc)[4:]
    for p_, g_ in zip(enc_params, g_enc):
        p_.grad = g_
    sgd.step()


report("random init ")
n = 0
for upto in (200, 800, 2000):
    while n < upto:
        step(n)
        n += 1
    torch.cuda.synchronize()
    report(f"{upto:5d} steps  ")
