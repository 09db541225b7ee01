"""Host time of the pieces of the end-to-end (batch-1) training step, measured WITHOUT a profiler (cProfile inflates some torch calls 10x): every autograd-Function
forward / backward of MuCon.fused_train_step wrapped in perf_counter stamps (queue drained before the loop so that no call waits for the GPU), plus micro-timings
of the Python idioms inside them.  Usage: python tools/experiments/e2e_host_sections.py"""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mucon_amd import ops, synth  # noqa: E402
from mucon_amd.config import get_cfg_defaults, update_config  # noqa: E402
from mucon_amd.core.datasets import Batch  # noqa: E402
from mucon_amd.mucon.models import create_model  # noqa: E402
from mucon_amd.mucon.trainers import SimpleTrainer  # noqa: E402

dev, T, N, C = "cuda", 2000, 6, 48
cfg = update_config(get_cfg_defaults(), [], [])
torch.manual_seed(0)
model = create_model(cfg, C, 31, 2048).to(dev)
tr = synth.transcript(3, N, C, allow_repeats=False)
trainer = SimpleTrainer(cfg, model, dev)
trainer.on_start_epoch(0)
model.train()
batch = Batch(feats=torch.randn(1, T, 2048), gt_label=torch.from_numpy(synth.segment_labels(4, T, tr)), transcript=torch.from_numpy(tr),
              transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name="s").to(dev)

acc = defaultdict(float)
orig_f, orig_b = ops.run_forward, ops.run_backward


def rf(fn, *a):
    t0 = time.perf_counter()
    r = orig_f(fn, *a)
    acc["fwd " + fn.__name__] += time.perf_counter() - t0
    return r


def rb(fn, ctx, *g):
    t0 = time.perf_counter()
    r = orig_b(fn, ctx, *g)
    acc["bwd " + fn.__name__] += time.perf_counter() - t0
    return r


def step():
    trainer._train_1_batch(0, batch)


for _ in range(20):
    step()
torch.cuda.synchronize()
ops.run_forward, ops.run_backward = rf, rb
n = 8          # few steps: the queue never fills
tot = 0.0
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    tot += time.perf_counter() - t0
torch.cuda.synchronize()
steps = 5 * n
print(f"host enqueue per step: {tot / steps * 1e6:.1f} us (of which inside the wrapped calls: {sum(acc.values()) / steps * 1e6:.1f})")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / steps * 1e6:8.1f} us")
ops.run_forward, ops.run_backward = orig_f, orig_b
# the optimizer step and the zero_grad, alone
for name, fn in (("trainer._zero_grad", trainer._zero_grad),):
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    print(f"  {name:28s} {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us")
# micro: 50 slices + views of a flat device buffer (what _EncoderFn.backward builds per step)
flat = torch.empty(2_000_000, device=dev)
shapes = [p.shape for p in model.ft.parameters()]
t0 = time.perf_counter()
for _ in range(200):
    off, out = 0, []
    for s in shapes:
        k = s.numel()
        out.append(flat[off: off + k].view(s))
        off += k
print(f"  {len(shapes)} slices + views            {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us")
t0 = time.perf_counter()
for _ in range(200):
    x = torch.empty(1000, device=dev)
print(f"  torch.empty (device)         {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us")
