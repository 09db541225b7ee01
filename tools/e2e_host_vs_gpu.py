"""The end-to-end (batch-1) training step: how long the HOST needs to enqueue one step (no synchronisation inside the loop) against what the
step takes on the GPU (the same loop, synchronised at the end), for runs short enough not to fill the launch queue (a full queue makes the host wait
for the GPU: both numbers are the GPU's then).  The leg is GPU-bound as long as the first is smaller than the second."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mucon_amd import synth  # noqa: E402
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


def step():
    trainer._train_1_batch(0, batch)


for _ in range(20):
    step()
torch.cuda.synchronize()
for n in (5, 5, 10, 20, 200):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n:3d} steps: host enqueue {1e3 * (t1 - t0) / n:.3f} ms per step; with the GPU drained {1e3 * (t2 - t0) / n:.3f} ms per step")

