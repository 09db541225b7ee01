"""The one-video training step enqueued without synchronisation: host time per window of 10 steps, to see where a host that runs ahead of the
GPU starts to pay (launch queue depth) -- and the same with a synchronisation every 40 steps (what bench.py's end_to_end leg does)."""
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
for _ in range(20):
    trainer._train_1_batch(0, batch)
torch.cuda.synchronize()
for sync_every in (0, 40):
    t_all = time.perf_counter()
    marks = []
    for w in range(80):
        t0 = time.perf_counter()
        for i in range(10):
            trainer._train_1_batch(0, batch)
        marks.append((time.perf_counter() - t0) / 10 * 1e3)
        if sync_every and (w + 1) * 10 % sync_every == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t_all) / 800 * 1e3
    print(f"sync every {sync_every or 'never'}: {total:.3f} ms per step overall; host ms per step by window of 10:", " ".join(f"{m:.1f}" for m in marks))
    print("   allocator: reserved %.0f MB, allocated %.0f MB" % (torch.cuda.memory_reserved() / 2 ** 20, torch.cuda.memory_allocated() / 2 ** 20))
