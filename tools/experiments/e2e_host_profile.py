"""cProfile of the host side of the end-to-end (batch-1) training step (MuCon.fused_train_step + the fused clip / SGD step): where the ~0.7 ms
of enqueue time per step go (the leg is host-bound or at parity: tools/e2e_host_vs_gpu.py).  Usage: python tools/e2e_host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
n = 40
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr).stats
rows = [(f"{os.path.basename(k[0])}:{k[1]}({k[2]})", v[1], v[2], v[3]) for k, v in st.items()]
for title, col in (("cumulative", 3), ("own", 2)):
    print(f"-- top 45 by {title} time, us per step (under cProfile: every Python call costs ~1 us extra)")
    for name, nc, tt, ct in sorted(rows, key=lambda r: -r[col])[:45]:
        print(f"{nc / n:8.1f} calls  own {tt / n * 1e6:8.1f}  cum {ct / n * 1e6:8.1f}  {name}")
