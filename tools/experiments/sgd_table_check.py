import os, sys, ctypes
sys.path.insert(0, "/root/repo")
import torch
from mucon_amd import synth
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.core.datasets import Batch
from mucon_amd.mucon.models import create_model
from mucon_amd.mucon.trainers import SimpleTrainer
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
prev = None
names = [n for n, _ in model.named_parameters()]
for i in range(8):
    trainer._train_1_batch(i, batch)
    fs = trainer.fused_step
    tab = fs._plan[3]
    cur = [(tab[k].param, tab[k].grad, tab[k].momentum_buf, tab[k].n, tab[k].group) for k in range(len(tab))]
    if prev is not None:
        diff = [k for k in range(len(cur)) if cur[k] != prev[k]]
        print("step", i, "records changed:", len(diff), [ (k) for k in diff[:10]])
    prev = cur
flat = fs._flat
idx = fs._plan[2]
print([ (k, [n for n, p in model.named_parameters() if p is flat[idx[k]][0]]) for k in (diff[:10] if prev else [])])
