"""Host-side cost of the end-to-end training step: issue time (no sync) vs total, and a cProfile of the step."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mucon_amd import synth
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.core.datasets import Batch
from mucon_amd.mucon.models import create_model
from mucon_amd.mucon.trainers import SimpleTrainer
dev = "cuda"; T, N, C = 2000, 6, 48
cfg = update_config(get_cfg_defaults(), [], [])
torch.manual_seed(0)
model = create_model(cfg, C, 31, 2048).to(dev)
trainer = SimpleTrainer(cfg, model, dev); trainer.on_start_epoch(0); model.train()
tr = synth.transcript(3, N, C, allow_repeats=False)
batch = Batch(feats=torch.randn(1, T, 2048), gt_label=torch.from_numpy(synth.segment_labels(4, T, tr)), transcript=torch.from_numpy(tr),
              transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name="s").to(dev)
for i in range(10): trainer._train_1_batch(i, batch)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for i in range(K): trainer._train_1_batch(10 + i, batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"issue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(K): trainer._train_1_batch(100 + i, batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
