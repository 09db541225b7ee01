"""Evaluation throughput: MuConEvaluator.evaluate() (eval-mode forward, greedy s-head decode, Viterbi, all metrics) on a
Breakfast-shaped synthetic test set kept resident in HBM."""
import sys, os, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.core.datasets import handel_dataset, make_resident, write_synthetic_breakfast
from mucon_amd.mucon.evaluators import MuConEvaluator
from mucon_amd.mucon.models import create_model
from mucon_amd.mucon.trainers import SimpleTrainer
root = tempfile.mkdtemp()
write_synthetic_breakfast(root, n_train=12, n_test=24, t_range=(1500, 2500), n_range=(4, 8))
cfg = update_config(get_cfg_defaults(), [], [["dataset.root", root, "trainer.learning_rate", "0.02"]])
torch.manual_seed(0)
train_db, test_db = make_resident(handel_dataset(cfg, True), "cuda"), make_resident(handel_dataset(cfg, False), "cuda")
model = create_model(cfg, train_db.get_num_classes(), 8, train_db.feat_dim).cuda()   # at most 8 decoding steps
tr = SimpleTrainer(cfg, model, "cuda", train_db)
t0 = time.perf_counter()
for e in range(6): losses = tr.train_epoch(e)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"train: {6 * len(train_db) / (t1 - t0):.1f} videos/s (6 epochs of {len(train_db)} videos, last mean loss {np.mean(losses):.3f})")
with torch.no_grad():   # throughput probe: a barely trained s-head would emit EOS first (the reference's evaluator then fails too)
    model.fs_decoder_transcript[2].bias[train_db.get_num_classes()] = -20.0
ev = MuConEvaluator(cfg, test_db, model, "cuda"); ev.viterbi_mode(True)
for rep in range(2):
    t0 = time.perf_counter()
    try:
        res = ev.evaluate()
    except Exception as e:   # the reference fails the same way when the s-head emits EOS first / no hypothesis survives
        print("evaluate raised", type(e).__name__, e); res = None; break
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"eval: {len(test_db) / (t1 - t0):.1f} videos/s ({(t1 - t0) / len(test_db) * 1e3:.2f} ms per video)")
if res: print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in list(res.items())[:8]})
