"""ON THE GPU BOX: host time to ENQUEUE one evaluation forward (MuCon.forward_deferred, no synchronisation), per part."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cProfile, pstats
from mucon_amd import synth
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.core.datasets import Batch
from mucon_amd.mucon.models import create_model
dev, T, C = "cuda:0", 2000, 48
cfg = update_config(get_cfg_defaults(), [], [])
model = create_model(cfg, C, 8, 2048).to(dev).eval()
model.set_teacher_forcing(False)
tr = synth.transcript(3, 6, C, allow_repeats=False)
batch = Batch(feats=torch.randn(1, T, 2048), gt_label=torch.from_numpy(synth.segment_labels(4, T, tr)), transcript=torch.from_numpy(tr),
              transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name="v").to(dev)
with torch.no_grad():
    for _ in range(20): model.forward_deferred(batch)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): model.forward_deferred(batch)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"enqueue {(t1-t0)/200*1e6:.0f} us per forward; with the GPU drained {(t2-t0)/200*1e6:.0f} us per forward")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): model.forward_deferred(batch)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
