"""GPU: the data-parallel path with the REAL model and trainer on what hardware exists -- one GPU.  Two ranks share cuda:0
under the gloo backend (RCCL refuses two ranks on one device; device tensors travel through a host copy,
trainers.dist_all_reduce), which exercises everything of the N > 1 path except RCCL itself: parameter broadcast, the
rank-independent gradient bucket (one collective whose layout does not depend on which gradients exist or which route -- the
fused HIP step or torch.autograd -- produced them), per-rank dropout masks, the fused clip + SGD on the averaged gradient.

  * world size 2, three steps, ranks on different videos, the two routes MIXED across ranks: parameters stay bit-equal across
    ranks, and equal (1e-5) to one process that computes both ranks' gradients itself, averages them and takes the same
    optimizer step;
  * the ranks' dropout masks differ (same video, same step, different rank -> different encoder output);
  * BASELINE config 4's launcher (tools/launch_splits.sh): two independent 2-rank groups train two splits concurrently."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIDEOS = [(420, 4, 11), (333, 3, 21), (510, 5, 31), (390, 4, 41), (450, 3, 51), (365, 5, 61)]
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg(dropout):
    from mucon_amd.config import get_cfg_defaults, update_config
    d = str(dropout)
    return update_config(get_cfg_defaults(), [], [["model.ft.dropout_rate", d, "model.ft.last_dropout_rate", d,
                                                   "model.fs.decoder.embedding_dropout", d, "trainer.learning_rate", "0.01"]])


def _model(cfg):
    from test_gpu_model import seeded_value
    from mucon_amd.mucon.models import create_model
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
    return model.cuda()


def _fused_on(rank, step):
    return not ((step == 1 and rank == 1) or (step == 2 and rank == 0))   # the two routes mixed across ranks


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_trajectory import make_batch
    from mucon_amd.mucon.trainers import SimpleTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = _cfg(0.0)
    model = _model(cfg)
    if rank == 1:                          # a replica that starts different: the broadcast at trainer start must repair it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
    trainer = SimpleTrainer(cfg, model, "cuda", world_size=world, rank=rank)
    trainer.on_start_epoch(0)
    model.train()
    for step in range(STEPS):
        trainer.fuse_step = _fused_on(rank, step)
        trainer._train_1_batch(step, make_batch(*VIDEOS[step * world + rank]).to("cuda"))
    params = {n: p.detach().cpu() for n, p in model.named_parameters()}
    # dropout masks: same video, same step counter, rank-specific key
    cfg_d = _cfg(0.25)
    m2 = _model(cfg_d)
    m2.dropout_rank = rank
    m2.train()
    with torch.no_grad():
        enc = m2.temporal_modeling_forward(make_batch(*VIDEOS[0]).to("cuda").feats)
    torch.save({"params": params, "enc": enc.cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_a_single_process_on_the_averaged_gradient(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n        # replicas stay bit-identical
    assert not torch.equal(r0["enc"], r1["enc"])                        # every rank draws its own dropout masks

    # one process: both ranks' gradients on the same weights, averaged, then the same optimizer tail
    from test_gpu_trajectory import make_batch
    from mucon_amd.mucon.trainers import SimpleTrainer
    cfg = _cfg(0.0)
    model = _model(cfg)
    trainer = SimpleTrainer(cfg, model, "cuda")
    trainer.on_start_epoch(0)
    model.train()
    params = [p for p in model.parameters()]
    for step in range(STEPS):
        grads = []
        for rank in range(world):
            trainer.optimizer.zero_grad()
            batch = make_batch(*VIDEOS[step * world + rank]).to("cuda")
            if _fused_on(rank, step):
                model.fused_train_step(batch)
            else:
                model.loss(batch, model.forward(batch)).main.backward()
            grads.append([None if p.grad is None else p.grad.detach().clone() for p in params])
        for p, g0, g1 in zip(params, *grads):
            if g0 is None and g1 is None:
                p.grad = None
            else:
                z = torch.zeros_like(p)
                p.grad = ((g0 if g0 is not None else z) + (g1 if g1 is not None else z)) / world
        trainer.fused_step.step()
    for n, p in model.named_parameters():
        a, b = r0["params"][n].double(), p.detach().cpu().double()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7, n


def test_config4_launcher_runs_two_independent_groups_concurrently(tmp_path):
    """tools/launch_splits.sh with two splits x two ranks on the one GPU (gloo): both groups train an epoch, evaluate with the
    Viterbi decode and write their own results -- independent rendezvous, nothing shared but the device."""
    from mucon_amd.core.datasets import write_synthetic_breakfast
    data, root = tmp_path / "datasets", tmp_path / "root"
    write_synthetic_breakfast(str(data), n_train=4, n_test=2, t_range=(200, 300), n_range=(2, 3), splits=(1, 2))
    overlay = tmp_path / "inside.yaml"
    overlay.write_text(f"trainer:\n  root: {root}\n  num_epochs: 1\n  eval_every: 0\ndataset:\n  root: {data}\n")
    env = dict(os.environ, SPLITS="1 2", GPUS_PER_GROUP="2", DEVICES="0", BASE_PORT=str(_free_port()), LOG_DIR=str(tmp_path / "logs"),
               MUCON_DIST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "launch_splits.sh"), "--cfg", str(overlay)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    logs = "".join(open(tmp_path / "logs" / f"split{s}.log").read()[-2000:] for s in (1, 2) if (tmp_path / "logs" / f"split{s}.log").exists())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:] + logs
    for s in (1, 2):
        assert (root / f"split{s}" / "1" / "results.json").exists(), logs
