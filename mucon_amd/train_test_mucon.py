"""Train, then evaluate with the Viterbi decode -- the counterpart of the reference's entry script
(src/train_test_mucon.py: same defaults tree, YAML overlays and KEY VALUE overrides, same order of work):

    python -m mucon_amd.train_test_mucon --cfg configs/docker/inside.yaml --set dataset.split 1 [--exp-name NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m mucon_amd.train_test_mucon --cfg ... --set ...

fandak's experiment services (run folders, metric files, tensorboard) are reduced to what the script itself needs:
<trainer.root>/<experiment_name>/<run>/{config.yaml, epoch_<n>.pt, data_test_eval.pkl, results.json}.  One process per
GPU; with WORLD_SIZE > 1 the videos are sharded over ranks and gradients are averaged with one all-reduce per step
(mucon_amd/mucon/trainers.py).  The tapes are cached in HBM unless --no-resident is given."""
import argparse
import json
import os
import pickle
from pathlib import Path

import torch

from .config import get_cfg_defaults, update_config
from .core.datasets import handel_dataset, make_resident
from .mucon.evaluators import MuConEvaluator
from .mucon.models import create_model
from .mucon.trainers import SimpleTrainer


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--cfg", action="append", default=[], help="YAML overlay (repeatable)")
    ap.add_argument("--set", nargs="+", action="append", default=[], metavar="KEY VALUE", help="config overrides")
    ap.add_argument("--exp-name", default="")
    ap.add_argument("--no-resident", action="store_true", help="read the .npy files every step instead of caching in HBM")
    return ap.parse_args(argv)


def new_run_folder(root: Path, exp_name: str) -> Path:
    base = root / exp_name
    base.mkdir(parents=True, exist_ok=True)
    n = 1 + max([int(p.name) for p in base.iterdir() if p.is_dir() and p.name.isdigit()], default=0)
    (base / str(n)).mkdir()
    return base / str(n)


def jsonable(v):
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return float(v)


def main(argv=None):
    args = parse(argv)
    cfg = update_config(get_cfg_defaults(), args.cfg, args.set)
    if args.exp_name:
        cfg.defrost()
        cfg.experiment_name = args.exp_name
        cfg.freeze()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = cfg.system.device
    if device.startswith("cuda"):
        # one process per GPU; more ranks than visible GPUs (tests on a 1-GPU box) wrap around and then need MUCON_DIST_BACKEND=gloo
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        device = f"cuda:{torch.cuda.current_device()}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("NCCL_IB_DISABLE", "1")     # one node: the gradient all-reduce stays on xGMI (an operator's own setting wins)
        dist.init_process_group(os.environ.get("MUCON_DIST_BACKEND", "nccl" if device.startswith("cuda") else "gloo"))
    if rank == 0:
        print(cfg.dump())
        # the reference pins PyTorch 1.1, whose affine_grid / grid_sample behave as align_corners=True; torch >= 1.3 defaults to
        # False.  The convention is an explicit key here (saved with config.yaml): say which one this run uses.
        print(f"model.loss.mucon.align_corners = {bool(cfg.model.loss.mucon.get('align_corners', True))} "
              f"(True = the reference's PyTorch 1.1 environment; a reference run on torch >= 1.3 corresponds to False)", flush=True)
    torch.manual_seed(int(cfg.system.seed))

    train_db, test_db = handel_dataset(cfg, train=True), handel_dataset(cfg, train=False)
    if device.startswith("cuda") and not args.no_resident:
        train_db, test_db = make_resident(train_db, device), make_resident(test_db, device)
    model = create_model(cfg, num_classes=train_db.get_num_classes(), max_decoding_steps=train_db.max_transcript_length + 1,
                         input_feature_size=train_db.feat_dim).to(device)         # + 1: EOS
    evaluator = MuConEvaluator(cfg, test_db, model, device)
    trainer = SimpleTrainer(cfg, model, device, train_db, world_size=world, rank=rank)
    run = new_run_folder(Path(cfg.trainer.root), cfg.experiment_name) if rank == 0 else None
    if rank == 0:
        (run / "config.yaml").write_text(cfg.dump())

    t = cfg.trainer
    for epoch in range(1, t.num_epochs + 1):
        losses = trainer.train_epoch(epoch)
        if rank == 0:
            print(f"epoch {epoch}: mean loss {sum(losses) / max(len(losses), 1):.4f} over {len(losses)} videos per rank", flush=True)
        if t.eval_every and epoch % t.eval_every == 0:
            evaluator.viterbi_mode(False)
            res = evaluator.evaluate(rank, world)
            trainer.step_scheduler_on_eval(res)       # scheduler.name == "plateau" (every rank sees the same all-reduced result)
            if rank == 0:
                print(f"epoch {epoch}: y_mof {res['y_mof']:.4f}  s_mof {res['s_mof']:.4f}  s_mat_score {res['s_mat_score']:.4f}", flush=True)
        if rank == 0 and t.save_every and epoch % t.save_every == 0:
            torch.save({"model": model.state_dict(), "optimizer": trainer.optimizer.state_dict(), "epoch": epoch}, run / f"epoch_{epoch}.pt")
    if rank == 0:
        torch.save({"model": model.state_dict(), "optimizer": trainer.optimizer.state_dict(), "epoch": t.num_epochs},
                   run / f"epoch_{t.num_epochs}.pt")

    evaluator.viterbi_mode(True)                      # full evaluation with viterbi
    result = evaluator.evaluate(rank, world)
    if rank == 0:
        print(result)
        with open(run / "data_test_eval.pkl", "wb") as f:
            pickle.dump(evaluator.to_save, f)
        (run / "results.json").write_text(json.dumps({k: jsonable(v) for k, v in result.items()}, indent=1))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
