"""Evaluate a saved run with the Viterbi decode -- the counterpart of the reference's src/test_mucon.py:

    python -m mucon_amd.test_mucon EXP_NAME/RUN/EPOCH [--root TRAINER_ROOT] [--data-root DATASETS]

Loads <root>/<exp>/<run>/config.yaml and epoch_<n>.pt (written by mucon_amd.train_test_mucon; a reference checkpoint's
model state_dict has the same keys), evaluates the test split and prints the result record."""
import argparse
from pathlib import Path

import torch

from .config import get_cfg_defaults
from .core.datasets import handel_dataset
from .mucon.evaluators import MuConEvaluator
from .mucon.models import create_model


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("identifier", help="exp-name/run-number/epoch-number")
    ap.add_argument("--root", default="")
    ap.add_argument("--data-root", default="")
    args = ap.parse_args(argv)
    cfg = get_cfg_defaults()
    root = args.root or cfg.trainer.root
    exp_name, run_number, epoch_number = args.identifier.split("/")
    run = Path(root) / exp_name / run_number
    cfg.merge_from_file(str(run / "config.yaml"))
    cfg.defrost()
    cfg.trainer.root = root
    if args.data_root:
        cfg.dataset.root = args.data_root
    cfg.freeze()
    test_db = handel_dataset(cfg, train=False)
    model = create_model(cfg, num_classes=test_db.get_num_classes(), max_decoding_steps=test_db.max_transcript_length + 1,
                         input_feature_size=test_db.feat_dim)
    state = torch.load(run / f"epoch_{int(epoch_number)}.pt", map_location="cpu")
    model.load_state_dict(state["model"] if "model" in state else state)
    device = cfg.system.device
    model = model.to(device)
    evaluator = MuConEvaluator(cfg, test_db, model, device)
    evaluator.viterbi_mode(True)
    result = evaluator.evaluate()
    print(result)
    return result


if __name__ == "__main__":
    main()
