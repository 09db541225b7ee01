"""Goldens for the fully- / mixed-supervised classes (build container only): the reference's MuConFullySupervised and
MuConMixedSupervision (src/mucon/models.py:781-911) on one seeded video in eval() mode with teacher forcing: the seven loss
values; and the supervised-flag draw of GeneralMixedSupervisionDataset (src/core/datasets/general_dataset.py:211-246)."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402

ref_harness.install()
from mucon_amd import synth  # noqa: E402
from make_golden_model import seeded_value  # noqa: E402


def main():
    from configs.mucon.default import get_cfg_defaults
    from core.datasets.general_dataset import FullySupervisedBatch, MixedSupervisionBatch
    from mucon.models import create_fully_supervised_model, create_mixed_supervision_model

    out = {}
    T, N, C, seed = 400, 4, 48, 5
    tr = synth.transcript(seed, N, C, allow_repeats=False)
    gt = synth.segment_labels(seed + 1, T, tr)
    feats = synth.uniform_pm1(seed + 2, (1, T, 2048))
    cuts = np.flatnonzero(np.diff(gt)) + 1
    lengths = np.diff(np.concatenate(([0], cuts, [T]))).astype(np.float32)
    base = dict(feats=torch.from_numpy(feats), gt_label=torch.from_numpy(gt), transcript=torch.from_numpy(tr),
                transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]),
                video_name="synthetic", absolute_lengths=torch.from_numpy(lengths))
    for kind, create in (("full", create_fully_supervised_model), ("mixed_on", create_mixed_supervision_model),
                         ("mixed_off", create_mixed_supervision_model)):
        cfg = get_cfg_defaults()
        model = create(cfg, num_classes=C, max_decoding_steps=31, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
        model.eval()
        model.set_teacher_forcing(True)
        batch = (FullySupervisedBatch(**base) if kind == "full"
                 else MixedSupervisionBatch(**base, fully_supervised=(kind == "mixed_on")))
        fo = model.forward(batch)
        loss = model.loss(batch, fo)
        out[f"{kind}__loss"] = np.asarray([loss.main.item(), loss.transcript_loss.item(), loss.length_loss.item(), loss.mucon_loss.item(),
                                           loss.smoothing_loss.item(), loss.classification_loss.item(),
                                           loss.supervised_length_loss.item()], dtype=np.float64)
        print(kind, out[f"{kind}__loss"])
    out["meta"] = np.asarray([T, N, seed])
    # the supervised-flag draw for a few (n, percentage) pairs at the default system seed
    seed0 = get_cfg_defaults().system.seed
    for n, pct in ((10, 50.0), (37, 20.0), (5, 1.0)):
        k = min(n, max(1, int(round(n * pct / 100.0))))
        flags = [False] * n
        flags[:k] = [True for _ in range(k)]
        random.seed(f"{seed0}-{k}")
        random.shuffle(flags)
        out[f"flags__{n}__{pct}"] = np.asarray(flags)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "supervised_cases.npz"), **out)


if __name__ == "__main__":
    main()
