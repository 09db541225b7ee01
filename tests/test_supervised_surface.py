"""Fully- and mixed-supervised variants (reference src/mucon/models.py:781-911,
src/core/datasets/general_dataset.py:36-43, 176-263): datasets with the reference's extra fields and supervised-flag draw
(CPU), and -- on the GPU -- the seven loss values against the reference's own (tests/golden/supervised_cases.npz)."""
import os

import numpy as np
import pytest
import torch

from helpers import seeded_model_value
from mucon_amd import synth
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.core import datasets as D

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "supervised_cases.npz"))


def test_supervised_datasets(tmp_path):
    root = tmp_path / "data"
    D.write_synthetic_breakfast(str(root), n_train=10, n_test=2, t_range=(40, 90), feat_dim=16)
    cfg = update_config(get_cfg_defaults(), [], [["dataset.root", str(root)]])
    full = D.handel_fully_supervised_dataset(cfg, train=True)
    b = full[3]
    assert isinstance(b, D.FullySupervisedBatch) and b.absolute_lengths.dtype == torch.float32
    assert int(b.absolute_lengths.sum()) == b.gt_label.shape[0] and b.absolute_lengths.shape[0] == b.transcript.shape[0]
    assert full.convenient_name == "fully_supervised_breakfast_split1_train"
    mixed = D.handel_mixed_supervision_dataset(cfg, train=True)
    assert mixed.is_it_supervised == GOLD["flags__10__50.0"].tolist() and sum(mixed.is_it_supervised) == 5
    assert isinstance(mixed[0], D.MixedSupervisionBatch) and mixed[0].fully_supervised == mixed.is_it_supervised[0]
    assert mixed.convenient_name == "mixed_supervision_percentage_50.0_breakfast_split1_train"
    on_device = mixed[0].to("cpu")
    assert on_device.fully_supervised == mixed.is_it_supervised[0]


@pytest.mark.parametrize("n,pct", [(10, 50.0), (37, 20.0), (5, 1.0)])
def test_mixed_supervision_flag_draw_is_the_references(tmp_path, n, pct):
    root = tmp_path / "data"
    D.write_synthetic_breakfast(str(root), n_train=n, n_test=1, t_range=(20, 30), feat_dim=4)
    cfg = update_config(get_cfg_defaults(), [], [["dataset.root", str(root), "dataset.mixed.full_supervision_percentage", str(pct)]])
    db = D.handel_mixed_supervision_dataset(cfg, train=True)
    assert db.is_it_supervised == GOLD[f"flags__{n}__{pct}"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["full", "mixed_on", "mixed_off"])
def test_supervised_losses_match_reference(kind):
    from mucon_amd.mucon import models as M
    T, N, seed = [int(v) for v in GOLD["meta"]]
    C = 48
    tr = synth.transcript(seed, N, C, allow_repeats=False)
    gt = synth.segment_labels(seed + 1, T, tr)
    cuts = np.flatnonzero(np.diff(gt)) + 1
    lengths = np.diff(np.concatenate(([0], cuts, [T]))).astype(np.float32)
    base = dict(feats=torch.from_numpy(synth.uniform_pm1(seed + 2, (1, T, 2048))), gt_label=torch.from_numpy(gt),
                transcript=torch.from_numpy(tr), transcript_tf_input=torch.tensor([C + 1] + tr.tolist()),
                transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name="synthetic",
                absolute_lengths=torch.from_numpy(lengths))
    create = M.create_fully_supervised_model if kind == "full" else M.create_mixed_supervision_model
    model = create(get_cfg_defaults(), num_classes=C, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_model_value(name, p.shape).astype(np.float32)))
    model = model.cuda().eval()
    model.set_teacher_forcing(True)
    batch = (D.FullySupervisedBatch(**base) if kind == "full"
             else D.MixedSupervisionBatch(**base, fully_supervised=(kind == "mixed_on"))).to("cuda")
    fo = model.forward(batch)
    loss = model.loss(batch, fo)
    got = [loss.main.item(), loss.transcript_loss.item(), loss.length_loss.item(), loss.mucon_loss.item(),
           loss.smoothing_loss.item(), loss.classification_loss.item(), loss.supervised_length_loss.item()]
    np.testing.assert_allclose(got, GOLD[f"{kind}__loss"], rtol=1e-3, atol=1e-6)
    loss.main.backward()
    assert model.conv_classifier.weight.grad is not None and torch.isfinite(model.conv_classifier.weight.grad).all()
