"""HBM-resident dataset cache (mucon_amd/core/datasets/resident.py): every Batch equals what the file-backed
GeneralDataset (the reference's behaviour, general_dataset.py:138-167) returns; views are zero-copy; the byte budget
spills the tail to the file-backed path.  CPU here (arena in host memory); the GPU variant is in the gpu-marked test."""
import dataclasses

import pytest
import torch

from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.core.datasets import handel_dataset, make_resident, write_synthetic_breakfast


def _db(tmp_path, **kw):
    root = tmp_path / "data"
    write_synthetic_breakfast(str(root), n_train=6, n_test=2, t_range=(40, 90), feat_dim=32, **kw)
    cfg = update_config(get_cfg_defaults(), [], [["dataset.root", str(root)]])
    return handel_dataset(cfg, train=True)


def _same(a, b):
    for f in dataclasses.fields(a):
        x, y = getattr(a, f.name), getattr(b, f.name)
        if isinstance(x, torch.Tensor):
            assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x.cpu(), y.cpu()), f.name
        else:
            assert x == y, f.name


def test_resident_batches_equal_file_backed(tmp_path):
    db = _db(tmp_path)
    res = make_resident(db, "cpu")
    assert len(res) == len(db) and res.resident_fraction == 1.0 and res.get_num_classes() == db.get_num_classes()
    for i in range(len(db)):
        _same(res[i], db[i])
    b = res[2]
    assert b.feats.data_ptr() == res[2].feats.data_ptr()            # a view of the arena, not a copy
    assert b.feats.untyped_storage().data_ptr() == res[0].feats.untyped_storage().data_ptr()
    assert res.collate_fn([b]) is b


def test_budget_spills_to_the_file_backed_path(tmp_path):
    db = _db(tmp_path)
    full = make_resident(db, "cpu")
    some = make_resident(db, "cpu", max_bytes=full.resident_bytes // 2)
    assert 0 < some.resident_fraction < 1
    assert some.resident_bytes <= full.resident_bytes // 2
    for i in range(len(db)):
        _same(some[i], db[i])
    assert any(not some.is_resident(i) for i in range(len(db)))


def test_subset_and_shape_errors(tmp_path):
    db = _db(tmp_path)
    sub = make_resident(db, "cpu", indices=[1, 3])
    assert sub.is_resident(1) and sub.is_resident(3) and not sub.is_resident(0)
    _same(sub[0], db[0])
    import numpy as np
    name = db.file_names[0]
    np.save(db.root / "labels" / f"{name}.npy", np.zeros(3, dtype=np.int64))
    with pytest.raises(ValueError):
        make_resident(db, "cpu")


@pytest.mark.gpu
def test_resident_on_the_gpu_and_trains(tmp_path):
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer
    root = tmp_path / "data"
    write_synthetic_breakfast(str(root), n_train=5, n_test=2, t_range=(200, 500))
    cfg = update_config(get_cfg_defaults(), [], [["dataset.root", str(root)]])
    db = handel_dataset(cfg, train=True)
    res = make_resident(db, "cuda:0")
    assert res.resident_fraction == 1.0
    for i in range(len(db)):
        assert res[i].feats.is_cuda
        _same(res[i], db[i])
    model = create_model(cfg, db.get_num_classes(), db.max_transcript_length, db.feat_dim).cuda()
    tr = SimpleTrainer(cfg, model, "cuda", train_db=res)
    losses = tr.train_epoch(0)
    assert len(losses) == len(db) and all(l == l for l in losses)
