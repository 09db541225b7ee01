"""GPU: the entry scripts (mucon_amd.train_test_mucon / mucon_amd.test_mucon, counterparts of the reference's
src/train_test_mucon.py and src/test_mucon.py) on a Breakfast-shaped synthetic tree: YAML overlay + KEY VALUE overrides,
two epochs with evaluation and checkpoints, final Viterbi evaluation, then re-evaluation of the saved run."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_then_test_scripts(tmp_path):
    from mucon_amd import test_mucon, train_test_mucon
    from mucon_amd.core.datasets import write_synthetic_breakfast
    from mucon_amd.mucon.evaluators import RESULT_FIELDS
    data, root = tmp_path / "datasets", tmp_path / "root"
    write_synthetic_breakfast(str(data), n_train=6, n_test=3, t_range=(200, 400), n_range=(2, 4))
    overlay = tmp_path / "inside.yaml"
    overlay.write_text(f"trainer:\n  root: {root}\n  save_every: 1\n  eval_every: 1\ndataset:\n  root: {data}\n")
    res = train_test_mucon.main(["--cfg", str(overlay), "--set", "trainer.num_epochs", "2", "dataset.split", "1",
                                 "--exp-name", "smoke"])
    assert set(RESULT_FIELDS) <= set(res)
    run = root / "smoke" / "1"
    assert (run / "config.yaml").exists() and (run / "epoch_1.pt").exists() and (run / "epoch_2.pt").exists()
    assert (run / "data_test_eval.pkl").exists()
    saved = json.loads((run / "results.json").read_text())
    assert abs(saved["y_mof"] - float(res["y_mof"])) < 1e-12
    again = test_mucon.main(["smoke/1/2", "--root", str(root)])
    assert abs(float(again["y_mof"]) - float(res["y_mof"])) < 1e-6       # same weights, same test videos
    sd = torch.load(run / "epoch_2.pt", map_location="cpu")["model"]
    assert "ft.first_conv.weight" in sd and "fs_decoder_attention_W1" in sd
