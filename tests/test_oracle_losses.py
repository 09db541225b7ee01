"""CPU: oracle/losses.py (explicit float64 restatement of MuCon.loss incl. the mask construction) against the
reference's own MuCon.loss values and gradients -- tests/golden/loss_cases.npz, made by tools/make_golden_loss.py.
The golden is float32 torch: values 2e-5 relative, gradients 5e-4 relative L2 (the mask boundary terms are
differences of O(1) float32 numbers times T)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import loss_inputs

HERE = os.path.dirname(__file__)
GOLD = np.load(os.path.join(HERE, "golden", "loss_cases.npz"))
with open(os.path.join(HERE, "golden", "loss_cases.json")) as f:
    CASES = json.load(f)


def _rel(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_run(case):
    from oracle import losses
    c = CASES[case]
    T, N, seed = c["T"], c["N"], c["seed"]
    cfg = losses.LossConfig.from_overrides(c["overrides"])
    seg, tl, ln, tr = loss_inputs(T, N, seed)
    seg_t = torch.from_numpy(seg).double().requires_grad_(True)
    tlogp = torch.log_softmax(torch.from_numpy(tl).double(), dim=1).requires_grad_(True)
    ln_t = torch.from_numpy(ln).double().requires_grad_(True)
    if c["teacher_forcing"]:
        target = torch.from_numpy(tr).long()
    else:
        target = tlogp[:-1].argmax(dim=1)
        target[target >= 48] = 0
    out = losses.loss(cfg, seg_t, tlogp, ln_t, target, torch.tensor(tr.tolist() + [48]))
    out[0].backward()
    return out, seg_t.grad, tlogp.grad, ln_t.grad


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_loss_matches_reference(case):
    out, d_seg, d_tlogp, d_len = oracle_run(case)
    got = np.asarray([float(v.detach()) for v in out])
    np.testing.assert_allclose(got, GOLD[f"{case}__losses"], rtol=2e-5, atol=1e-6)
    sub = 1 if CASES[case]["T"] <= 400 else 7
    scale = float(GOLD[f"{case}__d_seg_norm"])
    assert np.linalg.norm(d_seg.numpy()[::sub] - GOLD[f"{case}__d_seg"]) / scale < 5e-4
    assert abs(float(d_seg.norm()) - scale) / scale < 5e-4
    assert _rel(d_tlogp.numpy(), GOLD[f"{case}__d_tlogp"]) < 1e-5
    assert _rel(d_len.numpy(), GOLD[f"{case}__d_lengths"]) < 5e-4


def test_templates_match_the_product_masks_module():
    from oracle import losses
    from mucon_amd.mucon import masks
    like = torch.zeros(1)
    for kind in ("box", "gaussian", "trapezoid"):
        got = masks._template(kind, 1, like).reshape(-1).double()
        assert torch.equal(got, losses.template(kind)), kind
