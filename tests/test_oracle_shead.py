"""CPU: oracle/shead.py (explicit-formula restatement of the s-head) against the reference's own
MuCon.sequence_generation_forward -- tests/golden/shead_cases.npz, made by tools/make_golden_shead.py.
The golden is float32 torch on CPU, the oracle float64: 2e-5 absolute on outputs, 2e-4 relative L2 on gradients."""
import os

import numpy as np
import pytest
import torch

from helpers import shead_case, shead_params

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "shead_cases.npz"))
CASES = ["a", "b", "c", "d", "e"]


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("case", CASES)
def test_oracle_teacher_forced_outputs_and_gradients(case):
    from oracle import shead
    c = shead_case(GOLD, case)
    P = {k: v.double().requires_grad_(True) for k, v in shead_params(GOLD, case).items()}
    enc = c["enc"].double().requires_grad_(True)
    logp, lens = shead.shead(enc, P, c["tf_in"], c["N"] + 1, True, False, c["eos"])
    np.testing.assert_allclose(logp.detach().numpy(), GOLD[f"{case}__tf_logp"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(lens.detach().numpy(), GOLD[f"{case}__tf_lengths"], atol=2e-5, rtol=0)
    ((logp * c["R1"].double()).sum() + (lens * c["r2"].double()).sum()).backward()
    assert _rel(enc.grad.numpy(), GOLD[f"{case}__tf_d_enc"]) < 2e-4
    for n in GOLD["param_names"]:
        n = str(n)
        g = P[n].grad.numpy().reshape(-1)
        want = GOLD[f"{case}__tf_grad__{n}"]
        got = g if g.size <= 4096 else g[::29]
        scale = max(float(GOLD[f"{case}__tf_gnorm__{n}"]), 1e-30)
        assert np.linalg.norm(got - want) / scale < 2e-4, n
        assert abs(np.linalg.norm(g) - scale) / scale < 2e-4, n


@pytest.mark.parametrize("case", CASES)
def test_oracle_greedy_decode_and_eos_stop(case):
    from oracle import shead
    c = shead_case(GOLD, case)
    P = {k: v.double() for k, v in shead_params(GOLD, case).items()}
    with torch.no_grad():
        logp, lens = shead.shead(c["enc"].double(), P, c["tf_in"], 12, False, True, c["eos"])
    want = GOLD[f"{case}__greedy_logp"]
    assert logp.shape == want.shape
    np.testing.assert_allclose(logp.numpy(), want, atol=5e-5, rtol=0)
    np.testing.assert_allclose(lens.numpy(), GOLD[f"{case}__greedy_lengths"], atol=5e-5, rtol=0)
    if case == "e":
        assert logp.shape[0] == 4 and int(logp[-1].argmax()) == c["eos"]


def test_oracle_lstm_equals_torch_module():
    """The written-out LSTM is torch.nn.LSTM (the op the reference calls), to float64 round-off."""
    from oracle import shead
    torch.manual_seed(0)
    m = torch.nn.LSTM(128, 128, batch_first=True, bidirectional=True).double()
    x = torch.randn(1, 17, 128, dtype=torch.float64)
    out, (hn, cn) = m(x)
    P = {f"fs_encoder_lstm.{k}": v.detach() for k, v in m.named_parameters()}
    o, h, c = shead.lstm(x[0], P)
    assert (o - out[0]).abs().max() < 1e-12 and (h - hn[:, 0]).abs().max() < 1e-12 and (c - cn[:, 0]).abs().max() < 1e-12
