"""HIP bidirectional LSTM (csrc/lstm.hpp) against torch.nn.LSTM evaluated on the CPU in
float64 -- the very op the reference's s-head calls (reference src/mucon/models.py:195-201, :605-611).
Tolerances: forward 2e-5 absolute (activations are O(1), fp32 expf/tanhf), gradients 1e-4 relative L2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(T, ndir, seed):
    g = torch.Generator().manual_seed(seed)
    lstm = torch.nn.LSTM(128, 128, batch_first=True, bidirectional=(ndir == 2)).double()
    for p in lstm.parameters():  # larger than the default init so the gates leave the linear regime
        p.data = (torch.rand(p.shape, generator=g, dtype=torch.float64) * 2 - 1) * 0.25
    x = torch.randn(1, T, 128, generator=g, dtype=torch.float64, requires_grad=True)
    out, (hn, cn) = lstm(x)
    d_out = torch.randn(out.shape, generator=g, dtype=torch.float64)
    d_hn = torch.randn(hn.shape, generator=g, dtype=torch.float64)
    d_cn = torch.randn(cn.shape, generator=g, dtype=torch.float64)
    ((out * d_out).sum() + (hn * d_hn).sum() + (cn * d_cn).sum()).backward()
    return lstm, x, out, hn, cn, d_out, d_hn, d_cn


def _rel(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("T,ndir", [(1, 2), (2, 2), (7, 1), (31, 2), (125, 2), (300, 2), (1024, 1)])
def test_lstm_matches_torch_f64(T, ndir):
    from mucon_amd import ops
    lstm, x, out, hn, cn, d_out, d_hn, d_cn = _ref(T, ndir, seed=100 + T + ndir)
    dev = torch.device("cuda:0")
    ws = [p.detach().float().to(dev).requires_grad_(True) for p in lstm.parameters()]
    xg = x.detach()[0].float().to(dev).requires_grad_(True)
    o, h, c = ops.lstm_forward(xg, ws, bidirectional=(ndir == 2))
    assert o.shape == (T, ndir * 128) and h.shape == (ndir, 128) and c.shape == (ndir, 128)
    assert (o.double().cpu() - out[0]).abs().max() < 2e-5
    assert (h.double().cpu() - hn[:, 0]).abs().max() < 2e-5
    assert (c.double().cpu() - cn[:, 0]).abs().max() < 5e-5
    loss = (o * d_out[0].float().to(dev)).sum() + (h * d_hn[:, 0].float().to(dev)).sum() + (c * d_cn[:, 0].float().to(dev)).sum()
    loss.backward()
    assert _rel(xg.grad, x.grad[0]) < 1e-4
    for pg, pr in zip(ws, lstm.parameters()):
        assert _rel(pg.grad, pr.grad) < 1e-4


def test_lstm_partial_upstream_gradients():
    """Only h_n / c_n feed the loss (d_out absent), and only out (d_hn, d_cn absent)."""
    from mucon_amd import ops
    lstm, x, *_ = _ref(20, 2, seed=7)
    dev = torch.device("cuda:0")
    for which in ("hn", "out"):
        lstm.zero_grad()
        xr = x.detach().clone().requires_grad_(True)
        out, (hn, cn) = lstm(xr)
        (hn.sum() + 2 * cn.sum() if which == "hn" else out.square().sum()).backward()
        ws = [p.detach().float().to(dev).requires_grad_(True) for p in lstm.parameters()]
        xg = x.detach()[0].float().to(dev).requires_grad_(True)
        o, h, c = ops.lstm_forward(xg, ws)
        (h.sum() + 2 * c.sum() if which == "hn" else o.square().sum()).backward()
        assert _rel(xg.grad, xr.grad[0]) < 1e-4
        for pg, pr in zip(ws, lstm.parameters()):
            assert _rel(pg.grad, pr.grad) < 1e-4


def test_lstm_rejects_other_sizes():
    from mucon_amd import ops
    dev = torch.device("cuda:0")
    lstm = torch.nn.LSTM(64, 64, bidirectional=True).to(dev)
    with pytest.raises(RuntimeError):
        ops.lstm_forward(torch.zeros(5, 64, device=dev), list(lstm.parameters()))


def test_model_native_lstm_equals_miopen_path():
    """The model's s-head with the HIP LSTM and with torch's nn.LSTM give the same transcript / length heads."""
    from test_gpu_model import make_batch
    from mucon_amd.config import get_cfg_defaults
    from mucon_amd.mucon.models import create_model
    torch.manual_seed(3)
    model = create_model(get_cfg_defaults(), num_classes=48, max_decoding_steps=31, input_feature_size=2048).cuda().eval()
    model.set_teacher_forcing(True)
    batch = make_batch(640, 5).to("cuda")
    outs = []
    for native in (True, False):
        model.native_lstm = native
        with torch.no_grad():
            fo = model.forward(batch)
        outs.append((fo.transcript, fo.lengths))
    assert (outs[0][0] - outs[1][0]).abs().max() < 1e-4
    assert (outs[0][1] - outs[1][1]).abs().max() < 1e-4
