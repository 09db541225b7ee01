"""GPU: the fused HIP loss kernels (csrc/loss.hpp) through the C ABI against (1) the reference's own
MuCon.loss values and gradients (tests/golden/loss_cases.npz: every configuration under both conventions of affine_grid /
grid_sample -- "<case>@ac" = align_corners True, the PyTorch 1.1 the reference pins; "<case>" = False, a current torch's
default) and (2) oracle/losses.py in float64 at other sizes, both conventions.
Tolerances: loss values 3e-5 relative; gradients 1e-3 relative L2 against the float32 golden and 5e-4 against the
float64 oracle (the mask-boundary terms are O(T) multiples of float32 rounding in the frame coordinate)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import loss_inputs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
GOLD = np.load(os.path.join(HERE, "golden", "loss_cases.npz"))
with open(os.path.join(HERE, "golden", "loss_cases.json")) as f:
    CASES = json.load(f)
DEV = "cuda:0"


def _rel(a, b):
    a = a.detach().double().cpu().numpy().reshape(-1) if torch.is_tensor(a) else np.asarray(a, np.float64).reshape(-1)
    b = b.detach().double().cpu().numpy().reshape(-1) if torch.is_tensor(b) else np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _spec_and_consts(ocfg, M):
    from mucon_amd import ops
    from mucon_amd.mucon.masks import _template
    spec = ops.LossSpec(mucon_type=ocfg.mucon_type, overlap=ocfg.mucon_overlap, smoothing_clamp=ocfg.smoothing_clamp,
                        clamp_min=ocfg.smoothing_clamp_min, clamp_max=ocfg.smoothing_clamp_max, length_width=ocfg.length_width,
                        transcript_average=ocfg.transcript_average, mul_transcript=ocfg.mul_transcript,
                        mul_length=ocfg.mul_length, mul_mucon=ocfg.mul_mucon, mul_smoothing=ocfg.mul_smoothing,
                        align_corners=ocfg.mucon_align_corners)
    tmpl = _template(ocfg.mucon_template, 1, torch.zeros(1, device=DEV)).reshape(-1)
    mw = tw = None
    if ocfg.mucon_weight_background:
        mw = torch.ones(M, device=DEV)
        mw[ocfg.mucon_weight_background_index] = ocfg.mucon_weight_background_value
    if ocfg.transcript_weight_background:
        tw = torch.ones(M + 1, device=DEV)
        tw[ocfg.transcript_weight_background_index] = ocfg.transcript_weight_background_value
    return spec, tmpl, mw, tw


def _hip_loss(ocfg, seg, tlogp, lengths, mtarget, ttarget):
    from mucon_amd import ops
    spec, tmpl, mw, tw = _spec_and_consts(ocfg, seg.shape[1])
    sx = torch.log_softmax(seg, dim=1) if ocfg.smoothing_log_softmax_before else seg
    return ops.losses_forward(seg, sx, tlogp, lengths, spec, mtarget, ttarget, tmpl, mw, tw)


@pytest.mark.parametrize("case", sorted(CASES))
def test_fused_loss_matches_reference_golden(case):
    from oracle import losses
    c = CASES[case]
    T, N, seed = c["T"], c["N"], c["seed"]
    ocfg = losses.LossConfig.from_overrides(c["overrides"])
    seg, tl, ln, tr = loss_inputs(T, N, seed)
    seg_t = torch.from_numpy(seg).to(DEV).requires_grad_(True)
    tlogp = torch.log_softmax(torch.from_numpy(tl).to(DEV), dim=1).requires_grad_(True)
    ln_t = torch.from_numpy(ln).to(DEV).requires_grad_(True)
    if c["teacher_forcing"]:
        target = torch.from_numpy(tr).long().to(DEV)
    else:
        target = tlogp[:-1].argmax(dim=1)
        target[target >= 48] = 0
    main, parts = _hip_loss(ocfg, seg_t, tlogp, ln_t, target, torch.tensor(tr.tolist() + [48], device=DEV))
    got = np.asarray([main.item()] + parts.tolist())
    np.testing.assert_allclose(got, GOLD[f"{case}__losses"], rtol=3e-5, atol=2e-6)
    main.backward()
    sub = 1 if T <= 400 else 7
    scale = float(GOLD[f"{case}__d_seg_norm"])
    assert np.linalg.norm(seg_t.grad.cpu().numpy()[::sub].astype(np.float64) - GOLD[f"{case}__d_seg"]) / scale < 1e-3
    assert _rel(tlogp.grad, GOLD[f"{case}__d_tlogp"]) < 1e-5
    assert _rel(ln_t.grad, GOLD[f"{case}__d_lengths"]) < 1e-3


@pytest.mark.parametrize("T,N,M,over", [
    (2, 1, 48, []), (33, 64, 48, []), (5000, 30, 16, ["model.loss.mucon.template", "gaussian", "model.loss.mucon.overlap", 0.15]),
    (9000, 25, 64, ["model.loss.mucon.type", "arithmetic", "model.loss.mucon.template", "trapezoid"]),
    (777, 9, 48, ["model.loss.mucon.overlap", 0.3, "model.loss.mucon_weight_background", True,
                  "model.loss.smoothing.log_softmax_before", False, "model.loss.smoothing.clamp", False])])
@pytest.mark.parametrize("align_corners", [True, False])
def test_fused_loss_against_float64_oracle(T, N, M, over, align_corners):
    from oracle import losses
    ocfg = losses.LossConfig.from_overrides(list(over) + ["model.loss.mucon.align_corners", align_corners])
    g = torch.Generator().manual_seed(T + N)
    seg = ((torch.rand((T, M), generator=g) * 2 - 1) * 3).float()
    tl = torch.log_softmax(((torch.rand((N + 1, M + 1), generator=g) * 2 - 1) * 2).double(), dim=1).float()
    ln = ((torch.rand((N,), generator=g) * 2 - 1) * 2).float()
    mt = torch.randint(0, M, (N,), generator=g)
    tt = torch.cat([mt, torch.tensor([M])])
    a = [seg.double().requires_grad_(True), tl.double().requires_grad_(True), ln.double().requires_grad_(True)]
    want = losses.loss(ocfg, a[0], a[1], a[2], mt, tt)
    want[0].backward()
    b = [seg.to(DEV).requires_grad_(True), tl.to(DEV).requires_grad_(True), ln.to(DEV).requires_grad_(True)]
    main, parts = _hip_loss(ocfg, b[0], b[1], b[2], mt.to(DEV), tt.to(DEV))
    got = np.asarray([main.item()] + parts.tolist())
    np.testing.assert_allclose(got, [float(v.detach()) for v in want], rtol=3e-5, atol=2e-6)
    main.backward()
    assert _rel(b[0].grad, a[0].grad) < 5e-4
    assert _rel(b[1].grad, a[1].grad) < 1e-5
    assert _rel(b[2].grad, a[2].grad) < 5e-4


def test_fused_loss_is_bitwise_reproducible_and_scales_with_upstream_gradient():
    from oracle import losses
    ocfg = losses.LossConfig()
    seg, tl, ln, tr = loss_inputs(2000, 12, 29)
    outs = []
    for scale in (1.0, 1.0, 0.5):
        seg_t = torch.from_numpy(seg).to(DEV).requires_grad_(True)
        tlogp = torch.log_softmax(torch.from_numpy(tl).to(DEV), dim=1).requires_grad_(True)
        ln_t = torch.from_numpy(ln).to(DEV).requires_grad_(True)
        main, _ = _hip_loss(ocfg, seg_t, tlogp, ln_t, torch.from_numpy(tr).long().to(DEV), torch.tensor(tr.tolist() + [48], device=DEV))
        (main * scale).backward()
        outs.append((main.detach().clone(), seg_t.grad.clone(), ln_t.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.allclose(outs[2][2], outs[0][2] * 0.5, rtol=1e-6, atol=0)


def test_model_native_loss_equals_torch_formulation():
    from test_gpu_model import make_batch, seeded_value
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    for over in ([], ["model.loss.mucon.type", "arithmetic"], ["model.loss.mucon.template", "gaussian", "model.loss.mucon.overlap", "0.1"],
                 ["model.loss.mucon.align_corners", "False"], ["model.loss.mucon.align_corners", "False", "model.loss.mucon.type", "arithmetic"]):
        cfg = update_config(get_cfg_defaults(), [], [over])
        model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
        model = model.cuda().eval()
        model.set_teacher_forcing(True)
        batch = make_batch(640, 5).to("cuda")
        res = []
        for native in (True, False):
            model.native_loss = native
            model.zero_grad()
            fo = model.forward(batch)
            loss = model.loss(batch, fo)
            loss.main.backward()
            res.append(([loss.main.item(), loss.transcript_loss.item(), loss.length_loss.item(), loss.mucon_loss.item(),
                         loss.smoothing_loss.item()],
                        {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-5, atol=1e-6)
        assert set(res[0][1]) == set(res[1][1])
        for n in res[0][1]:
            a, b = res[0][1][n].double(), res[1][1][n].double()
            assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-6, n


def test_fused_loss_rejects_unsupported_sizes():
    from mucon_amd import ops, _lib
    spec = ops.LossSpec()
    z = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    with pytest.raises(_lib.MuconHipError):   # 65 segments
        ops.losses_forward(z(100, 48), z(100, 48), z(66, 49), z(65), spec, torch.zeros(65, dtype=torch.long, device=DEV),
                           torch.zeros(66, dtype=torch.long, device=DEV), torch.ones(100, device=DEV))
    with pytest.raises(Exception, match="Invalid mucon type"):
        ops.losses_forward(z(100, 48), z(100, 48), z(3, 49), z(2), ops.LossSpec(mucon_type="geometric"),
                           torch.zeros(2, dtype=torch.long, device=DEV), torch.zeros(3, dtype=torch.long, device=DEV),
                           torch.ones(100, device=DEV))
