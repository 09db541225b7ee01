"""cfg.model.ft.type = "noft" / "mstcnpp" (SURVEY 8f row 4): the non-default encoders of the reference
(src/core/modules/temporal.py:56-74, :150-204) behind MuCon.temporal_modeling_forward, against outputs of the reference's
own model on the same seeded parameters and tape (tests/golden/variant_cases.npz, tools/make_golden_variants.py).
They run on library ops (torch), so the comparison runs on the CPU here; the gpu-marked test repeats it on the device and
takes a full training step.  Tolerance 1e-5 (same float32 ops, different blocking)."""
import os

import numpy as np
import pytest
import torch

from helpers import seeded_model_value
from mucon_amd import synth
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.mucon.models import create_model

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "variant_cases.npz"))


def _model(kind):
    cfg = update_config(get_cfg_defaults(), [], [["model.ft.type", kind]])
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_model_value(name, p.shape).astype(np.float32)))
    return cfg, model


@pytest.mark.parametrize("kind", ["noft", "mstcnpp"])
def test_variant_matches_reference(kind):
    _, model = _model(kind)
    model.eval()
    assert [k for k in model.state_dict() if k.startswith("ft.")] == [str(k) for k in GOLD[f"{kind}__keys"]]
    T, Tz = [int(v) for v in GOLD[f"{kind}__meta"]]
    tape = torch.from_numpy(synth.uniform_pm1(55, (1, T, 2048)))
    with torch.no_grad():
        enc = model.temporal_modeling_forward(tape)
    assert tuple(enc.shape) == (1, Tz, 128)
    np.testing.assert_allclose(enc.numpy(), GOLD[f"{kind}__enc"], rtol=1e-5, atol=1e-5)


def test_invalid_type_raises_like_the_reference():
    cfg = update_config(get_cfg_defaults(), [], [["model.ft.type", "transformer"]])
    with pytest.raises(Exception, match="Invalid ft type"):
        create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["noft", "mstcnpp"])
def test_variant_on_gpu_and_training_step(kind):
    from test_gpu_model import make_batch
    from mucon_amd.mucon.trainers import SimpleTrainer
    cfg, model = _model(kind)
    model = model.cuda().eval()
    T, Tz = [int(v) for v in GOLD[f"{kind}__meta"]]
    tape = torch.from_numpy(synth.uniform_pm1(55, (1, T, 2048))).cuda()
    with torch.no_grad():
        enc = model.temporal_modeling_forward(tape)
    np.testing.assert_allclose(enc.cpu().numpy(), GOLD[f"{kind}__enc"], rtol=1e-4, atol=1e-4)
    tr = SimpleTrainer(cfg, model, "cuda")
    tr.on_start_epoch(0)
    model.train()
    batch = make_batch(640, 5).to("cuda")
    before = model.conv_classifier.weight.detach().clone()
    loss, _ = tr._train_1_batch(0, batch)
    assert torch.isfinite(loss.main) and not torch.equal(before, model.conv_classifier.weight)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(1, 640), (2, 4500)])
def test_noft_kernels_against_float64(B, T):
    """NoFt on first_conv's kernels through the C ABI (mucon_linear_fwd / _bwd): forward and the two gradients against float64,
    below and above the size where the forward switches to the split-bf16 kernel (8,192 frames)."""
    from mucon_amd import ops
    g = torch.Generator().manual_seed(3)
    tape = torch.randn(B, T, 2048, generator=g).cuda()
    w = (torch.randn(128, 2048, generator=g) * 0.03).cuda().requires_grad_()
    b = torch.randn(128, generator=g).cuda().requires_grad_()
    u = torch.randn(B, T, 128, generator=g).cuda()
    out = ops.linear_forward(tape, w, b)
    (out * u).sum().backward()
    ref = tape.double() @ w.detach().double().t() + b.detach().double()
    assert float((out.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    dw = (u.double().reshape(-1, 128).t() @ tape.double().reshape(-1, 2048))
    db = u.double().sum((0, 1))
    assert float((w.grad.double() - dw).abs().max()) <= 2e-5 * float(dw.abs().max())
    assert float((b.grad.double() - db).abs().max()) <= 2e-5 * float(db.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,taps,dilation", [(1, 640, 3, 1), (2, 333, 3, 64), (1, 200, 3, 512), (2, 4500, 1, 1), (1, 77, 3, 7)])
def test_conv128_kernels_against_float64(B, T, taps, dilation):
    """The 128-channel temporal convolution of MSTCNPPFirstStage through the C ABI (mucon_conv128_fwd / _dgrad / _wgrad): forward
    and all three gradients against a float64 nn.functional.conv1d, incl. a dilation that reaches past the sequence."""
    from mucon_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, T, 128, generator=g).cuda().requires_grad_()
    w = (torch.randn(128, 128, taps, generator=g) * 0.08).cuda().requires_grad_()
    b = torch.randn(128, generator=g).cuda().requires_grad_()
    u = torch.randn(B, T, 128, generator=g).cuda()
    y = ops.conv128_forward(x, w, b, dilation)
    (y * u).sum().backward()
    xd, wd, bd = (t.detach().double().cpu().requires_grad_() for t in (x, w, b))
    ref = torch.nn.functional.conv1d(xd.permute(0, 2, 1), wd, bd, padding=dilation * (taps // 2), dilation=dilation).permute(0, 2, 1)
    (ref * u.double().cpu()).sum().backward()
    for got, want in ((y, ref), (x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        assert float((got.detach().double().cpu() - want.detach()).abs().max()) <= 2e-5 * float(want.detach().abs().max())


@pytest.mark.gpu
def test_mstcnpp_hip_path_gradients_match_library_ops():
    """MSTCNPPFirstStage on the HIP kernels vs the same module on library ops in float64 (CPU): output and every parameter gradient."""
    from mucon_amd.core.modules.temporal import MSTCNPPFirstStage
    torch.manual_seed(5)
    m = MSTCNPPFirstStage(num_layers=5, num_f_maps=128, input_dim=256, output_dim=128, pooling_layers=(1, 3)).eval()
    ref = MSTCNPPFirstStage(num_layers=5, num_f_maps=128, input_dim=256, output_dim=128, pooling_layers=(1, 3)).double().eval()
    ref.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    m = m.cuda()
    x = torch.randn(2, 256, 203)
    u = torch.randn(2, 128, 203 // 4)
    out = m(x.cuda())
    (out * u.cuda()).sum().backward()
    want = ref(x.double())
    (want * u.double()).sum().backward()
    assert float((out.double().cpu() - want).abs().max()) <= 5e-5 * float(want.abs().max())
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert float((p.grad.double().cpu() - q.grad).abs().max()) <= 1e-4 * float(q.grad.abs().max()) + 1e-9, n


@pytest.mark.gpu
@pytest.mark.parametrize("H,groups", [(64, 32), (32, 8), (96, 32)])
def test_hidden_sizes_below_128_against_float64_oracle(H, groups):
    """WaveNetBlock with a hidden size below 128 (the reference's constructor default is 64) runs on the 128-channel kernels with
    zero-padded parameters: output of the encoder wrapper and every parameter gradient against the float64 oracle of the
    H-channel network (eval mode)."""
    from oracle import dense as od
    from mucon_amd.core.modules.temporal import WaveNetBlock
    cfg = od.EncoderConfig(in_dim=256, hidden=H, stages=[1, 2, 4, 8, 16], pooling_layers=[1, 3], last_gn_num_groups=groups)
    pn = od.seeded_params(cfg, 9)
    blk = WaveNetBlock(256, stages=cfg.stages, out_dims=H, pooling_layers=cfg.pooling_layers, dropout_rate=0.25).cuda().eval()
    sd = {k[len("ft."):]: torch.from_numpy(v) for k, v in pn.items() if k.startswith("ft.")}
    blk.load_state_dict(sd)
    gw = torch.from_numpy(pn["ft_last_gn.weight"]).cuda().requires_grad_()
    gb = torch.from_numpy(pn["ft_last_gn.bias"]).cuda().requires_grad_()
    B, T = 2, 333
    tape_np = synth.uniform_pm1(77, (B, T, 256))
    spec = blk.spec(last_gn=True, last_gn_num_groups=groups, last_relu=True, last_dropout=True, last_dropout_rate=0.25)
    z = blk.forward_time_major(torch.from_numpy(tape_np).cuda(), gw, gb, spec)
    u = torch.from_numpy(synth.uniform_pm1(78, tuple(z.shape)))
    (z * u.cuda()).sum().backward()
    pt = od.to_torch(pn, requires_grad=True)
    want = od.encoder_forward(torch.from_numpy(tape_np).double(), pt, cfg)
    (want * u.double()).sum().backward()
    assert tuple(z.shape) == tuple(want.shape) == (B, T // 4, H)
    assert float((z.double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    got = {"ft." + n: p.grad for n, p in blk.named_parameters()}
    got.update({"ft_last_gn.weight": gw.grad, "ft_last_gn.bias": gb.grad})
    for n, g in got.items():
        w = pt[n].grad
        assert tuple(g.shape) == tuple(w.shape), n
        assert float((g.double().cpu() - w).norm()) <= 2e-4 * float(w.norm()) + 1e-12, n


@pytest.mark.gpu
def test_model_with_hidden_64_trains():
    """cfg.model.ft.hidden_size = 64 end to end: forward, loss, backward, optimizer step on the GPU."""
    from test_gpu_model import make_batch
    from mucon_amd.mucon.trainers import SimpleTrainer
    cfg = update_config(get_cfg_defaults(), [], [["model.ft.hidden_size", "64"]])
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048).cuda()
    tr = SimpleTrainer(cfg, model, "cuda")
    tr.on_start_epoch(0)
    model.train()
    before = model.ft.first_conv.weight.detach().clone()
    loss, _ = tr._train_1_batch(0, make_batch(640, 5).to("cuda"))
    assert torch.isfinite(loss.main) and not torch.equal(before, model.ft.first_conv.weight)
    model.eval()
    with torch.no_grad():
        enc = model.temporal_modeling_forward(make_batch(640, 5).to("cuda").feats)
    assert tuple(enc.shape) == (1, 40, 64)


@pytest.mark.gpu
def test_unsupported_device_shapes_raise_instead_of_falling_back():
    """A device tensor never takes library ops silently: shapes the kernels are not built for raise (host tensors keep the
    cfg.system.device = "cpu" plumbing path); an input that wants a gradient is refused too (the HIP path produces none)."""
    from mucon_amd import _lib
    from mucon_amd.core.modules.temporal import MSTCNPPFirstStage, NoFt
    with pytest.raises(NotImplementedError):
        NoFt(in_chnnels=100, out_dims=128).cuda()(torch.randn(1, 100, 64, device="cuda"))
    with pytest.raises(NotImplementedError):
        NoFt(in_chnnels=256, out_dims=64).cuda()(torch.randn(1, 256, 64, device="cuda"))
    with pytest.raises(NotImplementedError):
        MSTCNPPFirstStage(num_layers=3, num_f_maps=64, input_dim=256, output_dim=64, pooling_layers=()).cuda()(torch.randn(1, 256, 64, device="cuda"))
    assert NoFt(in_chnnels=100, out_dims=128)(torch.randn(1, 100, 64)).shape == (1, 128, 64)      # host tensors: plumbing
    with pytest.raises(_lib.MuconHipError):
        NoFt(in_chnnels=256, out_dims=128).cuda()(torch.randn(1, 256, 64, device="cuda", requires_grad=True))


@pytest.mark.gpu
@pytest.mark.parametrize("pool, training, T", [(False, False, 130), (True, False, 203), (True, True, 96), (False, True, 257)])
def test_mstcn_fused_tail_against_float64(pool, training, T):
    """ops.mstcn_fuse_forward (one launch: conv_fusion over cat(a, b), ReLU, dropout, residual, max-pool -- reference
    temporal.py:196-201) against the same expression in float64 on the host, the kernel's counter-based dropout mask replayed in
    numpy; output and the gradients of all five inputs."""
    from helpers import dropout_keep_np
    from mucon_amd import ops
    B, p, seed = 2, 0.5, 0x1234567890ABCDEF
    g = torch.Generator().manual_seed(21)
    a, b, f = (torch.randn(B, T, 128, generator=g).cuda().requires_grad_() for _ in range(3))
    w = (torch.randn(128, 256, 1, generator=g) * 0.06).cuda().requires_grad_()
    bias = torch.randn(128, generator=g).cuda().requires_grad_()
    u = torch.randn(B, T // 2 if pool else T, 128, generator=g)
    y = ops.mstcn_fuse_forward(a, b, f, w, bias, p, seed, training, pool)
    (y * u.cuda()).sum().backward()
    ad, bd, fd, wd, biasd = (t.detach().double().cpu().requires_grad_() for t in (a, b, f, w, bias))
    pre = torch.nn.functional.conv1d(torch.cat((ad, bd), dim=2).permute(0, 2, 1), wd, biasd).permute(0, 2, 1)
    x = torch.relu(pre)
    if training:
        keep = torch.from_numpy(dropout_keep_np(B * T * 128, seed, 0, p).reshape(B, T, 128))
        x = x * keep.double() / (1 - p)
    s = fd + x
    ref = torch.nn.functional.max_pool1d(s.permute(0, 2, 1), 2).permute(0, 2, 1) if pool else s
    (ref * u.double()).sum().backward()
    # elements whose pre-activation is within float32 noise of the ReLU's kink may fall on either side: exclude nothing, bound loosely
    assert float((y.detach().double().cpu() - ref.detach()).abs().max()) <= 5e-5 * float(ref.detach().abs().max())
    for name, got, want in (("a", a.grad, ad.grad), ("b", b.grad, bd.grad), ("f", f.grad, fd.grad), ("w", w.grad, wd.grad), ("bias", bias.grad, biasd.grad)):
        err = float((got.double().cpu() - want).norm()) / (float(want.norm()) + 1e-30)
        assert err <= 2e-4, (name, err)
