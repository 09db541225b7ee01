"""GPU: ops.FusedClipSGD (csrc/optim.hpp: two launches) against what it replaces -- torch.nn.utils.clip_grad_norm_
per group followed by torch.optim.SGD.step() (reference src/mucon/trainers.py:137-140, :18-30).  float32 both
sides; the only difference is the summation order of the norm: 2e-6 relative."""
import pytest
import torch
from torch.nn.utils import clip_grad_norm_

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SIZES = [(1,), (7, 3), (4095,), (4096,), (4097,), (128, 2048, 1), (512, 128), (50, 128), (3, 5, 7)]


def _params(seed, scale):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in SIZES]
    grads = [(torch.randn(s, generator=g) * scale).to(DEV) for s in SIZES]
    return ps, grads


@pytest.mark.parametrize("scale,momentum,max_norm", [(1.0, 0.0, 100.0), (0.001, 0.0, 100.0), (1.0, 0.9, 5.0), (1.0, 0.0, None)])
def test_fused_clip_sgd_matches_torch(scale, momentum, max_norm):
    from mucon_amd import ops
    pa, ga = _params(1, scale)
    pb, _ = _params(1, scale)
    split = 4
    opt_a = torch.optim.SGD(pa, lr=0.01, weight_decay=0.005, momentum=momentum)
    opt_b = torch.optim.SGD(pb, lr=0.01, weight_decay=0.005, momentum=momentum)
    fused = ops.FusedClipSGD([pb[:split], pb[split:]], max_norm, opt_b)
    for step in range(3):
        for p, q, g in zip(pa, pb, ga):
            p.grad = g.clone() * (step + 1)
            q.grad = g.clone() * (step + 1)
        pb[2].grad = None   # a parameter without a gradient is skipped
        pa[2].grad = None
        norms = []
        if max_norm is not None:
            norms = [clip_grad_norm_(pa[:split], max_norm), clip_grad_norm_(pa[split:], max_norm)]
        opt_a.step()
        fused.step()
        for i, (p, q) in enumerate(zip(pa, pb)):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (step, i)
            if p.grad is not None:
                assert torch.allclose(p.grad, q.grad, rtol=2e-6, atol=1e-9), (step, i)   # clipped in place, as torch
        if norms:
            assert torch.allclose(torch.stack(norms), fused.last_norms, rtol=2e-6)


def test_fused_step_follows_the_scheduler_and_is_deterministic():
    from mucon_amd import ops
    outs = []
    for _ in range(2):
        ps, gs = _params(3, 1.0)
        opt = torch.optim.SGD(ps, lr=0.01, weight_decay=0.0)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
        fused = ops.FusedClipSGD([ps], 1e9, opt)
        before = [p.detach().clone() for p in ps]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        fused.step()
        d1 = (ps[3].detach() - before[3]).abs().max().item()
        opt.step = lambda *a, **k: None
        sched.step()
        mid = ps[3].detach().clone()
        fused.step()
        d2 = (ps[3].detach() - mid).abs().max().item()
        assert abs(d2 / d1 - 0.1) < 1e-3
        outs.append([p.detach().clone() for p in ps])
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_trainer_uses_the_fused_step_and_matches_the_torch_tail():
    """Two SimpleTrainers from the same seed, one with the fused tail, one forced onto torch's clip + SGD: same
    parameters after three steps on the same video (dropout off: eval-mode forward inside train step is not possible, so
    dropout rates are set to 0)."""
    import numpy as np
    from test_gpu_model import make_batch, seeded_value
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer
    cfg = update_config(get_cfg_defaults(), [], [["model.ft.dropout_rate", "0.0", "model.ft.last_dropout_rate", "0.0",
                                                   "model.fs.decoder.embedding_dropout", "0.0"]])
    finals = []
    for fused in (True, False):
        model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
        model = model.cuda()
        tr = SimpleTrainer(cfg, model, "cuda")
        assert tr.fused_step is not None
        if not fused:
            tr.fused_step = None
        tr.on_start_epoch(0)
        model.train()
        batch = make_batch(640, 5).to("cuda")
        for it in range(3):
            tr._train_1_batch(it, batch)
        finals.append({n: p.detach().clone() for n, p in model.named_parameters()})
    for n in finals[0]:
        a, b = finals[0][n].double(), finals[1][n].double()
        assert float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-7, n


@pytest.mark.parametrize("amsgrad,max_norm,wd", [(True, 5.0, 0.005), (False, 100.0, 0.0), (True, None, 0.005)])
def test_fused_clip_adam_matches_torch(amsgrad, max_norm, wd):
    """ops.FusedClipAdam against clip_grad_norm_ per group + torch.optim.Adam.step() (the reference's second optimizer,
    trainers.py:31-36: Adam with amsgrad=True) over five steps: parameters, moment buffers (torch's own state entries) and norms."""
    from mucon_amd import ops
    pa, ga = _params(5, 1.0)
    pb, _ = _params(5, 1.0)
    split = 4
    opt_a = torch.optim.Adam(pa, lr=0.003, weight_decay=wd, amsgrad=amsgrad)
    opt_b = torch.optim.Adam(pb, lr=0.003, weight_decay=wd, amsgrad=amsgrad)
    fused = ops.FusedClipAdam([pb[:split], pb[split:]], max_norm, opt_b)
    g = torch.Generator().manual_seed(9)
    for step in range(5):
        for p, q, g0 in zip(pa, pb, ga):
            noise = torch.randn(g0.shape, generator=g).to(DEV)
            p.grad = g0.clone() * (0.5 + step) + noise
            q.grad = p.grad.clone()
        norms = []
        if max_norm is not None:
            norms = [clip_grad_norm_(pa[:split], max_norm), clip_grad_norm_(pa[split:], max_norm)]
        opt_a.step()
        fused.step()
        for i, (p, q) in enumerate(zip(pa, pb)):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), (step, i, float((p - q).abs().max()))
            sa, sb = opt_a.state[p], opt_b.state[q]
            assert float(sa["step"]) == float(sb["step"]) == step + 1
            for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ()):
                assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=1e-7), (step, i, k)   # (lerp near a zero crossing: absolute, not relative)
        if norms:
            assert torch.allclose(torch.stack(norms), fused.last_norms, rtol=2e-6)
    # the state is torch's: a checkpoint of one loads into the other
    opt_a.load_state_dict(opt_b.state_dict())


def test_trainer_with_adam_takes_the_fused_step():
    """cfg.trainer.optimizer = "Adam": SimpleTrainer builds torch's Adam (amsgrad) and steps it through ops.FusedClipAdam."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_model import make_batch
    from mucon_amd import ops
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer
    cfg = update_config(get_cfg_defaults(), [], [["trainer.optimizer", "Adam", "trainer.learning_rate", "0.001"]])
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048).cuda()
    tr = SimpleTrainer(cfg, model, "cuda")
    assert isinstance(tr.fused_step, ops.FusedClipAdam)
    tr.on_start_epoch(0)
    model.train()
    before = model.ft.first_conv.weight.detach().clone()
    for i in range(2):
        loss, _ = tr._train_1_batch(i, make_batch(640, 5).to("cuda"))
    assert torch.isfinite(loss.main) and not torch.equal(before, model.ft.first_conv.weight)
    assert float(tr.optimizer.state[model.ft.first_conv.weight]["step"]) == 2


@pytest.mark.parametrize("optimizer", ["SGD", "Adam"])
def test_gradient_accumulation_with_the_fused_tail_matches_torch(optimizer):
    """cfg.trainer.accumulate_grad_every = 2 (reference trainers.py:113-149: zero_grad at the start of a group, the accumulated
    gradient clipped at EVERY iteration, the step at the last one): the fused tail (clip only / clip + step) against torch's
    clip_grad_norm_ + optimizer.step() on the same seeded model, four iterations, dropout off."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_model import make_batch, seeded_value
    import numpy as np
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer
    cfg = update_config(get_cfg_defaults(), [], [["trainer.optimizer", optimizer, "trainer.learning_rate", "0.002", "trainer.accumulate_grad_every", "2",
                                                   "trainer.clip_grad_norm", "True", "trainer.clip_grad_norm_value", "2.0",
                                                   "model.ft.dropout_rate", "0.0", "model.ft.last_dropout_rate", "0.0",
                                                   "model.fs.decoder.embedding_dropout", "0.0"]])
    results = []
    for fused in (True, False):
        model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
        model = model.cuda()
        tr = SimpleTrainer(cfg, model, "cuda")
        assert tr.fused_step is not None
        if not fused:
            tr.fused_step = None
        tr.fuse_step = False
        tr.on_start_epoch(0)
        model.train()
        for i in range(4):
            tr._train_1_batch(i, make_batch(400 + 37 * i, 4).to("cuda"))
        results.append({n: p.detach().clone() for n, p in model.named_parameters()})
    for n in results[0]:
        a, b = results[0][n], results[1][n]
        # (Adam divides by sqrt(v): where a gradient is rounding noise the update is +-lr whatever the optimizer, so the bound is a
        # fraction of one learning-rate step, not of the parameter)
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + (5e-6 if optimizer == "Adam" else 1e-7), n


@pytest.mark.parametrize("kind", ["sgd", "sgd_momentum", "adam"])
def test_a_non_finite_gradient_norm_is_not_applied_and_check_health_raises(kind):
    """A NaN gradient in one clipping group (what a failed decoder hand-over leaves behind: decoder_mw.hpp poisons its outputs): that
    group's parameters, gradients and optimizer state stay untouched on the device, the other group steps as usual, last_norms carries the
    NaN and ops.check_health -- the trainer's call at its stream drain -- raises.  (torch would write NaN into every parameter of the group.)"""
    from mucon_amd import _lib, ops
    ps, gs = _params(3, 1.0)
    split = 4
    if kind == "adam":
        opt = torch.optim.Adam(ps, lr=0.01, amsgrad=True)
        fused = ops.FusedClipAdam([ps[:split], ps[split:]], 100.0, opt)
    else:
        opt = torch.optim.SGD(ps, lr=0.01, weight_decay=0.005, momentum=0.9 if kind == "sgd_momentum" else 0.0)
        fused = ops.FusedClipSGD([ps[:split], ps[split:]], 100.0, opt)
    for p, g in zip(ps, gs):
        p.grad = g.clone()
    fused.step()                      # a healthy step first (creates the optimizer state)
    ops.check_health([fused])
    before = [p.detach().clone() for p in ps]
    for p, g in zip(ps, gs):
        p.grad = g.clone()
    ps[5].grad[0, 1] = float("nan")   # group 1
    grads_before = [p.grad.clone() for p in ps]
    fused.step()
    torch.cuda.synchronize()
    for i in range(split, len(ps)):
        assert torch.equal(ps[i].detach(), before[i]), i                                    # not applied
        assert torch.equal(ps[i].grad.nan_to_num(7.0), grads_before[i].nan_to_num(7.0)), i   # not scaled either
    for i in range(split):
        assert not torch.equal(ps[i].detach(), before[i]) and torch.isfinite(ps[i]).all(), i
    assert torch.isfinite(fused.last_norms[0]) and torch.isnan(fused.last_norms[1])
    # STICKY: healthy steps behind the bad one overwrite its norm, not the count of skipped steps -- a NaN step anywhere in a
    # 32-step drain window is reported, once, at the window's check
    for _ in range(3):
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        fused.step()
    assert torch.isfinite(fused.last_norms).all()
    with pytest.raises(_lib.MuconHipError, match=r"clipping group 1 met a non-finite gradient norm in 1 step\(s\)"):
        ops.check_health([fused])
    ops.check_health([fused])         # reported once: the counts were cleared
    for p, g in zip(ps, gs):          # the next healthy step goes through
        p.grad = g.clone()
    fused.step()
    ops.check_health([fused])
    assert all(torch.isfinite(p).all() for p in ps)
    # ... and the clipping-only launch of a gradient-accumulation group counts too
    for p, g in zip(ps, gs):
        p.grad = g.clone()
    ps[0].grad[0] = float("inf")      # group 0
    fused.clip_only()
    with pytest.raises(_lib.MuconHipError, match="clipping group 0"):
        ops.check_health([fused])


def test_fused_sgd_rebuilds_momentum_buffers_of_replaced_parameters():
    """ A parameter whose storage is replaced by one of another size keeps its id(): the rebuilt table must not point the kernel at
    the old, smaller momentum buffer."""
    from mucon_amd import ops
    ps, gs = _params(5, 1.0)
    opt = torch.optim.SGD(ps, lr=0.01, momentum=0.9)
    fused = ops.FusedClipSGD([ps], None, opt)
    for p, g in zip(ps, gs):
        p.grad = g.clone()
    fused.step()
    big = torch.randn(ps[2].shape[0] * 3, *ps[2].shape[1:], device=DEV)
    ps[2].data = big.clone()
    ps[2].grad = torch.ones_like(ps[2].data)
    fused.step()                       # rebuilds: fresh (zero) momentum buffer of the new size
    torch.cuda.synchronize()
    assert fused._mom[id(ps[2])].shape == ps[2].shape
    assert torch.allclose(ps[2].detach(), big - 0.01 * torch.ones_like(big))


def test_check_health_reads_the_teacher_forced_decoders_status_word():
    from mucon_amd import _lib, ops
    ops.check_health()
    ops._PENDING_DECODER_STATUS.append(torch.tensor([7], dtype=torch.int32, device=DEV))
    ops.check_health()
    assert not ops._PENDING_DECODER_STATUS
    ops._PENDING_DECODER_STATUS.extend([torch.tensor([7], dtype=torch.int32, device=DEV), torch.tensor([-1], dtype=torch.int32, device=DEV)])
    with pytest.raises(_lib.MuconHipError, match="hand-over"):
        ops.check_health()
    assert not ops._PENDING_DECODER_STATUS
