"""GPU: MuCon.fused_train_step (forward + loss + backward as direct calls of the autograd Functions, no graph) against
the autograd path loss.main.backward() on the same weights, video, dropout seeds: same losses and same gradients (same
kernels in the same order: differences are only where torch sums two contributions in another order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("over", [[], ["model.loss.mucon.type", "arithmetic", "model.loss.smoothing.log_softmax_before", "False"],
                                  ["model.teacher_forcing", "False"]])
def test_fused_step_equals_autograd(over):
    from test_gpu_model import make_batch, seeded_value
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    cfg = update_config(get_cfg_defaults(), [], [over])
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
    model = model.cuda().train()
    model.set_teacher_forcing(cfg.model.teacher_forcing)
    batch = make_batch(640, 5).to("cuda")
    assert model.can_fuse_step(batch)
    out = []
    for fused in (True, False):
        model.zero_grad(set_to_none=True)
        model._step = 41                 # same encoder dropout stream
        torch.manual_seed(7)             # same embedding-dropout mask
        if fused:
            loss, fo = model.fused_train_step(batch)
        else:
            fo = model.forward(batch)
            loss = model.loss(batch, fo)
            loss.main.backward()
        out.append(([loss.main.item(), loss.transcript_loss.item(), loss.length_loss.item(), loss.mucon_loss.item(),
                     loss.smoothing_loss.item()], fo.transcript.detach().clone(),
                    {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-6)
    assert torch.equal(out[0][1], out[1][1])
    assert set(out[0][2]) == set(out[1][2])
    for n in out[0][2]:
        a, b = out[0][2][n].double(), out[1][2][n].double()
        assert float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-8, n


def test_trainer_takes_the_fused_step_and_matches():
    from test_gpu_model import make_batch, seeded_value
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer
    cfg = update_config(get_cfg_defaults(), [], [["model.ft.dropout_rate", "0.0", "model.ft.last_dropout_rate", "0.0",
                                                   "model.fs.decoder.embedding_dropout", "0.0"]])
    finals = []
    for fuse in (True, False):
        model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
        model = model.cuda()
        tr = SimpleTrainer(cfg, model, "cuda")
        tr.fuse_step = fuse
        tr.on_start_epoch(0)
        model.train()
        batch = make_batch(640, 5).to("cuda")
        for it in range(3):
            tr._train_1_batch(it, batch)
        finals.append({n: p.detach().clone() for n, p in model.named_parameters()})
    for n in finals[0]:
        a, b = finals[0][n].double(), finals[1][n].double()
        assert float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-7, n


def test_fused_step_deferrals_change_no_bit():
    """(r6, ABI 8) MuCon.fused_train_step defers the y-head's slab sums into the encoder backward's first launch (mucon_head_bwd_defer) and the decoder's
    weight-gradient outer products into the LSTM backward's recurrence launch (mucon_decoder_bwd_defer), and hands out the previous step's gradient tensors
    again: the same step with all of that switched off (fused_step_deferrals = False) gives the same losses and EVERY gradient bit for bit, step after step
    (three steps each, the second and third on reused tensors)."""
    from test_gpu_model import make_batch, seeded_value
    from mucon_amd import _lib
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    cfg = update_config(get_cfg_defaults(), [], [])
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
    model = model.cuda().train()
    model.set_teacher_forcing(cfg.model.teacher_forcing)
    batches = [make_batch(640, 5).to("cuda"), make_batch(901, 7).to("cuda"), make_batch(640, 5).to("cuda")]
    runs = []
    for deferrals in (True, False):
        model.fused_step_deferrals = deferrals
        steps = []
        for k, batch in enumerate(batches):
            model.zero_grad(set_to_none=True)
            model._step = 41 + k
            torch.manual_seed(7 + k)
            loss, fo = model.fused_train_step(batch)
            torch.cuda.synchronize()
            steps.append((loss.main.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        runs.append(steps)
    lib = _lib.load()
    assert lib.mucon_decoder_bwd_flush() == 0 and lib.mucon_head_bwd_flush() == 0      # nothing left pending
    for (la, ga), (lb, gb) in zip(*runs):
        assert la == lb
        assert set(ga) == set(gb)
        for n in ga:
            assert torch.equal(ga[n], gb[n]), n
