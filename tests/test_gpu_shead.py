"""GPU: the HIP s-head (persistent biLSTM + persistent attention decoder) through the C ABI
against (1) the reference's own sequence_generation_forward outputs and gradients (tests/golden/shead_cases.npz)
and (2) oracle/shead.py in float64 on the same inputs.  Tolerances: outputs 5e-5 absolute (log-probs are O(4)),
gradients 3e-4 relative L2 against the float32 golden, 1e-4 against the float64 oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import shead_case, shead_params

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "shead_cases.npz"))
CASES = ["a", "b", "c", "d", "e"]
DEV = "cuda:0"


def _rel(a, b):
    a = a.detach().double().cpu().reshape(-1) if torch.is_tensor(a) else torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = b.detach().double().cpu().reshape(-1) if torch.is_tensor(b) else torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _hip_shead(P, enc, tf_in, steps, teacher_forcing, stop, eos, dropmask=None, bidirectional=True):
    from mucon_amd import ops
    lstm_w = [P[n] for n in ops.LSTM_STATE_NAMES[:8 if bidirectional else 4]]
    memory, h_n, c_n = ops.lstm_forward(enc, lstm_w, bidirectional=bidirectional)
    return ops.decoder_forward(memory, h_n, c_n, tf_in, [P[n] for n in ops.DECODER_STATE_NAMES], steps, teacher_forcing,
                               stop, eos, dropmask)


@pytest.mark.parametrize("case", CASES)
def test_teacher_forced_matches_reference_golden(case):
    c = shead_case(GOLD, case)
    P = {k: v.to(DEV).requires_grad_(True) for k, v in shead_params(GOLD, case).items()}
    enc = c["enc"].to(DEV).requires_grad_(True)
    logp, lens = _hip_shead(P, enc, c["tf_in"].to(DEV), c["N"] + 1, True, False, c["eos"])
    np.testing.assert_allclose(logp.detach().cpu().numpy(), GOLD[f"{case}__tf_logp"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(lens.detach().cpu().numpy(), GOLD[f"{case}__tf_lengths"], atol=5e-5, rtol=0)
    ((logp * c["R1"].to(DEV)).sum() + (lens * c["r2"].to(DEV)).sum()).backward()
    assert _rel(enc.grad, GOLD[f"{case}__tf_d_enc"]) < 3e-4
    for n in GOLD["param_names"]:
        n = str(n)
        g = P[n].grad.detach().cpu().numpy().reshape(-1)
        want = GOLD[f"{case}__tf_grad__{n}"]
        got = g if g.size <= 4096 else g[::29]
        scale = max(float(GOLD[f"{case}__tf_gnorm__{n}"]), 1e-30)
        assert np.linalg.norm(got - want) / scale < 3e-4, n
        assert abs(np.linalg.norm(g.astype(np.float64)) - scale) / scale < 3e-4, n


@pytest.mark.parametrize("case", CASES)
def test_greedy_decode_and_eos_stop_match_reference_golden(case):
    c = shead_case(GOLD, case)
    P = {k: v.to(DEV) for k, v in shead_params(GOLD, case).items()}
    with torch.no_grad():
        logp, lens = _hip_shead(P, c["enc"].to(DEV), c["tf_in"].to(DEV), 12, False, True, c["eos"])
    want = GOLD[f"{case}__greedy_logp"]
    assert tuple(logp.shape) == want.shape
    np.testing.assert_allclose(logp.cpu().numpy(), want, atol=1e-4, rtol=0)
    np.testing.assert_allclose(lens.cpu().numpy(), GOLD[f"{case}__greedy_lengths"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("Tz,N,bidir,classes,teacher,drop", [
    (1, 1, True, 48, True, False), (2, 3, True, 48, True, True), (77, 6, False, 48, True, True),
    (700, 9, True, 16, True, True), (130, 5, True, 48, False, True), (33, 30, True, 100, True, False)])
def test_against_float64_oracle(Tz, N, bidir, classes, teacher, drop):
    """Sizes the goldens do not cover (one direction, other class counts, long memories, training without teacher
    forcing, an embedding-dropout mask), against the formulas of oracle/shead.py evaluated in float64."""
    from oracle import shead
    from mucon_amd import ops
    g = torch.Generator().manual_seed(Tz * 31 + N)
    NC, ME = classes + 1, 256 if bidir else 128
    shapes = {"fs_encoder_hidden_out.weight": (128, ME), "fs_encoder_hidden_out.bias": (128,),
              "fs_encoder_cn_out.weight": (128, ME), "fs_encoder_cn_out.bias": (128,),
              "fs_decoder_attention_W1": (ME, 128), "fs_decoder_attention_l2.weight": (128, 128),
              "fs_decoder_attention_l2.bias": (128,), "fs_decoder_attention_V": (128,),
              "fs_decoder_embedding.weight": (classes + 2, 128),
              "fs_decoder_attn_combine.weight": (128, 128 + ME), "fs_decoder_attn_combine.bias": (128,),
              "fs_decoder_lstm.weight_ih_l0": (512, 128), "fs_decoder_lstm.weight_hh_l0": (512, 128),
              "fs_decoder_lstm.bias_ih_l0": (512,), "fs_decoder_lstm.bias_hh_l0": (512,),
              "fs_decoder_transcript.0.weight": (128, 128), "fs_decoder_transcript.0.bias": (128,),
              "fs_decoder_transcript.2.weight": (NC, 128), "fs_decoder_transcript.2.bias": (NC,),
              "fs_decoder_length.0.weight": (64, 128 + NC), "fs_decoder_length.0.bias": (64,),
              "fs_decoder_length.2.weight": (1, 64), "fs_decoder_length.2.bias": (1,)}
    for n in ops.LSTM_STATE_NAMES[:8 if bidir else 4]:
        shapes[n] = (512, 128) if "weight" in n else (512,)
    P64 = {}
    for n, sh in shapes.items():
        fan = sh[1] if len(sh) > 1 else 16
        P64[n] = ((torch.rand(sh, generator=g, dtype=torch.float64) * 2 - 1) * (2.0 / fan ** 0.5)).float().double()
    enc64 = (torch.rand((Tz, 128), generator=g, dtype=torch.float64) * 2 - 1).float().double()
    tf_in = torch.cat([torch.tensor([classes + 1]), torch.randint(0, classes, (N,), generator=g)])
    steps = N + 1
    mask = ((torch.rand((steps, 128), generator=g) >= 0.25).float() / 0.75) if drop else None
    R1 = torch.rand((steps, NC), generator=g, dtype=torch.float64) * 2 - 1
    r2 = torch.rand((steps,), generator=g, dtype=torch.float64) * 2 - 1

    P = {k: v.clone().requires_grad_(True) for k, v in P64.items()}
    enc = enc64.clone().requires_grad_(True)
    memory, h_n, c_n = shead.lstm(enc, P, bidirectional=bidir)
    lo, le = shead.decoder(memory, h_n, c_n, P, tf_in, steps, teacher, False, classes, mask.double() if drop else None)
    ((lo * R1).sum() + (le * r2).sum()).backward()

    Pg = {k: v.float().to(DEV).requires_grad_(True) for k, v in P64.items()}
    encg = enc64.float().to(DEV).requires_grad_(True)
    lg, leg = _hip_shead(Pg, encg, tf_in.to(DEV), steps, teacher, False, classes, mask.to(DEV) if drop else None, bidir)
    assert (lg.double().cpu() - lo.detach()).abs().max() < 5e-5
    assert (leg.double().cpu() - le.detach()).abs().max() < 5e-5
    ((lg * R1.float().to(DEV)).sum() + (leg * r2.float().to(DEV)).sum()).backward()
    assert _rel(encg.grad, enc.grad) < 1e-4
    for n in P:
        assert _rel(Pg[n].grad, P[n].grad) < 1e-4, n


def test_model_native_decoder_equals_torch_loop():
    """MuCon.forward + loss.backward with the persistent decoder kernel and with the torch decoding loop (eval mode:
    no dropout draw) give the same heads and the same parameter gradients."""
    from test_gpu_model import make_batch, seeded_value
    from mucon_amd.config import get_cfg_defaults
    from mucon_amd.mucon.models import create_model
    model = create_model(get_cfg_defaults(), num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
    model = model.cuda().eval()
    model.set_teacher_forcing(True)
    batch = make_batch(640, 5).to("cuda")
    res = []
    with torch.backends.cudnn.flags(enabled=False):
        for native in (True, False):
            model.native_decoder = native
            model.zero_grad()
            fo = model.forward(batch)
            model.loss(batch, fo).main.backward()
            res.append((fo.transcript.detach().clone(), fo.lengths.detach().clone(),
                        {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert (res[0][0] - res[1][0]).abs().max() < 5e-5
    assert (res[0][1] - res[1][1]).abs().max() < 5e-5
    assert set(res[0][2]) == set(res[1][2])
    for n in res[0][2]:
        a, b = res[0][2][n].double(), res[1][2][n].double()   # (length.2.bias: the softmax over lengths makes it 0)
        assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-6, n


def test_decoder_rejects_bad_arguments():
    from mucon_amd import ops, _lib
    P = {k: v.to(DEV) for k, v in shead_params(GOLD, "a").items()}
    params = [P[n] for n in ops.DECODER_STATE_NAMES]
    mem, h, c = torch.zeros(5, 256, device=DEV), torch.zeros(2, 128, device=DEV), torch.zeros(2, 128, device=DEV)
    with pytest.raises(ValueError):
        ops.decoder_forward(mem, h, c, torch.tensor([49, 1], device=DEV), params, 5, True, False, 48)   # too few tokens
    with pytest.raises(_lib.MuconHipError):
        ops.decoder_forward(torch.zeros(5, 512, device=DEV), h, c, torch.tensor([49], device=DEV), params, 1, True, False, 48)


@pytest.mark.parametrize("Tz,steps,teacher", [(125, 7, True), (125, 12, False), (3, 4, True), (192, 31, False), (64, 31, True)])
def test_eight_workgroup_step_kernel_against_the_one_workgroup_kernel(Tz, steps, teacher):
    """The decoder's forward step loop runs on eight workgroups with LDS-resident operand slices and three exchanges per step
    (csrc/decoder_mw.hpp) where the shape allows (ME = 256, Tz <= 192); MUCON_DEC_MW=0 keeps the one-workgroup kernel.  Both against
    each other (sums are taken in a different order: 2e-5), every saved activation the backward reads included -- checked through
    the gradients the (shared) backward kernel derives from them -- and the eight-workgroup kernel against itself bitwise (its
    exchanges add partial sums in workgroup order: results do not depend on timing)."""
    from mucon_amd import _lib
    c = shead_case(GOLD, "a")
    torch.manual_seed(Tz)
    enc = torch.randn(Tz, 128, device=DEV)
    tf_in = torch.randint(0, 48, (steps,), device=DEV)
    tf_in[0] = 49
    R1 = torch.randn(steps, 49, device=DEV)
    r2 = torch.randn(steps, device=DEV)

    def run(mw):
        _lib.set_knob("MUCON_DEC_MW", mw)
        P = {k: v.to(DEV).requires_grad_(True) for k, v in shead_params(GOLD, "a").items()}
        e = enc.clone().requires_grad_(True)
        logp, lens = _hip_shead(P, e, tf_in, steps, teacher, not teacher, c["eos"])
        n = logp.shape[0]
        ((logp * R1[:n]).sum() + (lens * r2[:n]).sum()).backward()
        return logp.detach(), lens.detach(), e.grad.detach(), {k: v.grad.detach() for k, v in P.items() if v.grad is not None}

    try:
        a, b, a2 = run(1), run(0), run(1)
    finally:
        _lib.set_knob("MUCON_DEC_MW", 1)
    assert a[0].shape == b[0].shape
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), atol=2e-5, rtol=0)
    assert _rel(a[2], b[2]) < 1e-4
    for k in a[3]:
        assert _rel(a[3][k], b[3][k]) < 1e-4, k
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1]) and torch.equal(a[2], a2[2])
    for k in a[3]:
        assert torch.equal(a[3][k], a2[3][k]), k


def test_eight_workgroup_kernels_under_concurrent_load():
    """The exchanges of the eight-workgroup decoder kernels must not depend on timing or placement: eight decodes enqueued round-robin on
    four streams while a fifth stream keeps every CU busy with large matmuls (uneven load: workgroups of one decode start at different
    times, on whatever CUs are free) give, bitwise, what each decode gives alone on an idle GPU -- outputs and parameter gradients."""
    from mucon_amd import ops
    c = shead_case(GOLD, "a")
    P = {k: v.to(DEV) for k, v in shead_params(GOLD, "a").items()}
    dec = [P[n] for n in ops.DECODER_STATE_NAMES]
    torch.manual_seed(11)
    jobs = []
    for i in range(8):
        Tz, steps = (125, 7) if i % 2 == 0 else (64 + 8 * i, 5 + i)
        tf = torch.randint(0, 48, (steps,), device=DEV)
        tf[0] = 49
        jobs.append(dict(memory=torch.randn(Tz, 256, device=DEV), hn=torch.randn(256, device=DEV), cn=torch.randn(256, device=DEV), tf=tf,
                         steps=steps, R1=torch.randn(steps, 49, device=DEV), r2=torch.randn(steps, device=DEV)))

    def run(j):
        (logp, lens), ctx = ops.run_forward(ops._DecoderFn, j["memory"], j["hn"], j["cn"], j["tf"], None, (j["steps"], True, False, c["eos"]), *dec)
        grads = ops.run_backward(ops._DecoderFn, ctx, j["R1"], j["r2"])
        return [logp, lens] + [g for g in grads if torch.is_tensor(g)]

    alone = []
    for j in jobs:
        alone.append([t.clone() for t in run(j)])
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    load = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    busy = []
    with torch.cuda.stream(load):
        for _ in range(12):
            busy.append(a @ a)
    for rep in range(3):
        res = [None] * len(jobs)
        for i, j in enumerate(jobs):
            with torch.cuda.stream(streams[i % 4]):
                res[i] = run(j)
        with torch.cuda.stream(load):
            for _ in range(6):
                busy.append(a @ a)
        torch.cuda.synchronize()
        for i, (got, want) in enumerate(zip(res, alone)):
            assert len(got) == len(want)
            for k, (g, w) in enumerate(zip(got, want)):
                assert torch.equal(g, w), (rep, i, k)
