"""GPU fuzz tests against the oracles: seeded random cases beyond the fixed lists of the other test files.  The default sizes keep the three
tests at a few seconds each; MUCON_FUZZ=k multiplies the iteration counts and MUCON_FUZZ_SEED moves the seeds (round 4 ran k = 5 .. 10 on
several seeds: 14,837 decodes without a beam, 9,028 under beams, 164 dense and 64 s-head forward / backward cases -- 0 mismatches).

  * the Viterbi decode without a beam (csrc/viterbi.hip through Viterbi.decode_batch: 1 .. 12 videos per call, so the latency kernels and the
    throughput kernels; frame_sampling 1 .. 30, 2 .. 66 length slots, fewer columns than states, integer-valued / constant / zero / Gaussian
    emissions and length scores -- ties everywhere -- and -inf entries) against the literal C oracle: score bits, labels, segments, errors;
  * the decode under the reference's beam (csrc/viterbi_beam.hip) against the oracle's literal prune();
  * encoder + y-head forward and backward against the float64 oracle at random (B, T, config) on both sides of every kernel-selection
    threshold (tests/test_gpu_dense.py's two oracle tests, called with random arguments);
  * the s-head forward and backward against the float64 formulas of oracle/shead.py at random sizes;
  * the four losses and their gradients against oracle/losses.py in float64 (and against the oracle's own float32 evaluation where float32 itself is the limit);
  * the device metric counters against the host metric classes; the batched evaluation against the one-video-at-a-time path on random test sets."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import C, f64_bits

pytestmark = pytest.mark.gpu
SCALE = max(1, int(os.environ.get("MUCON_FUZZ", "1")))
SEED = int(os.environ.get("MUCON_FUZZ_SEED", "0"))


class _Table:
    def __init__(self, P, max_len):
        self.P, self.max_len = P, max_len

    def max_length(self):
        return self.max_len

    def rows_for(self, transcript, fs):
        return self.P


def _emissions_and_table(rng, T, J, N, inf_rate):
    mode = int(rng.integers(0, 4))
    lp = (rng.integers(-3, 1, (T, C)).astype(np.float32) if mode == 0 else rng.standard_normal((T, C)).astype(np.float32) if mode == 1
          else np.full((T, C), -1.0, np.float32) if mode == 2 else np.zeros((T, C), np.float32))
    P = rng.integers(-2, 1, (J, N)).astype(np.float64) if mode != 1 else rng.standard_normal((J, N))
    if mode == 3:
        P = np.zeros((J, N))
    P[rng.random((J, N)) < inf_rate] = -np.inf
    return lp, P


def test_viterbi_decode_fuzz():
    from mucon_amd.core.viterbi import Viterbi
    from mucon_amd.core.viterbi.viterbi import NoHypothesisError, ShortSequenceError
    rng = np.random.default_rng(100 + SEED)
    n = 0
    for it in range(150 * SCALE):
        fs = int(rng.choice([1, 2, 3, 7, 30]))
        J = int(rng.integers(2, 67))
        max_len = J * fs + int(rng.integers(0, fs))
        lps, trs, lms, wants = [], [], [], []
        for v in range(int(rng.choice([1, 1, 2, 3, 8, 9, 12]))):
            N = int(rng.integers(1, 21))
            K = int(rng.integers(1, min(J * N, 400) + 1))
            T = K * fs + int(rng.integers(0, fs))
            tr = rng.integers(0, C, N).astype(np.int32)
            lp, P = _emissions_and_table(rng, T, J, N, 0.05)
            try:
                w = oracle.viterbi_decode_table(lp, tr, P, fs, max_len)
            except oracle.OracleDecodeError as e:
                w = e.status
            lps.append(torch.from_numpy(lp).cuda())
            trs.append([int(x) for x in tr])
            lms.append(_Table(P, max_len))
            wants.append(w)
        got = Viterbi(None, None, frame_sampling=fs).decode_batch(lps, trs, lms, return_exceptions=True)
        for g, w, lp in zip(got, wants, lps):
            n += 1
            where = (it, fs, J, tuple(lp.shape))
            if isinstance(w, int):
                assert isinstance(g, NoHypothesisError if w == oracle.ST_NO_HYPOTHESIS else ShortSequenceError), where
            else:
                assert not isinstance(g, Exception), (where, g)
                assert f64_bits(g[0]) == f64_bits(w[0]), (where, g[0], w[0])
                np.testing.assert_array_equal(np.asarray(g[1]), w[1], err_msg=str(where))
                assert [s.length for s in g[2]] == w[3].tolist() and [s.label for s in g[2]] == w[2].tolist(), where
    assert n >= 150 * SCALE


def test_beam_decode_fuzz():
    from mucon_amd import _lib, ops
    rng = np.random.default_rng(200 + SEED)
    n = 0
    for it in range(250 * SCALE):
        fs = int(rng.choice([1, 2, 3, 7, 30]))
        J = int(rng.integers(2, 40))
        max_len = J * fs + int(rng.integers(0, fs))
        mh = int(rng.choice([1, 2, 3, 5, 9, 20, 50, 150]))
        lps, trs, Ps, wants = [], [], [], []
        for v in range(int(rng.integers(1, 6))):
            N = int(rng.integers(1, 12))
            K = int(rng.integers(1, J * N + 2))
            T = K * fs + int(rng.integers(0, fs))
            tr = rng.integers(0, C, N).astype(np.int32)
            lp, P = _emissions_and_table(rng, T, J, N, 0.1)
            try:
                w = oracle.viterbi_decode_table(lp, tr, P, fs, max_len, max_hypotheses=mh)
            except oracle.OracleDecodeError as e:
                w = e.status
            lps.append(torch.from_numpy(lp).cuda())
            trs.append(tr)
            Ps.append(P)
            wants.append(w)
        for r, w, lp in zip(ops.viterbi_decode_beam(lps, trs, Ps, fs, max_len, mh), wants, lps):
            n += 1
            where = (it, fs, J, mh, tuple(lp.shape))
            if isinstance(w, int):
                assert w == oracle.ST_NO_HYPOTHESIS and r.status == _lib.VIT_NO_HYPOTHESIS, (where, w, r)
            else:
                assert r.status in (_lib.VIT_OK, _lib.VIT_TRUNCATED) and f64_bits(r.score) == f64_bits(w[0]), (where, r, w[0])
                np.testing.assert_array_equal(r.seg_len, w[3], err_msg=str(where))
                np.testing.assert_array_equal(r.labels, w[1], err_msg=str(where))
    assert n >= 250 * SCALE


def test_dense_forward_backward_fuzz():
    import test_gpu_dense as td
    rng = np.random.default_rng(300 + SEED)
    overs = [{}, {}, {}, {"pooling_type": "sum"}, {"leaky_relu": True}, {"last_gn": False}, {"last_relu": False}, {"last_gn_num_groups": 16}]
    for i in range(6 * SCALE):
        B = int(rng.choice([1, 1, 2, 3, 4, 8]))
        rows = int(rng.choice([rng.integers(16, 400), rng.integers(400, 4200), rng.integers(4000, 8300), rng.integers(8000, 17000),
                               rng.integers(16000, 34000)]))
        T = max(16, rows // B + int(rng.integers(0, 3)))
        over = overs[int(rng.integers(0, len(overs)))]
        td.test_forward_matches_oracle_f64(B, T, over)
        td.test_backward_matches_oracle_f64(B, T, over)


def test_shead_forward_backward_fuzz():
    """The s-head (persistent biLSTM + attention decoder, forward and backward) against the float64 formulas of oracle/shead.py at random
    memory lengths (1 .. 250: the eight-workgroup kernels up to 192 states, the one-workgroup kernels above and for one direction), transcript
    lengths, class counts, with / without teacher forcing and an embedding-dropout mask (tests/test_gpu_shead.py's oracle test, random arguments)."""
    import test_gpu_shead as ts
    rng = np.random.default_rng(400 + SEED)
    for i in range(8 * SCALE):
        Tz = int(rng.choice([rng.integers(1, 8), rng.integers(8, 130), rng.integers(120, 193), rng.integers(193, 251)]))
        N = int(rng.integers(1, 31))
        bidir = bool(rng.random() < 0.8)
        classes = int(rng.choice([7, 16, 48, 100]))
        ts.test_against_float64_oracle(Tz, N, bidir, classes, bool(rng.random() < 0.6), bool(rng.random() < 0.5))


def test_losses_fuzz():
    """The four losses and their gradients (csrc/loss.hpp) against the float64 formulation of oracle/losses.py at random (frames, segments,
    classes, config): both mucon types, the three templates, overlap, smoothing on logits / log-probs, both align_corners conventions.
    Tolerances as in tests/test_gpu_losses.py, except for the gradient of the length logits with many segments: there float32 itself is
    the limit (the softmax backward subtracts nearly equal numbers) -- the oracle's own formulas evaluated in float32, i.e. what the
    reference computes, are 2.5e-4 .. 9e-4 from float64 at 48 .. 64 segments -- so the kernels must stay within 5e-4 or twice that distance."""
    import test_gpu_losses as tl
    from oracle import losses
    rng = np.random.default_rng(500 + SEED)
    for i in range(12 * SCALE):
        T = int(rng.choice([rng.integers(2, 40), rng.integers(40, 3000), rng.integers(3000, 12000)]))
        N = int(rng.integers(1, 65))
        M = int(rng.choice([3, 16, 48, 64]))
        over = ["model.loss.mucon.type", str(rng.choice(["flint", "arithmetic"])), "model.loss.mucon.template", str(rng.choice(["box", "gaussian", "trapezoid"])),
                "model.loss.mucon.overlap", float(rng.choice([0.0, 0.1, 0.3])), "model.loss.smoothing.log_softmax_before", bool(rng.random() < 0.5),
                "model.loss.smoothing.clamp", bool(rng.random() < 0.5), "model.loss.mucon.align_corners", bool(rng.random() < 0.5)]
        ocfg = losses.LossConfig.from_overrides(over)
        g = torch.Generator().manual_seed(T + N)
        seg = ((torch.rand((T, M), generator=g) * 2 - 1) * 3).float()
        tlp = torch.log_softmax(((torch.rand((N + 1, M + 1), generator=g) * 2 - 1) * 2).double(), dim=1).float()
        ln = ((torch.rand((N,), generator=g) * 2 - 1) * 2).float()
        mt = torch.randint(0, M, (N,), generator=g)
        tt = torch.cat([mt, torch.tensor([M])])
        a = [seg.double().requires_grad_(True), tlp.double().requires_grad_(True), ln.double().requires_grad_(True)]
        want = losses.loss(ocfg, a[0], a[1], a[2], mt, tt)
        want[0].backward()
        a32 = [seg.clone().requires_grad_(True), tlp.clone().requires_grad_(True), ln.clone().requires_grad_(True)]
        losses.loss(ocfg, a32[0], a32[1], a32[2], mt, tt)[0].backward()
        b = [seg.cuda().requires_grad_(True), tlp.cuda().requires_grad_(True), ln.cuda().requires_grad_(True)]
        main, parts = tl._hip_loss(ocfg, b[0], b[1], b[2], mt.cuda(), tt.cuda())
        where = (T, N, M, over)
        np.testing.assert_allclose(np.asarray([main.item()] + parts.tolist()), [float(v.detach()) for v in want], rtol=3e-5, atol=2e-6, err_msg=str(where))
        main.backward()
        assert tl._rel(b[0].grad, a[0].grad) < 5e-4, where
        assert tl._rel(b[1].grad, a[1].grad) < 1e-5, where
        assert tl._rel(b[2].grad, a[2].grad) < max(5e-4, 2.0 * tl._rel(a32[2].grad, a[2].grad)), where


def test_metrics_fuzz():
    """The device metric counters (MoF / IoD / IoU / edit / F1 in one launch, csrc/metrics.hip) against the host metric classes (the reference's
    formulas, mucon_amd/core/metrics) on random labelling pairs: 1 .. 6,000 frames, 1 .. 48 classes, runs of 1 .. 500 frames, predictions that are
    noisy copies of the target, several ignore sets -- bit for bit."""
    from mucon_amd.core.metrics import Edit, F1Score, IoDMetric, IoUMetric, MoFAccuracyMetric
    from mucon_amd.core.metrics.device import segmental_counters
    rng = np.random.default_rng(600 + SEED)

    def labelling(T, ncls, run):
        out = []
        while len(out) < T:
            out += [int(rng.integers(0, ncls))] * int(max(1, rng.poisson(run)))
        return np.asarray(out[:T], dtype=np.int64)

    for it in range(6 * SCALE):
        ignore = [(), (0,), (0, 3), (1, 2, 5)][int(rng.integers(0, 4))]
        pairs = []
        for _ in range(int(rng.integers(1, 24))):
            T = int(rng.choice([rng.integers(1, 20), rng.integers(20, 1500), rng.integers(1500, 6000)]))
            ncls = int(rng.integers(1, 49))
            t = labelling(T, ncls, float(rng.choice([1, 3, 20, 120, 500])))
            if rng.random() < 0.5:                                   # a noisy copy: shifted boundaries, some frames re-labelled
                p = np.roll(t, int(rng.integers(-30, 31)))
                flip = rng.random(T) < 0.05
                p = np.where(flip, rng.integers(0, ncls, T), p)
            else:
                p = labelling(T, ncls, float(rng.choice([1, 3, 20, 120, 500])))
            pairs.append((t, p.astype(np.int64)))
        got = segmental_counters([torch.from_numpy(t).cuda() for t, _ in pairs], [torch.from_numpy(p).cuda() for _, p in pairs], ignore)
        for (t, p), g in zip(pairs, got):
            if g.get("over_limit"):
                continue                                             # more than 1,024 runs: the caller scores such a pair on the host
            mof, mof_i = MoFAccuracyMetric(), MoFAccuracyMetric(ignore_ids=ignore)
            mof.add(t, p), mof_i.add(t, p)
            assert (g["correct"], g["total"], g["correct_nbg"], g["total_nbg"]) == (mof.correct, mof.total, mof_i.correct, mof_i.total)
            with np.errstate(all="ignore"):
                for key, metric in (("iod", IoDMetric()), ("iou", IoUMetric()), ("iod_nbg", IoDMetric(ignore_ids=ignore)), ("iou_nbg", IoUMetric(ignore_ids=ignore))):
                    want = metric.add(targets=t, predictions=p)
                    assert (np.isnan(want) and np.isnan(g[key])) or np.float64(want).tobytes() == np.float64(g[key]).tobytes(), (key, want, g[key], len(t), ignore)
                want = Edit().add(targets=t, predictions=p)
                assert np.float64(want).tobytes() == np.float64(g["edit"]).tobytes(), (want, g["edit"], len(t))
            f1 = F1Score()
            f1.add(targets=t, predictions=p)
            assert [x[0] for x in g["f1"]] == f1.tp and [x[1] for x in g["f1"]] == f1.fp and [x[2] for x in g["f1"]] == f1.fn, (len(t), ignore)


def test_evaluation_fuzz():
    """MuConEvaluator: the batched path (pooled round trips, trimmed evaluation forward, one Viterbi launch and one metrics launch per chunk) against
    the one-video-at-a-time path on random test sets -- videos of 40 .. 5,000 frames, random model seeds and EOS biases (how soon the untrained
    s-head stops decoding: videos that end at the first word are skipped by both), chunk sizes 1 .. 7: every result field and saved list equal."""
    import test_gpu_eval_batched as te
    from mucon_amd import synth
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.core.datasets import Batch
    from mucon_amd.mucon.evaluators import RESULT_FIELDS, MuConEvaluator
    from mucon_amd.mucon.models import create_model
    rng = np.random.default_rng(700 + SEED)
    dev = "cuda:0"
    cfg = update_config(get_cfg_defaults(), [], [])
    for it in range(2 * SCALE):
        torch.manual_seed(int(rng.integers(0, 1 << 30)))
        model = create_model(cfg, C, int(rng.integers(4, 12)), 2048).to(dev)
        with torch.no_grad():
            model.fs_decoder_transcript[2].bias[C] = float(rng.choice([-20.0, -3.0, -0.5, 0.3]))
        db = te._Videos(0, C, dev, 0)
        for v in range(int(rng.integers(3, 10))):
            T = int(rng.choice([rng.integers(40, 150), rng.integers(150, 1500), rng.integers(1500, 5000)]))
            N = int(rng.integers(1, 8))
            tr = synth.transcript(int(rng.integers(0, 10 ** 6)), N, C, allow_repeats=False)
            db.items.append(Batch(feats=torch.randn(1, T, 2048), gt_label=torch.from_numpy(synth.segment_labels(int(rng.integers(0, 10 ** 6)), T, tr)),
                                  transcript=torch.from_numpy(tr), transcript_tf_input=torch.tensor([C + 1] + tr.tolist()),
                                  transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name=f"v{v}").to(dev))
        records = []
        for batched in (False, True):
            ev = MuConEvaluator(cfg, db, model, dev)
            ev.batched, ev.chunk_videos = batched, int(rng.integers(1, 8))
            ev.viterbi_mode(bool(it % 2 == 0 or rng.random() < 0.7))
            if batched:
                ev.viterbi_mode(records[0][4])
            res = ev.evaluate()
            records.append((res, ev.to_save, ev.skipped, list(ev._evaluated), ev.enable_viterbi))
        (ra, sa, ka, ea, _), (rb, sb, kb, eb, _) = records
        assert ka == kb and ea == eb
        for k in RESULT_FIELDS:
            a, b = np.asarray(ra[k], dtype=np.float64), np.asarray(rb[k], dtype=np.float64)
            assert a.shape == b.shape and all((np.isnan(x) and np.isnan(y)) or x.tobytes() == y.tobytes() for x, y in zip(a.ravel(), b.ravel())), (k, ra[k], rb[k])
        for name in sa:
            assert len(sa[name]) == len(sb[name]), name
            for x, y in zip(sa[name], sb[name]):
                np.testing.assert_array_equal(np.asarray(x), np.asarray(y), err_msg=name)


def test_head_fuzz():
    """The y-head alone (nearest upsample Tz -> Tf with torch's float32 index rule, 1x1 conv, log-softmax; forward and backward) at random
    (B, Tz, Tf, H, C): both head kernels (H % 16 == 0 and not), frame counts that are no multiple of the encoded length, Tf = Tz, one encoded
    row (tests/test_gpu_dense.py's head test, random arguments).  The index rule itself is torch's: checked against F.interpolate on the host."""
    import torch.nn.functional as F
    import test_gpu_dense as td
    rng = np.random.default_rng(800 + SEED)
    for i in range(12 * SCALE):
        Tz = int(rng.choice([1, rng.integers(2, 40), rng.integers(40, 700)]))
        Tf = int(rng.integers(Tz, 16 * Tz + 16))
        x = torch.arange(Tz, dtype=torch.float32).view(1, 1, Tz)
        idx = torch.clamp(torch.floor(torch.arange(Tf, dtype=torch.float32) * (torch.tensor(Tz, dtype=torch.float32) / torch.tensor(Tf, dtype=torch.float32))), max=Tz - 1)
        assert torch.equal(F.interpolate(x, size=Tf, mode="nearest").view(-1), idx)
        td.test_head_kernels_against_torch(int(rng.integers(1, 5)), Tz, Tf, int(rng.choice([128, 128, 48, 40, 20, 64])), int(rng.choice([3, 7, 48, 64])))


def test_optimizer_fuzz():
    """The fused clip + SGD / Adam step (csrc/optim.hpp) against clip_grad_norm_ + torch.optim at random parameter shapes (1 .. 300,000
    elements, up to 12 tensors, sizes around the kernels' 4,096-element blocks), gradient scales, momentum, clipping thresholds
    (tests/test_gpu_optim.py's two comparison tests on random shape lists; gradient scales up to 1: with 30x larger gradients and momentum an entry
    that the update carries through zero differs by 1.5e-7 absolute -- one float32 rounding of an operand of size ~1, the kernels' FMA against torch's
    multiply-then-subtract -- which is above that test's 1e-7 absolute tolerance and nothing else; Adam without weight decay: with it, one entry in 10^5 has g + wd p
    cancel to ~1e-10, where Adam's g / (|g| + eps) amplifies the summation order of the clip norm to 0.6 % of a step)."""
    import test_gpu_optim as to
    rng = np.random.default_rng(900 + SEED)
    saved = to.SIZES
    try:
        for i in range(6 * SCALE):
            sizes = []
            for _ in range(int(rng.integers(5, 13))):     # (the tests split the list at 4 and drop the gradient of entry 2)
                kind = int(rng.integers(0, 4))
                sizes.append((int(rng.integers(1, 9)),) if kind == 0 else (int(rng.integers(4090, 4100)),) if kind == 1
                             else (int(rng.integers(1, 300)), int(rng.integers(1, 1000))) if kind == 2 else (int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 4))))
            to.SIZES = sizes
            to.test_fused_clip_sgd_matches_torch(float(rng.choice([1.0, 1e-3, 0.05])), float(rng.choice([0.0, 0.9])), [100.0, 5.0, 0.5, None][int(rng.integers(0, 4))])
            to.test_fused_clip_adam_matches_torch(bool(rng.random() < 0.5), [5.0, 100.0, None][int(rng.integers(0, 3))], 0.0)
    finally:
        to.SIZES = saved
