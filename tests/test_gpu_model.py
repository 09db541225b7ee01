"""GPU: MuCon.forward / predict / loss / backward of mucon_amd (HIP hot path + PyTorch-ROCm s-head and
losses) against the outputs of the reference's own MuCon on the same seeded parameters and video
(tests/golden/model_cases.npz, made by tools/make_golden_model.py).  Tolerances: 1e-4 on y-head
outputs (fp32 kernels), 2e-4 on s-head log-probs (LSTM on GPU vs CPU), 1e-3 relative on losses and
gradient norms."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

from mucon_amd import synth

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_cases.npz"))


def seeded_value(name, shape):  # same recipe as tools/make_golden_model.py
    u = synth.uniform_pm1(zlib.crc32(name.encode()), tuple(shape))
    if name == "ft_last_gn.weight":
        return np.float32(1.0) + np.float32(0.25) * u
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return u * np.float32(2.0 ** -int(round(np.log2(np.sqrt(max(fan_in, 1))))))
    return u * np.float32(0.125)


def make_batch(T, N, C=48, seed=5):
    from mucon_amd.core.datasets import Batch
    tr = synth.transcript(seed, N, C, allow_repeats=False)
    gt = synth.segment_labels(seed + 1, T, tr)
    feats = synth.uniform_pm1(seed + 2, (1, T, 2048))
    return Batch(feats=torch.from_numpy(feats), gt_label=torch.from_numpy(gt), transcript=torch.from_numpy(tr),
                 transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]),
                 video_name="synthetic")


@pytest.mark.parametrize("case,over", [("base", []), ("arith", ["model.loss.mucon.type", "arithmetic"]),
                                       ("gauss", ["model.loss.mucon.template", "gaussian", "model.loss.mucon.overlap", "0.1"])])
def test_model_forward_loss_backward_match_reference(case, over):
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.models import create_model
    T, N = [int(x) for x in GOLD[f"{case}__meta"]]
    cfg = update_config(get_cfg_defaults(), [], [over])
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
    model = model.cuda().eval()
    model.set_teacher_forcing(True)
    batch = make_batch(T, N).to("cuda")
    # eval()-mode gradients (the golden's setting): MIOpen's fused LSTM refuses backward outside training
    # mode, so the s-head runs on torch's native LSTM for this check
    with torch.backends.cudnn.flags(enabled=False):
        fo = model.forward(batch)
        loss = model.loss(batch, fo)
        pred = model.predict(batch, fo)
        loss.main.backward()
    np.testing.assert_allclose(fo.segmentation.detach().cpu().numpy()[::7], GOLD[f"{case}__segmentation_sub"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pred.segmentation_logits.detach().cpu().numpy()[::7], GOLD[f"{case}__pred_logp_sub"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(fo.transcript.detach().cpu().numpy(), GOLD[f"{case}__transcript"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(fo.lengths.detach().cpu().numpy(), GOLD[f"{case}__lengths"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(pred.lengths.detach().cpu().numpy(), GOLD[f"{case}__pred_lengths"], rtol=2e-4, atol=1e-5)
    assert pred.transcript == batch.transcript_tf_target.cpu().tolist()
    got = np.asarray([loss.main.item(), loss.transcript_loss.item(), loss.mucon_loss.item(), loss.length_loss.item(),
                      loss.smoothing_loss.item()])
    np.testing.assert_allclose(got, GOLD[f"{case}__loss"], rtol=1e-3, atol=1e-6)
    named = dict(model.named_parameters())
    norms = np.asarray([named[str(k)].grad.norm().item() for k in GOLD["grad_names"]])
    np.testing.assert_allclose(norms, GOLD[f"{case}__grad_norms"], rtol=2e-3)


def test_train_steps_and_viterbi_eval_on_synthetic_breakfast(tmp_path):
    """Plumbing of BASELINE config 1/2 on a Breakfast-shaped synthetic tree: a few training steps
    (loss goes down), then evaluation with the Viterbi decode; the decode of every test video is
    checked bit-for-bit against the oracle on the same log-probs."""
    import oracle
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.core.datasets import handel_dataset, write_synthetic_breakfast
    from mucon_amd.mucon.evaluators import MuConEvaluator
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer
    write_synthetic_breakfast(tmp_path, n_train=6, n_test=3, t_range=(200, 500), n_range=(2, 5))
    cfg = update_config(get_cfg_defaults(), [], [["dataset.root", str(tmp_path), "trainer.learning_rate", "0.02"]])
    torch.manual_seed(0)
    train_db, test_db = handel_dataset(cfg, True), handel_dataset(cfg, False)
    model = create_model(cfg, train_db.get_num_classes(), train_db.max_transcript_length + 1, train_db.feat_dim).cuda()
    trainer = SimpleTrainer(cfg, model, "cuda", train_db)
    first = np.mean(trainer.train_epoch(0))
    for e in range(1, 4):
        last = np.mean(trainer.train_epoch(e))
    assert np.isfinite(last) and last < first, (first, last)
    ev = MuConEvaluator(cfg, test_db, model, "cuda")
    ev.viterbi_mode(True)
    model.eval()
    model.set_teacher_forcing(False)
    decoded = 0
    with torch.no_grad():
        for i in range(len(test_db)):
            batch = test_db[i].to("cuda")
            try:
                fo = model.forward(batch)
            except RuntimeError as e:   # s-head emitted EOS first: torch.stack([]) -- the reference fails the same way (models.py:351)
                assert "non-empty" in str(e)
                continue
            pred = model.predict(batch, fo)
            try:
                r = ev.batch_eval_calculation(batch, fo)
            except (AttributeError, IndexError):
                continue   # the reference raises on these inputs too (e.g. every hypothesis outlived max_length)
            decoded += 1
            transcript, lm = ev.viterbi_inputs(pred, batch.feats.shape[1])
            want = oracle.viterbi_decode_table(pred.segmentation_logits.cpu().numpy(), transcript,
                                               lm.rows_for(transcript, 30), 30, 2000)
            assert np.float64(r["viterbi_score"]).view(np.uint64) == np.float64(want[0]).view(np.uint64)
            assert r["viterbi_labels"] == want[1].tolist()
    if decoded == len(test_db):
        res = ev.evaluate()
        from mucon_amd.mucon.evaluators import RESULT_FIELDS
        assert set(res) == set(RESULT_FIELDS) and 0.0 <= res["vit_mof"] <= 1.0 and len(res["vit_f1_score"]) == 3
        assert len(ev.to_save["vit_segs"]) == len(test_db) and 0.0 <= res["s_mat_score"] <= 1.0
        ev.device_overlap = True          # y-head MoF / IoD / IoU from device counters: the same record
        res_dev = ev.evaluate()
        for k in RESULT_FIELDS:
            assert np.array_equal(np.asarray(res[k], dtype=np.float64), np.asarray(res_dev[k], dtype=np.float64), equal_nan=True), k
