"""CPU: BASELINE config 1 -- `--cfg inside.yaml --set system.device cpu trainer.num_epochs 1` (reference
src/core/config.py:16, src/train_test_mucon.py:41,51) trains an epoch and runs the final Viterbi evaluation on the host.

mucon_amd/cpu_plumbing.py is plumbing, not a fallback of the HIP path: these tests check that the command runs, that the
host decode agrees with the reference's golden decodes (a sanity check of the plumbing, not the product's parity claim --
that is tests/test_gpu_viterbi.py through the C ABI), that device tensors are refused there, and that the C-ABI wrappers
still refuse host tensors."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import f64_bits, load_viterbi_golden, viterbi_case_inputs

Z, META = load_viterbi_golden()
REF_CFG = "/root/reference/src/configs/docker/inside.yaml"


def _overlay(tmp_path, root, data):
    """The reference's docker overlay (configs/docker/inside.yaml: trainer.root, save_every, eval_every, dataset.root) with the
    two paths pointed at the temporary tree; the reference file itself is loaded first when it exists (build container)."""
    overlay = tmp_path / "inside.yaml"
    overlay.write_text(f"trainer:\n  root: {root}\n  save_every: 30\n  eval_every: 30\ndataset:\n  root: {data}\n")
    return ([REF_CFG] if os.path.exists(REF_CFG) else []) + [str(overlay)]


def test_config1_device_cpu_trains_one_epoch_and_evaluates(tmp_path):
    from mucon_amd import train_test_mucon
    from mucon_amd.core.datasets import write_synthetic_breakfast
    from mucon_amd.mucon.evaluators import RESULT_FIELDS

    data, root = tmp_path / "datasets", tmp_path / "root"
    write_synthetic_breakfast(str(data), n_train=4, n_test=2, t_range=(200, 320), n_range=(2, 4))
    argv = []
    for f in _overlay(tmp_path, root, data):
        argv += ["--cfg", f]
    argv += ["--set", "system.device", "cpu", "trainer.num_epochs", "1", "dataset.split", "1", "--exp-name", "config1"]
    res = train_test_mucon.main(argv)
    assert set(RESULT_FIELDS) <= set(res)
    run = root / "config1" / "1"
    assert (run / "config.yaml").exists() and (run / "epoch_1.pt").exists() and (run / "data_test_eval.pkl").exists()
    assert "device: cpu" in (run / "config.yaml").read_text()
    saved = json.loads((run / "results.json").read_text())
    assert abs(saved["y_mof"] - float(res["y_mof"])) < 1e-12
    sd = torch.load(run / "epoch_1.pt", map_location="cpu")["model"]
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


@pytest.mark.parametrize("cs", META["cases"][::3], ids=[c["name"] for c in META["cases"][::3]])
def test_host_decode_agrees_with_reference_goldens(cs):
    from mucon_amd import cpu_plumbing
    from mucon_amd.core.viterbi.viterbi import last_in_dict_order

    nm = cs["name"]
    lp, tr, P = viterbi_case_inputs(Z, cs), Z[f"{nm}__transcript"], Z[f"{nm}__P"]
    K, J, N = cs["T"] // META["fs"], P.shape[0], len(tr)
    nan_cols = np.isnan(P).any(axis=0)
    n_lim = int(np.argmax(nan_cols)) if nan_cols.any() else N
    force = last_in_dict_order(K, J, n_lim) if (n_lim < N or K < N) else None
    score, labels, seg_len, alive = cpu_plumbing.viterbi_decode(lp, tr, P, META["fs"], force)
    assert alive
    assert f64_bits(score) == f64_bits(Z[f"{nm}__score"][0])
    np.testing.assert_array_equal(labels, Z[f"{nm}__labels"])
    np.testing.assert_array_equal(seg_len, Z[f"{nm}__seg_len"])


def test_viterbi_surface_takes_host_tensors():
    from mucon_amd import synth
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi

    T, C, tr = 400, 48, [3, 7, 3]
    lp = torch.log_softmax(torch.from_numpy(synth.emissions(5, T, C)), dim=1)
    v = Viterbi(SingleTranscriptGrammar(tr, C), PoissonModel(np.full(C, T / 3)), frame_sampling=30)
    score, labels, segs = v.decode(lp)
    assert len(labels) == T and sum(s.length for s in segs) == T and [s.label for s in segs] == tr
    assert np.isfinite(score)


def test_plumbing_is_not_a_fallback():
    """The C-ABI wrappers refuse host tensors; the plumbing refuses device tensors (checked without a GPU on a meta-free
    stand-in: a tensor subclass that reports is_cuda)."""
    from mucon_amd import _lib, cpu_plumbing, ops

    with pytest.raises(_lib.MuconHipError):
        ops.encoder_forward(torch.zeros(1, 64, 2048), [torch.zeros(1)] * 50, ops.EncoderSpec())
    with pytest.raises(_lib.MuconHipError):
        ops.head_forward(torch.zeros(1, 8, 128), torch.zeros(48, 128, 1), torch.zeros(48), 130)

    class FakeDevice(torch.Tensor):
        @property
        def is_cuda(self):
            return True

    fake = torch.zeros(1, 8, 128).as_subclass(FakeDevice)
    with pytest.raises(cpu_plumbing.DevicePlumbingError):
        cpu_plumbing.head_forward(fake, torch.zeros(48, 128, 1), torch.zeros(48), 130)
    src = open(cpu_plumbing.__file__).read()
    assert "import oracle" not in src and "from oracle" not in src


@pytest.mark.gpu
def test_device_tensors_never_reach_the_plumbing(monkeypatch):
    """On the GPU box: a model on cuda:0 steps and evaluates with every plumbing entry point booby-trapped."""
    from mucon_amd import cpu_plumbing, synth
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.core.datasets import Batch
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
    from mucon_amd.mucon.models import create_model

    def boom(*a, **k):
        raise AssertionError("cpu_plumbing reached with device tensors")

    for name in ("wavenet_forward", "head_forward", "viterbi_decode"):
        monkeypatch.setattr(cpu_plumbing, name, boom)
    dev, T, C = "cuda:0", 300, 48
    cfg = update_config(get_cfg_defaults(), [], [])
    model = create_model(cfg, C, 8, 2048).to(dev).train()
    tr = synth.transcript(3, 3, C, allow_repeats=False)
    batch = Batch(feats=torch.randn(1, T, 2048), gt_label=torch.from_numpy(synth.segment_labels(4, T, tr)),
                  transcript=torch.from_numpy(tr), transcript_tf_input=torch.tensor([C + 1] + tr.tolist()),
                  transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name="v").to(dev)
    out = model.forward(batch)
    model.loss(batch, out).main.backward()
    v = Viterbi(SingleTranscriptGrammar(tr.tolist(), C), PoissonModel(np.full(C, T / 3)), frame_sampling=30)
    _, labels, _ = v.decode(model.predict(batch, out).segmentation_logits.detach())
    assert len(labels) == T


def test_nearest_resize_equals_torch():
    """make_same_size_interpolate (NumPy) against what the reference calls, F.interpolate(mode="nearest") (core/utils.py:34-47)."""
    from mucon_amd.mucon.evaluators import make_same_size_interpolate
    rng = np.random.default_rng(0)
    for it in range(1500):
        L, n = int(rng.integers(1, 12000)), int(rng.integers(1, 12000))
        if it % 5 == 0:
            n = max(1, L + int(rng.integers(-3, 4)))
        x = rng.integers(0, 48, L)
        want = torch.nn.functional.interpolate(torch.tensor(x[None, None]).float(), size=n, mode="nearest")[0, 0].long().numpy()
        np.testing.assert_array_equal(make_same_size_interpolate(x, n), want)


def test_alignment_evaluator_and_teacher_forcing_trainer(tmp_path):
    """The two small harness subclasses of the reference (evaluators.py:343-347, trainers.py:166-191) on the host plumbing: the alignment
    evaluator keeps the s-head teacher-forced during evaluation -- its predicted transcript IS the given one, so the transcript metrics are
    perfect whatever the weights -- and TrainerForTFExperiments switches teacher forcing off from the given epoch on."""
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.core.datasets import handel_dataset, write_synthetic_breakfast
    from mucon_amd.mucon.evaluators import MuConAlignmentEvaluator, MuConEvaluator
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer, TrainerForTFExperiments

    data, root = tmp_path / "datasets", tmp_path / "root"
    write_synthetic_breakfast(str(data), n_train=3, n_test=3, t_range=(200, 320), n_range=(2, 4))
    cfgs = _overlay(tmp_path, root, data)
    cfg = update_config(get_cfg_defaults(), cfgs, ["system.device", "cpu", "dataset.split", "1"])
    torch.manual_seed(3)
    train_db, test_db = handel_dataset(cfg, train=True), handel_dataset(cfg, train=False)
    model = create_model(cfg, num_classes=train_db.get_num_classes(), max_decoding_steps=train_db.max_transcript_length + 1,
                         input_feature_size=train_db.feat_dim)
    seen = []
    real = model.set_teacher_forcing
    model.set_teacher_forcing = lambda teacher_forcing=True: (seen.append(bool(teacher_forcing)), real(teacher_forcing))[1]

    ev = MuConAlignmentEvaluator(cfg, test_db, model, "cpu")
    assert isinstance(ev, MuConEvaluator)
    ev.viterbi_mode(True)
    res = ev.evaluate()
    assert seen[-2:] == [False, True] and model.teacher_forcing is True          # super().on_start_eval(), then forced on
    assert res.get("skipped_videos", 0) == 0
    for got, want in zip(ev.s_transcript, ev.target_transcripts):                  # teacher-forced: the decoder is fed, and scored on, the given transcript
        assert len(got) == len(want)
    plain = MuConEvaluator(cfg, test_db, model, "cpu")
    plain.on_start_eval()
    assert model.teacher_forcing is False

    tr = TrainerForTFExperiments(cfg, model, "cpu", train_db, turnoff_tf_after_epoch=2)
    assert isinstance(tr, SimpleTrainer) and tr.turnoff_tf_after_epoch == 2
    for epoch, want in ((0, True), (1, True), (2, False), (7, False)):
        tr.on_start_epoch(epoch)
        assert model.teacher_forcing is want
    assert TrainerForTFExperiments(cfg, model, "cpu").turnoff_tf_after_epoch == 1000
