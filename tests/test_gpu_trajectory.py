"""GPU: the training loop end to end against a trajectory of the REFERENCE's own loop (tests/golden/trajectory.npz, made by
tools/make_golden_trajectory.py: the reference's SimpleTrainer._train_1_batch -- forward, MuCon.loss, backward, the two
clip_grad_norm_ calls, SGD.step -- for 24 optimizer steps over three synthetic videos with every dropout rate 0, then the
evaluator's Viterbi decode of every video with the trained weights).

Real Breakfast is not in the container, so BASELINE config 2's vit_mof parity cannot be measured; this is what stands in for
it: every step's five MuConLoss fields (1e-3 relative), parameter norms after training (1e-4), the teacher-forced s-head
outputs, the y-head log-probs and the Viterbi labelling of every video (equal).  A 1e-6 relative perturbation of the initial
weights moves the reference's own losses by 2e-7 over these 24 steps (longer / hotter runs are chaotic: 36 steps at twice the
learning rate carry the same perturbation to 2e-2) -- tools/make_golden_trajectory.py, TRAJ_PERTURB.

Both routes through the product run: MuCon.fused_train_step (one straight line of HIP launches + fused clip/SGD) and
forward / loss / backward through torch.autograd with torch's clip_grad_norm_ and SGD."""
import os

import numpy as np
import pytest
import torch

from mucon_amd import synth
from test_gpu_model import seeded_value

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "trajectory.npz"))


def make_batch(T, N, seed, C=48):   # the recipe of tools/make_golden_trajectory.py
    from mucon_amd.core.datasets import Batch
    tr = synth.transcript(seed, N, C, allow_repeats=False)
    gt = synth.segment_labels(seed + 1, T, tr)
    feats = synth.uniform_pm1(seed + 2, (1, T, 2048)).copy()
    onehot = np.zeros((T, 64), dtype=np.float32)
    onehot[np.arange(T), gt % 64] = 1.0
    feats[0, :, :64] += 2.0 * onehot
    return Batch(feats=torch.from_numpy(feats), gt_label=torch.from_numpy(gt), transcript=torch.from_numpy(tr),
                 transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]),
                 video_name=f"synthetic{seed}")


@pytest.mark.parametrize("fused", [True, False], ids=["fused_step", "autograd"])
def test_training_trajectory_and_final_decode_match_the_reference_loop(fused):
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.mucon.evaluators import MuConEvaluator
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer

    steps, lr = int(GOLD["steps"]), float(GOLD["lr"])
    cfg = update_config(get_cfg_defaults(), [], [["model.ft.dropout_rate", "0.0", "model.ft.last_dropout_rate", "0.0",
                                                   "model.fs.decoder.embedding_dropout", "0.0", "trainer.learning_rate", str(lr)]])
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
    model = model.cuda()
    trainer = SimpleTrainer(cfg, model, "cuda")
    trainer.fuse_step = fused
    if not fused:
        trainer.fused_step = None       # torch's clip_grad_norm_ x 2 + SGD.step
    trainer.on_start_epoch(0)
    model.train()
    batches = [make_batch(*[int(x) for x in v]).to("cuda") for v in GOLD["videos"]]
    losses = []
    for i in range(steps):
        loss, _ = trainer._train_1_batch(i, batches[i % len(batches)])
        losses.append([float(loss.main), float(loss.transcript_loss), float(loss.mucon_loss), float(loss.length_loss),
                       float(loss.smoothing_loss)])
    got, want = np.asarray(losses), GOLD["losses"]
    worst = np.abs(got - want) / np.maximum(np.abs(want), 1e-2)
    print("largest relative loss deviation per field over the trajectory:", worst.max(0))
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5)
    named = dict(model.named_parameters())
    norms = np.asarray([named[str(k)].detach().double().norm().item() for k in GOLD["param_names"]])
    np.testing.assert_allclose(norms, GOLD["param_norms"], rtol=1e-4)

    # the evaluator's decode with the trained weights (teacher-forced s-head, as the golden: see the generator)
    class DB:
        background_class_ids = [0]

        def get_num_classes(self):
            return 48

        def __len__(self):
            return len(batches)

        def __getitem__(self, i):
            return batches[i]

    ev = MuConEvaluator(cfg, DB(), model, "cuda")
    ev.viterbi_mode(True)
    model.eval()
    model.set_teacher_forcing(True)
    with torch.no_grad():
        for v, batch in enumerate(batches):
            fo = model.forward(batch)
            pred = model.predict(batch, fo)
            assert list(pred.transcript) == GOLD[f"v{v}__transcript"].tolist()
            np.testing.assert_allclose(fo.transcript.cpu().numpy(), GOLD[f"v{v}__transcript_logp"], rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(pred.lengths.cpu().numpy(), GOLD[f"v{v}__rel_lengths"], rtol=1e-3, atol=1e-5)
            np.testing.assert_allclose(pred.segmentation_logits.cpu().numpy()[::5], GOLD[f"v{v}__logp_sub"], rtol=1e-3, atol=2e-3)
            r = ev.batch_eval_calculation(batch, fo)
            assert np.array_equal(np.asarray(r["viterbi_labels"], dtype=np.int32), GOLD[f"v{v}__viterbi_labels"]), \
                f"video {v}: {(np.asarray(r['viterbi_labels']) != GOLD[f'v{v}__viterbi_labels']).sum()} frames differ"
            np.testing.assert_allclose(r["viterbi_score"], float(GOLD[f"v{v}__viterbi_score"]), rtol=1e-3)
