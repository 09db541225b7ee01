"""Training-trajectory golden from the reference (build container only): the reference's own SimpleTrainer._train_1_batch
(src/mucon/trainers.py:108-155 -- forward, MuCon.loss, backward, the two clip_grad_norm_ calls, SGD.step) driven for
STEPS optimizer steps over three synthetic videos, every dropout rate 0 (so that the trajectory is a deterministic
function of the seeded parameters), then the evaluator's Viterbi decode (src/mucon/evaluators.py:121-180) of every video
with the trained weights.

Real Breakfast is not in this container, so BASELINE config 2's `vit_mof` parity cannot be measured; this pins the
same loop end to end instead: per-step MuConLoss fields, parameter norms at the end, the final s-head transcripts and
Viterbi labellings.  tests/test_gpu_trajectory.py replays it through mucon_amd's SimpleTrainer on the GPU.

The reference runs under the grid convention of its pinned PyTorch 1.1 (ref_harness.set_grid_convention(True), the
default of cfg.model.loss.mucon.align_corners here)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402

ref_harness.install()
from make_golden_model import seeded_value  # noqa: E402
from mucon_amd import synth  # noqa: E402

STEPS = int(os.environ.get("TRAJ_STEPS", "24"))
VIDEOS = [(420, 4, 11), (333, 3, 21), (510, 5, 31)]     # (T, N, seed)
LR = float(os.environ.get("TRAJ_LR", "0.01"))
PERTURB = float(os.environ.get("TRAJ_PERTURB", "0"))
OUT = os.environ.get("TRAJ_OUT", os.path.join(ROOT, "tests", "golden", "trajectory.npz"))


def make_batch(T, N, seed, C=48):
    """A video whose features carry its labels: class-dependent offsets on a few channels (so that 24 steps move the losses)."""
    from core.datasets.general_dataset import Batch
    tr = synth.transcript(seed, N, C, allow_repeats=False)
    gt = synth.segment_labels(seed + 1, T, tr)
    feats = synth.uniform_pm1(seed + 2, (1, T, 2048)).copy()
    onehot = np.zeros((T, 64), dtype=np.float32)
    onehot[np.arange(T), gt % 64] = 1.0
    feats[0, :, :64] += 2.0 * onehot
    return Batch(feats=torch.from_numpy(feats), gt_label=torch.from_numpy(gt), transcript=torch.from_numpy(tr),
                 transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]),
                 video_name=f"synthetic{seed}")


def main():
    from configs.mucon.default import get_cfg_defaults
    from core.viterbi.grammar import SingleTranscriptGrammar
    from core.viterbi.length_model import PoissonModel
    from core.viterbi.viterbi import Viterbi
    from mucon.models import create_model
    from mucon.trainers import SimpleTrainer

    cfg = get_cfg_defaults()
    cfg.model.ft.dropout_rate = 0.0
    cfg.model.ft.last_dropout_rate = 0.0
    cfg.model.fs.decoder.embedding_dropout = 0.0
    cfg.trainer.learning_rate = LR
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
            if PERTURB:      # sensitivity probe (TRAJ_PERTURB): how far do rounding-sized differences carry?
                p.mul_(1.0 + PERTURB * (torch.rand(p.shape, generator=torch.Generator().manual_seed(1)) * 2 - 1))
    trainer = object.__new__(SimpleTrainer)        # fandak's Trainer.__init__ (run folders, tensorboard) is not the loop under test
    trainer.cfg, trainer.model, trainer.device = cfg, model, "cpu"
    trainer.optimizer = trainer.figure_optimizer()
    trainer.clip_grad_norm = trainer.figure_clip_grad_norm()
    trainer.iter_num = 0
    trainer.on_start_batch = lambda *a, **k: None
    trainer.on_finish_batch = lambda *a, **k: None
    trainer.on_start_epoch(0)
    model.train()
    batches = [make_batch(*v) for v in VIDEOS]
    losses = []
    for i in range(STEPS):
        loss, _ = trainer._train_1_batch(i, batches[i % len(batches)])
        losses.append([loss.main.item(), loss.transcript_loss.item(), loss.mucon_loss.item(), loss.length_loss.item(),
                       loss.smoothing_loss.item()])
        trainer.iter_num += 1
        print(i, ["%.5f" % x for x in losses[-1]])
    out = {"losses": np.asarray(losses, dtype=np.float64), "videos": np.asarray(VIDEOS), "steps": np.asarray(STEPS),
           "lr": np.asarray(LR)}
    names = ["ft.first_conv.weight", "ft.l_0.dilated_conv.weight", "ft.l_10.conv_1x1.weight", "ft_last_gn.weight",
             "conv_classifier.weight", "fs_encoder_lstm.weight_hh_l0", "fs_decoder_transcript.2.weight"]
    named = dict(model.named_parameters())
    out["param_names"] = np.asarray(names)
    out["param_norms"] = np.asarray([named[k].detach().double().norm().item() for k in names])

    # evaluation of every video with the trained weights, as evaluators.py:121-180 does it -- except that the s-head is teacher
    # forced (the reference's evaluator decodes greedily, evaluators.py:316-318; after a few dozen steps a greedy decode still
    # emits EOS first, on which the reference fails in torch.stack([]), models.py:351)
    model.eval()
    model.set_teacher_forcing(True)
    decoder = Viterbi(None, None, frame_sampling=30)
    with torch.no_grad():
        for v, batch in enumerate(batches):
            fo = model.forward(batch)
            pred = model.predict(batch, fo)
            transcript = pred.transcript[:-1]
            Tf = batch.feats.shape[1]
            actions = np.eye(48)[np.array(transcript).reshape(-1)]
            lengths = np.dot(pred.lengths.detach().numpy(), actions)
            lengths *= Tf
            k = actions.sum(0)
            k[k == 0] = 1
            lengths /= k
            lengths[lengths == 0] = 1
            decoder.grammar = SingleTranscriptGrammar(transcript, 48)
            decoder.length_model = PoissonModel(lengths)
            decoder.set_multi_length(False)
            score, labels, segments = decoder.decode(pred.segmentation_logits.numpy())
            out[f"v{v}__transcript"] = np.asarray(pred.transcript)
            out[f"v{v}__transcript_logp"] = fo.transcript.detach().numpy()
            out[f"v{v}__rel_lengths"] = pred.lengths.detach().numpy()
            out[f"v{v}__logp_sub"] = pred.segmentation_logits.numpy()[::5]
            out[f"v{v}__y_argmax"] = pred.segmentation_logits.argmax(dim=1).numpy().astype(np.int32)
            out[f"v{v}__viterbi_labels"] = np.asarray(labels, dtype=np.int32)
            out[f"v{v}__viterbi_score"] = np.asarray(score, dtype=np.float64)
            out[f"v{v}__gt"] = batch.gt_label.numpy().astype(np.int32)
            mof = float((np.asarray(labels) == batch.gt_label.numpy()).mean())
            print(f"video {v}: transcript {pred.transcript} target {batch.transcript.tolist()} viterbi score {score:.3f} MoF {mof:.3f}")
    np.savez_compressed(OUT, **out)


if __name__ == "__main__":
    main()
