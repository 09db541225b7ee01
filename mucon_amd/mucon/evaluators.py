"""Evaluation glue of the hot path (reference src/mucon/evaluators.py:121-180, 225-228): the per-video
grammar and Poisson length model built from the s-head's prediction, the Viterbi decode on the
y-head's log-probs (which stay on the device), nearest-neighbour resizing to the ground-truth length
(src/core/utils.py:34-47) and MoF (src/core/metrics/segmentation.py:16-44).  The other 20-odd
metrics of the reference's evaluator are CPU post-processing outside this round's scope."""
from typing import Iterable, List

import numpy as np
import torch

from ..core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi


def one_hot(a: np.ndarray, num_classes: int) -> np.ndarray:
    return np.eye(num_classes)[a.reshape(-1)]


def mean_lengths_from_s_head(predicted_relative_lengths: np.ndarray, transcript: List[int], feature_length: int,
                             num_classes: int) -> np.ndarray:
    """Per-class mean length from the s-head's relative lengths (reference evaluators.py:155-165)."""
    actions = one_hot(np.array(transcript), num_classes)
    lengths = np.dot(predicted_relative_lengths, actions)
    lengths *= feature_length
    k = actions.sum(0)
    k[k == 0] = 1
    lengths /= k
    lengths[lengths == 0] = 1
    return lengths


def make_same_size_interpolate(prediction: np.ndarray, target_len: int) -> np.ndarray:
    """Nearest-neighbour resize of a label array (reference core/utils.py:34-47)."""
    p = torch.tensor(np.asarray(prediction)[None, None]).float()
    return torch.nn.functional.interpolate(p, size=target_len, mode="nearest")[0, 0].long().numpy()


class MoFAccuracyMetric:
    def __init__(self, ignore_ids: Iterable[int] = ()):
        self.ignore_ids = list(ignore_ids)
        self.reset()

    def reset(self):
        self.total, self.correct = 0, 0

    def add(self, targets, predictions) -> float:
        targets, predictions = np.array(targets), np.array(predictions)
        assert len(targets) == len(predictions)
        mask = np.logical_not(np.isin(targets, self.ignore_ids))
        targets, predictions = targets[mask], predictions[mask]
        cur_total, cur_correct = len(targets), int((targets == predictions).sum())
        self.correct += cur_correct
        self.total += cur_total
        return cur_correct / cur_total if cur_total else 0.0

    def summary(self) -> float:
        return self.correct / self.total if self.total else 0.0


class MuConEvaluator:
    def __init__(self, cfg, test_db, model, device):
        self.cfg, self.test_db, self.model, self.device = cfg, test_db, model, device
        self.enable_viterbi = False
        if cfg.evaluator.viterbi.multi_length:
            raise NotImplementedError("evaluator.viterbi.multi_length is broken in the reference "
                                      "(MultiPoissonModel.score raises); not supported")
        self.vi_decoder = Viterbi(None, None, frame_sampling=30)
        bg = getattr(test_db, "background_class_ids", [0])
        self.y_mof_metric, self.vit_mof_metric = MoFAccuracyMetric(), MoFAccuracyMetric()
        self.vit_mof_nbg_metric = MoFAccuracyMetric(ignore_ids=bg)

    def viterbi_mode(self, mode=True):
        self.enable_viterbi = mode

    def viterbi_inputs(self, prediction_out, feature_length: int):
        """(transcript without EOS, PoissonModel) as reference evaluators.py:131, 147-167."""
        C = self.test_db.get_num_classes()
        transcript = prediction_out.transcript[:-1]
        lengths = mean_lengths_from_s_head(prediction_out.lengths.detach().cpu().numpy(), transcript, feature_length, C)
        with np.errstate(all="ignore"):
            return transcript, PoissonModel(lengths)

    def batch_eval_calculation(self, batch, forward_out):
        pred = self.model.predict(batch, forward_out)
        Tf = batch.feats.shape[1]
        target = batch.gt_label.detach().cpu().numpy()
        y_pred = pred.segmentation_logits.argmax(dim=1).cpu().numpy()
        self.y_mof_metric.add(target, make_same_size_interpolate(y_pred, len(target)))
        result = {"y_prediction": y_pred}
        if self.enable_viterbi:
            transcript, lm = self.viterbi_inputs(pred, Tf)
            self.vi_decoder.grammar = SingleTranscriptGrammar(transcript, self.test_db.get_num_classes())
            self.vi_decoder.length_model = lm
            self.vi_decoder.set_multi_length(False)
            score, labels, segments = self.vi_decoder.decode(pred.segmentation_logits)  # device tensor: no D2H of emissions
            vit = make_same_size_interpolate(np.array(labels), len(target))
            self.vit_mof_metric.add(target, vit)
            self.vit_mof_nbg_metric.add(target, vit)
            result.update(viterbi_score=score, viterbi_labels=labels, viterbi_segments=segments)
        return result

    @torch.no_grad()
    def evaluate(self, rank: int = 0, world_size: int = 1):
        """Test videos sharded over ranks; MoF counters are all-reduced (a few scalars)."""
        self.model.eval()
        self.model.set_teacher_forcing(False)
        for m in (self.y_mof_metric, self.vit_mof_metric, self.vit_mof_nbg_metric):
            m.reset()
        for i in range(rank, len(self.test_db), world_size):
            batch = self.test_db[i].to(self.device)
            self.batch_eval_calculation(batch, self.model.forward(batch))
        counts = torch.tensor([self.y_mof_metric.correct, self.y_mof_metric.total, self.vit_mof_metric.correct,
                               self.vit_mof_metric.total, self.vit_mof_nbg_metric.correct, self.vit_mof_nbg_metric.total],
                              dtype=torch.float64, device=self.device)
        if world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(counts)
        c = counts.cpu().numpy()
        div = lambda a, b: float(a / b) if b else 0.0  # noqa: E731
        return {"y_mof": div(c[0], c[1]), "vit_mof": div(c[2], c[3]), "vit_mof_nbg": div(c[4], c[5])}
