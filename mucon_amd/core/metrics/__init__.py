"""Evaluation metrics of the reference's evaluator, host-side NumPy.

  MoFAccuracyMetric, IoDMetric, IoUMetric     reference src/core/metrics/segmentation.py:16-91 (+ isba_code.py:23-109)
  Edit, F1Score                                reference src/core/metrics/fully_supervised.py:9-94 (+ mstcn_code.py:6-81)
  MatchingScoreMetric, AbsLenDiffMetric        reference src/core/metrics/transcript.py:9-50

Same class names, constructor arguments, add()/summary()/reset() behaviour and return values.  The reference walks the
frames in Python loops; here a labelling is run-length encoded once (`runs`) and the segment-pair scores are array
expressions.  Every metric also exposes its accumulator as a flat vector (`state()` / `load_state()`), so a multi-GPU
evaluation all-reduces one small tensor instead of gathering per-video lists.

Not vendored upstream: MatchingScoreMetric depends on the PyPI package edit_distance==1.0.3 (reference
requirements.txt:13), which is absent here; `matching_ratio` restates its published algorithm (Levenshtein DP that
prefers the diagonal step on cost ties, then insertion, then deletion, counting matches along the chosen path;
ratio = 2 * matches / (len(a) + len(b))) -- parity for this one metric is unpinned.
"""
from typing import Iterable, List, Sequence

import numpy as np


class Metric:
    def add(self, *args, **kwargs):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.add(*args, **kwargs)

    def summary(self):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError


def runs(labels, ignore: Iterable[int] = ()):
    """Run-length encoding of a frame labelling: (label, start, end) arrays of its segments, without the segments whose
    label is in `ignore`."""
    y = np.asarray(labels)
    if y.shape[0] == 0:
        e = np.zeros(0, dtype=np.int64)
        return y[:0], e, e
    cut = np.flatnonzero(y[1:] != y[:-1]) + 1
    starts = np.concatenate(([0], cut))
    ends = np.concatenate((cut, [y.shape[0]]))
    vals = y[starts]
    ignore = list(ignore) if ignore is not None else []
    if len(ignore):
        keep = ~np.isin(vals, ignore)
        vals, starts, ends = vals[keep], starts[keep], ends[keep]
    return vals, starts, ends


def _careful_divide(a, b, zero_value: float = 0.0):
    return zero_value if b == 0 else a / b


# ------------------------------------------------------------------------------------ frame / segment overlap
class MoFAccuracyMetric(Metric):
    def __init__(self, ignore_ids: Iterable[int] = ()):
        self.ignore_ids = list(ignore_ids)
        self.reset()

    def reset(self):
        self.total, self.correct = 0, 0

    def add(self, targets, predictions) -> float:
        assert len(targets) == len(predictions)
        targets, predictions = np.array(targets), np.array(predictions)
        mask = np.logical_not(np.isin(targets, self.ignore_ids))
        targets, predictions = targets[mask], predictions[mask]
        cur_total, cur_correct = len(targets), (targets == predictions).sum()
        self.correct += cur_correct
        self.total += cur_total
        return _careful_divide(cur_correct, cur_total)

    def summary(self) -> float:
        return _careful_divide(self.correct, self.total)

    def state(self):
        return [float(self.correct), float(self.total)]

    def load_state(self, v):
        self.correct, self.total = v[0], v[1]


class MoFAccuracyFromLogitsMetric(MoFAccuracyMetric):
    def add(self, targets, logits) -> float:
        return super().add(targets, logits.argmax(-1))


def segment_overlap_score(predictions, targets, ignore_ids=(), union: bool = False) -> float:
    """Mean over the target segments of the best same-label overlap with a predicted segment: intersection over the
    predicted segment's length (IoD) or over the union span (IoU).  nan when no target segment is left (the reference
    takes the mean of an empty array)."""
    pv, ps, pe = runs(predictions, ignore_ids)
    tv, ts, te = runs(targets, ignore_ids)
    if tv.shape[0] == 0:
        return float("nan")
    if pv.shape[0] == 0:
        return 0.0
    inter = np.minimum(pe[None, :], te[:, None]) - np.maximum(ps[None, :], ts[:, None])
    denom = (np.maximum(pe[None, :], te[:, None]) - np.minimum(ps[None, :], ts[:, None])) if union else (pe - ps)[None, :]
    score = np.where(tv[:, None] == pv[None, :], inter / denom, 0.0)
    return float(np.maximum(score.max(axis=1), 0.0).mean())


class IoDMetric(Metric):
    union = False

    def __init__(self, ignore_ids: Iterable[int] = ()):
        self.ignore_ids = list(ignore_ids)
        self.reset()

    def reset(self):
        self.values = []

    def add(self, targets, predictions) -> float:
        assert len(targets) == len(predictions)
        result = segment_overlap_score(predictions, targets, self.ignore_ids, union=self.union)
        self.values.append(result)
        return result

    def summary(self) -> float:
        return sum(self.values) / len(self.values) if self.values else 0.0

    def state(self):
        return [float(sum(self.values)), float(len(self.values))]

    def load_state(self, v):
        self.values = [v[0] / v[1]] * int(round(v[1])) if v[1] else []


class IoUMetric(IoDMetric):
    union = True


# ------------------------------------------------------------------------------------ segmental edit / F1
def levenshtein(p: Sequence, y: Sequence) -> float:
    """Unit-cost edit distance between two label sequences (one DP row at a time)."""
    m, n = len(p), len(y)
    prev = np.arange(m + 1, dtype=np.float64)
    pa = np.asarray(p)
    for j in range(1, n + 1):
        cur = np.empty(m + 1, dtype=np.float64)
        cur[0] = j
        sub = prev[:-1] + (pa != y[j - 1])
        best = np.minimum(sub, prev[1:] + 1)       # substitution / step in y
        for i in range(1, m + 1):                   # step in p depends on the cell to the left
            cur[i] = min(best[i - 1], cur[i - 1] + 1)
        prev = cur
    return float(prev[m])


def edit_score(recognized, ground_truth, norm: bool = True, bg_class=()) -> float:
    P = runs(recognized, bg_class)[0]
    Y = runs(ground_truth, bg_class)[0]
    d = levenshtein(P, Y)
    if not norm:
        return d
    with np.errstate(all="ignore"):
        return float((1 - np.float64(d) / max(len(P), len(Y))) * 100)


def f_scores(recognized, ground_truth, overlaps: Sequence[float], bg_class=()):
    """[(tp, fp, fn)] of the segmental F1 at each IoU threshold: predicted segments claim, in order, the ground-truth
    segment of the same label they overlap most (first one on ties); a segment can be claimed once.  The segment-pair IoU
    matrix is built once for all thresholds."""
    pv, ps, pe = runs(recognized, bg_class)
    yv, ys, ye = runs(ground_truth, bg_class)
    P, Y = pv.shape[0], yv.shape[0]
    if Y == 0:   # the reference's argmax over an empty array raises here; count false positives instead
        return [(0.0, float(P), 0.0) for _ in overlaps]
    if P == 0:
        return [(0.0, 0.0, float(Y)) for _ in overlaps]
    inter = np.minimum(pe[:, None], ye[None, :]) - np.maximum(ps[:, None], ys[None, :])
    union = np.maximum(pe[:, None], ye[None, :]) - np.minimum(ps[:, None], ys[None, :])
    iou = (1.0 * inter / union) * (pv[:, None] == yv[None, :])
    best = iou.argmax(axis=1)
    best_val = iou[np.arange(P), best].tolist()
    best = best.tolist()
    out = []
    for ov in overlaps:
        hits, tp = set(), 0
        for k, v in zip(best, best_val):
            if v >= ov and k not in hits:
                tp += 1
                hits.add(k)
        out.append((float(tp), float(P - tp), float(Y - len(hits))))
    return out


def f_score(recognized, ground_truth, overlap: float, bg_class=()):
    return f_scores(recognized, ground_truth, [overlap], bg_class)[0]


class Edit(Metric):
    def __init__(self, ignore_ids: Iterable[int] = ()):
        self.ignore_ids = list(ignore_ids)
        self.reset()

    def reset(self):
        self.values = []

    def add(self, targets: List[int], predictions: List[int]) -> float:
        score = edit_score(recognized=predictions, ground_truth=targets, bg_class=self.ignore_ids)
        self.values.append(score)
        return score

    def summary(self) -> float:
        return float(np.array(self.values).mean()) if self.values else 0.0

    def state(self):
        return [float(np.sum(self.values)) if self.values else 0.0, float(len(self.values))]

    def load_state(self, v):
        self.values = [v[0] / v[1]] * int(round(v[1])) if v[1] else []


class F1Score(Metric):
    def __init__(self, overlaps: Sequence[float] = (0.1, 0.25, 0.5), ignore_ids: Sequence[int] = ()):
        self.overlaps, self.ignore_ids = list(overlaps), list(ignore_ids)
        self.reset()

    def reset(self):
        n = len(self.overlaps)
        self.tp, self.fp, self.fn = [0.0] * n, [0.0] * n, [0.0] * n

    def add(self, targets: List[int], predictions: List[int]) -> List[float]:
        out = []
        for s, (tp, fp, fn) in enumerate(f_scores(predictions, targets, self.overlaps, bg_class=self.ignore_ids)):
            self.tp[s] += tp
            self.fp[s] += fp
            self.fn[s] += fn
            out.append(self.get_f1_score(tp, fp, fn))
        return out

    def summary(self) -> List[float]:
        return [self.get_f1_score(tp=self.tp[s], fp=self.fp[s], fn=self.fn[s]) for s in range(len(self.overlaps))]

    @staticmethod
    def get_f1_score(tp: float, fp: float, fn: float) -> float:
        precision = tp / (tp + fp) if tp + fp != 0.0 else 0.0
        recall = tp / (tp + fn) if tp + fp != 0.0 else 0.0
        return 2.0 * (precision * recall) / (precision + recall) * 100 if precision + recall != 0.0 else 0.0

    def state(self):
        return list(self.tp) + list(self.fp) + list(self.fn)

    def load_state(self, v):
        n = len(self.overlaps)
        self.tp, self.fp, self.fn = list(v[:n]), list(v[n:2 * n]), list(v[2 * n:3 * n])


# ------------------------------------------------------------------------------------ transcripts
def matching_ratio(a: Sequence, b: Sequence) -> float:
    """edit_distance.SequenceMatcher(a=a, b=b).ratio() (edit_distance 1.0.3, restated; see the module docstring)."""
    na, nb = len(a), len(b)
    cost = [[0] * (nb + 1) for _ in range(na + 1)]
    match = [[0] * (nb + 1) for _ in range(na + 1)]
    for i in range(1, na + 1):
        cost[i][0] = i
    for j in range(1, nb + 1):
        cost[0][j] = j
    for i in range(1, na + 1):
        for j in range(1, nb + 1):
            eq = a[i - 1] == b[j - 1]
            sc, ic, dc = cost[i - 1][j - 1] + (0 if eq else 1), cost[i][j - 1] + 1, cost[i - 1][j] + 1
            lo = min(sc, ic, dc)
            if lo == sc:
                cost[i][j], match[i][j] = sc, match[i - 1][j - 1] + (1 if eq else 0)
            elif lo == ic:
                cost[i][j], match[i][j] = ic, match[i][j - 1]
            else:
                cost[i][j], match[i][j] = dc, match[i - 1][j]
    return 2.0 * match[na][nb] / (na + nb)      # ZeroDivisionError for two empty transcripts, as upstream


class MatchingScoreMetric(Metric):
    def __init__(self):
        self.reset()

    def reset(self):
        self.values = []

    def add(self, target_transcript: List[int], predicted_transcript: List[int]) -> float:
        score = matching_ratio(list(target_transcript), list(predicted_transcript))
        self.values.append(score)
        return score

    def summary(self) -> float:
        return float(np.array(self.values).mean())

    def state(self):
        return [float(np.sum(self.values)) if self.values else 0.0, float(len(self.values))]

    def load_state(self, v):
        self.values = [v[0] / v[1]] * int(round(v[1])) if v[1] else []


class AbsLenDiffMetric(MatchingScoreMetric):
    def add(self, target_transcript: List[int], predicted_transcript: List[int]) -> float:
        score = abs(len(predicted_transcript) - len(target_transcript))
        self.values.append(score)
        return score
