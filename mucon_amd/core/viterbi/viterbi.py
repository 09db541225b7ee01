"""Viterbi decoder with the reference's surface (src/core/viterbi/viterbi.py:10-65):

    v = Viterbi(grammar, length_model, frame_sampling=30)
    v.grammar = SingleTranscriptGrammar(transcript, n_classes)       # as evaluators.py:148-150
    v.length_model = PoissonModel(mean_lengths)                      # as evaluators.py:167
    score, labels, segments = v.decode(log_frame_probs)              # as evaluators.py:178-180

The dynamic programme runs in the gfx950 kernels of mucon_amd/csrc/viterbi.hip through
mucon_viterbi_decode_batch (include/mucon_hip.h); results are bit-identical to the reference's
(score bits, labels, segments).  `log_frame_probs` may be a device tensor (the y-head's log-probs
never leave HBM) or, as in the reference, a float32 numpy array [T x C] (uploaded first).

Host logic kept here: building the length table rows the kernel reads, and resolving the
reference's degenerate outcomes, which depend on the iteration order of its hypothesis dict:
  * T < frame_sampling                      -> IndexError            (viterbi.py:87)
  * all hypotheses outlived max_length      -> AttributeError        (viterbi.py:147)
  * fewer columns than transcript states, or a NaN length model (mean length < 0.5,
    length_model.py:56-58): every final score is -inf / NaN and the `>=` fold of
    finalize_decoding (viterbi.py:135) returns the LAST non-NaN hypothesis in dict order with
    score -inf and a truncated transcript.  `last_in_dict_order` reproduces that order in
    closed form (verified against the literal oracle in tests/test_viterbi_host.py).
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .grammar import SingleTranscriptGrammar
from .length_model import PoissonParams


def last_in_dict_order(K: int, J: int, n_lim: int) -> Optional[Tuple[int, int]]:
    """(n, j) of the last hypothesis, in the reference's dict iteration order at the final column
    K-1, among transcript states n < n_lim; None when none is alive.

    A hypothesis is (n, k0) (state n entered at column k0; k0 = 0 for n = 0, n <= k0 <= J*n
    otherwise) and is alive at column c while c - k0 <= J-1.  decode_frame (viterbi.py:92-123)
    rebuilds the dict every column: children keep their parents' order, and the new entry of
    state n+1 is inserted right behind the stay-child of the FIRST hypothesis of state n, i.e. of
    its most recent entry.  That makes the order a depth-first pre-order of the tree
    parent(n, k0) = (n-1, min(k0-1, J*(n-1))) with later-born children first; dead hypotheses
    keep their place in the structure.  The last alive node is then: per state the OLDEST alive
    entry, and among states the one whose root path (c_1, c_2, ...) is smallest at the first
    difference, a longer path winning over its own prefix."""
    c = K - 1
    best_key, best = None, None
    for n in range(min(n_lim, K)):
        if n == 0:
            if c > J - 1:
                continue
            k0 = 0
        else:
            k0 = max(n, c - J + 1)
            if k0 > min(J * n, c):
                continue
        path = [k0]
        for m in range(n, 1, -1):  # c_{m-1} = min(c_m - 1, J*(m-1))
            path.append(min(path[-1] - 1, J * (m - 1)))
        key = tuple(-x for x in reversed(path)) if n >= 1 else ()
        if best_key is None or key > best_key:
            best_key, best = key, (n, c - k0)
    return best


class NoHypothesisError(AttributeError):
    """The decode ran out of hypotheses (empty transcript, every hypothesis past max_length, NaN length model from the first
    state on).  The reference fails with this AttributeError text in its traceback (viterbi.py:147); the subclass lets callers
    (the evaluator) catch the decoder's own degenerate outcomes and nothing else."""


class ShortSequenceError(IndexError):
    """Fewer frames than one frame_sampling step: the reference's IndexError (viterbi.py:87)."""


class Viterbi(object):
    class Segment(object):
        def __init__(self, label, length=0):
            self.label, self.length = label, length

        def __repr__(self):
            return f"Segment(label={self.label}, length={self.length})"

    def __init__(self, grammar, length_model, frame_sampling=1, max_hypotheses=np.inf):
        self.grammar = grammar
        self.length_model = length_model
        self.frame_sampling = frame_sampling
        self.max_hypotheses = max_hypotheses

    def set_multi_length(self, mode=True):  # no-op in the reference too (viterbi.py:40-41)
        pass

    # ------------------------------------------------------------------------------ host logic
    def _table(self, transcript) -> np.ndarray:
        lm, fs = self.length_model, self.frame_sampling
        max_len = lm.max_length()
        if not np.isfinite(max_len):
            raise NotImplementedError("length models without a finite max_length() are not supported by the HIP decoder")
        if hasattr(lm, "rows_for"):
            return lm.rows_for(transcript, fs)
        J = int(max_len) // fs
        P = np.empty((J, len(transcript)), dtype=np.float64)
        for j in range(J):
            for n, a in enumerate(transcript):
                P[j, n] = lm.score((j + 1) * fs, int(a))
        return P

    def _prepare(self, T: int):
        """-> (transcript, table P[J x N], force (n, j) or None).  Raises what the reference raises."""
        if not isinstance(self.grammar, SingleTranscriptGrammar):
            raise NotImplementedError("the HIP decoder implements SingleTranscriptGrammar only (the one the "
                                      "reference decodes with, evaluators.py:148-150)")
        fs = self.frame_sampling
        tr = np.asarray(self.grammar.transcript, dtype=np.int32)
        N = len(tr)
        if self._beam(N) is not None:
            raise AssertionError("_prepare is the no-beam path: decode_batch routes beams to _prepare_beam")
        if N == 0:
            raise NoHypothesisError("'NoneType' object has no attribute 'label'")  # empty transcript: no hypothesis
        if T < fs:
            raise ShortSequenceError(f"index {fs - 1} is out of bounds for axis 0 with size {T}")
        lm, P = self.length_model, None
        if isinstance(lm, PoissonParams) and lm.frame_sampling == fs and lm.params.shape[1] == N:
            # The rows are built on the device (csrc/viterbi.hip: VitTab).  What this function needs of them is which states' scores are NaN
            # (a mean length < 0.5: the reference's norms are NaN, length_model.py:56-58).  The length enters a score only through `l * log(mu)`
            # with l > 0, so a column is NaN in every row or in none -- as long as no row is cut off at -inf by max_length.
            J = lm.max_len // fs
            if J >= 1 and J * fs < lm.max_len:
                first, last = np.isnan(lm.row(0)), np.isnan(lm.row(J - 1))
                if (first == last).all():
                    P, nan_cols = lm.params, first
        if P is None:
            P = self._table(tr)
            J = P.shape[0]
            nan_cols = np.isnan(P).any(axis=0)
            if nan_cols.any() and not (np.isnan(P).all(axis=0) == nan_cols).all():
                raise NotImplementedError("length table with partially-NaN columns")
        K = T // fs
        n_lim = int(np.argmax(nan_cols)) if nan_cols.any() else N
        force = None
        if n_lim < N or K < N:
            force = last_in_dict_order(K, J, n_lim)
            if force is None:
                raise NoHypothesisError("'NoneType' object has no attribute 'label'")
        elif K > J * N:
            raise NoHypothesisError("'NoneType' object has no attribute 'label'")
        return tr, P, force

    def _beam(self, N: int) -> Optional[int]:
        """max_hypotheses as the beam the decode runs under, or None when prune() (viterbi.py:74-79) can never act: inf, 0 (Python's
        `tmp[0:-0]` is empty: nothing is ever deleted) and anything from N * J on, the most hypotheses a transcript of N states has alive."""
        mh = self.max_hypotheses
        if not np.isfinite(mh):
            return None
        if not isinstance(mh, (int, np.integer)):      # the reference slices a list with it: TypeError for a float, also an integral one
            raise TypeError("slice indices must be integers or None or have an __index__ method")
        if mh < 0:
            raise ValueError("max_hypotheses < 0 (the reference would delete its |max_hypotheses| best... lowest hypotheses every column: not supported)")
        max_len = self.length_model.max_length()
        if mh == 0 or (np.isfinite(max_len) and mh >= N * (int(max_len) // self.frame_sampling)):
            return None
        return int(mh)

    def _prepare_beam(self, T: int):
        """-> (transcript, table P[J x N]) for a decode under a beam (csrc/viterbi_beam.hip keeps the hypothesis list itself: the degenerate
        outcomes _prepare resolves on the host -- fewer columns than states, every hypothesis dead -- fall out of that list)."""
        from ... import _lib
        if not isinstance(self.grammar, SingleTranscriptGrammar):
            raise NotImplementedError("the HIP decoder implements SingleTranscriptGrammar only (the one the "
                                      "reference decodes with, evaluators.py:148-150)")
        tr = np.asarray(self.grammar.transcript, dtype=np.int32)
        N = len(tr)
        if N == 0:
            raise NoHypothesisError("'NoneType' object has no attribute 'label'")
        if T < self.frame_sampling:
            raise ShortSequenceError(f"index {self.frame_sampling - 1} is out of bounds for axis 0 with size {T}")
        P = self._table(tr)
        if np.isnan(P).any():
            raise NotImplementedError("a beam (max_hypotheses) over a length model with NaN scores: Python's sort of NaN scores is not an order")
        if self.max_hypotheses + N > _lib.VIT_BEAM_MAX_ITEMS or N > 128 or P.shape[0] > 128:
            raise NotImplementedError(f"max_hypotheses + N = {self.max_hypotheses + N}: the beam kernel holds {_lib.VIT_BEAM_MAX_ITEMS} hypotheses "
                                      f"(N <= 128 states, <= 128 length slots)")
        return tr, P

    # ------------------------------------------------------------------------------ decode
    def decode_batch(self, log_frame_probs: Sequence, transcripts: Sequence[Sequence[int]],
                     length_models: Sequence, return_exceptions: bool = False, labels_as_arrays: bool = False) -> List[tuple]:
        """Decode several videos in one kernel launch (one workgroup per video).  Each result is
        what decode() returns.  Not in the reference (it decodes one video at a time).

        return_exceptions: a video the reference's decode raises for (ShortSequenceError, NoHypothesisError) yields that
        exception object in its place instead of ending the whole batch (the batched evaluation skips such videos one by one).
        labels_as_arrays: the labels of each triple as an int32 numpy array instead of the reference's Python list (a caller that goes
        on with numpy saves building a T-element list per video and parsing it again)."""
        import torch
        from ... import _lib, ops

        fs = self.frame_sampling
        if len(log_frame_probs) and all(isinstance(lp, torch.Tensor) and not lp.is_cuda for lp in log_frame_probs):
            # cfg.system.device = "cpu" (reference core/config.py:16): the plumbing path, mucon_amd/cpu_plumbing.py
            out = []
            for lp, tr, lm in zip(log_frame_probs, transcripts, length_models):
                try:
                    out.append(self._decode_host(lp, tr, lm))
                except (ShortSequenceError, NoHypothesisError) as e:
                    if not return_exceptions:
                        raise
                    out.append(e)
            return out
        lps, trs, tabs, forces, slots, beams = [], [], [], [], [], []
        log_facts = []      # per queued video: the shared log-factorial row of its PoissonParams, or None (a host-built table)
        out: List = [None] * len(log_frame_probs)
        max_len = None
        length_models_by_slot = list(length_models)
        for i, (lp, tr, lm) in enumerate(zip(log_frame_probs, transcripts, length_models_by_slot)):
            if isinstance(lp, np.ndarray):
                lp = torch.from_numpy(np.ascontiguousarray(lp, dtype=np.float32)).cuda()
            v = Viterbi(SingleTranscriptGrammar(tr, lp.shape[1]), lm, fs, self.max_hypotheses)
            try:
                if v._beam(len(tr)) is not None:
                    t, P = v._prepare_beam(int(lp.shape[0]))
                    beams.append((i, lp, t, P, int(lm.max_length())))
                    continue
                t, P, force = v._prepare(int(lp.shape[0]))
            except (ShortSequenceError, NoHypothesisError) as e:
                if not return_exceptions:
                    raise
                out[i] = e
                continue
            ml = int(lm.max_length())
            if max_len is not None and ml != max_len:
                raise ValueError("all length models of a batch must share max_length()")
            max_len = ml
            lps.append(lp)
            trs.append(t)
            tabs.append(P)
            log_facts.append(lm.log_fact if (isinstance(lm, PoissonParams) and P is lm.params) else None)
            forces.append(force)
            slots.append(i)
        lf = None
        if lps:
            if all(f is not None for f in log_facts) and all(f is log_facts[0] or np.array_equal(f, log_facts[0]) for f in log_facts):
                lf = log_facts[0]          # every video's scores are built on the device from [3, N] parameters
            else:                          # a mixed batch: the parameter blocks become host-built tables
                tabs = [length_models_by_slot[i].rows_for(t, fs) if f is not None else P for i, t, P, f in zip(slots, trs, tabs, log_facts)]
        res = list(ops.viterbi_decode_batch(lps, trs, tabs, fs, max_len, forces, log_fact=lf)) if lps else []
        if beams:      # the reference's beam search (max_hypotheses below N * J): csrc/viterbi_beam.hip, one call per max_length
            for ml in sorted({b[4] for b in beams}):
                grp = [b for b in beams if b[4] == ml]
                rb = ops.viterbi_decode_beam([b[1] for b in grp], [b[2] for b in grp], [b[3] for b in grp], fs, ml, int(self.max_hypotheses))
                slots += [b[0] for b in grp]
                trs += [b[2] for b in grp]
                res += rb
        for i, r, t in zip(slots, res, trs):
            err = None
            if r.status == _lib.VIT_INDEX_ERROR:
                err = ShortSequenceError("frame_sampling exceeds the sequence length")
            elif r.status == _lib.VIT_NO_HYPOTHESIS:
                err = NoHypothesisError("'NoneType' object has no attribute 'label'")
            if err is not None:
                if not return_exceptions:
                    raise err
                out[i] = err
                continue
            segs = [Viterbi.Segment(int(t[s]), int(r.seg_len[s])) for s in range(r.n_seg)]
            out[i] = (r.score, r.labels if labels_as_arrays else r.labels.tolist(), segs)
        return out

    def _decode_host(self, lp, transcript, length_model):
        """One video on the host (CPU torch tensor in, the reference's triple out): plumbing only, no parity claim."""
        from ... import cpu_plumbing

        v = Viterbi(SingleTranscriptGrammar(transcript, lp.shape[1]), length_model, self.frame_sampling, self.max_hypotheses)
        if v._beam(len(transcript)) is not None:
            raise NotImplementedError("the CPU plumbing path has no beam (max_hypotheses): it exists on the device, csrc/viterbi_beam.hip")
        t, P, force = v._prepare(int(lp.shape[0]))
        score, labels, seg_len, alive = cpu_plumbing.viterbi_decode(lp.detach().numpy(), t, P, self.frame_sampling, force)
        if not alive:
            raise NoHypothesisError("'NoneType' object has no attribute 'label'")
        return score, labels, [Viterbi.Segment(int(t[s]), int(n)) for s, n in enumerate(seg_len)]

    def decode(self, log_frame_probs):
        """-> (score: np.float64, labels: list[int] of len T, segments: list of Segment(label, length))."""
        assert log_frame_probs.shape[1] == self.grammar.n_classes()
        return self.decode_batch([log_frame_probs], [self.grammar.transcript], [self.length_model])[0]
