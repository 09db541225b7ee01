"""Evaluation glue of the hot path (reference src/mucon/evaluators.py:121-180, 225-228): the per-video
grammar and Poisson length model built from the s-head's prediction, the Viterbi decode on the
y-head's log-probs (which stay on the device), nearest-neighbour resizing to the ground-truth length
(src/core/utils.py:34-47) and the reference's full metric set (y-head, s-head and Viterbi variants of MoF / IoD /
IoU / edit / F1, transcript matching score and length difference: evaluators.py:84-111, 197-244, 268-296), computed
by mucon_amd/core/metrics."""
from typing import Iterable, List

import numpy as np
import torch

from .models import EmptyTranscriptError
from ..core.metrics import (AbsLenDiffMetric, Edit, F1Score, IoDMetric, IoUMetric, MatchingScoreMetric,  # noqa: F401
                            MoFAccuracyMetric)
from ..core.viterbi import (NoHypothesisError, PoissonModel, ShortSequenceError, SingleTranscriptGrammar, Viterbi,
                            poisson_params_for_many)


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
    """Nearest-neighbour resize of a label array (reference core/utils.py:34-47: F.interpolate(mode="nearest") on the float
    copy).  torch's rule, in ITS arithmetic: source index = min(floor(i * scale), L - 1) with scale = float32(L) / float32(n)
    and the product in float32 -- restated in NumPy (a torch CPU call costs 4x as much per video); equality with
    F.interpolate over thousands of (L, n) pairs is tests/test_cpu_plumbing.py::test_nearest_resize_equals_torch."""
    p = np.asarray(prediction)
    L, n = int(p.shape[0]), int(target_len)
    scale = np.float32(L) / np.float32(n)
    idx = np.minimum(np.floor(np.arange(n, dtype=np.float32) * scale).astype(np.int64), L - 1)
    return p[idx].astype(np.int64)


def create_segmentation_from_segments(actions: np.ndarray, lengths: np.ndarray, n_frames: int) -> np.ndarray:
    """The s-head's own segmentation: action n repeated round(length_n * n_frames) times (reference evaluators.py:28-35)."""
    counts = np.around(lengths * n_frames).astype(int)
    counts[counts < 0] = 0
    return np.repeat(actions, counts)


# result fields of the reference's MuConEvaluatorResult (evaluators.py:38-68), in its order
RESULT_FIELDS = ("y_mof", "y_mof_nbg", "y_iod", "y_iou", "s_mof", "s_mof_nbg", "s_iod", "s_iou", "s_iod_nbg", "s_iou_nbg",
                 "s_mat_score", "s_len_diff", "vit_mof", "vit_mof_nbg", "vit_iod", "vit_iou", "vit_iod_nbg", "vit_iou_nbg",
                 "vit_edit_score", "vit_f1_score", "y_edit_score", "y_f1_score", "s_edit_score", "s_f1_score")


class MuConEvaluator:
    def __init__(self, cfg, test_db, model, device):
        self.cfg, self.test_db, self.model, self.device = cfg, test_db, model, device
        self.enable_viterbi = False
        if cfg.evaluator.viterbi.multi_length:
            raise NotImplementedError("evaluator.viterbi.multi_length is broken in the reference "
                                      "(MultiPoissonModel.score raises); not supported")
        self.vi_decoder = Viterbi(None, None, frame_sampling=30)
        # y-head MoF / IoD / IoU from the device-resident arg max and ground truth (mucon_metrics_overlap) instead of the host
        # arrays: same values bit for bit (tests/test_gpu_metrics_device.py); the host keeps edit / F1 and the saved lists
        self.device_overlap = False
        bg = getattr(test_db, "background_class_ids", [0])
        m = self.metrics = {}
        for head in ("y", "s", "vit"):
            m[f"{head}_mof"], m[f"{head}_mof_nbg"] = MoFAccuracyMetric(), MoFAccuracyMetric(ignore_ids=bg)
            m[f"{head}_iod"], m[f"{head}_iou"] = IoDMetric(), IoUMetric()
            if head != "y":
                m[f"{head}_iod_nbg"], m[f"{head}_iou_nbg"] = IoDMetric(ignore_ids=bg), IoUMetric(ignore_ids=bg)
            m[f"{head}_edit_score"], m[f"{head}_f1_score"] = Edit(), F1Score()
        m["s_mat_score"], m["s_len_diff"] = MatchingScoreMetric(), AbsLenDiffMetric()
        # the attribute names of the reference's evaluator (y_mof_metric, vit_f1_score_metric, ...)
        for k, v in m.items():
            setattr(self, f"{k}_metric", v)
        self.s_abs_len_diff_metric = m["s_len_diff"]
        self._reset_lists()

    def _reset_lists(self):
        self.y_segs, self.s_segs, self.vit_segs, self.s_lens = [], [], [], []
        self.s_transcript, self.target_segs, self.target_transcripts = [], [], []

    def viterbi_mode(self, mode=True):
        self.enable_viterbi = mode

    def viterbi_inputs(self, prediction_out, feature_length: int):
        """(transcript without EOS, PoissonModel) as reference evaluators.py:131, 147-167."""
        C = self.test_db.get_num_classes()
        transcript = prediction_out.transcript[:-1]
        lengths = mean_lengths_from_s_head(prediction_out.lengths.detach().cpu().numpy(), transcript, feature_length, C)
        with np.errstate(all="ignore"):
            return transcript, PoissonModel(lengths)

    def _add(self, head: str, target, prediction, with_nbg_overlap: bool):
        names = ["mof", "mof_nbg", "iod", "iou", "edit_score", "f1_score"] + (["iod_nbg", "iou_nbg"] if with_nbg_overlap else [])
        with np.errstate(all="ignore"):
            for n in names:
                self.metrics[f"{head}_{n}"](targets=target, predictions=prediction)

    def _add_y_from_device(self, target_dev, pred_dev, target, y_same):
        from ..core.metrics.device import add_to_metrics, overlap_counters
        m = self.metrics
        plain = overlap_counters([target_dev], [pred_dev], ())[0]
        add_to_metrics(plain, m["y_mof"], m["y_iod"], m["y_iou"])
        nbg = overlap_counters([target_dev], [pred_dev], m["y_mof_nbg"].ignore_ids)[0]
        add_to_metrics(nbg, m["y_mof_nbg"])
        with np.errstate(all="ignore"):
            for n in ("edit_score", "f1_score"):
                m[f"y_{n}"](targets=target, predictions=y_same)

    def batch_eval_calculation(self, batch, forward_out):
        """One test video (reference evaluators.py:121-257)."""
        pred = self.model.predict(batch, forward_out)
        Tf = batch.feats.shape[1]
        target = batch.gt_label.detach().cpu().numpy()
        target_transcript = batch.transcript.detach().cpu().numpy().tolist()
        s_transcript = pred.transcript[:-1]                       # the last word should be EOS
        rel_lengths = pred.lengths.detach().cpu().numpy()
        y_pred = pred.segmentation_logits.argmax(dim=1).cpu().numpy()
        result = {"y_prediction": y_pred}
        if self.enable_viterbi:
            transcript, lm = self.viterbi_inputs(pred, Tf)
            self.vi_decoder.grammar = SingleTranscriptGrammar(transcript, self.test_db.get_num_classes())
            self.vi_decoder.length_model = lm
            self.vi_decoder.set_multi_length(False)
            score, labels, segments = self.vi_decoder.decode(pred.segmentation_logits)  # device tensor: no D2H of emissions
            result.update(viterbi_score=score, viterbi_labels=labels, viterbi_segments=segments)
        # every per-video value exists from here on: only now do the metrics see the video (a decode that raises above leaves
        # all of them untouched, so a skipped video is skipped by every metric alike)
        self.metrics["s_mat_score"].add(target_transcript=target_transcript, predicted_transcript=s_transcript)
        self.metrics["s_len_diff"].add(target_transcript=target_transcript, predicted_transcript=s_transcript)
        s_pred = create_segmentation_from_segments(np.array(s_transcript), rel_lengths, Tf)
        s_same = make_same_size_interpolate(s_pred, len(target))
        y_same = make_same_size_interpolate(y_pred, len(target))
        self._add("s", target, s_same, True)
        if self.device_overlap and batch.gt_label.is_cuda and len(y_pred) == len(target):
            self._add_y_from_device(batch.gt_label.reshape(-1), pred.segmentation_logits.argmax(dim=1).reshape(-1), target, y_same)
        else:
            self._add("y", target, y_same, False)
        if self.enable_viterbi:
            vit_same = make_same_size_interpolate(np.array(result["viterbi_labels"]), len(target))
            self._add("vit", target, vit_same, True)
        self.vit_segs.append(vit_same if self.enable_viterbi else s_same)
        self.y_segs.append(y_same)
        self.s_segs.append(s_same)
        self.s_lens.append(rel_lengths)
        self.s_transcript.append(s_transcript)
        self.target_segs.append(target)
        self.target_transcripts.append(target_transcript)
        return result

    def on_finish_eval(self):
        """The reference's result record (evaluators.py:268-296) as a dict, plus `to_save` (evaluators.py:259-267)."""
        self.to_save = {"y_segs": self.y_segs, "s_segs": self.s_segs, "vit_segs": self.vit_segs, "s_lens": self.s_lens,
                        "s_transcript": self.s_transcript, "target_segs": self.target_segs,
                        "target_transcripts": self.target_transcripts}
        import warnings
        with np.errstate(all="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore")       # the mean of an empty list (no video evaluated) is nan, as upstream
            return {k: self.metrics[k].summary() for k in RESULT_FIELDS}

    # ------------------------------------------------------------------------------------------ batched evaluation
    forward_streams = 4          # streams the forwards of a chunk rotate over (0: all on the current stream).  One forward is ~0.18 ms of host time to enqueue and ~0.36 ms of GPU time on a stream of its own (batch-1 launches, persistent LSTM / decoder kernels on one or two CUs): measured 0 / 2 / 4 streams: 0.68 / 0.51-0.52 / 0.51-0.52 ms per video (tools/eval_repeat.py)

    def _forward_streams(self, dev):
        if self.forward_streams <= 0:
            return []
        if getattr(self, "_fw_streams", None) is None or len(self._fw_streams) != self.forward_streams:
            self._fw_streams = [torch.cuda.Stream(device=dev) for _ in range(self.forward_streams)]
        return self._fw_streams

    def _evaluate_chunk(self, idxs):
        """idxs: [(dataset index, Batch)] of videos MuCon.can_defer_eval accepts.  _evaluate_chunk_on(idxs, ...) with the chunk's forwards rotating over self.forward_streams side streams.  What they allocate
        is read on the current stream until the chunk is done; it is handed back to the side streams' pools only after those
        streams have been made to wait for the current one (no per-tensor record_stream: its deferred frees made the caching
        allocator grow the pools with hipMalloc for the first passes -- 1.2-1.6 ms per video instead of 0.55)."""
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        streams = self._forward_streams(dev)
        try:
            self._evaluate_chunk_on(idxs, cur, streams)
        finally:
            for st in streams:
                st.wait_stream(cur)

    def _evaluate_chunk_on(self, idxs, cur, streams):
        """A chunk of test videos with the host round trips of batch_eval_calculation pooled: every forward is enqueued first
        (MuCon.forward_deferred: the number of decoded words stays on the device), ONE copy fetches the transcripts of the chunk,
        ONE launch decodes all its videos (Viterbi.decode_batch), ONE launch scores all its labellings (mucon_metrics_segmental:
        MoF / IoD / IoU / edit / F1 for the y-head, the s-head and the Viterbi segmentations).  Feeds the same metric objects
        and per-video lists, in dataset order, with the same values as the per-video path (tests/test_gpu_eval_batched.py)."""
        from ..core.metrics.device import segmental_counters
        dev, model, C = self.device, self.model, self.test_db.get_num_classes()
        vids = []
        # the forwards of a chunk are independent chains of small launches (batch 1: a persistent LSTM / decoder kernel occupies one
        # or two CUs): round-robin over a few streams they overlap on the chip; the current stream waits for all of them below
        if streams:
            start = cur.record_event()
            for st in streams:
                st.wait_event(start)
        # (r6) the chunk's first forward validates the parameters (addresses, types, sizes); nothing between the forwards of a chunk can replace them
        ef = None
        if getattr(model, "fast_eval_forward", False):
            from .eval_forward import for_model
            ef = for_model(model)
        try:
            for k, (i, batch) in enumerate(idxs):
                if ef is not None:
                    ef.trusted = k > 0
                if streams:
                    with torch.cuda.stream(streams[k % len(streams)]):
                        batch = batch.to(dev)
                        out = model.forward_deferred(batch)
                else:
                    batch = batch.to(dev)
                    out = model.forward_deferred(batch)
                vids.append({"i": i, "batch": batch, "out": out})
        finally:
            if ef is not None:
                ef.trusted = False
        for st in streams:
            cur.wait_stream(st)
        # -- sync 1: how many words every video decoded, and which
        S = model.max_decoding_steps
        n_steps = torch.cat([v["out"]["n_steps"] for v in vids]).cpu().numpy()
        if (n_steps < 0).any():
            from .. import _lib, ops
            raise _lib.MuconHipError(ops.DECODER_HANDOVER_FAILED)
        alive = []
        for v, n in zip(vids, n_steps):
            n = int(n)
            if n < 2:        # EOS as the first word: the reference dies in torch.stack([]) (models.py:351)
                self.skipped += 1
                continue
            v["n"] = n
            lens = v["out"]["lengths"][:n - 1]
            e = torch.exp(lens - lens.max())                   # MuCon.predict's softmax, op for op
            v["rel_d"] = e / e.sum()
            v["words_d"] = v["out"]["transcript"][:n].argmax(dim=1)
            v["y_d"] = v["out"]["logp"].argmax(dim=1)
            alive.append(v)
        if not alive:
            return
        # -- sync 2: words, relative lengths, y-head labels and ground truth of the chunk in one copy each
        words = torch.cat([v["words_d"] for v in alive]).cpu().numpy()
        rels = torch.cat([v["rel_d"] for v in alive]).cpu().numpy()
        ys = torch.cat([v["y_d"] for v in alive]).cpu().numpy()
        gts = torch.cat([v["batch"].gt_label.reshape(-1) for v in alive]).cpu().numpy()
        wo = ro = yo = go = 0
        for v in alive:
            n, Tf, Tg = v["n"], int(v["out"]["logp"].shape[0]), int(v["batch"].gt_label.numel())
            v["transcript"] = words[wo: wo + n].tolist()
            v["rel"] = rels[ro: ro + n - 1].copy()
            v["y_pred"] = ys[yo: yo + Tf]
            v["target"] = gts[go: go + Tg]
            wo, ro, yo, go = wo + n, ro + n - 1, yo + Tf, go + Tg
        tts = torch.cat([v["batch"].transcript.reshape(-1) for v in alive]).cpu().numpy()
        to = 0
        for v in alive:
            k = int(v["batch"].transcript.numel())
            v["target_transcript"] = tts[to: to + k].tolist()
            to += k
        # -- Viterbi: one launch for the chunk (the emissions stay where the y-head wrote them)
        if self.enable_viterbi:
            # the chunk's length tables in one go (PoissonModel(lengths).rows_for(transcript, fs) per video, bit for bit: poisson_rows_for_many)
            fs = self.vi_decoder.frame_sampling
            mus = [mean_lengths_from_s_head(v["rel"], v["transcript"][:-1], int(v["out"]["logp"].shape[0]), C) for v in alive]
            # (r6) ... as [3, N] parameter blocks: the rows themselves are built on the device, bit for bit (core/viterbi/length_model.py: PoissonParams)
            lms = poisson_params_for_many(mus, [v["transcript"][:-1] for v in alive], fs, 2000)      # (2000: PoissonModel's default max_length, evaluators.py:167)
            res = self.vi_decoder.decode_batch([v["out"]["logp"] for v in alive], [v["transcript"][:-1] for v in alive], lms,
                                               return_exceptions=True, labels_as_arrays=True)
            kept = []
            for v, r in zip(alive, res):
                if isinstance(r, Exception):
                    self.skipped += 1
                    continue
                v["vit"] = r
                kept.append(v)
            alive = kept
            if not alive:
                return
        # -- the three labellings per video at the ground truth's length, scored in one launch
        preds, tgts = [], []
        for v in alive:
            s_tr, Tf, target = v["transcript"][:-1], int(v["out"]["logp"].shape[0]), v["target"]
            s_pred = create_segmentation_from_segments(np.array(s_tr), v["rel"], Tf)
            v["s_same"] = make_same_size_interpolate(s_pred, len(target))
            v["y_same"] = v["y_pred"] if len(v["y_pred"]) == len(target) else make_same_size_interpolate(v["y_pred"], len(target))
            heads = [v["y_same"], v["s_same"]]
            if self.enable_viterbi:
                vl = v["vit"][1].astype(np.int64)          # (what np.array of the reference's label list is)
                v["vit_same"] = vl if len(vl) == len(target) else make_same_size_interpolate(vl, len(target))
                heads.append(v["vit_same"])
            for h in heads:
                preds.append(np.asarray(h, dtype=np.int32))
                tgts.append(v["batch"].gt_label.reshape(-1))
        nh = 3 if self.enable_viterbi else 2
        flat = torch.from_numpy(np.concatenate(preds)).to(dev)
        pred_d, off = [], 0
        for p in preds:
            pred_d.append(flat[off: off + len(p)])
            off += len(p)
        bg = self.metrics["y_mof_nbg"].ignore_ids
        cnt = segmental_counters(tgts, pred_d, bg, self.metrics["y_f1_score"].overlaps)
        # -- feed the metric objects, video by video in dataset order (what batch_eval_calculation does with the host classes)
        m = self.metrics
        for k, v in enumerate(alive):
            s_tr = v["transcript"][:-1]
            m["s_mat_score"].add(target_transcript=v["target_transcript"], predicted_transcript=s_tr)
            m["s_len_diff"].add(target_transcript=v["target_transcript"], predicted_transcript=s_tr)
            for hi, head in enumerate(("y", "s", "vit")[:nh]):
                c = cnt[k * nh + hi]
                if c.get("over_limit"):      # more runs than the kernel's tables hold: this pair on the host metric objects
                    self._add(head, v["target"], preds[k * nh + hi], head != "y")
                    continue
                m[f"{head}_mof"].correct += c["correct"]
                m[f"{head}_mof"].total += c["total"]
                m[f"{head}_mof_nbg"].correct += c["correct_nbg"]
                m[f"{head}_mof_nbg"].total += c["total_nbg"]
                m[f"{head}_iod"].values.append(c["iod"])
                m[f"{head}_iou"].values.append(c["iou"])
                if head != "y":
                    m[f"{head}_iod_nbg"].values.append(c["iod_nbg"])
                    m[f"{head}_iou_nbg"].values.append(c["iou_nbg"])
                m[f"{head}_edit_score"].values.append(c["edit"])
                f1 = m[f"{head}_f1_score"]
                for s_, (tp, fp, fn) in enumerate(c["f1"]):
                    f1.tp[s_] += tp
                    f1.fp[s_] += fp
                    f1.fn[s_] += fn
            self.vit_segs.append(v["vit_same"] if self.enable_viterbi else v["s_same"])
            self.y_segs.append(v["y_same"])
            self.s_segs.append(v["s_same"])
            self.s_lens.append(v["rel"])
            self.s_transcript.append(s_tr)
            self.target_segs.append(v["target"])
            self.target_transcripts.append(v["target_transcript"])
            self._evaluated.append(v["i"])

    def on_start_eval(self):
        """Inference decodes greedily (reference evaluators.py:316-318: "should not happen if we want alignment performance")."""
        self.model.set_teacher_forcing(False)

    batched = True          # pooled host round trips + device metrics for device-resident test videos (False: one video at a time)
    chunk_videos = 32

    @torch.no_grad()
    def evaluate(self, rank: int = 0, world_size: int = 1):
        """Test videos sharded over ranks; every metric's accumulator is a few scalars, all-reduced as one vector."""
        self.model.eval()
        self.on_start_eval()
        for m in self.metrics.values():
            m.reset()
        self._reset_lists()
        self.skipped = 0
        self._evaluated = []     # dataset indices of the videos that made it into the lists
        mine = list(range(rank, len(self.test_db), world_size))
        on_cuda = str(self.device).startswith("cuda")
        may_chunk = self.batched and on_cuda and hasattr(self.model, "can_defer_eval")
        # Whether a video can take the batched path depends on the VIDEO (its encoded length must fit the native decoder: 1..4096
        # positions), so it is decided per video: deferrable ones collect into chunks, any other one first flushes the chunk in
        # front of it (the per-video lists stay in dataset order) and then runs through forward() / batch_eval_calculation.
        chunk = []
        for i in mine:
            batch = self.test_db[i]
            if may_chunk and self.model.can_defer_eval(batch, on_device=True):
                chunk.append((i, batch))
                if len(chunk) == self.chunk_videos:
                    self._evaluate_chunk(chunk)
                    chunk = []
                continue
            if chunk:
                self._evaluate_chunk(chunk)
                chunk = []
            batch = batch.to(self.device)
            try:
                forward_out = self.model.forward(batch)
                self.batch_eval_calculation(batch, forward_out)
                self._evaluated.append(i)
            except (EmptyTranscriptError, NoHypothesisError, ShortSequenceError):
                # The three degenerate outcomes the reference's evaluation dies on: an s-head that emits EOS as its first word
                # (models.py:351), a transcript the decoder runs out of hypotheses for (viterbi.py:147), fewer frames than one
                # decoding step (viterbi.py:87).  Here the video is counted and skipped, so that an early-epoch evaluation of a
                # barely trained model still reports.  Anything else is a bug and propagates.
                self.skipped += 1
        if chunk:
            self._evaluate_chunk(chunk)
        if on_cuda:
            # the status words the teacher-forced decoder launches of this evaluation left on the device (ops._PENDING_DECODER_STATUS: the alignment
            # evaluator appends one per video and nothing else reads them): read here, where the host has the results anyway -- a failed hand-over
            # raises from ITS evaluation, not from an unrelated forward 4096 videos later
            from .. import ops
            ops.check_health()
        if world_size > 1:
            import torch.distributed as dist
            keys = sorted(self.metrics)
            sizes = [len(self.metrics[k].state()) for k in keys]
            # NaN accumulators (an IoD / IoU over a video without target segments) stay NaN through the sum, as on one rank
            flat = torch.tensor([x for k in keys for x in np.asarray(self.metrics[k].state(), dtype=np.float64)] + [float(self.skipped)],
                                dtype=torch.float64, device=self.device)
            from .trainers import dist_all_reduce
            dist_all_reduce(flat)
            vals, off = flat.cpu().numpy(), 0
            for k, n in zip(keys, sizes):
                self.metrics[k].load_state(list(vals[off:off + n]))
                off += n
            self.skipped = int(round(vals[off]))
            # the per-video records (to_save, evaluators.py:259-267) of every rank, back in dataset order on every rank
            lists = ("y_segs", "s_segs", "vit_segs", "s_lens", "s_transcript", "target_segs", "target_transcripts")
            mine = {name: getattr(self, name) for name in lists}
            mine["order"] = self._evaluated
            gathered = [None] * world_size
            dist.all_gather_object(gathered, mine)
            order = np.argsort(np.concatenate([np.asarray(g["order"], dtype=np.int64) for g in gathered]), kind="stable")
            for name in lists:
                merged = [x for g in gathered for x in g[name]]
                setattr(self, name, [merged[i] for i in order])
        result = self.on_finish_eval()
        if self.skipped:
            result["skipped_videos"] = self.skipped
        return result


class MuConAlignmentEvaluator(MuConEvaluator):
    """Alignment instead of inference: the s-head is teacher-forced with the ground-truth transcript during evaluation, so the Viterbi
    decode aligns the GIVEN action sequence to the frames (reference evaluators.py:343-347).  Teacher-forced videos take the per-video
    forward() path (MuCon.can_defer_eval is False under teacher forcing)."""

    def on_start_eval(self):
        super().on_start_eval()
        self.model.set_teacher_forcing(True)      # because we are doing alignment here
