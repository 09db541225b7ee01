"""MoF / IoD / IoU from labellings that live on the device (mucon_metrics_overlap, include/mucon_hip.h): one launch for a list
of videos, a few numbers per video back to the host.  The results feed the host metric objects (MoFAccuracyMetric, IoDMetric,
IoUMetric of this package) and equal what their own add() would have computed from the same labellings, bit for bit: the
kernel returns integer counts and single float64 quotients, the mean over a video's target segments is taken here, in the
order NumPy takes it."""
from typing import Iterable, List, Sequence

import numpy as np
import torch

from . import IoDMetric, IoUMetric, MoFAccuracyMetric


def overlap_counters(targets: Sequence[torch.Tensor], predictions: Sequence[torch.Tensor], ignore_ids: Iterable[int] = ()) -> List[dict]:
    """targets[v], predictions[v]: device int tensors of equal length.  -> per video {"correct", "total", "iod", "iou"} with the
    values MoFAccuracyMetric(ignore_ids).add, IoDMetric(ignore_ids).add and IoUMetric(ignore_ids).add return / accumulate."""
    from ... import _lib
    lib = _lib.load()
    n = len(targets)
    if n == 0:
        return []
    dev = targets[0].device
    assert all(t.is_cuda and p.is_cuda and t.shape == p.shape and t.dim() == 1 for t, p in zip(targets, predictions))
    lens = [int(t.shape[0]) for t in targets]
    off = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int64).to(dev)
    # (one conversion per side, not one per pair; pairs of one video usually share the target tensor and a cat of views is one launch)
    tg = (torch.cat(list(targets)) if len({t.dtype for t in targets}) == 1 else torch.cat([t.to(torch.int64) for t in targets])).to(torch.int32).contiguous()
    pr = (torch.cat(list(predictions)) if len({p.dtype for p in predictions}) == 1 else torch.cat([p.to(torch.int64) for p in predictions])).to(torch.int32).contiguous()
    ign = sorted(int(i) for i in (ignore_ids or ()))
    ign_d = torch.tensor(ign, dtype=torch.int32).to(dev) if ign else None
    R = _lib.METRICS_MAX_RUNS
    mof = torch.empty((n, 2), dtype=torch.int64, device=dev)
    n_runs = torch.zeros((n, 3), dtype=torch.int32, device=dev)
    run_label = torch.empty((n, R), dtype=torch.int32, device=dev)
    iod = torch.empty((n, R), dtype=torch.float64, device=dev)
    iou = torch.empty((n, R), dtype=torch.float64, device=dev)
    _lib.check(lib.mucon_metrics_overlap(n, _lib.ptr(off), _lib.ptr(tg), _lib.ptr(pr), _lib.ptr(ign_d), len(ign), _lib.ptr(mof),
                                         _lib.ptr(n_runs), _lib.ptr(run_label), _lib.ptr(iod), _lib.ptr(iou),
                                         _lib.current_stream_ptr()), "mucon_metrics_overlap")
    mof_h, runs_h = mof.cpu().numpy(), n_runs.cpu().numpy()
    nmax = int(runs_h[:, 0].max()) if n else 0
    if int(runs_h[:, :2].max()) > R:
        raise ValueError(f"a labelling with more than {R} segments: use the host metrics for it")
    lab_h = run_label[:, :nmax].cpu().numpy()
    iod_h, iou_h = iod[:, :nmax].cpu().numpy(), iou[:, :nmax].cpu().numpy()
    out = []
    for v in range(n):
        nt, kept_pred = int(runs_h[v, 0]), int(runs_h[v, 2])
        keep = ~np.isin(lab_h[v, :nt], ign) if ign else np.ones(nt, dtype=bool)
        res = {"correct": int(mof_h[v, 0]), "total": int(mof_h[v, 1])}
        for name, vals in (("iod", iod_h), ("iou", iou_h)):
            if lens[v] == 0 or not keep.any():
                res[name] = float("nan")
            elif kept_pred == 0:
                res[name] = 0.0
            else:
                res[name] = float(np.maximum(vals[v, :nt][keep], 0.0).mean())
        out.append(res)
    return out


def add_to_metrics(counters: dict, mof: MoFAccuracyMetric = None, iod: IoDMetric = None, iou: IoUMetric = None):
    """Feed one video's device counters into host metric objects built with the same ignore_ids."""
    if mof is not None:
        mof.correct += counters["correct"]
        mof.total += counters["total"]
    if iod is not None:
        iod.values.append(counters["iod"])
    if iou is not None:
        iou.values.append(counters["iou"])


def segmental_counters(targets: Sequence[torch.Tensor], predictions: Sequence[torch.Tensor], ignore_ids: Iterable[int] = (),
                       overlaps: Sequence[float] = (0.1, 0.25, 0.5)) -> List[dict]:
    """One launch (mucon_metrics_segmental) for a list of (target, prediction) pairs of device int tensors: per pair everything
    the evaluator's metric objects accumulate --
      correct / total, correct_nbg / total_nbg      MoFAccuracyMetric() / MoFAccuracyMetric(ignore_ids)
      iod, iou, iod_nbg, iou_nbg                     IoDMetric / IoUMetric without and with ignore_ids
      edit                                           Edit().add's score        f1: [(tp, fp, fn)] per overlap threshold (F1Score)
    with the values the host classes compute from the same labellings, bit for bit.
    A pair in which either labelling has more than METRICS_MAX_RUNS (1,024) runs -- a noisy y-head labelling of a long video --
    is beyond the kernel's LDS tables: its record is {"over_limit": True} and nothing else, and the caller scores that pair with
    the host metric objects (MuConEvaluator._evaluate_chunk_on does)."""
    from ... import _lib
    lib = _lib.load()
    n = len(targets)
    if n == 0:
        return []
    dev = targets[0].device
    assert all(t.is_cuda and p.is_cuda and t.shape == p.shape and t.dim() == 1 for t, p in zip(targets, predictions))
    assert len(overlaps) <= 4
    lens = [int(t.shape[0]) for t in targets]
    off = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int64)).to(dev, non_blocking=True)
    # (one conversion per side, not one per pair; pairs of one video usually share the target tensor and a cat of views is one launch)
    tg = (torch.cat(list(targets)) if len({t.dtype for t in targets}) == 1 else torch.cat([t.to(torch.int64) for t in targets])).to(torch.int32).contiguous()
    pr = (torch.cat(list(predictions)) if len({p.dtype for p in predictions}) == 1 else torch.cat([p.to(torch.int64) for p in predictions])).to(torch.int32).contiguous()
    ign = sorted(int(i) for i in (ignore_ids or ()))
    ign_d = torch.tensor(ign, dtype=torch.int32).to(dev) if ign else None
    thr_d = torch.tensor(list(overlaps), dtype=torch.float64).to(dev)
    R = _lib.METRICS_MAX_RUNS
    # the small per-pair records share one buffer (one copy back): mof int64 [n][4] | n_runs int32 [n][3] | seg int32 [n][13]
    small = torch.zeros(n * (32 + 12 + 52), dtype=torch.uint8, device=dev)
    mof = small[: 32 * n].view(torch.int64).view(n, 4)
    n_runs = small[32 * n: 44 * n].view(torch.int32).view(n, 3)
    seg = small[44 * n:].view(torch.int32).view(n, 13)
    run_label = torch.empty((n, R), dtype=torch.int32, device=dev)
    iod = torch.empty((n, R), dtype=torch.float64, device=dev)
    iou = torch.empty((n, R), dtype=torch.float64, device=dev)
    _lib.check(lib.mucon_metrics_segmental(n, _lib.ptr(off), _lib.ptr(tg), _lib.ptr(pr), _lib.ptr(ign_d), len(ign), _lib.ptr(thr_d),
                                           len(overlaps), _lib.ptr(mof), _lib.ptr(n_runs), _lib.ptr(run_label), _lib.ptr(iod),
                                           _lib.ptr(iou), _lib.ptr(seg), _lib.current_stream_ptr()), "mucon_metrics_segmental")
    small_h = small.cpu().numpy()                      # (synchronises)
    mof_h = small_h[: 32 * n].view(np.int64).reshape(n, 4)
    runs_h = small_h[32 * n: 44 * n].view(np.int32).reshape(n, 3)
    seg_h = small_h[44 * n:].view(np.int32).reshape(n, 13)
    over = (runs_h[:, :2] > R).any(axis=1)
    nmax = max(int(np.minimum(runs_h[:, 0], R).max()), 1)
    per_run = torch.cat([iod[:, :nmax], iou[:, :nmax], run_label[:, :nmax].to(torch.float64)], dim=1).cpu().numpy()   # labels are exact in f64
    iod_h, iou_h, lab_h = per_run[:, :nmax], per_run[:, nmax: 2 * nmax], per_run[:, 2 * nmax:].astype(np.int64)
    # the clamp and the ignore mask once for the chunk; the means stay per pair, over the compressed values, as the host classes take them
    # (np.mean's pairwise grouping depends on the array it is handed: a masked full-width sum would round differently)
    iod_m, iou_m = np.maximum(iod_h, 0.0), np.maximum(iou_h, 0.0)
    not_ign = ~np.isin(lab_h, ign) if ign else None
    out = []
    nan = float("nan")
    with np.errstate(all="ignore"):
        for v in range(n):
            if over[v]:
                out.append({"over_limit": True})
                continue
            nt, npred, kept_pred = int(runs_h[v, 0]), int(runs_h[v, 1]), int(runs_h[v, 2])
            res = {"correct": int(mof_h[v, 0]), "total": int(mof_h[v, 1]), "correct_nbg": int(mof_h[v, 2]), "total_nbg": int(mof_h[v, 3])}
            keep = not_ign[v, :nt] if not_ign is not None else None
            any_kept = nt > 0 and (keep is None or bool(keep.any()))
            for suffix, mask, kp, some in (("", None, npred, nt > 0), ("_nbg", keep, kept_pred, any_kept)):
                for name, vals in (("iod", iod_m), ("iou", iou_m)):
                    if lens[v] == 0 or not some:
                        res[name + suffix] = nan
                    elif kp == 0:
                        res[name + suffix] = 0.0
                    else:
                        row = vals[v, :nt] if mask is None else vals[v, :nt][mask]
                        res[name + suffix] = float(np.add.reduce(row) / row.size)       # ndarray.mean() of a float64 row, minus its Python wrapper
            res["edit"] = float((1 - np.float64(float(seg_h[v, 0])) / max(npred, nt)) * 100)
            res["f1"] = [(float(seg_h[v, 1 + 3 * s]), float(seg_h[v, 2 + 3 * s]), float(seg_h[v, 3 + 3 * s])) for s in range(len(overlaps))]
            out.append(res)
    return out
