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
    tg = torch.cat([t.to(torch.int32) for t in targets]).contiguous()
    pr = torch.cat([p.to(torch.int32) for p in predictions]).contiguous()
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
