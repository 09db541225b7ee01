"""GPU: MoF / IoD / IoU counters computed on the device (mucon_metrics_overlap) == the host metric classes on the same labellings,
bit for bit -- random segmentations, background ignored or not, a labelling made of ignored labels only, predictions that never
hit the target's labels, single-frame videos, many short runs."""
import numpy as np
import pytest
import torch

from mucon_amd import synth
from mucon_amd.core.metrics import IoDMetric, IoUMetric, MoFAccuracyMetric

pytestmark = pytest.mark.gpu


def _labelling(seed, T, n_seg, C):
    tr = synth.integers(seed, n_seg, 0, C)
    return synth.segment_labels(seed + 1, T, tr).astype(np.int64)


def _cases():
    cases = []
    for i, (T, n_t, n_p) in enumerate([(2000, 6, 6), (640, 3, 9), (9741, 30, 22), (1, 1, 1), (57, 20, 25), (4096, 200, 180), (300, 5, 1)]):
        cases.append((_labelling(100 + 7 * i, T, n_t, 12), _labelling(200 + 11 * i, T, n_p, 12)))
    cases.append((np.zeros(500, dtype=np.int64), _labelling(5, 500, 4, 12)))                 # target = background only
    cases.append((_labelling(6, 500, 4, 12), np.zeros(500, dtype=np.int64)))                 # prediction = background only
    cases.append((_labelling(7, 400, 5, 6), _labelling(8, 400, 5, 6) + 6))                   # disjoint label sets
    cases.append((np.arange(900) % 7, (np.arange(900) // 3) % 7))                            # hundreds of short runs
    return cases


@pytest.mark.parametrize("ignore", [(), (0,), (0, 3, 11)])
def test_device_counters_equal_the_host_metrics(ignore):
    from mucon_amd.core.metrics.device import add_to_metrics, overlap_counters
    cases = _cases()
    got = overlap_counters([torch.from_numpy(t).cuda() for t, _ in cases], [torch.from_numpy(p).cuda() for _, p in cases], ignore)
    dm, dd, du = MoFAccuracyMetric(ignore), IoDMetric(ignore), IoUMetric(ignore)
    hm, hd, hu = MoFAccuracyMetric(ignore), IoDMetric(ignore), IoUMetric(ignore)
    with np.errstate(all="ignore"):
        for (t, p), g in zip(cases, got):
            hm.add(t, p)
            want_d, want_u = hd.add(t, p), hu.add(t, p)
            add_to_metrics(g, dm, dd, du)
            for a, b in ((g["iod"], want_d), (g["iou"], want_u)):
                assert (np.isnan(a) and np.isnan(b)) or np.float64(a).tobytes() == np.float64(b).tobytes(), (a, b)
    assert (dm.correct, dm.total) == (hm.correct, hm.total)
    assert np.array_equal(np.asarray(dd.values), np.asarray(hd.values), equal_nan=True)
    assert np.array_equal(np.asarray(du.values), np.asarray(hu.values), equal_nan=True)


def test_more_runs_than_the_kernel_holds_is_reported():
    from mucon_amd.core.metrics.device import overlap_counters
    y = torch.arange(4000, device="cuda") % 2
    with pytest.raises(ValueError):
        overlap_counters([y], [y], ())
