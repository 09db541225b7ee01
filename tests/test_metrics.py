"""CPU: mucon_amd/core/metrics against the reference's own metric classes (src/core/metrics/*.py) on seeded random
labellings -- tests/golden/metric_cases.npz, made by tools/make_golden_metrics.py: every per-video add() result and
the final summary(), with and without the background class; exact to float64 round-off."""
import os

import numpy as np
import pytest

from mucon_amd.core import metrics as M

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "metric_cases.npz"))
NV = int(GOLD["n_videos"])


def _make(ignore):
    return {"mof": M.MoFAccuracyMetric(ignore_ids=ignore), "iod": M.IoDMetric(ignore_ids=ignore), "iou": M.IoUMetric(ignore_ids=ignore),
            "edit": M.Edit(ignore_ids=ignore), "f1": M.F1Score(ignore_ids=ignore)}


@pytest.mark.parametrize("tag,ignore", [("all", ()), ("nbg", (0,))])
def test_metrics_match_reference(tag, ignore):
    ms = _make(ignore)
    with np.errstate(all="ignore"):
        for v in range(NV):
            t, p = GOLD[f"v{v}__target"], GOLD[f"v{v}__pred"]
            for k, m in ms.items():
                want = GOLD[f"{tag}__{k}__per_video"][v]
                if k == "f1" and np.all(np.isnan(want)):
                    continue   # the reference raises on a target without segments; here it counts false positives
                got = np.asarray(m(targets=t, predictions=p), dtype=np.float64)
                np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, equal_nan=True, err_msg=f"{k} video {v}")
    for k, m in ms.items():
        if k == "f1" and tag == "nbg":
            continue
        np.testing.assert_allclose(np.asarray(m.summary(), dtype=np.float64), GOLD[f"{tag}__{k}__summary"], rtol=1e-12,
                                   equal_nan=True, err_msg=k)


def test_state_vectors_merge_like_one_evaluator():
    """Two shards, summed state vectors == one metric that saw every video (what the multi-GPU evaluator all-reduces)."""
    whole, a, b = _make(()), _make(()), _make(())
    for v in range(NV):
        t, p = GOLD[f"v{v}__target"], GOLD[f"v{v}__pred"]
        for k in whole:
            whole[k](targets=t, predictions=p)
            (a if v % 2 == 0 else b)[k](targets=t, predictions=p)
    for k in whole:
        merged = _make(())[k]
        merged.load_state(list(np.asarray(a[k].state()) + np.asarray(b[k].state())))
        np.testing.assert_allclose(np.asarray(merged.summary(), dtype=np.float64), np.asarray(whole[k].summary(), dtype=np.float64),
                                   rtol=1e-12)


def test_transcript_metrics():
    ld = M.AbsLenDiffMetric()
    for a, b in (([1, 2, 3], [1, 2]), ([4], [4, 4, 4, 4]), ([], [1])):
        ld.add(target_transcript=a, predicted_transcript=b)
    assert abs(ld.summary() - float(GOLD["len_diff_summary"])) < 1e-12
    assert M.matching_ratio([1, 2, 3], [1, 2, 3]) == 1.0
    assert M.matching_ratio([1, 2, 3], [4, 5, 6]) == 0.0
    assert abs(M.matching_ratio([1, 2, 3, 4], [1, 3, 4]) - 2 * 3 / 7) < 1e-12
    with pytest.raises(ZeroDivisionError):
        M.matching_ratio([], [])


def test_runs_and_levenshtein():
    v, s, e = M.runs([3, 3, 0, 0, 0, 5, 3])
    assert v.tolist() == [3, 0, 5, 3] and s.tolist() == [0, 2, 5, 6] and e.tolist() == [2, 5, 6, 7]
    assert M.runs([3, 3, 0, 5], ignore=[0])[0].tolist() == [3, 5]
    assert M.runs([])[0].shape == (0,)
    assert M.levenshtein([1, 2, 3], [1, 3]) == 1 and M.levenshtein([], [1, 2]) == 2 and M.levenshtein([1, 2], [2, 1]) == 2
