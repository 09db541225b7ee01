"""Metric goldens from the reference (build container only): its metric classes (src/core/metrics/*.py) on seeded
random labellings -- per-video add() results and the final summary().  MatchingScoreMetric is left out: it needs the
un-vendored edit_distance package.  Pins mucon_amd/core/metrics (tests/test_metrics.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402

ref_harness.install()
from mucon_amd import synth  # noqa: E402


def labelling(seed, T, n_seg, C):
    tr = synth.integers(seed, n_seg, 0, C)
    return synth.segment_labels(seed + 1, T, tr)


def videos():
    out = []
    for k in range(14):
        T = int(synth.integers(900 + k, 1, 20, 400)[0])
        tgt = labelling(1000 + 10 * k, T, 2 + k % 7, 6)
        if k % 4 == 0:      # a noisy copy of the target
            pred = tgt.copy()
            flip = synth.integers(2000 + k, T, 0, 10) == 0
            pred[flip] = synth.integers(3000 + k, int(flip.sum()), 0, 6)
        elif k == 5:
            pred = np.zeros(T, dtype=tgt.dtype)          # all background
        else:
            pred = labelling(4000 + 10 * k, T, 1 + k % 9, 6)
        out.append((tgt, pred))
    out.append((np.zeros(30, dtype=np.int64), labelling(77, 30, 3, 6)))   # target is background only
    return out


def main():
    from core.metrics.fully_supervised import Edit, F1Score
    from core.metrics.segmentation import IoDMetric, IoUMetric, MoFAccuracyMetric
    from core.metrics.transcript import AbsLenDiffMetric
    import warnings
    warnings.simplefilter("ignore")
    out = {}
    vids = videos()
    for v, (t, p) in enumerate(vids):
        out[f"v{v}__target"], out[f"v{v}__pred"] = t, p
    for tag, ignore in (("all", ()), ("nbg", (0,))):
        ms = {"mof": MoFAccuracyMetric(ignore_ids=ignore), "iod": IoDMetric(ignore_ids=ignore), "iou": IoUMetric(ignore_ids=ignore),
              "edit": Edit(ignore_ids=ignore), "f1": F1Score(ignore_ids=ignore)}
        per = {k: [] for k in ms}
        for v, (t, p) in enumerate(vids):
            for k, m in ms.items():
                if k == "f1" and tag == "nbg" and v == len(vids) - 1:
                    per[k].append([np.nan] * 3)     # the reference raises (argmax of an empty array): skipped there
                    continue
                per[k].append(np.asarray(m.add(targets=t, predictions=p), dtype=np.float64))
        for k, m in ms.items():
            out[f"{tag}__{k}__per_video"] = np.asarray(per[k], dtype=np.float64)
            out[f"{tag}__{k}__summary"] = np.asarray(m.summary(), dtype=np.float64)
    ld = AbsLenDiffMetric()
    for a, b in (([1, 2, 3], [1, 2]), ([4], [4, 4, 4, 4]), ([], [1])):
        ld.add(target_transcript=a, predicted_transcript=b)
    out["len_diff_summary"] = np.asarray(ld.summary())
    out["n_videos"] = np.asarray(len(vids))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metric_cases.npz"), **out)
    print({k: v for k, v in out.items() if k.endswith("summary")})


if __name__ == "__main__":
    main()
