#!/usr/bin/env python3
"""HERE (reference mounted): what the REFERENCE's beam search returns for finite max_hypotheses (src/core/viterbi/viterbi.py:34, :74-79),
for cases of tests/golden/viterbi_cases.npz -> tests/golden/viterbi_pruned.json: per (case, max_hypotheses) the pruned score, whether the
labelling equals the unpruned one, and the unpruned score.  The HIP decoder does not prune (it returns the exact optimum and says so in a
warning); tests/test_gpu_viterbi.py checks the relation the fixture pins: score >= the beam's, the same result wherever the beam kept the
best path.  Data only -- no reference source."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402

ref_harness.install()
from mucon_amd import synth  # noqa: E402

FS, MAXLEN, C = 30, 2000, 48


def main():
    from core.viterbi.grammar import SingleTranscriptGrammar
    from core.viterbi.length_model import PoissonModel
    from core.viterbi.viterbi import Viterbi

    gold = np.load(os.path.join(ROOT, "tests", "golden", "viterbi_cases.npz"))
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "viterbi_cases.json")))
    by_name = {c["name"]: c for c in meta["cases"]}
    out = []
    for name in ("rand03", "rand05", "rand07", "rand16", "repeat_adjacent", "uninformative", "stress_small"):
        cs = by_name[name]
        T, seed, kind = cs["T"], cs["seed"], cs["kind"]
        tr = gold[f"{name}__transcript"].astype(np.int64)
        gt = synth.segment_labels(seed + 11, T, tr)
        lp = synth.emissions(seed, T, C, labels=None if kind == "poisson_noise" else gt)
        with np.errstate(all="ignore"):
            lm = PoissonModel(gold[f"{name}__mu"])
        full_score = float(gold[f"{name}__score"][0])
        full_labels = gold[f"{name}__labels"]
        for mh in (1, 3, 10, 40, 200, len(tr) * (MAXLEN // FS)):
            v = Viterbi(None, None, frame_sampling=FS, max_hypotheses=mh)
            v.grammar = SingleTranscriptGrammar([int(x) for x in tr], C)
            v.length_model = lm
            try:
                score, labels, _ = v.decode(lp)
                rec = dict(case=name, max_hypotheses=mh, score=float(score) if np.isfinite(score) else str(score),
                           same_labels=bool(np.array_equal(np.asarray(labels), full_labels)), unpruned_score=full_score)
            except Exception as e:  # noqa: BLE001  (the beam can lose every hypothesis that reaches the last state)
                rec = dict(case=name, max_hypotheses=mh, exception=type(e).__name__, unpruned_score=full_score)
            out.append(rec)
            print(rec)
    with open(os.path.join(ROOT, "tests", "golden", "viterbi_pruned.json"), "w") as f:
        json.dump(dict(fs=FS, max_length=MAXLEN, C=C, numpy=np.__version__, cases=out), f, indent=1)


if __name__ == "__main__":
    main()
