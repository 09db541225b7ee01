#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

    python tools/make_golden.py [viterbi] [dense] [glue]

Imports /root/reference/src through tools/ref_harness.py (stubs for the absent third-party
packages), runs the reference's own classes on seeded inputs from mucon_amd/synth.py, and
writes small fixtures under tests/golden/.  Fixtures hold data only -- seeds, small inputs,
expected outputs -- never reference source.  The reference never travels to the GPU box.

Reference entry points exercised:
  core.viterbi.viterbi.Viterbi.decode            (src/core/viterbi/viterbi.py:49-65)
  core.viterbi.grammar.SingleTranscriptGrammar   (src/core/viterbi/grammar.py:196-217)
  core.viterbi.length_model.PoissonModel         (src/core/viterbi/length_model.py:42-83)
  core.modules.temporal.WaveNetBlock             (src/core/modules/temporal.py:77-147)
  mucon.models.MuCon.temporal_modeling_forward / frame_classifier_forward / predict
                                                 (src/mucon/models.py:746-773, 567-582, 360-374)
  mucon.evaluators (mean-length glue, restated from :155-165 -- it is inline code, not a function)
"""
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

GOLD = os.path.join(ROOT, "tests", "golden")
FS, MAXLEN, C = 30, 2000, 48


# ------------------------------------------------------------------------------- viterbi
def _ref_decode(lp, transcript, length_model):
    from core.viterbi.grammar import SingleTranscriptGrammar
    from core.viterbi.viterbi import Viterbi

    v = Viterbi(None, None, frame_sampling=FS)  # as reference evaluators.py:80
    v.grammar = SingleTranscriptGrammar([int(x) for x in transcript], C)
    v.length_model = length_model
    score, labels, segments = v.decode(lp)
    return (np.float64(score), np.asarray(labels, dtype=np.int32),
            np.asarray([s.label for s in segments], dtype=np.int32),
            np.asarray([s.length for s in segments], dtype=np.int32))


def _table_rows(length_model, transcript):
    J = MAXLEN // FS
    P = np.empty((J, len(transcript)), dtype=np.float64)
    for j in range(J):
        for n, a in enumerate(transcript):
            P[j, n] = length_model.score((j + 1) * FS, int(a))
    return P


def viterbi_cases():
    """(name, T, N, emission seed, kind) -- SURVEY.md 8c case list."""
    cases = []
    # random sizes
    sizes = [(200, 1), (333, 2), (512, 3), (777, 4), (1000, 3), (1234, 5), (1500, 2), (1999, 6),
             (2000, 6), (2048, 7), (2400, 4), (2999, 7), (3000, 5), (900, 1), (1980, 1), (450, 7),
             (660, 6), (1111, 3), (2718, 6), (3141, 4)]
    for i, (T, N) in enumerate(sizes):
        cases.append(dict(name=f"rand{i:02d}", T=T, N=N, seed=100 + i, kind="poisson"))
    cases += [
        dict(name="repeat_adjacent", T=1500, N=5, seed=201, kind="poisson", transcript=[7, 10, 10, 3, 3]),
        dict(name="mod0", T=1800, N=4, seed=202, kind="poisson"),
        dict(name="mod29", T=1829, N=4, seed=203, kind="poisson"),
        dict(name="k_lt_n", T=100, N=6, seed=204, kind="poisson"),        # K=3 < N: score -inf, truncated
        dict(name="k_eq_n", T=150, N=5, seed=205, kind="poisson"),        # K=5 == N
        dict(name="t30", T=30, N=1, seed=206, kind="poisson"),            # K=1
        dict(name="t30_n3", T=30, N=3, seed=207, kind="poisson"),         # K=1 < N
        dict(name="t59", T=59, N=2, seed=208, kind="poisson"),
        dict(name="long", T=9741, N=30, seed=209, kind="poisson"),        # Breakfast max length
        dict(name="stress_small", T=6000, N=40, seed=210, kind="poisson"),
        dict(name="ties_flat", T=3000, N=3, seed=0, kind="flat_const"),   # all ties
        dict(name="ties_flat2", T=2500, N=4, seed=0, kind="flat_const"),
        dict(name="flat_rand", T=1700, N=4, seed=211, kind="flat"),
        dict(name="uninformative", T=1600, N=5, seed=212, kind="poisson_noise"),  # emissions carry no signal
        dict(name="tiny_mu", T=1200, N=4, seed=213, kind="poisson", mu_override=(2, 0.7)),
        dict(name="huge_mu", T=2500, N=3, seed=214, kind="poisson", mu_override=(1, 1900.0)),
        dict(name="single_long", T=1980, N=1, seed=215, kind="poisson"),  # N=1, exactly fits
        # mean length < 0.5 -> NaN norms (length_model.py:56-58): the reference does NOT raise, it
        # returns score -inf and a truncated labelling (NaN never wins a `<=`/`>=` comparison)
        dict(name="nan_mu_n1", T=900, N=3, seed=216, kind="poisson", mu_override=(1, 0.3)),
        dict(name="nan_mu_n2", T=1500, N=4, seed=217, kind="poisson", mu_override=(2, 0.3)),
        dict(name="nan_mu_n3of6", T=2600, N=6, seed=218, kind="poisson", mu_override=(3, 0.2)),
        dict(name="nan_mu_last", T=1300, N=3, seed=219, kind="poisson", mu_override=(2, 0.4)),
        dict(name="nan_mu_long", T=5000, N=5, seed=220, kind="poisson", mu_override=(3, 0.3)),
        dict(name="k_lt_n_b", T=250, N=12, seed=221, kind="poisson"),
        dict(name="k_eq_66n", T=3960, N=2, seed=222, kind="poisson"),     # K = 132 = 66*N: last column that still has hypotheses
    ]
    return cases


def make_viterbi():
    from core.viterbi.length_model import LengthModel, PoissonModel

    class FlatF64(LengthModel):  # a float64 length model with no preference (tie cases)
        def score(self, length, label):
            return -np.inf if length >= MAXLEN else np.float64(0.0)

        def max_length(self):
            return MAXLEN

    out, meta = {}, []
    for cs in viterbi_cases():
        T, N, seed, kind = cs["T"], cs["N"], cs["seed"], cs["kind"]
        tr = np.asarray(cs.get("transcript") or synth.transcript(seed + 7, N, C), dtype=np.int64)
        if kind in ("poisson", "poisson_noise"):
            gt = synth.segment_labels(seed + 11, T, tr)
            lp = synth.emissions(seed, T, C, labels=None if kind == "poisson_noise" else gt)
            rel = synth.uniform01(seed + 13, (N,)) + np.float32(0.1)
            rel = (rel / rel.sum()).astype(np.float32)
            mu = synth.mean_lengths(tr, rel, T, C)
            if "mu_override" in cs:
                n_, v_ = cs["mu_override"]
                mu[tr[n_]] = v_
            with np.errstate(all="ignore"):
                lm = PoissonModel(mu)
        else:
            lp = np.full((T, C), np.float32(-1.0), np.float32) if kind == "flat_const" else synth.emissions(seed, T, C)
            mu = np.zeros(C)
            lm = FlatF64()
        score, labels, seg_label, seg_len = _ref_decode(lp, tr, lm)
        P = _table_rows(lm, tr)
        nm = cs["name"]
        out[f"{nm}__transcript"] = tr.astype(np.int32)
        out[f"{nm}__mu"] = mu
        out[f"{nm}__P"] = P
        out[f"{nm}__score"] = np.asarray([score], dtype=np.float64)
        out[f"{nm}__labels"] = labels
        out[f"{nm}__seg_label"] = seg_label
        out[f"{nm}__seg_len"] = seg_len
        meta.append(dict(name=nm, T=T, N=N, seed=seed, kind=kind, score=float(score) if np.isfinite(score) else str(score)))
        print(f"  {nm:18s} T={T:5d} N={N:2d} score={score!r} segs={list(zip(seg_label.tolist(), seg_len.tolist()))[:4]}...")

    # a few cases with REAL log-softmax emissions stored explicitly (their exp/log is host dependent)
    import torch
    for i, (T, N) in enumerate([(300, 3), (615, 4), (1000, 5)]):
        seed = 300 + i
        tr = synth.transcript(seed + 7, N, C)
        gt = synth.segment_labels(seed + 11, T, tr)
        logits = 3.0 * torch.from_numpy(synth.uniform_pm1(seed, (T, C)))
        logits[torch.arange(T), torch.from_numpy(gt)] += 2.0
        lp = torch.log_softmax(logits, dim=1).numpy().astype(np.float32)
        rel = synth.uniform01(seed + 13, (N,)) + np.float32(0.1)
        rel = (rel / rel.sum()).astype(np.float32)
        mu = synth.mean_lengths(tr, rel, T, C)
        lm = PoissonModel(mu)
        score, labels, seg_label, seg_len = _ref_decode(lp, tr, lm)
        nm = f"stored{i}"
        out[f"{nm}__lp"] = lp
        out[f"{nm}__transcript"] = tr.astype(np.int32)
        out[f"{nm}__mu"] = mu
        out[f"{nm}__P"] = _table_rows(lm, tr)
        out[f"{nm}__score"] = np.asarray([score], dtype=np.float64)
        out[f"{nm}__labels"] = labels
        out[f"{nm}__seg_label"] = seg_label
        out[f"{nm}__seg_len"] = seg_len
        meta.append(dict(name=nm, T=T, N=N, seed=seed, kind="stored"))
        print(f"  {nm:18s} T={T:5d} N={N:2d} score={score!r}")

    # error behaviour of the reference (SURVEY.md 8a-6 "failure modes")
    errs = []
    for nm, T, N, mu_small in [("err_t_lt_fs", 29, 2, False), ("err_too_long", 2100, 1, False),
                               ("err_too_long2", 4100, 2, False), ("err_nan_mu_first", 900, 3, True),
                               ("err_k_gt_66n", 3990, 2, False)]:
        tr = synth.transcript(400 + len(errs), N, C)
        lp = synth.emissions(400 + len(errs), T, C)
        mu = np.full(C, 300.0)
        if mu_small:
            mu[tr[0]] = 0.3
        try:
            with np.errstate(all="ignore"):
                _ref_decode(lp, tr, PoissonModel(mu))
            exc = "none"
        except Exception as e:  # noqa: BLE001
            exc = type(e).__name__
        errs.append(dict(name=nm, T=T, N=N, seed=400 + len(errs), mu_small=mu_small, exception=exc,
                         transcript=[int(x) for x in tr]))
        print(f"  {nm:18s} T={T:5d} N={N:2d} -> {exc}")

    np.savez_compressed(os.path.join(GOLD, "viterbi_cases.npz"), **out)
    with open(os.path.join(GOLD, "viterbi_cases.json"), "w") as f:
        json.dump(dict(fs=FS, max_length=MAXLEN, C=C, cases=meta, errors=errs,
                       numpy=np.__version__), f, indent=1)


# ------------------------------------------------------------------------------- glue
def make_glue():
    """Poisson table + evaluator mean-length glue (reference length_model.py:43-71, evaluators.py:155-165)."""
    from core.viterbi.length_model import PoissonModel

    out = {}
    for i, (N, Tf) in enumerate([(3, 900), (6, 2000), (12, 5000), (25, 9741)]):
        seed = 500 + i
        tr = synth.transcript(seed, N, C)
        rel = synth.uniform01(seed + 1, (N,)) + np.float32(0.05)
        rel = (rel / rel.sum()).astype(np.float32)
        # reference evaluators.py:155-165, executed verbatim on numpy arrays
        actions = np.eye(C)[np.array([int(x) for x in tr]).reshape(-1)]
        lengths = np.dot(rel, actions)
        lengths *= Tf
        k = actions.sum(0)
        k[k == 0] = 1
        lengths /= k
        lengths[lengths == 0] = 1
        pm = PoissonModel(lengths)
        out[f"g{i}__transcript"] = tr.astype(np.int32)
        out[f"g{i}__rel"] = rel
        out[f"g{i}__Tf"] = np.asarray([Tf])
        out[f"g{i}__mu"] = lengths
        out[f"g{i}__poisson_rows"] = pm.poisson[FS:MAXLEN:FS, :].copy()  # rows 30,60,..,1980
        out[f"g{i}__norms"] = pm.norms.copy()
    np.savez_compressed(os.path.join(GOLD, "glue_cases.npz"), **out)
    print("  glue: wrote", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["viterbi", "glue", "dense"]
    os.makedirs(GOLD, exist_ok=True)
    if "viterbi" in which:
        print("viterbi goldens:")
        make_viterbi()
    if "glue" in which:
        make_glue()
    if "dense" in which:
        import make_golden_dense
        make_golden_dense.main(GOLD)
