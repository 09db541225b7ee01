"""CPU: the C oracle (oracle/viterbi_oracle.c) against the golden vectors produced by the
reference's own Viterbi.decode (tools/make_golden.py).  Bit-exact: score bits, labels, segments."""
import numpy as np
import pytest

import oracle
from mucon_amd import synth
from helpers import C, f64_bits, load_viterbi_golden, viterbi_case_inputs

Z, META = load_viterbi_golden()
FS, MAXLEN = META["fs"], META["max_length"]


@pytest.mark.parametrize("cs", META["cases"], ids=[c["name"] for c in META["cases"]])
def test_oracle_matches_reference_golden(cs):
    nm = cs["name"]
    lp = viterbi_case_inputs(Z, cs)
    tr = Z[f"{nm}__transcript"]
    score, labels, seg_label, seg_len = oracle.viterbi_decode_table(lp, tr, Z[f"{nm}__P"], FS, MAXLEN)
    assert f64_bits(score) == f64_bits(Z[f"{nm}__score"][0]), (score, Z[f"{nm}__score"][0])
    np.testing.assert_array_equal(labels, Z[f"{nm}__labels"])
    np.testing.assert_array_equal(seg_label, Z[f"{nm}__seg_label"])
    np.testing.assert_array_equal(seg_len, Z[f"{nm}__seg_len"])
    assert len(labels) == cs["T"] and seg_len.sum() == cs["T"]


@pytest.mark.parametrize("er", META["errors"], ids=[e["name"] for e in META["errors"]])
def test_oracle_error_behaviour(er):
    """Reference failure modes: T < fs -> IndexError; empty hypothesis set / all-NaN -> AttributeError."""
    lp = synth.emissions(er["seed"], er["T"], C)
    tr = np.asarray(er["transcript"])
    mu = np.full(C, 300.0)
    if er["mu_small"]:
        mu[tr[0]] = 0.3
    want = {"IndexError": oracle.ST_INDEX_ERROR, "AttributeError": oracle.ST_NO_HYPOTHESIS}[er["exception"]]
    with pytest.raises(oracle.OracleDecodeError) as e:
        oracle.viterbi_decode(lp, tr, mu, FS, MAXLEN)
    assert e.value.status == want


def test_oracle_poisson_table_matches_reference():
    """oracle.poisson_table / mean_lengths_from_s_head against reference PoissonModel + evaluator glue."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_cases.npz"))
    i = 0
    while f"g{i}__mu" in g:
        tr, rel, Tf = g[f"g{i}__transcript"], g[f"g{i}__rel"], int(g[f"g{i}__Tf"][0])
        mu = oracle.mean_lengths_from_s_head(rel, tr, Tf, C)
        np.testing.assert_array_equal(mu, g[f"g{i}__mu"])
        tab = oracle.poisson_table(mu, MAXLEN)
        got, want = tab[FS:MAXLEN:FS, :], g[f"g{i}__poisson_rows"]
        # np.log may differ by an ulp between hosts with different SIMD dispatch: equal here, close elsewhere
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-9)
        i += 1
    assert i == 4


def test_oracle_frame_scores_sequential_f32():
    """cumsum must be a sequential float32 chain (reference viterbi.py:51), not an f64 scan."""
    lp = synth.emissions(7, 700, C)
    F = oracle.frame_scores(lp, 30)
    cs = np.cumsum(lp, axis=0)  # numpy's f32 cumsum is sequential
    want = np.empty_like(F)
    want[0] = cs[29]
    for k in range(1, 700 // 30):
        want[k] = cs[(k + 1) * 30 - 1] - cs[(k + 1) * 30 - 1 - 30]
    np.testing.assert_array_equal(F.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------------------------------------- the beam (max_hypotheses)
from helpers import load_pruned_golden, pruned_case_inputs  # noqa: E402

PZ, PMETA = load_pruned_golden()


@pytest.mark.parametrize("rec", PMETA["cases"], ids=[r["name"] for r in PMETA["cases"]])
def test_oracle_beam_matches_the_references(rec):
    """oracle/viterbi_oracle.c:dict_prune (a literal restatement of prune(), viterbi.py:74-79, with Python's tuple comparison of the
    (score, key) pairs) against the reference's own results for finite max_hypotheses: bit-exact, including the decodes whose beam lost
    every path to the last transcript state (score -inf, the labelling of the last hypothesis in dict order) and those that end with no
    hypothesis at all (AttributeError in the reference)."""
    lp, tr, P = pruned_case_inputs(rec)
    nm = rec["name"]
    if rec["exception"] is not None:
        assert rec["exception"] == "AttributeError"
        with pytest.raises(oracle.OracleDecodeError) as e:
            oracle.viterbi_decode_table(lp, tr, P, rec["fs"], rec["max_len"], max_hypotheses=rec["max_hypotheses"])
        assert e.value.status == oracle.ST_NO_HYPOTHESIS
        return
    score, labels, seg_label, seg_len = oracle.viterbi_decode_table(lp, tr, P, rec["fs"], rec["max_len"], max_hypotheses=rec["max_hypotheses"])
    assert f64_bits(score) == f64_bits(PZ[f"{nm}__score"][0]), (score, PZ[f"{nm}__score"][0])
    np.testing.assert_array_equal(labels, PZ[f"{nm}__labels"])
    np.testing.assert_array_equal(seg_label, PZ[f"{nm}__seg_label"])
    np.testing.assert_array_equal(seg_len, PZ[f"{nm}__seg_len"])


def test_oracle_beam_that_cannot_prune_equals_no_beam():
    cs = META["cases"][5]
    nm = cs["name"]
    lp, tr = viterbi_case_inputs(Z, cs), Z[f"{nm}__transcript"]
    a = oracle.viterbi_decode_table(lp, tr, Z[f"{nm}__P"], FS, MAXLEN)
    for mh in (len(tr) * (MAXLEN // FS), 10 ** 6, 0):          # (0: Python's tmp[0:-0] deletes nothing)
        b = oracle.viterbi_decode_table(lp, tr, Z[f"{nm}__P"], FS, MAXLEN, max_hypotheses=mh)
        assert f64_bits(a[0]) == f64_bits(b[0]) and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
