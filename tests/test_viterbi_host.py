"""CPU: host-side logic of the product's Viterbi wrapper (no kernel is launched here).

* PoissonModel (vectorised) reproduces the reference's table (golden rows + the literal oracle loop).
* last_in_dict_order(): the closed form for the reference's degenerate outcomes is checked against
  the LITERAL oracle (ordered-dict emulation) over a sweep of column counts, transcript lengths,
  slot counts and NaN positions."""
import os

import numpy as np
import pytest

import oracle
from mucon_amd import synth
from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
from mucon_amd.core.viterbi.viterbi import last_in_dict_order

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_poisson_model_matches_reference_rows():
    g = np.load(os.path.join(GOLD, "glue_cases.npz"))
    for i in range(4):
        mu = g[f"g{i}__mu"]
        pm = PoissonModel(mu)
        np.testing.assert_allclose(pm.norms, g[f"g{i}__norms"], rtol=1e-13, atol=1e-9)
        np.testing.assert_allclose(pm.poisson[30:2000:30], g[f"g{i}__poisson_rows"], rtol=1e-13, atol=1e-9)
        tr = g[f"g{i}__transcript"]
        P = pm.rows_for(tr, 30)
        assert P.shape == (66, len(tr))
        np.testing.assert_array_equal(P, pm.poisson[30:2000:30][:, tr])


def test_poisson_model_bits_equal_literal_loop():
    rng = np.random.default_rng(0)
    for it in range(10):
        mu = rng.uniform(0.2, 2500, size=48)
        mu[:5] = [1.0, 0.3, 0.5, 1.5, 2.0]
        with np.errstate(all="ignore"):
            a = oracle.poisson_table(mu, 2000)
        b = PoissonModel(mu).poisson
        same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), np.argwhere(~same)[:4]
        for l in (0, 1, 30, 1999, 2000, 2500):
            x, y = PoissonModel(mu).score(l, 7), (-np.inf if l >= 2000 else a[l, 7])
            assert (np.isnan(x) and np.isnan(y)) or x == y


def test_grammar_surface():
    g = SingleTranscriptGrammar([3, 5, 5], 48)
    assert g.n_classes() == 48 and g.start_symbol() == -1 and g.end_symbol() == -2
    assert g.possible_successors((-1,)) == {3}
    assert g.possible_successors((-1, 3, 5)) == {5}
    assert g.possible_successors((-1, 3, 5, 5)) == {-2}
    assert g.possible_successors((-1, 4)) == set()
    assert g.score((-1, 3), 5) == 0.0 and g.score((-1, 3), 6) == -np.inf


@pytest.mark.parametrize("J", [1, 2, 3, 5, 8])
def test_last_in_dict_order_against_literal_oracle(J):
    """fs = 1, max_len = J: K = T columns, J slots.  A NaN column at n* makes every state >= n*
    incomparable, so the reference returns the last hypothesis (dict order) among states < n*."""
    C = 6
    checked = 0
    for N in range(1, 7):
        for K in range(1, J * N + 3):
            lp = synth.emissions(1000 + 31 * N + K, K, C)
            tr = synth.transcript(7 * N + K, N, C)
            for nstar in [None] + list(range(N)):
                P = np.zeros((J, N))
                if J * 1 >= J:  # length J*fs == max_len scores -inf in the reference (length_model.py:77)
                    P[J - 1, :] = -np.inf
                if nstar is not None:
                    P[:, nstar] = np.nan
                n_lim = N if nstar is None else nstar
                degenerate = (nstar is not None) or (K < N)
                if not degenerate:
                    continue
                want = last_in_dict_order(K, J, n_lim)
                try:
                    score, labels, seg_label, seg_len = oracle.viterbi_decode_table(lp, tr, P, 1, J)
                except oracle.OracleDecodeError as e:
                    assert e.status == oracle.ST_NO_HYPOTHESIS
                    assert want is None, (J, N, K, nstar, want)
                    continue
                assert want is not None, (J, N, K, nstar)
                assert score == -np.inf
                assert (len(seg_len) - 1, int(seg_len[-1]) - 1) == want, (J, N, K, nstar, want, seg_len)
                checked += 1
    assert checked > 20


def test_wrapper_raises_like_reference():
    v = Viterbi(None, None, frame_sampling=30)
    v.grammar = SingleTranscriptGrammar([1, 2], 48)
    v.length_model = PoissonModel(np.full(48, 300.0))
    with pytest.raises(IndexError):
        v._prepare(29)
    with pytest.raises(AttributeError):
        v._prepare(30 * (66 * 2 + 1))
    tr, P, force = v._prepare(3000)
    assert force is None and P.shape == (66, 2)
    tr, P, force = v._prepare(45)       # K = 1 < N = 2
    assert force == (0, 0)
    mu = np.full(48, 300.0)
    mu[1] = 0.3                          # NaN norms for the first transcript label
    with np.errstate(all="ignore"):
        v.length_model = PoissonModel(mu)
    with pytest.raises(AttributeError):
        v._prepare(900)


def test_max_hypotheses_routing():
    """prune() (reference viterbi.py:74-79) only acts when more than max_hypotheses hypotheses are alive; a single transcript of N
    states has at most N * (max_length // frame_sampling) of them: from that bound on (and for inf, and for 0 -- Python's tmp[0:-0] is
    empty) the decode is the unpruned one and takes the ordinary kernels; below it the decode runs under the reference's beam
    (csrc/viterbi_beam.hip).  A float max_hypotheses makes the reference's slice raise TypeError."""
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
    tr = [3, 7, 3, 1]
    mu = np.full(12, 200.0)
    for mh, beam in ((np.inf, None), (4 * 66, None), (10 ** 6, None), (0, None), (4 * 66 - 1, 263), (50, 50), (np.int64(7), 7)):
        v = Viterbi(SingleTranscriptGrammar(tr, 12), PoissonModel(mu), frame_sampling=30, max_hypotheses=mh)
        assert v._beam(len(tr)) == beam
        if beam is None:
            t, P, force = v._prepare(900)
            assert P.shape == (66, 4) and force is None
        else:
            t, P = v._prepare_beam(900)
            assert P.shape == (66, 4) and t.dtype == np.int32
            with pytest.raises(IndexError):
                v._prepare_beam(29)
    for mh, exc in ((50.0, TypeError), (50.5, TypeError), (-3, ValueError)):
        with pytest.raises(exc):
            Viterbi(SingleTranscriptGrammar(tr, 12), PoissonModel(mu), frame_sampling=30, max_hypotheses=mh)._beam(len(tr))
    long_tr = list(range(40)) + list(range(24))                  # N = 64: 64 x 66 = 4,224 hypotheses at most
    v = Viterbi(SingleTranscriptGrammar(long_tr, 48), PoissonModel(np.full(48, 200.0)), frame_sampling=30, max_hypotheses=4100)
    assert v._beam(64) == 4100
    with pytest.raises(NotImplementedError, match="holds 4096"):
        v._prepare_beam(9000)
    mu_nan = mu.copy()
    mu_nan[7] = 0.3                                              # mean length < 0.5: NaN length scores (length_model.py:56-58)
    with np.errstate(all="ignore"):
        v = Viterbi(SingleTranscriptGrammar(tr, 12), PoissonModel(mu_nan), frame_sampling=30, max_hypotheses=20)
    with pytest.raises(NotImplementedError, match="NaN"):
        v._prepare_beam(900)


def test_the_beam_fixture_says_what_the_docs_say():
    """tests/golden/viterbi_pruned.* (the reference's own beam search, tools/make_golden_pruned.py): a beam never scores above the
    exact decode, equals it from N * J hypotheses on, and a beam of one loses every path (score -inf)."""
    import os

    from helpers import load_pruned_golden
    pz, pmeta = load_pruned_golden()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "viterbi_cases.npz"))
    seen = 0
    for r in pmeta["cases"]:
        base = r["name"].rsplit("_m", 1)[0]
        if f"{base}__score" not in gold.files:
            continue                                        # (the tie / small-sampling cases have no unpruned golden of their own)
        assert r["exception"] is None
        seen += 1
        score, exact = float(pz[f"{r['name']}__score"][0]), float(gold[f"{base}__score"][0])
        assert score <= exact
        N, J = len(r["transcript"]), r["max_len"] // r["fs"]
        if r["max_hypotheses"] >= N * J:
            assert score == exact and np.array_equal(pz[f"{r['name']}__labels"], gold[f"{base}__labels"])
        if r["max_hypotheses"] == 1:
            assert score == -np.inf
    assert seen == 42


def test_poisson_rows_for_many_equals_a_model_per_video():
    """poisson_rows_for_many (the evaluation's chunk-wide table builder) against PoissonModel(mu).rows_for(transcript, fs) per video, bit for bit:
    mean lengths from 0.2 (NaN norms, length_model.py:56-58) to 5,000, repeated classes, one-state transcripts, frame_sampling 1 / 7 / 30."""
    from mucon_amd.core.viterbi import PoissonModel, PoissonRows, poisson_rows_for_many
    rng = np.random.default_rng(3)
    for fs, max_len in ((30, 2000), (7, 300), (1, 40)):
        mus, trs = [], []
        for v in range(25):
            mu = rng.uniform(0.6, 900.0, 48)
            mu[rng.integers(0, 48, 3)] = rng.choice([0.2, 0.49, 1.0, 1.5, 2.0, 2.5, 4999.7])
            mus.append(mu)
            trs.append(rng.integers(0, 48, int(rng.integers(1, 20))).tolist())
        got = poisson_rows_for_many(mus, trs, fs, max_len)
        for mu, tr, g in zip(mus, trs, got):
            with np.errstate(all="ignore"):
                want = PoissonModel(mu, max_length=max_len).rows_for(tr, fs)
            assert g.shape == want.shape and g.flags.c_contiguous
            assert g.tobytes() == want.tobytes()
            lm = PoissonRows(g, max_len, fs)
            assert lm.max_length() == max_len and lm.rows_for(tr, fs) is g
            with pytest.raises(ValueError):
                lm.rows_for(tr + [1], fs)
    assert poisson_rows_for_many([], [], 30, 2000) == []
