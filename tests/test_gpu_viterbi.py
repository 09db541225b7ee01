"""GPU parity of the Viterbi decode through the C ABI: bit-exact (score bits, labels, segments)
against the golden vectors produced by the reference's Viterbi.decode, and against the literal C
oracle on further seeded cases (batched launch, BASELINE config 5 size, small slot counts)."""
import numpy as np
import pytest
import torch

import oracle
from mucon_amd import synth
from helpers import C, f64_bits, load_viterbi_golden, viterbi_case_inputs

pytestmark = pytest.mark.gpu

Z, META = load_viterbi_golden()
FS, MAXLEN = META["fs"], META["max_length"]


class TableModel:
    """Length model backed by a golden table P[J x N] (rows = lengths fs, 2fs, ...)."""

    def __init__(self, P, max_len=MAXLEN):
        self.P, self.max_len = P, max_len

    def max_length(self):
        return self.max_len

    def rows_for(self, transcript, fs):
        return self.P


def _decode(lp, tr, P, fs=FS, max_len=MAXLEN):
    from mucon_amd.core.viterbi import SingleTranscriptGrammar, Viterbi
    v = Viterbi(None, None, frame_sampling=fs)
    v.grammar = SingleTranscriptGrammar([int(x) for x in tr], lp.shape[1])
    v.length_model = TableModel(P, max_len)
    return v.decode(lp)


def _check(got, score, labels, seg_label, seg_len):
    g_score, g_labels, g_segs = got
    assert f64_bits(g_score) == f64_bits(score), (g_score, score)
    np.testing.assert_array_equal(np.asarray(g_labels, dtype=np.int32), labels)
    np.testing.assert_array_equal(np.asarray([s.label for s in g_segs], dtype=np.int32), seg_label)
    np.testing.assert_array_equal(np.asarray([s.length for s in g_segs], dtype=np.int32), seg_len)


@pytest.mark.parametrize("cs", META["cases"], ids=[c["name"] for c in META["cases"]])
def test_decode_matches_reference_golden(cs):
    nm = cs["name"]
    lp = viterbi_case_inputs(Z, cs)
    got = _decode(torch.from_numpy(lp).cuda(), Z[f"{nm}__transcript"], Z[f"{nm}__P"])
    _check(got, Z[f"{nm}__score"][0], Z[f"{nm}__labels"], Z[f"{nm}__seg_label"], Z[f"{nm}__seg_len"])
    assert isinstance(got[0], np.float64) and isinstance(got[1], list) and len(got[1]) == cs["T"]


def test_decode_accepts_numpy_like_the_reference():
    cs = META["cases"][4]
    nm = cs["name"]
    got = _decode(viterbi_case_inputs(Z, cs), Z[f"{nm}__transcript"], Z[f"{nm}__P"])
    _check(got, Z[f"{nm}__score"][0], Z[f"{nm}__labels"], Z[f"{nm}__seg_label"], Z[f"{nm}__seg_len"])


@pytest.mark.parametrize("er", META["errors"], ids=[e["name"] for e in META["errors"]])
def test_error_behaviour_matches_reference(er):
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
    lp = torch.from_numpy(synth.emissions(er["seed"], er["T"], C)).cuda()
    mu = np.full(C, 300.0)
    if er["mu_small"]:
        mu[er["transcript"][0]] = 0.3
    v = Viterbi(None, None, frame_sampling=FS)
    v.grammar = SingleTranscriptGrammar(er["transcript"], C)
    with np.errstate(all="ignore"):
        v.length_model = PoissonModel(mu)
    with pytest.raises({"IndexError": IndexError, "AttributeError": AttributeError}[er["exception"]]):
        v.decode(lp)


def test_full_pipeline_poisson_model_against_oracle():
    """PoissonModel built by the product from mean lengths (evaluator glue) + decode, vs the oracle."""
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
    for i, (T, N) in enumerate([(1500, 4), (2222, 6), (640, 3), (9000, 22)]):
        tr = synth.transcript(50 + i, N, C)
        gt = synth.segment_labels(60 + i, T, tr)
        lp = synth.emissions(70 + i, T, C, labels=gt)
        rel = synth.uniform01(80 + i, (N,)) + np.float32(0.1)
        rel = (rel / rel.sum()).astype(np.float32)
        mu = oracle.mean_lengths_from_s_head(rel, tr, T, C)
        want = oracle.viterbi_decode(lp, tr, mu, FS, MAXLEN)
        v = Viterbi(SingleTranscriptGrammar([int(x) for x in tr], C), PoissonModel(mu), frame_sampling=FS)
        _check(v.decode(torch.from_numpy(lp).cuda()), *want)


def test_batched_launch_many_videos():
    """64 ragged videos in one launch == each decoded by the oracle."""
    from mucon_amd.core.viterbi import PoissonModel, Viterbi
    lps, trs, lms, wants = [], [], [], []
    for i in range(64):
        T = 200 + int(synth.integers(900 + i, 1, 0, 3800)[0])
        N = 1 + int(synth.integers(901 + i, 1, 0, 12)[0])
        N = max(min(N, T // 30), -(-(T // 30) // 66))  # 1 <= N <= K, and K <= 66 N so that hypotheses survive
        tr = synth.transcript(902 + i, N, C)
        lp = synth.emissions(903 + i, T, C, labels=synth.segment_labels(904 + i, T, tr))
        mu = np.full(C, float(T) / N)
        wants.append(oracle.viterbi_decode(lp, tr, mu, FS, MAXLEN))
        lps.append(torch.from_numpy(lp).cuda())
        trs.append([int(x) for x in tr])
        lms.append(PoissonModel(mu))
    got = Viterbi(None, None, frame_sampling=FS).decode_batch(lps, trs, lms)
    for g, w in zip(got, wants):
        _check(g, *w)


@pytest.mark.parametrize("nv,n_hi,t_hi", [(2, 6, 3000), (3, 16, 6000), (5, 40, 9000), (8, 90, 12000), (1, 14, 30000), (2, 3, 700)])
def test_latency_calls_a_few_ragged_videos(nv, n_hi, t_hi):
    """Calls of 1..8 videos take the one-launch kernel (one video, <= 16 states, <= 640 columns) or the pair kernel (both phases as two
    workgroups of one launch, the decode behind the published column count; back-pointers in LDS): ragged lengths and transcript
    sizes -- the launch's lane layout is the largest transcript's -- against the oracle, repeated (the scratch, and with it the
    published count of the call before, is reused)."""
    from mucon_amd.core.viterbi import PoissonModel, Viterbi
    for rep in range(3):
        lps, trs, lms, wants = [], [], [], []
        for i in range(nv):
            seed = 7000 + 100 * rep + 10 * i + nv
            T = 90 + int(synth.integers(seed, 1, 0, t_hi)[0])
            N = 1 + int(synth.integers(seed + 1, 1, 0, n_hi)[0]) if i else n_hi
            N = max(min(N, T // 30), -(-(T // 30) // 66))  # 1 <= N <= K, and K <= 66 N so that hypotheses survive
            tr = synth.transcript(seed + 2, N, C)
            lp = synth.emissions(seed + 3, T, C, labels=synth.segment_labels(seed + 4, T, tr))
            mu = np.ones(C)
            mu[np.unique(tr)] = float(T) / N
            wants.append(oracle.viterbi_decode(lp, tr, mu, FS, MAXLEN))
            lps.append(torch.from_numpy(lp).cuda())
            trs.append([int(x) for x in tr])
            lms.append(PoissonModel(mu))
        got = Viterbi(None, None, frame_sampling=FS).decode_batch(lps, trs, lms)
        for g, w in zip(got, wants):
            _check(g, *w)


def test_baseline_config5_long_video():
    """BASELINE config 5: T = 16384, 64-state transcript (K = 546 columns, 64 x 66 hypotheses)."""
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
    T, N = 16384, 64
    tr = synth.transcript(11, N, C)
    lp = synth.emissions(12, T, C, labels=synth.segment_labels(13, T, tr))
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    want = oracle.viterbi_decode(lp, tr, mu, FS, MAXLEN)
    v = Viterbi(SingleTranscriptGrammar([int(x) for x in tr], C), PoissonModel(mu), frame_sampling=FS)
    _check(v.decode(torch.from_numpy(lp).cuda()), *want)
    # uninformative emissions: the length model and the tie rules decide
    lp2 = synth.emissions(14, T, C)
    _check(v.decode(torch.from_numpy(lp2).cuda()), *oracle.viterbi_decode(lp2, tr, mu, FS, MAXLEN))


@pytest.mark.parametrize("fs,max_len", [(1, 7), (3, 20), (10, 500), (30, 2000), (50, 6400)])
def test_other_sampling_and_slot_counts(fs, max_len):
    """Slot counts J = max_len // fs from 6 to 128, incl. lengths that hit max_len (score -inf)."""
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
    J = max_len // fs
    for i, N in enumerate([1, 2, 5, 9]):
        K = max(N, min(J * N, 3 * J + i))
        T = K * fs + (i % fs)
        tr = synth.transcript(30 + i, N, 12)
        lp = synth.emissions(31 + i, T, 12, labels=synth.segment_labels(32 + i, T, tr))
        mu = np.full(12, max(1.0, T / N))
        with np.errstate(all="ignore"):
            P = oracle.length_rows(oracle.poisson_table(mu, max_len), tr, fs, max_len)
        want = oracle.viterbi_decode_table(lp, tr, P, fs, max_len)
        v = Viterbi(SingleTranscriptGrammar([int(x) for x in tr], 12), PoissonModel(mu, max_length=max_len), frame_sampling=fs)
        _check(v.decode(torch.from_numpy(lp).cuda()), *want)


def _oracle_case(seed, T, N, informative=True):
    tr = synth.transcript(seed, N, C)
    lp = synth.emissions(seed + 1, T, C, labels=synth.segment_labels(seed + 2, T, tr) if informative else None)
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    return lp, tr, mu


@pytest.mark.parametrize("N", [9, 16, 17, 32, 33, 48, 64, 65, 100, 128])
def test_every_lane_layout_of_the_register_kernel(N):
    """The DP keeps the hypotheses in registers, 8 / 4 / 2 / 1 lanes per transcript state on one wave (N <= 8, 16, 32) or two
    (N <= 64, 128); each layout against the oracle, with informative and with uninformative emissions (ties, -inf lengths)."""
    from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
    for seed, T, informative in ((400 + N, 45 * N + 300, True), (500 + N, 120 * N + 17, False)):
        lp, tr, mu = _oracle_case(seed, T, N, informative)
        want = oracle.viterbi_decode(lp, tr, mu, FS, MAXLEN)
        v = Viterbi(SingleTranscriptGrammar([int(x) for x in tr], C), PoissonModel(mu), frame_sampling=FS)
        _check(v.decode(torch.from_numpy(lp).cuda()), *want)


def test_lds_kernel_on_the_golden_cases():
    """The one-wave-per-state LDS kernel (more than 66 length slots, or more than 128 states) on the cases the register kernel
    normally takes: both must give the reference's bits."""
    from mucon_amd import _lib
    _lib.set_knob("MUCON_VIT_LANES", 0)
    try:
        for cs in META["cases"]:
            nm = cs["name"]
            got = _decode(torch.from_numpy(viterbi_case_inputs(Z, cs)).cuda(), Z[f"{nm}__transcript"], Z[f"{nm}__P"])
            _check(got, Z[f"{nm}__score"][0], Z[f"{nm}__labels"], Z[f"{nm}__seg_label"], Z[f"{nm}__seg_len"])
        lp, tr, mu = _oracle_case(77, 16384, 64)
        from mucon_amd.core.viterbi import PoissonModel, SingleTranscriptGrammar, Viterbi
        v = Viterbi(SingleTranscriptGrammar([int(x) for x in tr], C), PoissonModel(mu), frame_sampling=FS)
        _check(v.decode(torch.from_numpy(lp).cuda()), *oracle.viterbi_decode(lp, tr, mu, FS, MAXLEN))
    finally:
        _lib.set_knob("MUCON_VIT_LANES", 1)


def test_finite_max_hypotheses_decodes_exactly_against_the_references_beam():
    """The reference prunes to a beam for finite max_hypotheses (viterbi.py:74-79); the kernels do not prune.  Against the
    reference's own beam results (tests/golden/viterbi_pruned.*, also at frame_sampling 1 and 7 and with all-ties inputs): the decode
    equals the UNPRUNED result (the golden one where there is one, the oracle's otherwise) bit for bit for every max_hypotheses, its
    score is never below the beam's, and with informative emissions it is the beam's labelling wherever the beam kept the best path."""
    import warnings

    from helpers import load_pruned_golden, pruned_case_inputs
    from mucon_amd.core.viterbi import SingleTranscriptGrammar, Viterbi
    pz, pmeta = load_pruned_golden()
    same = 0
    for r in pmeta["cases"]:
        nm, base = r["name"], r["name"].rsplit("_m", 1)[0]
        lp, tr, P = pruned_case_inputs(r)
        v = Viterbi(None, None, frame_sampling=r["fs"], max_hypotheses=r["max_hypotheses"])
        v.grammar = SingleTranscriptGrammar([int(x) for x in tr], lp.shape[1])
        v.length_model = TableModel(P, r["max_len"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            got = v.decode(torch.from_numpy(lp).cuda())
        if f"{base}__score" in Z.files:
            want = (Z[f"{base}__score"][0], Z[f"{base}__labels"], Z[f"{base}__seg_label"], Z[f"{base}__seg_len"])
        else:
            want = oracle.viterbi_decode_table(lp, tr, P, r["fs"], r["max_len"])
        _check(got, *want)
        if r["exception"] is not None:
            continue                                         # the reference's beam lost every hypothesis; the exact decode has a result
        beam = float(pz[f"{nm}__score"][0])
        assert got[0] >= beam
        if f"{base}__score" in Z.files and beam == float(got[0]):
            np.testing.assert_array_equal(np.asarray(got[1], dtype=np.int32), pz[f"{nm}__labels"])
            same += 1
    assert same >= 25
