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


# ---------------------------------------------------------------------------------------------- the beam (max_hypotheses)
from helpers import load_pruned_golden, pruned_case_inputs  # noqa: E402

PZ, PMETA = load_pruned_golden()


def _beam_decode(lp, tr, P, fs, max_len, mh):
    from mucon_amd.core.viterbi import SingleTranscriptGrammar, Viterbi
    v = Viterbi(None, None, frame_sampling=fs, max_hypotheses=mh)
    v.grammar = SingleTranscriptGrammar([int(x) for x in tr], lp.shape[1])
    v.length_model = TableModel(P, max_len)
    return v.decode(torch.from_numpy(lp).cuda() if isinstance(lp, np.ndarray) else lp)


@pytest.mark.parametrize("rec", PMETA["cases"], ids=[r["name"] for r in PMETA["cases"]])
def test_beam_decode_matches_the_references_beam(rec):
    """Viterbi(max_hypotheses = M).decode through csrc/viterbi_beam.hip against the reference's own beam search (74 decodes,
    tools/make_golden_pruned.py): score bits, labels, segments -- also where every score ties and the key tuples decide, at
    frame_sampling 1 and 7, where the beam loses every path into the last state (score -inf, truncated labelling of the last hypothesis
    in dict order) and where it loses every hypothesis (AttributeError)."""
    from mucon_amd.core.viterbi.viterbi import NoHypothesisError
    lp, tr, P = pruned_case_inputs(rec)
    nm = rec["name"]
    if rec["exception"] is not None:
        with pytest.raises(NoHypothesisError):
            _beam_decode(lp, tr, P, rec["fs"], rec["max_len"], rec["max_hypotheses"])
        return
    got = _beam_decode(lp, tr, P, rec["fs"], rec["max_len"], rec["max_hypotheses"])
    _check(got, PZ[f"{nm}__score"][0], PZ[f"{nm}__labels"], PZ[f"{nm}__seg_label"], PZ[f"{nm}__seg_len"])


def test_beam_decode_against_the_oracle_random_and_tied():
    """Further seeded cases against the literal oracle (oracle/viterbi_oracle.c:dict_prune): small and large beams, integer-valued
    emissions and length scores (ties everywhere), -inf length scores, frame_sampling 1 / 2 / 7 / 30, several videos per call."""
    from mucon_amd.core.viterbi import SingleTranscriptGrammar, Viterbi
    from mucon_amd.core.viterbi.viterbi import NoHypothesisError
    rng = np.random.default_rng(7)
    for fs in (1, 2, 7, 30):
        J = int(rng.integers(3, 20))
        max_len = J * fs + int(rng.integers(0, fs))
        for mh in (1, 2, 5, 17, 60):
            lps, trs, lms, wants = [], [], [], []
            for i in range(6):
                N = int(rng.integers(1, 9))
                if mh >= N * J:
                    N = mh // J + 1
                K = int(rng.integers(1, J * N + 1))
                T = K * fs + int(rng.integers(0, fs))
                tr = rng.integers(0, C, N).astype(np.int32)
                mode = i % 3
                lp = (rng.integers(-3, 1, (T, C)).astype(np.float32) if mode == 0 else
                      rng.standard_normal((T, C)).astype(np.float32) if mode == 1 else np.full((T, C), -1.0, np.float32))
                P = rng.integers(-2, 1, (J, N)).astype(np.float64) if mode != 1 else rng.standard_normal((J, N))
                P[rng.random((J, N)) < 0.1] = -np.inf
                try:
                    wants.append(oracle.viterbi_decode_table(lp, tr, P, fs, max_len, max_hypotheses=mh))
                except oracle.OracleDecodeError as e:
                    assert e.status == oracle.ST_NO_HYPOTHESIS
                    wants.append(None)
                lps.append(torch.from_numpy(lp).cuda())
                trs.append([int(x) for x in tr])
                lms.append(TableModel(P, max_len))
            got = Viterbi(None, None, frame_sampling=fs, max_hypotheses=mh).decode_batch(lps, trs, lms, return_exceptions=True)
            for g, w in zip(got, wants):
                if w is None:
                    assert isinstance(g, NoHypothesisError), g
                else:
                    _check(g, *w)


def test_beam_decode_long_video_wide_beam():
    """T = 9,741 / N = 30 (the longest Breakfast video) under beams of 100 and 1,000, and a config-5-sized video (T = 16,384, N = 64)
    under 500 and 4,000 (the list nearly full), a one-state transcript, a one-column video, against the oracle."""
    for T, N, mh, seed in ((9741, 30, 100, 1), (9741, 30, 1000, 2), (16384, 64, 500, 3), (16384, 64, 4000, 4), (2000, 1, 20, 5), (45, 3, 2, 6)):
        tr = synth.transcript(seed, N, C)
        lp = synth.emissions(seed + 1, T, C, labels=synth.segment_labels(seed + 2, T, tr))
        mu = np.full(C, float(T) / N)
        P = oracle.length_rows(oracle.poisson_table(mu, MAXLEN), tr, FS, MAXLEN)
        want = oracle.viterbi_decode_table(lp, tr, P, FS, MAXLEN, max_hypotheses=mh)
        _check(_beam_decode(lp, tr, P, FS, MAXLEN, mh), *want)


def test_beam_entry_argument_checks():
    import ctypes

    from mucon_amd import _lib
    lib = _lib.load()
    vids = (_lib.ViterbiVideo * 1)()
    assert lib.mucon_viterbi_decode_beam(1, vids, C, FS, MAXLEN, 10, None, None, None, None, None) == _lib.E_ARG
    sc, ns, st, sg = np.zeros(1), np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(8, np.int32)
    lp = torch.zeros(90, C, device="cuda")
    tr, P = np.zeros(3, np.int32), np.zeros((66, 3))
    q = vids[0]
    q.lp, q.transcript, q.table, q.T, q.N, q.force_n, q.force_j = lp.data_ptr(), tr.ctypes.data, P.ctypes.data, 90, 3, -1, -1
    args = (sc.ctypes.data, ns.ctypes.data, st.ctypes.data, sg.ctypes.data, None)
    assert lib.mucon_viterbi_decode_beam(1, vids, C, FS, MAXLEN, 0, *args) == _lib.E_ARG           # 0 never prunes: the other entry
    assert lib.mucon_viterbi_decode_beam(1, vids, C, FS, MAXLEN, 4094, *args) == _lib.E_ARG        # 4094 + 3 > 4096
    assert lib.mucon_viterbi_decode_beam(1, vids, 65, FS, MAXLEN, 10, *args) == _lib.E_ARG
    assert lib.mucon_viterbi_decode_beam(1, vids, C, 1, 200, 10, *args) == _lib.E_ARG              # 200 length slots
    assert lib.mucon_viterbi_decode_beam(1, vids, C, FS, MAXLEN, 10, *args) == _lib.OK and st[0] == _lib.VIT_OK and ns[0] == 3
    q.T = 20
    assert lib.mucon_viterbi_decode_beam(1, vids, C, FS, MAXLEN, 10, *args) == _lib.OK and st[0] == _lib.VIT_INDEX_ERROR
    del ctypes


# ---- (r6, ABI 7) the PoissonModel's length scores built ON THE DEVICE from [3][N] parameters (mucon_viterbi_decode_host_poisson) ----------------
def _device_rows(params, log_fact, J, fs, max_len):
    from mucon_amd import _lib
    lib = _lib.load()
    N = params.shape[1]
    p_d, lf_d = torch.from_numpy(np.ascontiguousarray(params)).cuda(), torch.from_numpy(np.ascontiguousarray(log_fact)).cuda()
    out = torch.full((J, N), 123.0, dtype=torch.float64, device="cuda")
    _lib.check(lib.mucon_test_vit_rows(_lib.ptr(p_d), _lib.ptr(lf_d), N, J, fs, max_len, _lib.ptr(out), _lib.current_stream_ptr()), "test_vit_rows")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _same_rows(got, want):
    """bit for bit -- except that a NaN only has to be a NaN: x86 and gfx950 generate different default NaNs (sign bit) for inf - inf / 0 * inf, and
    no comparison or arg-max of the decode can tell two NaNs apart"""
    nan = np.isnan(want)
    return got.shape == want.shape and (np.isnan(got) == nan).all() and (got.view(np.int64)[~nan] == want.view(np.int64)[~nan]).all()


def test_device_built_length_rows():
    """The rows the kernels build -- ((l * ln mu - mu) - logFak_l) - norm as four single IEEE double operations on host-computed ln mu, mu, norm
    and the shared log-factorial row -- equal, BIT FOR BIT, (a) the rows the reference's own PoissonModel produced for every golden case
    (tests/golden: `__P`, generated by importing the reference) and (b) PoissonModel.rows_for of this package, incl. a mean length < 0.5 (the
    reference's norms are NaN, length_model.py:56-58: NaN rows), mu = 0 and inf, and lengths cut off by max_length (-inf, :76-80)."""
    from mucon_amd.core.viterbi import PoissonModel, poisson_params_for_many
    n = 0
    for cs in META["cases"]:
        nm = cs["name"]
        if cs["kind"] != "poisson" or f"{nm}__mu" not in Z.files:
            continue
        mu, tr, P_ref = Z[f"{nm}__mu"], Z[f"{nm}__transcript"], Z[f"{nm}__P"]
        pp = poisson_params_for_many([mu], [tr], FS, MAXLEN)[0]
        got = _device_rows(pp.params, pp.log_fact, MAXLEN // FS, FS, MAXLEN)
        assert _same_rows(got, P_ref), nm
        n += 1
    assert n >= 30
    rng = np.random.default_rng(5)
    for fs, max_len in ((30, 2000), (30, 1980), (1, 7), (3, 20), (50, 6400), (10, 500)):
        mus = [rng.uniform(0.05, 1200.0, C) for _ in range(6)]
        mus[1][:8] = [0.3, 0.49999, 0.5, 0.0, np.inf, 1e-300, 1999.7, 2.5]
        trs = [rng.integers(0, C, int(rng.integers(1, 65))) for _ in range(6)]
        trs[1] = np.arange(16) % 8
        with np.errstate(all="ignore"):
            for mu, tr, pp in zip(mus, trs, poisson_params_for_many(mus, trs, fs, max_len)):
                want = PoissonModel(mu, max_length=max_len).rows_for(tr, fs)
                got = _device_rows(pp.params, pp.log_fact, max_len // fs, fs, max_len)
                assert _same_rows(got, want), (fs, max_len)
                assert _same_rows(pp.rows_for(tr, fs), want)
        if max_len % fs == 0:
            assert np.isneginf(got[-1]).all()       # the row at l = max_len


def _decode_params(lp, tr, mu, fs=FS, max_len=MAXLEN):
    from mucon_amd.core.viterbi import SingleTranscriptGrammar, Viterbi, poisson_params_for_many
    with np.errstate(all="ignore"):
        pp = poisson_params_for_many([mu], [tr], fs, max_len)[0]
    v = Viterbi(SingleTranscriptGrammar([int(x) for x in tr], lp.shape[1]), pp, frame_sampling=fs)
    return v.decode(lp)


@pytest.mark.parametrize("cs", [c for c in META["cases"] if c["kind"] == "poisson"], ids=[c["name"] for c in META["cases"] if c["kind"] == "poisson"])
def test_decode_with_device_built_rows_matches_reference_golden(cs):
    nm = cs["name"]
    lp = viterbi_case_inputs(Z, cs)
    got = _decode_params(torch.from_numpy(lp).cuda(), Z[f"{nm}__transcript"], Z[f"{nm}__mu"])
    _check(got, Z[f"{nm}__score"][0], Z[f"{nm}__labels"], Z[f"{nm}__seg_label"], Z[f"{nm}__seg_len"])


@pytest.mark.parametrize("er", META["errors"], ids=[e["name"] for e in META["errors"]])
def test_error_behaviour_with_device_built_rows(er):
    lp = torch.from_numpy(synth.emissions(er["seed"], er["T"], C)).cuda()
    mu = np.full(C, 300.0)
    if er["mu_small"]:
        mu[er["transcript"][0]] = 0.3
    with pytest.raises({"IndexError": IndexError, "AttributeError": AttributeError}[er["exception"]]):
        _decode_params(lp, np.asarray(er["transcript"]), mu)


def test_batched_decodes_with_device_built_rows_against_oracle():
    """64 ragged videos (every kernel family: the latency kernels of 1..8 videos, the throughput launches), a NaN-norm state in the middle of
    a transcript (the host resolves the truncated outcome from the parameters alone), BASELINE config 5 x 12 -- parameter form == the oracle,
    == the host-table form; and the all-device entry (mucon_viterbi_decode_batch_poisson)."""
    from mucon_amd import ops
    from mucon_amd.core.viterbi import PoissonModel, Viterbi, poisson_params_for_many
    lps, trs, mus, wants = [], [], [], []
    for i in range(64):
        T = 200 + int(synth.integers(900 + i, 1, 0, 3800)[0])
        N = 1 + int(synth.integers(901 + i, 1, 0, 12)[0])
        N = max(min(N, T // 30), -(-(T // 30) // 66))
        tr = synth.transcript(902 + i, N, C)
        lp = synth.emissions(903 + i, T, C, labels=synth.segment_labels(904 + i, T, tr))
        mu = np.full(C, float(T) / N)
        if i % 9 == 4 and N >= 3:
            mu[tr[N // 2]] = 0.3          # NaN norms from that state on
        wants.append(mu)
        lps.append(torch.from_numpy(lp).cuda())
        trs.append([int(x) for x in tr])
        mus.append(mu)
    dec = Viterbi(None, None, frame_sampling=FS)
    with np.errstate(all="ignore"):
        pps = poisson_params_for_many(mus, trs, FS, MAXLEN)
        ref = dec.decode_batch(lps, trs, [PoissonModel(mu) for mu in mus], return_exceptions=True)
    for nv in (64, 1, 3, 8):
        got = dec.decode_batch(lps[:nv], trs[:nv], pps[:nv], return_exceptions=True)
        for g, r in zip(got, ref[:nv]):
            if isinstance(r, Exception):
                assert type(g) is type(r)
            else:
                _check(g, r[0], np.asarray(r[1], dtype=np.int32), np.asarray([s.label for s in r[2]], dtype=np.int32), np.asarray([s.length for s in r[2]], dtype=np.int32))
    # against the literal oracle on the healthy ones
    for i in (0, 7, 21, 40, 63):
        if np.isfinite(mus[i]).all() and (mus[i] >= 0.5).all():
            lp = lps[i].cpu().numpy()
            _check(dec.decode_batch([lps[i]], [trs[i]], [pps[i]])[0], *oracle.viterbi_decode(lp, np.asarray(trs[i], dtype=np.int32), mus[i], FS, MAXLEN))
    # BASELINE config 5 x 12 through both host entries and the all-device entry
    T, N = 16384, 64
    tr = synth.transcript(11, N, C)
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    lp5 = [torch.from_numpy(synth.emissions(300 + k, T, C, labels=synth.segment_labels(13, T, tr))).cuda() for k in range(12)]
    pp = poisson_params_for_many([mu] * 12, [tr] * 12, FS, MAXLEN)
    a = dec.decode_batch(lp5, [tr] * 12, pp)
    b = dec.decode_batch(lp5, [tr] * 12, [PoissonModel(mu)] * 12)
    for x, y in zip(a, b):
        assert f64_bits(x[0]) == f64_bits(y[0]) and x[1] == y[1]
    _check(a[0], *oracle.viterbi_decode(lp5[0].cpu().numpy(), tr, mu, FS, MAXLEN))
    P = PoissonModel(mu).rows_for(tr, FS)
    d_par = ops.viterbi_decode_batch_device(lp5, [tr] * 12, [p.params for p in pp], FS, MAXLEN, log_fact=pp[0].log_fact)
    d_tab = ops.viterbi_decode_batch_device(lp5, [tr] * 12, [P] * 12, FS, MAXLEN)
    torch.cuda.synchronize()
    assert torch.equal(d_par.score.view(torch.int64), d_tab.score.view(torch.int64)) and torch.equal(d_par.seg_len, d_tab.seg_len)
    assert torch.equal(d_par.labels, d_tab.labels) and torch.equal(d_par.status, d_tab.status)
