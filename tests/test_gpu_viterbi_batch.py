"""GPU parity of the THROUGHPUT schedule of the Viterbi decode -- what BENCH's batch64 / batch256 legs time -- bit-exact
against the literal C oracle (reference src/core/viterbi/viterbi.py:92-158):

  * mucon_viterbi_decode_batch, the all-device entry point (ops.viterbi_decode_batch_device: device job table, no completion
    flag, sizes unknown to the launch code): always viterbi_framescore_cols_kernel / viterbi_framescore_kernel +
    viterbi_dp_lanes_kernel<G, JG, NW> with the back-pointers in HBM scratch -- every instantiation <8,9,1>, <4,17,1>, <8,9,4>,
    <4,17,4>, <4,17,8> (N <= 8, 16, 32, 64, 128), single videos and ragged batches;
  * the same ragged batches through mucon_viterbi_decode_host (ops.viterbi_decode_batch, > 8 videos: the same two launches,
    results written into pinned host memory);
  * the three label formats of ABI 5 (int32 / uint8 / none + host-side expansion) against each other and the oracle.
"""
import numpy as np
import pytest
import torch

import oracle
from mucon_amd import _lib, ops, synth
from helpers import C, f64_bits

pytestmark = pytest.mark.gpu

FS, MAXLEN = 30, 2000
J = MAXLEN // FS


def _case(seed, T, N, kind):
    """(lp f32 [T, C], transcript i32 [N], table f64 [J, N]).  kind: 'informative' (emissions follow a segmentation),
    'noise' (the length model and the tie rules decide), 'ties' (constant emissions AND a flat length table: every comparison
    of the DP is a tie -- the reference's `<=` / `>=` rules alone pick the path)."""
    tr = synth.transcript(seed, N, C).astype(np.int32)
    if kind == "ties":
        return np.full((T, C), np.float32(-1.0), np.float32), tr, np.zeros((J, N), np.float64)
    lp = synth.emissions(seed + 1, T, C, labels=synth.segment_labels(seed + 2, T, tr) if kind == "informative" else None)
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    return lp, tr, oracle.length_rows(oracle.poisson_table(mu, MAXLEN), tr, FS, MAXLEN)


def _want(lp, tr, P):
    return oracle.viterbi_decode_table(lp, tr, P, FS, MAXLEN)


def _check_host(r, want, T):
    score, labels, seg_label, seg_len = want
    assert r.status == _lib.VIT_OK
    assert f64_bits(r.score) == f64_bits(score), (r.score, score)
    np.testing.assert_array_equal(r.seg_len, seg_len)
    assert r.n_seg == len(seg_len)
    assert r.labels.dtype == np.int32 and r.labels.shape == (T,)
    np.testing.assert_array_equal(r.labels, labels)


def _check_device(res, cases, wants, label_dtype):
    torch.cuda.synchronize()
    status, n_seg = res.status.cpu().numpy(), res.n_seg.cpu().numpy()
    score, seg = res.score.cpu().numpy(), res.seg_len.cpu().numpy()
    labels = res.labels.cpu().numpy() if res.labels is not None else None
    for v, ((lp, tr, P), (w_score, w_labels, w_seg_label, w_seg_len)) in enumerate(zip(cases, wants)):
        T = lp.shape[0]
        assert status[v] == _lib.VIT_OK, (v, status[v])
        assert f64_bits(score[v]) == f64_bits(w_score), (v, score[v], w_score)
        ns = int(n_seg[v])
        got_seg = seg[res.seg_off[v]: res.seg_off[v] + ns]
        np.testing.assert_array_equal(got_seg, w_seg_len)
        np.testing.assert_array_equal(tr[:ns], w_seg_label)
        if labels is not None:
            got = labels[res.label_off[v]: res.label_off[v] + T]
            assert got.dtype == (np.uint8 if label_dtype == torch.uint8 else np.int32)
            np.testing.assert_array_equal(got.astype(np.int32), w_labels)
        np.testing.assert_array_equal(ops.expand_labels(tr, got_seg, ns, T, FS), w_labels)     # the host-side expansion of the segments


def _device(cases, label_dtype):
    return ops.viterbi_decode_batch_device([torch.from_numpy(lp).cuda() for lp, _, _ in cases], [tr for _, tr, _ in cases],
                                           [P for _, _, P in cases], FS, MAXLEN, label_dtype=label_dtype)


@pytest.mark.parametrize("N", [3, 8, 9, 16, 17, 32, 33, 64, 65, 100, 128])
def test_all_device_entry_every_lane_layout(N):
    """One video per call through mucon_viterbi_decode_batch: the launch code knows neither T nor K (max_K = 0, no completion
    flag), so every N takes the two-launch schedule with viterbi_dp_lanes_kernel<G, JG, NW> and back-pointers in HBM."""
    for i, (T, kind) in enumerate(((45 * N + 300, "informative"), (120 * N + 17, "noise"), (40 * N + 90, "ties"))):
        case = _case(4000 + 10 * N + i, T, N, kind)
        want = _want(*case)
        for dt in (torch.uint8, torch.int32, None):
            _check_device(_device([case], dt), [case], [want], dt)


def _ragged(seed, nv, n_max, t_hi):
    cases = []
    for i in range(nv):
        s = seed + 10 * i
        T = 90 + int(synth.integers(s, 1, 0, t_hi)[0])
        N = n_max if i == 0 else 1 + int(synth.integers(s + 1, 1, 0, n_max)[0])
        N = max(min(N, T // FS), -(-(T // FS) // J))          # 1 <= N <= K, and K <= J N so that hypotheses survive
        cases.append(_case(s + 2, T, N, ("informative", "noise", "ties")[i % 3]))
    return cases


@pytest.mark.parametrize("n_max,t_hi", [(12, 3000), (64, 9000), (100, 12000)])
def test_ragged_batches_both_entries(n_max, t_hi):
    """12 ragged videos per call (the launch's lane layout is the LARGEST transcript's: <4,17,1>, <4,17,4>, <4,17,8> with shorter
    transcripts riding along) through the all-device entry and through the host entry, every label format."""
    cases = _ragged(5000 + n_max, 12, n_max, t_hi)
    wants = [_want(*c) for c in cases]
    for dt in (torch.uint8, torch.int32, None):
        _check_device(_device(cases, dt), cases, wants, dt)
    lps = [torch.from_numpy(lp).cuda() for lp, _, _ in cases]
    for fmt in ("lazy", "uint8", "int32"):
        res = ops.viterbi_decode_batch(lps, [tr for _, tr, _ in cases], [P for _, _, P in cases], FS, MAXLEN, labels=fmt)
        for r, (lp, _, _), w in zip(res, cases, wants):
            _check_host(r, w, lp.shape[0])
            raw = r.labels_raw
            assert (raw is not None) and raw.dtype == np.int32     # after .labels: the widened / expanded array is cached


def test_config5_batch_of_twelve():
    """BASELINE config 5 (T = 16,384, N = 64: viterbi_dp_lanes_kernel<4,17,4>, K = 546 columns) x 12 videos per call -- the
    configuration behind BENCH.viterbi.ms_per_video_batch64 / batch256 -- both entries, every label format."""
    T, N = 16384, 64
    cases = [_case(6000 + 10 * i, T, N, "informative" if i % 2 == 0 else "noise") for i in range(12)]
    wants = [_want(*c) for c in cases]
    for dt in (torch.uint8, torch.int32, None):
        _check_device(_device(cases, dt), cases, wants, dt)
    lps = [torch.from_numpy(lp).cuda() for lp, _, _ in cases]
    for fmt in ("lazy", "uint8", "int32"):
        res = ops.viterbi_decode_batch(lps, [tr for _, tr, _ in cases], [P for _, _, P in cases], FS, MAXLEN, labels=fmt)
        for r, w in zip(res, wants):
            _check_host(r, w, T)


def test_label_formats_of_the_latency_paths():
    """Calls of 1..8 videos (one-launch kernel / pair kernel): the uint8 and the none format against the int32 one and the oracle,
    with video lengths that leave leftover frames (T mod fs != 0: they carry the last label, at the START of the video) and
    label runs whose ends are not 16-byte aligned in the uint8 array."""
    for nv, n_max, t_hi in ((1, 6, 2500), (1, 40, 9000), (3, 16, 5000), (8, 90, 7000)):
        cases = _ragged(7000 + nv + n_max, nv, n_max, t_hi)
        wants = [_want(*c) for c in cases]
        lps = [torch.from_numpy(lp).cuda() for lp, _, _ in cases]
        for fmt in ("int32", "uint8", "lazy"):
            res = ops.viterbi_decode_batch(lps, [tr for _, tr, _ in cases], [P for _, _, P in cases], FS, MAXLEN, labels=fmt)
            for r, (lp, _, _), w in zip(res, cases, wants):
                if fmt == "uint8":
                    assert r.labels_raw.dtype == np.uint8
                if fmt == "lazy":
                    assert r.labels_raw is None
                _check_host(r, w, lp.shape[0])


def test_error_statuses_leave_labels_alone():
    """T < fs and K > J N through the all-device entry: status / n_seg / score only, in every label format; argument errors raise."""
    tr = synth.transcript(1, 1, C).astype(np.int32)
    P = np.zeros((J, 1))
    short = torch.from_numpy(synth.emissions(2, 20, C)).cuda()
    long_ = torch.from_numpy(synth.emissions(3, 2100, C)).cuda()      # K = 70 > J * 1
    for dt in (torch.uint8, torch.int32, None):
        res = ops.viterbi_decode_batch_device([short, long_], [tr, tr], [P, P], FS, MAXLEN, label_dtype=dt)
        torch.cuda.synchronize()
        assert res.status.cpu().tolist() == [_lib.VIT_INDEX_ERROR, _lib.VIT_NO_HYPOTHESIS]
        assert res.n_seg.cpu().tolist() == [0, 0]
    lib = _lib.load()
    assert lib.mucon_viterbi_decode_batch(1, None, C, FS, MAXLEN, 1, None, None, None, 7, None, None, None, None, None, None) == _lib.E_ARG
    assert lib.mucon_viterbi_decode_batch(1, None, C, FS, MAXLEN, 1, None, None, None, _lib.VIT_LABELS_U8, None, None, None, None, None,
                                          None) == _lib.E_ARG      # labels NULL with a format that writes them


def test_plain_framescore_kernel_with_odd_class_count():
    """C not a multiple of 4 (rows of an aligned base are then not 16-byte aligned): viterbi_framescore_kernel, the plain
    version of phase 1, in front of every lane layout's DP -- both entries."""
    Cc = 10
    cases = []
    for i, (T, N) in enumerate(((700, 3), (2500, 12), (5000, 20), (9000, 40), (6100, 70), (3000, 5), (4000, 9), (1200, 2), (2000, 17))):
        tr = synth.transcript(8000 + i, N, Cc).astype(np.int32)
        lp = synth.emissions(8100 + i, T, Cc, labels=synth.segment_labels(8200 + i, T, tr))
        mu = np.ones(Cc)
        mu[np.unique(tr)] = T / N
        cases.append((lp, tr, oracle.length_rows(oracle.poisson_table(mu, MAXLEN), tr, FS, MAXLEN)))
    wants = [_want(*c) for c in cases]
    _check_device(_device(cases, torch.uint8), cases, wants, torch.uint8)
    for c, w in zip(cases, wants):                         # one video per call too (the layout of ITS transcript)
        _check_device(_device([c], torch.int32), [c], [w], torch.int32)
    lps = [torch.from_numpy(lp).cuda() for lp, _, _ in cases]
    res = ops.viterbi_decode_batch(lps, [tr for _, tr, _ in cases], [P for _, _, P in cases], FS, MAXLEN, labels="uint8")
    for r, (lp, _, _), w in zip(res, cases, wants):
        _check_host(r, w, lp.shape[0])
