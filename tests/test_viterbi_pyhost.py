"""CPU: the host side of ops.viterbi_decode_batch without a GPU -- csrc/pyhost.c (the C loop over the Python lists, one crossing
of the ctypes boundary) and the lazy result sequence -- against a stand-in for mucon_viterbi_decode_host that records what it was
handed and writes known results.  What the stand-in sees must be exactly the caller's arrays (pointed at, not copied), in the
C ABI's record layout; inputs that are not int32 / float64 C-contiguous buffers are converted; the flat result layout, the
offsets and the host-side label expansion are checked against the reference's traceback rule
(src/core/viterbi/viterbi.py:140-158: leftover frames carry the LAST label and sit at the START)."""
import ctypes

import numpy as np
import pytest
import torch

from mucon_amd import _lib, ops

FS, MAXLEN, C = 30, 2000, 48
J = MAXLEN // FS


class FakeTensor:
    """Stands in for a device tensor: shape, pointer and layout are all the binding reads."""
    is_cuda, dtype = True, torch.float32

    def __init__(self, T, ptr):
        self.shape, self._ptr = (T, C), ptr

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self._ptr


PROTO = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int32, ctypes.POINTER(_lib.ViterbiVideo), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)
PROTO_POISSON = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int32, ctypes.POINTER(_lib.ViterbiVideo), ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)


@pytest.fixture
def fake_decode(monkeypatch):
    """Replaces the library's mucon_viterbi_decode_host by a callback: every video 'decodes' to segments of equal length
    (the last one takes the remainder), score = -(v + 1), labels written in the asked format."""
    seen = {}

    def decode_poisson(nv, vids, log_fact, Cc, fs, max_len, score, n_seg, status, labels, fmt, seg, stream):
        seen["log_fact"] = np.ctypeslib.as_array((ctypes.c_double * (max_len // fs)).from_address(log_fact)).copy()
        seen["log_fact_ptr"] = log_fact
        return decode(nv, vids, Cc, fs, max_len, score, n_seg, status, labels, fmt, seg, stream, rows=3)

    def decode(nv, vids, Cc, fs, max_len, score, n_seg, status, labels, fmt, seg, stream, rows=J):
        seen.update(nv=nv, C=Cc, fs=fs, max_len=max_len, fmt=fmt, records=[], labels_null=not labels)
        sc = (ctypes.c_double * nv).from_address(score)
        ns = (ctypes.c_int32 * nv).from_address(n_seg)
        st = (ctypes.c_int32 * nv).from_address(status)
        lab_off = seg_off = 0
        for v in range(nv):
            q = vids[v]
            tr = np.ctypeslib.as_array((ctypes.c_int32 * q.N).from_address(q.transcript)).copy()
            tab = np.ctypeslib.as_array((ctypes.c_double * (rows * q.N)).from_address(q.table)).reshape(rows, q.N).copy()
            seen["records"].append(dict(lp=q.lp, T=q.T, N=q.N, tr=tr, tab=tab, tr_ptr=q.transcript, tab_ptr=q.table, force=(q.force_n, q.force_j)))
            sg = (ctypes.c_int32 * q.N).from_address(seg + 4 * seg_off)
            if q.T < fs:
                st[v], ns[v], sc[v] = _lib.VIT_INDEX_ERROR, 0, -np.inf
            else:
                K = q.T // fs
                per = max(K // q.N, 1)
                lens = [per * fs] * q.N
                lens[-1] = (K - per * (q.N - 1)) * fs + (q.T - K * fs)          # the last segment: remaining columns + leftover frames
                for i, l in enumerate(lens):
                    sg[i] = l
                st[v], ns[v], sc[v] = _lib.VIT_OK, q.N, -(v + 1.0)
                if fmt != _lib.VIT_LABELS_NONE:
                    want = ops.expand_labels(tr, np.asarray(lens, np.int32), q.N, q.T, fs)
                    ct = ctypes.c_int32 if fmt == _lib.VIT_LABELS_I32 else ctypes.c_uint8
                    arr = np.ctypeslib.as_array((ct * q.T).from_address(labels + ctypes.sizeof(ct) * lab_off))
                    arr[:] = want
            lab_off += max(q.T, 1)
            seg_off += q.N
        return 0

    cb = PROTO(decode)
    cb_p = PROTO_POISSON(decode_poisson)

    class Lib:
        mucon_viterbi_decode_host = cb
        mucon_viterbi_decode_host_poisson = cb_p

    monkeypatch.setattr(_lib, "load", lambda *a, **k: Lib)
    monkeypatch.setattr(ops, "_VIT_FN_ADDR", [0, 0])

    monkeypatch.setattr(_lib, "current_stream_raw", lambda: 0x1234)
    return seen


def _inputs(nv, seed=0):
    rng = np.random.default_rng(seed)
    lps, trs, tabs = [], [], []
    for v in range(nv):
        N = int(rng.integers(1, 9))
        T = int(rng.integers(N * FS, N * FS * 20)) + int(rng.integers(0, FS))
        lps.append(FakeTensor(T, 0x7F0000000000 + 0x100000 * v))
        trs.append(rng.integers(0, C, N).astype(np.int32))
        tabs.append(rng.standard_normal((J, N)))
    return lps, trs, tabs


@pytest.mark.parametrize("nv", [1, 3, 8, 40])
def test_records_point_at_the_callers_arrays(fake_decode, nv):
    lps, trs, tabs = _inputs(nv)
    forces = [None if v % 3 else (v % 2, 5) for v in range(nv)]
    res = ops.viterbi_decode_batch(lps, trs, tabs, FS, MAXLEN, forces, labels="lazy")
    assert fake_decode["nv"] == nv and fake_decode["fmt"] == _lib.VIT_LABELS_NONE and fake_decode["labels_null"]
    assert (fake_decode["C"], fake_decode["fs"], fake_decode["max_len"]) == (C, FS, MAXLEN)
    for v, r in enumerate(fake_decode["records"]):
        assert r["lp"] == lps[v].data_ptr() and r["T"] == lps[v].shape[0] and r["N"] == len(trs[v])
        assert r["tr_ptr"] == trs[v].ctypes.data and r["tab_ptr"] == tabs[v].ctypes.data          # pointed at, not copied
        assert r["force"] == ((-1, -1) if forces[v] is None else forces[v])
    assert len(res) == nv
    for v, r in enumerate(res):
        assert r.status == _lib.VIT_OK and r.n_seg == len(trs[v]) and r.score == -(v + 1.0)
        assert r.labels_raw is None                                     # nothing was written: expanded on access
        lab = r.labels
        T = lps[v].shape[0]
        assert lab.dtype == np.int32 and lab.shape == (T,)
        missing = T - (T // FS) * FS
        assert (lab[:missing] == trs[v][-1]).all()                      # leftover frames: last label, at the start
        assert int(r.seg_len.sum()) == T
        # run-length decode of the expansion gives the segments back (adjacent equal labels merge)
        body = lab[missing:]
        want = np.repeat(trs[v], np.asarray(r.seg_len) - np.eye(1, r.n_seg, r.n_seg - 1, dtype=np.int64)[0] * missing)
        np.testing.assert_array_equal(body, want)


@pytest.mark.parametrize("fmt", ["int32", "uint8"])
def test_label_formats_small_calls_come_back_in_the_result_buffer(fake_decode, fmt):
    lps, trs, tabs = _inputs(5, seed=1)
    res = ops.viterbi_decode_batch(lps, trs, tabs, FS, MAXLEN, labels=fmt)
    assert fake_decode["fmt"] == (_lib.VIT_LABELS_I32 if fmt == "int32" else _lib.VIT_LABELS_U8)
    for v, r in enumerate(res):
        raw = r.labels_raw
        assert raw.dtype == (np.int32 if fmt == "int32" else np.uint8) and raw.shape == (lps[v].shape[0],)
        np.testing.assert_array_equal(r.labels, ops.expand_labels(trs[v], r.seg_len, r.n_seg, lps[v].shape[0], FS))


def test_other_input_kinds_are_converted(fake_decode):
    lps, trs, tabs = _inputs(6, seed=2)
    trs[1] = trs[1].tolist()                                   # a Python list
    trs[3] = trs[3].astype(np.int64)                           # another integer width
    tabs[2] = np.asfortranarray(tabs[2])                       # not C-contiguous
    tabs[4] = tabs[4].astype(np.float32)                       # another float width
    tabs[5] = np.concatenate([tabs[5], tabs[5]], axis=1)[:, ::2]   # a strided view
    want_tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in tabs]
    res = ops.viterbi_decode_batch(lps, trs, tabs, FS, MAXLEN)
    for v, r in enumerate(fake_decode["records"]):
        np.testing.assert_array_equal(r["tr"], np.asarray(trs[v], dtype=np.int32))
        np.testing.assert_array_equal(r["tab"], want_tabs[v])
    assert [r.n_seg for r in res] == [len(t) for t in trs]


def test_shape_errors_and_error_statuses(fake_decode):
    lps, trs, tabs = _inputs(3, seed=3)
    with pytest.raises(ValueError, match="length table"):
        ops.viterbi_decode_batch(lps, trs, [tabs[0], tabs[1][:, :0], tabs[2]], FS, MAXLEN)
    with pytest.raises(ValueError, match="length table"):
        ops.viterbi_decode_batch(lps, trs, [tabs[0], tabs[1].astype(np.float32)[:, :0], tabs[2]], FS, MAXLEN)
    lps[1] = FakeTensor(7, 0x7F1000000000)                      # fewer frames than one decoding step
    res = ops.viterbi_decode_batch(lps, trs, tabs, FS, MAXLEN)
    assert res[1].status == _lib.VIT_INDEX_ERROR and res[1].labels.shape == (0,) and res[1].n_seg == 0
    assert res[0].status == res[2].status == _lib.VIT_OK and len(res[2].labels) == lps[2].shape[0]
    assert [r.status for r in res[0:3]] == [r.status for r in res]          # slices and iteration agree
    with pytest.raises(_lib.MuconHipError):
        ops.viterbi_decode_batch([torch.zeros(40, C)], [trs[0]], [tabs[0]], FS, MAXLEN)      # a host tensor: no CPU fallback


def test_expand_labels_rule():
    tr = np.array([7, 3, 9], np.int32)
    lab = ops.expand_labels(tr, np.array([60, 30, 100], np.int32), 3, 190, 30)       # K = 6 columns, 10 leftover frames
    assert lab.tolist() == [9] * 10 + [7] * 60 + [3] * 30 + [9] * 90
    assert ops.expand_labels(tr, np.array([60, 30], np.int32), 2, 90, 30).tolist() == [7] * 60 + [3] * 30
    assert ops.expand_labels(tr, np.array([], np.int32), 0, 10, 30).shape == (0,)


def test_other_dtypes_and_shapes_are_converted_in_one_pass_and_one_crossing(fake_decode, monkeypatch):
    """int64 / list / (N, 1)-shaped transcripts and float32 / Fortran-ordered tables are normalised before the first C call (one crossing
    for the whole list -- the C loop reports one offending video per call); a wrong table shape raises instead of looping."""
    lps, trs, tabs = _inputs(6, seed=3)
    want_tr = [t.copy() for t in trs]
    want_tab = [t.copy() for t in tabs]
    trs[0] = trs[0].astype(np.int64)
    trs[1] = trs[1].reshape(-1, 1)
    trs[2] = [int(x) for x in trs[2]]
    tabs[3] = np.asfortranarray(tabs[3])
    tabs[4] = tabs[4].astype(np.float32)
    want_tab[4] = tabs[4].astype(np.float64)
    calls = []
    real = _lib.pyhost().mucon_py_viterbi_decode
    monkeypatch.setattr(ops, "_vit_entry", lambda lib: (ops._VIT_FN_ADDR.__setitem__(0, ctypes.cast(lib.mucon_viterbi_decode_host, ctypes.c_void_p).value),
                                                        ops._VIT_FN_ADDR.__setitem__(1, ctypes.cast(lib.mucon_viterbi_decode_host_poisson, ctypes.c_void_p).value),
                                                        lambda *a: (calls.append(1), real(*a))[1])[2])
    res = ops.viterbi_decode_batch(lps, trs, tabs, FS, MAXLEN, labels="lazy")
    assert len(calls) == 1 and len(res) == 6
    for v, r in enumerate(fake_decode["records"]):
        np.testing.assert_array_equal(r["tr"], want_tr[v])
        np.testing.assert_array_equal(r["tab"], want_tab[v])
    tabs[5] = tabs[5][:, :-1] if tabs[5].shape[1] > 1 else np.zeros((J, 3))
    with pytest.raises(ValueError, match="length table"):
        ops.viterbi_decode_batch(lps, trs, tabs, FS, MAXLEN, labels="lazy")


@pytest.mark.parametrize("nv", [1, 9])
def test_poisson_parameter_form_hands_over_parameter_blocks_and_the_shared_row(fake_decode, nv):
    """(ABI 7) log_fact given: the records point at the callers' [3, N] parameter blocks (not copied), the entry called is
    mucon_viterbi_decode_host_poisson and it receives the shared log-factorial row; a [J, N] table is refused then."""
    from mucon_amd.core.viterbi import poisson_params_for_many
    lps, trs, _ = _inputs(nv, seed=4)
    rng = np.random.default_rng(9)
    pps = poisson_params_for_many([rng.uniform(1.0, 500.0, C) for _ in range(nv)], trs, FS, MAXLEN)
    lf = pps[0].log_fact
    res = ops.viterbi_decode_batch(lps, trs, [p.params for p in pps], FS, MAXLEN, labels="lazy", log_fact=lf)
    assert len(res) == nv and fake_decode["log_fact_ptr"] == lf.ctypes.data
    np.testing.assert_array_equal(fake_decode["log_fact"], lf)
    for v, r in enumerate(fake_decode["records"]):
        assert r["tab_ptr"] == pps[v].params.ctypes.data and r["tab"].shape == (3, len(trs[v]))
        np.testing.assert_array_equal(r["tab"], pps[v].params)
    with pytest.raises(ValueError, match="length table"):
        ops.viterbi_decode_batch(lps, trs, [p.rows_for(t, FS) for p, t in zip(pps, trs)], FS, MAXLEN, labels="lazy", log_fact=lf)
