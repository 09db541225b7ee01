/* libmucon_pyhost.so -- host-side helper of the Python binding (mucon_amd/ops.py), plain C against the CPython API.
 *
 * ops.viterbi_decode_batch takes what the reference's call site has (src/mucon/evaluators.py:147-180, one video at a time):
 * per video a device tensor of emissions, a NumPy transcript and a NumPy length table.  Turning 256 of those into the C ABI's
 * mucon_viterbi_video records cost ~2.5 us of interpreter time EACH (ndarray.__array_interface__ alone builds a dict per
 * array: ~1 us) -- 0.65 ms per 256-video call whose GPU work is 0.08 ms (T = 2,000) to 0.63 ms (T = 16,384): r4 profile,
 * tools/vit_host_breakdown.py.  Here the per-video part is a C loop over the Python lists through the buffer protocol (~50 ns
 * per array).  Loaded with ctypes.PyDLL (the GIL stays held; exceptions propagate).  No device code, no HIP calls.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#include "../../include/mucon_hip.h"

static int is_format(const char *fmt, char want) {
    /* struct-module format of a one-element type, optionally with a byte-order / alignment prefix */
    if (!fmt) return 0;
    if (*fmt == '@' || *fmt == '=' || *fmt == '<') ++fmt;
    return fmt[0] == want && fmt[1] == '\0';
}

typedef int (*decode_host_fn)(int32_t, const mucon_viterbi_video *, int32_t, int32_t, int32_t, double *, int32_t *, int32_t *, void *, int32_t,
                              int32_t *, void *);
typedef int (*decode_host_poisson_fn)(int32_t, const mucon_viterbi_video *, const double *, int32_t, int32_t, int32_t, double *, int32_t *, int32_t *,
                                      void *, int32_t, int32_t *, void *);

static mucon_viterbi_video *g_rec = NULL;   /* record scratch, grown on demand (one Python thread calls at a time: the GIL) */
static Py_ssize_t g_rec_cap = 0;

static size_t up8(size_t n) { return (n + 7) & ~(size_t)7; }

/* ops.viterbi_decode_batch's whole call in one crossing of the ctypes boundary (each costs ~3 us, a NumPy `.ctypes.data` ~1 us):
 * builds the mucon_viterbi_video records from
 *   lp_ptrs      list[int]   device pointers of the emission tensors (contiguous float32 [T, C])
 *   Ts           list[int]   their frame counts
 *   transcripts  list        C-contiguous int32 buffers [N]      (NumPy arrays; anything else: see `bad` below)
 *   tables       list        C-contiguous float64 buffers [J, N] -- or, with log_fact, the [3, N] PoissonModel parameter blocks (include/mucon_hip.h, ABI 7)
 *   log_fact     None, or a C-contiguous float64 buffer [J]: the length scores are built on the device, `decode_fn` is mucon_viterbi_decode_host_poisson
 *   forces       None, or list of None | (n, j)
 * (the arrays' memory is only pointed at: the caller keeps the lists alive), calls mucon_viterbi_decode_host (its address in
 * `decode_fn`: this library does not link against libmucon_hip.so), and returns
 *   (rc, bad, sum_T, sum_N, out)
 * out = a bytearray [score f64 nv][n_seg i32 nv][status i32 nv][seg_len i32 sum_N, padded to 8 bytes][label offsets i64 nv + 1]
 * [segment offsets i64 nv + 1][labels, when label_format != NONE and labels_addr == 0]; labels_addr != 0: the caller's
 * own (pinned) label array is written instead.
 * bad >= 0 (rc = 0, out = None, nothing decoded): video `bad` has a transcript / table that is not such a buffer, or an emission
 * pointer that is not 16-byte aligned while need_align is set -- the caller converts / copies that one and calls again with
 * start = bad (records below `start` are kept). */
PyObject *mucon_py_viterbi_decode(PyObject *lp_ptrs, PyObject *Ts, PyObject *transcripts, PyObject *tables, PyObject *forces, PyObject *log_fact, long C,
                                  long fs, long max_len, long label_format, long need_align, long start,
                                  unsigned long long labels_addr, unsigned long long decode_fn, unsigned long long stream) {
    if (!PyList_Check(lp_ptrs) || !PyList_Check(Ts) || !PyList_Check(transcripts) || !PyList_Check(tables)) {
        PyErr_SetString(PyExc_TypeError, "mucon_py_viterbi_decode: lists expected");
        return NULL;
    }
    const Py_ssize_t nv = PyList_GET_SIZE(lp_ptrs);
    if (PyList_GET_SIZE(Ts) != nv || PyList_GET_SIZE(transcripts) != nv || PyList_GET_SIZE(tables) != nv ||
        (forces != Py_None && (!PyList_Check(forces) || PyList_GET_SIZE(forces) != nv))) {
        PyErr_SetString(PyExc_ValueError, "mucon_py_viterbi_decode: lists of different lengths");
        return NULL;
    }
    if (fs <= 0 || nv <= 0) {
        PyErr_SetString(PyExc_ValueError, "mucon_py_viterbi_decode: no videos / bad frame sampling");
        return NULL;
    }
    if (nv > g_rec_cap) {
        if (start > 0) {
            PyErr_SetString(PyExc_RuntimeError, "mucon_py_viterbi_decode: record scratch lost between retries");
            return NULL;
        }
        free(g_rec);
        g_rec_cap = 2 * nv > 64 ? 2 * nv : 64;
        g_rec = (mucon_viterbi_video *)malloc(sizeof(mucon_viterbi_video) * (size_t)g_rec_cap);
        if (!g_rec) {
            g_rec_cap = 0;
            return PyErr_NoMemory();
        }
    }
    const long J = max_len / fs;
    const double *lf_ptr = NULL;
    Py_buffer blf;
    int have_lf = 0;
    if (log_fact != Py_None) {
        if (PyObject_GetBuffer(log_fact, &blf, PyBUF_FORMAT | PyBUF_ND | PyBUF_C_CONTIGUOUS) != 0) return NULL;
        if (!(blf.ndim == 1 && blf.itemsize == 8 && is_format(blf.format, 'd') && blf.shape[0] == J)) {
            PyBuffer_Release(&blf);
            PyErr_Format(PyExc_ValueError, "mucon_py_viterbi_decode: log_fact must be a C-contiguous float64 array [%ld]", J);
            return NULL;
        }
        lf_ptr = (const double *)blf.buf;     /* (the caller keeps the array alive across the call) */
        PyBuffer_Release(&blf);
        have_lf = 1;
    }
    const long tab_rows = have_lf ? 3 : J;
    mucon_viterbi_video *rec = g_rec;
    long long sum_T = 0, sum_N = 0;
    for (Py_ssize_t v = 0; v < nv; ++v) {
        mucon_viterbi_video *q = rec + v;
        if (v >= start) {
            const unsigned long long lp = PyLong_AsUnsignedLongLong(PyList_GET_ITEM(lp_ptrs, v));
            const long T = PyLong_AsLong(PyList_GET_ITEM(Ts, v));
            if (PyErr_Occurred()) return NULL;
            Py_buffer bt, bp;
            if (PyObject_GetBuffer(PyList_GET_ITEM(transcripts, v), &bt, PyBUF_FORMAT | PyBUF_ND | PyBUF_C_CONTIGUOUS) != 0) {
                PyErr_Clear();
                return Py_BuildValue("inLLO", 0, v, 0LL, 0LL, Py_None);
            }
            int ok = bt.ndim == 1 && bt.itemsize == 4 && is_format(bt.format, 'i');
            const long N = ok ? (long)bt.shape[0] : 0;
            const void *tr_ptr = bt.buf;
            PyBuffer_Release(&bt);
            if (ok) {
                if (PyObject_GetBuffer(PyList_GET_ITEM(tables, v), &bp, PyBUF_FORMAT | PyBUF_ND | PyBUF_C_CONTIGUOUS) != 0) {
                    PyErr_Clear();
                    return Py_BuildValue("inLLO", 0, v, 0LL, 0LL, Py_None);
                }
                ok = bp.ndim == 2 && bp.itemsize == 8 && is_format(bp.format, 'd');
                if (ok && (bp.shape[0] != tab_rows || bp.shape[1] != N)) {
                    PyErr_Format(PyExc_ValueError, "video %zd: length table [%zd, %zd], expected [%ld, %ld]", v, bp.shape[0], bp.shape[1], tab_rows, N);
                    PyBuffer_Release(&bp);
                    return NULL;
                }
                q->table = (const double *)bp.buf;
                PyBuffer_Release(&bp);
            }
            if (!ok || (need_align && (lp & 15))) return Py_BuildValue("inLLO", 0, v, 0LL, 0LL, Py_None);
            q->lp = (const float *)(uintptr_t)lp;
            q->transcript = (const int32_t *)tr_ptr;
            q->T = (int32_t)T;
            q->N = (int32_t)N;
            q->force_n = q->force_j = -1;
            if (forces != Py_None) {
                PyObject *f = PyList_GET_ITEM(forces, v);
                if (f != Py_None) {
                    if (!PyTuple_Check(f) || PyTuple_GET_SIZE(f) != 2) {
                        PyErr_SetString(PyExc_TypeError, "mucon_py_viterbi_decode: forces[v] must be None or (n, j)");
                        return NULL;
                    }
                    q->force_n = (int32_t)PyLong_AsLong(PyTuple_GET_ITEM(f, 0));
                    q->force_j = (int32_t)PyLong_AsLong(PyTuple_GET_ITEM(f, 1));
                    if (PyErr_Occurred()) return NULL;
                }
            }
        }
        sum_T += q->T > 0 ? q->T : 1;
        sum_N += q->N;
    }
    const size_t lab_elem = label_format == MUCON_VIT_LABELS_I32 ? 4 : (label_format == MUCON_VIT_LABELS_U8 ? 1 : 0);
    const size_t o_nseg = 8 * (size_t)nv, o_stat = o_nseg + 4 * (size_t)nv, o_seg = o_stat + 4 * (size_t)nv;
    const size_t o_laboff = up8(o_seg + 4 * (size_t)sum_N), o_segoff = o_laboff + 8 * (size_t)(nv + 1);
    const size_t o_lab = o_segoff + 8 * (size_t)(nv + 1);
    const size_t total = o_lab + ((lab_elem && !labels_addr) ? lab_elem * (size_t)sum_T : 0);
    PyObject *out = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)total);
    if (!out) return NULL;
    char *b = PyByteArray_AS_STRING(out);
    int64_t *lo = (int64_t *)(b + o_laboff), *so = (int64_t *)(b + o_segoff);
    int64_t a = 0, c = 0;
    for (Py_ssize_t v = 0; v < nv; ++v) {
        lo[v] = a;
        so[v] = c;
        a += rec[v].T > 0 ? rec[v].T : 1;
        c += rec[v].N;
    }
    lo[nv] = a;
    so[nv] = c;
    void *labels = lab_elem ? (labels_addr ? (void *)(uintptr_t)labels_addr : (void *)(b + o_lab)) : NULL;
    /* (the GIL stays held: the library keeps per-device staging state and this file a record scratch, both written for ONE host thread
     * per process -- include/mucon_hip.h; a second Python thread entering here during a 0.05 - 0.9 ms decode would share them) */
    const int rc = have_lf ? ((decode_host_poisson_fn)(uintptr_t)decode_fn)((int32_t)nv, rec, lf_ptr, (int32_t)C, (int32_t)fs, (int32_t)max_len, (double *)b,
                                                                            (int32_t *)(b + o_nseg), (int32_t *)(b + o_stat), labels, (int32_t)label_format,
                                                                            (int32_t *)(b + o_seg), (void *)(uintptr_t)stream)
                           : ((decode_host_fn)(uintptr_t)decode_fn)((int32_t)nv, rec, (int32_t)C, (int32_t)fs, (int32_t)max_len, (double *)b,
                                                                    (int32_t *)(b + o_nseg), (int32_t *)(b + o_stat), labels, (int32_t)label_format,
                                                                    (int32_t *)(b + o_seg), (void *)(uintptr_t)stream);
    PyObject *res = Py_BuildValue("inLLO", rc, (Py_ssize_t)-1, sum_T, sum_N, out);
    Py_DECREF(out);
    return res;
}
