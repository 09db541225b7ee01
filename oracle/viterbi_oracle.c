/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (mucon_amd/), only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * CPU restatement, in plain C, of the reference's transcript-constrained Viterbi decode:
 *   reference src/core/viterbi/viterbi.py:49-158 (Viterbi.decode and helpers),
 *   with SingleTranscriptGrammar (src/core/viterbi/grammar.py:196-217) as the grammar and an
 *   f64 length table (PoissonModel, src/core/viterbi/length_model.py:42-83) as the length model,
 *   as driven by src/mucon/evaluators.py:147-180.
 *
 * The restatement is LITERAL: it keeps the reference's insertion-ordered hypothesis dictionary
 * (an ordered key list + position map), walks it in the same order, and applies the same
 * `<=` / `>=` update rules (viterbi.py:26-28, 135), so ties and the "no final state reached"
 * case resolve exactly as in the reference.  Floating point follows the reference as it runs
 * under NumPy 2.x: a sequential float32 cumsum (viterbi.py:51), float32 score chain for the
 * first transcript state, float64 everywhere after the first length-model add.
 * Build WITHOUT -ffast-math and with -ffp-contract=off (see oracle/Makefile).
 *
 * Parity pin: tests/golden/viterbi_*.npz were produced by tools/make_golden.py importing the
 * reference itself in the build container; tests/test_oracle_viterbi.py checks this file
 * against every one of them (score bits, labels, segments).  The beam (max_hypotheses, prune():
 * viterbi.py:74-79) is restated too and pinned by tests/golden/viterbi_pruned.* (74 decodes of the
 * reference under finite beams, tools/make_golden_pruned.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ST_OK 0
#define ST_INDEX_ERROR 1     /* T < frame_sampling: reference raises IndexError (viterbi.py:87) */
#define ST_NO_HYPOTHESIS 2   /* final traceback is None: reference raises AttributeError (viterbi.py:147) */
#define ST_BAD_ARG 3

typedef struct {
    int n;        /* transcript position of the hypothesis' current label              */
    int j;        /* length index: segment length so far = (j+1)*fs frames             */
    double score; /* float32-valued while n == 0 (kept exactly), float64 afterwards    */
} hyp_t;

typedef struct {
    hyp_t *items; /* insertion-ordered, like the reference's HypDict (a Python dict)   */
    int count;
    int *pos;     /* pos[n*J + j] = index into items, or -1                            */
} hypdict_t;

static void dict_clear(hypdict_t *d, int N, int J) {
    d->count = 0;
    for (int i = 0; i < N * J; ++i) d->pos[i] = -1;
}

/* HypDict.update (viterbi.py:26-28): insert, or overwrite in place iff old.score <= score.
 * Returns 1 when the stored hypothesis is now the given one. */
static int dict_update(hypdict_t *d, int J, int n, int j, double score) {
    int p = d->pos[n * J + j];
    if (p < 0) {
        p = d->count++;
        d->pos[n * J + j] = p;
        d->items[p].n = n;
        d->items[p].j = j;
        d->items[p].score = score;
        return 1;
    }
    if (d->items[p].score <= score) {
        d->items[p].score = score;
        return 1;
    }
    return 0;
}

/* ---- prune (viterbi.py:74-79): `sorted([(hyps[key].score, key) for key in hyps])`, the first len - max_hypotheses entries deleted.
 * Python sorts the (score, key) tuples: by score, ties by the KEY TUPLE key = (-1, a_0, .., a_n, length) compared element by element, a tuple
 * that is a prefix of the other being the smaller one.  Restated literally (no closed form), so that it can check the device kernel's. */
typedef struct {
    const int32_t *tr; /* transcript */
    int fs;
} keyctx_t;
static keyctx_t g_keyctx;   /* (qsort has no context argument; the oracle is single-threaded test infrastructure) */
static int key_elem(const hyp_t *h, int i, int *len_out) {   /* element i of the key tuple of h, its length in *len_out */
    *len_out = h->n + 3;                  /* -1, a_0 .. a_n, length */
    if (i == 0) return -1;
    if (i <= h->n + 1) return g_keyctx.tr[i - 1];
    return (h->j + 1) * g_keyctx.fs;
}
static int cmp_score_key(const void *pa, const void *pb) {
    const hyp_t *a = (const hyp_t *)pa, *b = (const hyp_t *)pb;
    if (a->score < b->score) return -1;
    if (a->score > b->score) return 1;
    int la, lb;
    key_elem(a, 0, &la);
    key_elem(b, 0, &lb);
    const int l = la < lb ? la : lb;
    for (int i = 0; i < l; ++i) {
        const int ea = key_elem(a, i, &la), eb = key_elem(b, i, &lb);
        if (ea != eb) return ea < eb ? -1 : 1;
    }
    return la < lb ? -1 : (la > lb ? 1 : 0);
}
/* Returns 0, or 1 when a score is NaN (Python's sort of tuples with NaN is not an order: outside what the oracle restates). */
static int dict_prune(hypdict_t *d, int J, long max_hyp) {
    if (max_hyp < 0 || d->count <= max_hyp) return 0;       /* `if len(hyps) > self.max_hypotheses` */
    const int ndel = max_hyp == 0 ? 0 : d->count - (int)max_hyp;   /* tmp[0:-max_hypotheses]: -0 is 0, the slice is empty */
    if (ndel == 0) return 0;
    for (int i = 0; i < d->count; ++i)
        if (d->items[i].score != d->items[i].score) return 1;
    hyp_t *tmp = (hyp_t *)malloc(sizeof(hyp_t) * (size_t)d->count);
    memcpy(tmp, d->items, sizeof(hyp_t) * (size_t)d->count);
    qsort(tmp, (size_t)d->count, sizeof(hyp_t), cmp_score_key);
    for (int i = 0; i < ndel; ++i) d->pos[tmp[i].n * J + tmp[i].j] = -2;   /* marked */
    free(tmp);
    int w = 0;                                                /* `del hyps[key]`: the others keep their order */
    for (int i = 0; i < d->count; ++i) {
        const hyp_t h = d->items[i];
        if (d->pos[h.n * J + h.j] == -2) {
            d->pos[h.n * J + h.j] = -1;
            continue;
        }
        d->items[w] = h;
        d->pos[h.n * J + h.j] = w;
        ++w;
    }
    d->count = w;
    return 0;
}

/* lp: [T x C] row-major float32.  P: [J x N] float64, P[j*N+n] = length_model.score((j+1)*fs, a_n)
 * with J = max_len / fs (rows for lengths fs, 2fs, .. J*fs).
 * Outputs: labels[T]; seg_label/seg_len[<=N], *n_seg; *score.  Returns a status code. */
static int decode_impl(const float *lp, int T, int C, const int32_t *transcript, int N,
                       const double *P, int fs, int max_len, long max_hyp, int32_t *labels,
                       int32_t *seg_label, int32_t *seg_len, int32_t *n_seg,
                       double *score_out) {
    if (T < 0 || C <= 0 || N <= 0 || fs <= 0 || max_len < fs) return ST_BAD_ARG;
    g_keyctx.tr = transcript;
    g_keyctx.fs = fs;
    int nan_in_prune = 0;
    if (T < fs) return ST_INDEX_ERROR; /* frame_scores[fs-1] out of range (viterbi.py:87) */
    const int J = max_len / fs;
    const int K = T / fs;

    /* frame_scores = np.cumsum(log_frame_probs, axis=0): sequential float32 adds (viterbi.py:51).
     * Only the sampled rows t = (k+1)*fs-1 are kept. */
    float *cs_s = (float *)malloc(sizeof(float) * (size_t)K * C); /* cs at sampled rows */
    float *run = (float *)calloc((size_t)C, sizeof(float));
    {
        int k = 0;
        for (int t = 0; t < K * fs; ++t) {
            const float *row = lp + (size_t)t * C;
            if (t == 0) {
                for (int c = 0; c < C; ++c) run[c] = row[c];
            } else {
                for (int c = 0; c < C; ++c) {
                    volatile float s = run[c] + row[c];
                    run[c] = s;
                }
            }
            if ((t + 1) % fs == 0) {
                memcpy(cs_s + (size_t)k * C, run, sizeof(float) * C);
                ++k;
            }
        }
    }
    free(run);
/* frame_score(t_k, label) (viterbi.py:68-72), float32 */
#define FRAME(k, c) ((k) == 0 ? cs_s[(c)] : (float)(cs_s[(size_t)(k) * C + (c)] - cs_s[(size_t)((k) - 1) * C + (c)]))

    hypdict_t a, b;
    a.items = (hyp_t *)malloc(sizeof(hyp_t) * (size_t)N * J);
    b.items = (hyp_t *)malloc(sizeof(hyp_t) * (size_t)N * J);
    a.pos = (int *)malloc(sizeof(int) * (size_t)N * J);
    b.pos = (int *)malloc(sizeof(int) * (size_t)N * J);
    /* bp[k*N+n]: length index j of the (n-1)-hypothesis of column k-1 that the (n, j=0)
     * hypothesis of column k points back to (the TracebackNode chain, viterbi.py:13-17,119-121). */
    int16_t *bp = (int16_t *)malloc(sizeof(int16_t) * (size_t)K * N);
    memset(bp, 0xff, sizeof(int16_t) * (size_t)K * N);
    hypdict_t *old = &a, *cur = &b;

    /* init_decoding (viterbi.py:81-90): score = 0.0 + frame_score  -> float32 */
    dict_clear(old, N, J);
    {
        volatile float s0 = 0.0f + FRAME(0, transcript[0]);
        dict_update(old, J, 0, 0, (double)s0);
    }

    /* decode_frame for t = 2fs-1, 3fs-1, ...  (viterbi.py:57-61, 92-123), each followed by prune (:61) */
    for (int k = 1; k < K; ++k) {
        dict_clear(cur, N, J);
        for (int i = 0; i < old->count; ++i) {
            const hyp_t h = old->items[i];
            const float f = FRAME(k, transcript[h.n]);
            /* hyp.score + frame_score: float32 + float32 for n == 0, float64 + float32 otherwise */
            double stay;
            if (h.n == 0) {
                volatile float t32 = (float)h.score + f;
                stay = (double)t32;
            } else {
                volatile double t64 = h.score + (double)f;
                stay = t64;
            }
            /* stay in the same label (viterbi.py:96-104) */
            if ((h.j + 1) * fs + fs <= max_len) dict_update(cur, J, h.n, h.j + 1, stay);
            /* go to the next label (viterbi.py:105-121); the end symbol is skipped */
            if (h.n + 1 < N) {
                volatile double s1 = stay + P[(size_t)h.j * N + h.n]; /* + length_model.score(length, label) */
                volatile double s2 = s1 + 0.0;                        /* + grammar.score(...) == 0.0         */
                if (dict_update(cur, J, h.n + 1, 0, s2)) bp[(size_t)k * N + h.n + 1] = (int16_t)h.j;
            }
        }
        nan_in_prune |= dict_prune(cur, J, max_hyp);
        hypdict_t *tmp = old;
        old = cur;
        cur = tmp;
    }

    /* finalize_decoding (viterbi.py:125-138) */
    double best = -INFINITY;
    int best_n = -1, best_j = -1;
    for (int i = 0; i < old->count; ++i) {
        const hyp_t h = old->items[i];
        volatile double s1 = h.score + P[(size_t)h.j * N + h.n];
        volatile double s2 = s1 + ((h.n == N - 1) ? 0.0 : -INFINITY);
        if (s2 >= best) {
            best = s2;
            best_n = h.n;
            best_j = h.j;
        }
    }
    int status = ST_OK;
    if (nan_in_prune) {
        status = ST_BAD_ARG;
    } else if (best_n < 0) {
        status = ST_NO_HYPOTHESIS;
    } else {
        /* traceback (viterbi.py:140-158): every node covers fs frames; leftover frames are
         * prepended with the LAST segment's label and added to the last segment's length. */
        int nseg = best_n + 1;
        int n = best_n, j = best_j, k = K - 1;
        for (int s = nseg - 1; s >= 0; --s) {
            seg_label[s] = transcript[n];
            seg_len[s] = (j + 1) * fs;
            const int k0 = k - j; /* column at which state n was entered */
            if (n > 0) {
                j = bp[(size_t)k0 * N + n];
                k = k0 - 1;
                n = n - 1;
            }
        }
        const int missing = T - K * fs;
        seg_len[nseg - 1] += missing;
        int t = 0;
        for (int m = 0; m < missing; ++m) labels[t++] = seg_label[nseg - 1];
        for (int s = 0; s < nseg; ++s) {
            const int len = seg_len[s] - (s == nseg - 1 ? missing : 0);
            for (int m = 0; m < len; ++m) labels[t++] = seg_label[s];
        }
        *n_seg = nseg;
        *score_out = best;
    }
    free(cs_s);
    free(a.items);
    free(b.items);
    free(a.pos);
    free(b.pos);
    free(bp);
    return status;
}

int mucon_oracle_viterbi_decode(const float *lp, int T, int C, const int32_t *transcript, int N,
                                const double *P, int fs, int max_len, int32_t *labels,
                                int32_t *seg_label, int32_t *seg_len, int32_t *n_seg,
                                double *score_out) {
    return decode_impl(lp, T, C, transcript, N, P, fs, max_len, -1, labels, seg_label, seg_len, n_seg, score_out);
}

/* The same with Viterbi(max_hypotheses = max_hyp) (viterbi.py:34): max_hyp >= 0; 0 never deletes anything (Python's tmp[0:-0]). */
int mucon_oracle_viterbi_decode_pruned(const float *lp, int T, int C, const int32_t *transcript, int N,
                                       const double *P, int fs, int max_len, long max_hyp, int32_t *labels,
                                       int32_t *seg_label, int32_t *seg_len, int32_t *n_seg,
                                       double *score_out) {
    if (max_hyp < 0) return ST_BAD_ARG;
    return decode_impl(lp, T, C, transcript, N, P, fs, max_len, max_hyp, labels, seg_label, seg_len, n_seg, score_out);
}

/* Frame scores only (for unit-testing the GPU kernel's first phase): F[K x C] float32. */
int mucon_oracle_frame_scores(const float *lp, int T, int C, int fs, float *F) {
    if (T < fs) return ST_INDEX_ERROR;
    const int K = T / fs;
    float *run = (float *)calloc((size_t)C, sizeof(float));
    float *prev = (float *)calloc((size_t)C, sizeof(float));
    int k = 0;
    for (int t = 0; t < K * fs; ++t) {
        const float *row = lp + (size_t)t * C;
        for (int c = 0; c < C; ++c) {
            if (t == 0) run[c] = row[c];
            else {
                volatile float s = run[c] + row[c];
                run[c] = s;
            }
        }
        if ((t + 1) % fs == 0) {
            for (int c = 0; c < C; ++c) {
                volatile float d = run[c] - prev[c];
                F[(size_t)k * C + c] = (k == 0) ? run[c] : d;
                prev[c] = run[c];
            }
            ++k;
        }
    }
    free(run);
    free(prev);
    return ST_OK;
}
