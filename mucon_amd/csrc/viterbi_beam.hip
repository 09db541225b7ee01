// Viterbi decode UNDER THE REFERENCE'S BEAM: Viterbi(max_hypotheses = M) with M below the N * J hypotheses a transcript can have alive
// (reference src/core/viterbi/viterbi.py:34, :57-61, :74-79: after every column `prune` sorts the (score, key) pairs and deletes the
// lowest len - M of them).  Off by default in the reference (its callers never pass max_hypotheses, evaluators.py:80); this is the exact
// device version of it, bit for bit (score, segments) -- a separate, small translation unit: the decode kernels of viterbi.hip keep the
// hypotheses of a state in fixed (state, length) slots and know which of them are alive in closed form, which is what a beam destroys.
//
// What a beam makes of the algorithm.  Which hypothesis wins a `<=` tie (HypDict.update, viterbi.py:26-28) or the `>=` fold of
// finalize_decoding (:125-138) depends on the ITERATION ORDER of the hypothesis dict.  Without a beam that order is, within a transcript
// state, increasing segment length.  With a beam it is not: decode_frame inserts the new entry of state n+1 right behind the stay-child of
// the FIRST SURVIVING hypothesis of state n, and that one can sit behind older entries of state n+1.  So the kernel carries the dict as
// what it is -- a LIST in iteration order:
//   per column, for the hypothesis at rank r (state n, length slot j, score s):
//     stay   (n, j+1, s + f_n)                            if (j+2) * fs <= max_length            (viterbi.py:96-104)
//     enter  (n+1, 0, max_c) behind it                    if r is the first hypothesis of state n and n+1 < N, where max_c is the largest
//            c = (s' + f_n + P[j'][n]) + 0.0 over the hypotheses of state n, the LAST of them in list order on ties      (:105-121)
//   new ranks = an exclusive scan of (stay + enter) over the old ranks; then prune: the D = len - M smallest under Python's tuple order of
//   (score, key) are dropped, the others keep their order (a radix select over an 88-bit composite, then a compaction scan).
// The key tuple is (-1, a_0 .. a_n, length): two keys of different depth differ where the shorter one's LENGTH meets the longer one's
// next LABEL, so the order has a closed form (beam_tie_key): verified against the literal oracle (oracle/viterbi_oracle.c:dict_prune) on
// the reference's own beam results (tests/golden/viterbi_pruned.*), all-ties inputs and frame_sampling 1 included.
// finalize_decoding is the same fold over the list (the last maximum wins): a beam that lost every path into the last transcript state
// returns, like the reference, score -inf and the truncated labelling of the LAST hypothesis in the list.
//
// One workgroup per video, the list in LDS (at most M + N entries between two prunes: M + N <= BEAM_MAX_ITEMS), two launches per call
// (frame scores: a sequential float32 cumsum per class, viterbi.py:51, 68-72; then the decode).  Throughput is not the point of this entry:
// a column costs ~10 barriers and, when it prunes, 11 histogram passes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "../../include/mucon_hip.h"

void mucon_internal_set_error(const char *msg);  // mucon_hip.hip: feeds mucon_last_error()

static int vit_fail(int code, const char *fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mucon_internal_set_error(buf);
    return code;
}

namespace {

constexpr int VB_T = 256;          // (the radix select clears its 256 histogram bins with one store per thread)
constexpr int VB_MAX_ITEMS = MUCON_VIT_BEAM_MAX_ITEMS;
constexpr int VB_MAX_N = 128, VB_MAX_J = 128, VB_MAX_C = 64;

struct VbJob {
    const float *lp;
    int32_t T, N;
    int64_t tr_off, p_off;       // elements into transcripts / tables
    int64_t f_off, bp_off;       // bytes into the scratch
    int64_t seg_off;             // elements into seg_len
};

// ---- frame scores: F[k][c] = cs[(k+1) fs - 1][c] - cs[k fs - 1][c], cs = the sequential float32 cumsum over frames (viterbi.py:51, :68-72)
__global__ __launch_bounds__(64) void vb_framescore_kernel(const VbJob *jobs, char *ws, int C, int fs) {
    const VbJob job = jobs[blockIdx.x];
    const int c = threadIdx.x, K = job.T / fs;
    if (c >= C || K < 1) return;
    float *F = reinterpret_cast<float *>(ws + job.f_off);
    const float *lp = job.lp + c;
    float run = 0.f, prev = 0.f;
    int in_col = 0, k = 0;
    const long frames = (long)K * fs;
    for (long t0 = 0; t0 < frames; t0 += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = lp[min(t0 + u, frames - 1) * C];     // eight loads in flight; the adds below stay sequential
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t0 + u >= frames) break;
            run = (t0 + u == 0) ? x[u] : run + x[u];
            if (++in_col == fs) {
                F[(long)k * C + c] = k == 0 ? run : run - prev;
                prev = run;
                in_col = 0;
                ++k;
            }
        }
    }
}

__device__ __forceinline__ unsigned long long vb_sortable(double x) {   // order-preserving map to u64; -0.0 and +0.0 compare equal in Python
    if (x == 0.0) x = 0.0;
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// hyp.score + frame_score: float32 + float32 while the hypothesis is in the first transcript state, float64 afterwards
__device__ __forceinline__ double vb_add_frame(double s, float f, int n) { return n == 0 ? (double)((float)s + f) : s + (double)f; }
// rank of the key tuple (-1, a_0 .. a_n, (j+1) fs) among all keys of one transcript: a key whose length is <= the label that follows its
// prefix in the transcript precedes every deeper key (and its own deeper continuations), one whose length is larger follows them all
__device__ __forceinline__ unsigned vb_tie_key(int n, int j, int N, int fs, const int *a) {
    const bool low = n == N - 1 || (j + 1) * fs <= a[n + 1];
    return low ? (unsigned)(n * 128 + j) : (1u << 20) + (unsigned)((N - 1 - n) * 128 + j);
}

// exclusive scan of one value per thread over the workgroup (256 threads = 4 waves); *total = the sum
__device__ __forceinline__ unsigned vb_block_exscan(unsigned v, unsigned *s_wave, unsigned *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return base + inc - v;
}

// grid (n_videos), 256 threads, dynamic LDS = 2 x VB_MAX_ITEMS x (8 + 1 + 1) + VB_MAX_ITEMS bytes
__global__ __launch_bounds__(VB_T) void vb_decode_kernel(const VbJob *jobs, const int32_t *transcripts, const double *tables, char *ws, int C, int fs,
                                                         int J, long long M, double *score, int32_t *n_seg, int32_t *status, int32_t *seg_len) {
    extern __shared__ __attribute__((aligned(16))) char vb_dyn[];
    double *s_sc = reinterpret_cast<double *>(vb_dyn);                                   // [2][VB_MAX_ITEMS]
    uint8_t *s_n = reinterpret_cast<uint8_t *>(vb_dyn + 2 * VB_MAX_ITEMS * 8);           // [2][VB_MAX_ITEMS]
    uint8_t *s_j = s_n + 2 * VB_MAX_ITEMS;                                               // [2][VB_MAX_ITEMS]
    uint8_t *s_dead = s_j + 2 * VB_MAX_ITEMS;                                            // [VB_MAX_ITEMS]
    __shared__ unsigned long long s_cmax[VB_MAX_N], s_key;
    __shared__ unsigned s_first[VB_MAX_N], s_wrank[VB_MAX_N], s_hist[256], s_wave[4], s_rank, s_lo, s_rem, s_digit;
    __shared__ int s_a[VB_MAX_N];
    __shared__ float s_f[VB_MAX_C];
    const VbJob job = jobs[blockIdx.x];
    const int vid = blockIdx.x, tid = threadIdx.x, T = job.T, N = job.N, K = T / fs;
    if (K < 1) {   // frame_scores[fs-1] does not exist: IndexError in the reference (viterbi.py:87)
        if (tid == 0) {
            status[vid] = MUCON_VIT_INDEX_ERROR;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }
    const float *F = reinterpret_cast<const float *>(ws + job.f_off);
    uint8_t *bp = reinterpret_cast<uint8_t *>(ws + job.bp_off);                          // [K][N]: the length slot an entry came from
    const double *P = tables + job.p_off;                                                // [J][N]
    for (int e = tid; e < N; e += VB_T) s_a[e] = transcripts[job.tr_off + e];
    __syncthreads();
    int cur = 0, m = 1;
    if (tid == 0) {   // init_decoding (viterbi.py:81-90): score = 0.0 + frame_score(fs-1, a_0), float32
        s_n[0] = 0;
        s_j[0] = 0;
        s_sc[0] = (double)(0.0f + F[s_a[0]]);
    }
    __syncthreads();

    for (int k = 1; k < K; ++k) {
        double *sc = s_sc + cur * VB_MAX_ITEMS, *sc2 = s_sc + (cur ^ 1) * VB_MAX_ITEMS;
        uint8_t *sn = s_n + cur * VB_MAX_ITEMS, *sj = s_j + cur * VB_MAX_ITEMS, *sn2 = s_n + (cur ^ 1) * VB_MAX_ITEMS, *sj2 = s_j + (cur ^ 1) * VB_MAX_ITEMS;
        for (int e = tid; e < N; e += VB_T) {
            s_first[e] = 0xFFFFFFFFu;
            s_cmax[e] = 0ull;
            s_wrank[e] = 0u;
        }
        for (int e = tid; e < C; e += VB_T) s_f[e] = F[(long)k * C + e];
        __syncthreads();
        // -- per state: its first hypothesis in list order, the largest entering candidate
        for (int r = tid; r < m; r += VB_T) {
            const int n = sn[r], j = sj[r];
            atomicMin(&s_first[n], (unsigned)r);
            if (n + 1 < N) {
                const double c = (vb_add_frame(sc[r], s_f[s_a[n]], n) + P[(long)j * N + n]) + 0.0;
                atomicMax(&s_cmax[n], vb_sortable(c));
            }
        }
        __syncthreads();
        // -- ... and the LAST hypothesis in list order that reaches it (HypDict.update replaces on `<=`)
        for (int r = tid; r < m; r += VB_T) {
            const int n = sn[r], j = sj[r];
            if (n + 1 < N) {
                const double c = (vb_add_frame(sc[r], s_f[s_a[n]], n) + P[(long)j * N + n]) + 0.0;
                if (vb_sortable(c) == s_cmax[n]) atomicMax(&s_wrank[n], (unsigned)r);
            }
        }
        __syncthreads();
        // -- the new list: a thread takes a contiguous run of ranks, an exclusive scan places its output
        const int per = (m + VB_T - 1) / VB_T, r0 = min(tid * per, m), r1 = min(r0 + per, m);
        unsigned cnt = 0;
        for (int r = r0; r < r1; ++r) {
            const int n = sn[r], j = sj[r];
            cnt += (j + 1 < J ? 1u : 0u) + ((s_first[n] == (unsigned)r && n + 1 < N) ? 1u : 0u);
        }
        unsigned m2;
        unsigned pos = vb_block_exscan(cnt, s_wave, &m2);
        for (int r = r0; r < r1; ++r) {
            const int n = sn[r], j = sj[r];
            const float f = s_f[s_a[n]];
            if (j + 1 < J) {                          // `length + frame_sampling <= max_length()`
                sn2[pos] = (uint8_t)n;
                sj2[pos] = (uint8_t)(j + 1);
                sc2[pos] = vb_add_frame(sc[r], f, n);
                ++pos;
            }
            if (s_first[n] == (unsigned)r && n + 1 < N) {
                const int w = (int)s_wrank[n], jw = sj[w];
                sn2[pos] = (uint8_t)(n + 1);
                sj2[pos] = 0;
                sc2[pos] = (vb_add_frame(sc[w], f, n) + P[(long)jw * N + n]) + 0.0;
                bp[(long)k * N + n + 1] = (uint8_t)jw;
                ++pos;
            }
        }
        __syncthreads();
        cur ^= 1;
        m = (int)m2;
        // -- prune (viterbi.py:74-79)
        if (M > 0 && (long long)m > M) {
            const unsigned D = (unsigned)((long long)m - M);     // the D smallest composites go
            if (tid == 0) {
                s_key = 0ull;
                s_lo = 0u;
                s_rem = D;
            }
            __syncthreads();
            for (int pass = 0; pass < 11; ++pass) {              // bytes 7..0 of the score key, then bytes 2..0 of the tie key
                s_hist[tid] = 0u;
                __syncthreads();
                const unsigned long long pk = s_key;
                const unsigned plo = s_lo;
                for (int r = tid; r < m; r += VB_T) {
                    const unsigned long long key = vb_sortable(sc2[r]);
                    const unsigned lo = vb_tie_key(sn2[r], sj2[r], N, fs, s_a);
                    bool match;
                    unsigned digit;
                    if (pass < 8) {
                        const int sh = 8 * (7 - pass);
                        match = pass == 0 || (key >> (sh + 8)) == (pk >> (sh + 8));
                        digit = (unsigned)(key >> sh) & 255u;
                    } else {
                        const int sh = 8 * (10 - pass);
                        match = key == pk && (pass == 8 || (lo >> (sh + 8)) == (plo >> (sh + 8)));
                        digit = (lo >> sh) & 255u;
                    }
                    if (match) atomicAdd(&s_hist[digit], 1u);
                }
                __syncthreads();
                if (tid < 64) {                                  // the digit in which the remaining rank falls: four bins per lane
                    const unsigned h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
                    unsigned inc = h0 + h1 + h2 + h3;
                    const unsigned mine = inc;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned up = __shfl_up(inc, o);
                        if (tid >= o) inc += up;
                    }
                    const unsigned before = inc - mine, rem = s_rem;
                    if (before < rem && rem <= inc) {
                        unsigned acc = before;
                        int d = 0;
                        if (acc + h0 >= rem) d = 0;
                        else if ((acc += h0) + h1 >= rem) d = 1;
                        else if ((acc += h1) + h2 >= rem) d = 2;
                        else { acc += h2; d = 3; }
                        s_digit = (unsigned)(4 * tid + d);
                        s_rank = rem - acc;                      // rank inside that digit's items
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    if (pass < 8) s_key |= (unsigned long long)s_digit << (8 * (7 - pass));
                    else s_lo |= s_digit << (8 * (10 - pass));
                    s_rem = s_rank;
                }
                __syncthreads();
            }
            const unsigned long long tkey = s_key;
            const unsigned tlo = s_lo;
            const int per2 = (m + VB_T - 1) / VB_T, q0 = min(tid * per2, m), q1 = min(q0 + per2, m);
            unsigned keep = 0;
            for (int r = q0; r < q1; ++r) {
                const unsigned long long key = vb_sortable(sc2[r]);
                const unsigned lo = vb_tie_key(sn2[r], sj2[r], N, fs, s_a);
                const bool dead = key < tkey || (key == tkey && lo <= tlo);
                s_dead[r] = dead ? 1 : 0;
                keep += dead ? 0u : 1u;
            }
            unsigned mk;
            unsigned at = vb_block_exscan(keep, s_wave, &mk);
            for (int r = q0; r < q1; ++r)            // the survivors, in order, into the other buffer (`del hyps[key]` keeps the others' order)
                if (!s_dead[r]) {
                    sn[at] = sn2[r];
                    sj[at] = sj2[r];
                    sc[at] = sc2[r];
                    ++at;
                }
            __syncthreads();
            cur ^= 1;
            m = (int)mk;
        }
    }

    // -- finalize_decoding (viterbi.py:125-138): the `>=` fold over the list = its largest final score, the last one on ties
    const double *sc = s_sc + cur * VB_MAX_ITEMS;
    const uint8_t *sn = s_n + cur * VB_MAX_ITEMS, *sj = s_j + cur * VB_MAX_ITEMS;
    if (tid == 0) {
        s_key = 0ull;
        s_rank = 0u;
    }
    __syncthreads();
    for (int r = tid; r < m; r += VB_T) {
        const int n = sn[r], j = sj[r];
        const double v = (sc[r] + P[(long)j * N + n]) + (n == N - 1 ? 0.0 : -INFINITY);
        atomicMax(&s_key, vb_sortable(v));
    }
    __syncthreads();
    for (int r = tid; r < m; r += VB_T) {
        const int n = sn[r], j = sj[r];
        const double v = (sc[r] + P[(long)j * N + n]) + (n == N - 1 ? 0.0 : -INFINITY);
        if (vb_sortable(v) == s_key) atomicMax(&s_rank, (unsigned)r);
    }
    __syncthreads();
    if (tid == 0) {
        if (m == 0) {   // no hypothesis left: final_hyp.traceback is None, AttributeError in the reference (viterbi.py:147)
            status[vid] = MUCON_VIT_NO_HYPOTHESIS;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
            return;
        }
        const int r = (int)s_rank;
        int n = sn[r], j = sj[r], k = K - 1;
        const int nseg = n + 1;
        const double v = (sc[r] + P[(long)j * N + n]) + (n == N - 1 ? 0.0 : -INFINITY);
        int32_t *seg = seg_len + job.seg_off;
        // traceback (viterbi.py:140-158): every node covers fs frames; the frames beyond K fs belong to the last segment
        for (int s = nseg - 1; s >= 0; --s) {
            seg[s] = (j + 1) * fs;
            const int k0 = k - j;      // the column at which state n was entered
            if (n > 0) {
                j = bp[(long)k0 * N + n];
                k = k0 - 1;
                --n;
            }
        }
        seg[nseg - 1] += T - K * fs;
        n_seg[vid] = nseg;
        score[vid] = v;
        status[vid] = nseg == N ? MUCON_VIT_OK : MUCON_VIT_TRUNCATED;
    }
}

struct VbState {
    std::mutex mu;
    char *dev = nullptr, *pin = nullptr;
    size_t dev_cap = 0, pin_cap = 0;
    int device = -1;          // the device `dev` lives on and the kernel attribute was set for
    bool attr_set = false;
};
VbState g_vb;

int vb_grow(char **buf, size_t *cap, size_t need, bool pinned) {
    if (need <= *cap) return MUCON_OK;
    if (*buf) {
        if (pinned) (void)hipHostFree(*buf);
        else (void)hipFree(*buf);
        *buf = nullptr;
        *cap = 0;
    }
    const size_t want = need + need / 2 + 4096;
    const hipError_t e = pinned ? hipHostMalloc(reinterpret_cast<void **>(buf), want, hipHostMallocDefault) : hipMalloc(reinterpret_cast<void **>(buf), want);
    if (e != hipSuccess) return vit_fail(MUCON_E_HIP, "viterbi beam: allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
    *cap = want;
    return MUCON_OK;
}
inline size_t vb_al(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int mucon_viterbi_decode_beam(int32_t n_videos, const mucon_viterbi_video *videos, int32_t C, int32_t fs, int32_t max_len,
                                         int64_t max_hypotheses, double *score, int32_t *n_seg, int32_t *status, int32_t *seg_len, void *stream) {
    if (n_videos < 1 || !videos || !score || !n_seg || !status || !seg_len) return vit_fail(MUCON_E_ARG, "viterbi beam: null argument or no videos");
    if (C < 1 || C > VB_MAX_C || fs < 1 || max_len < fs) return vit_fail(MUCON_E_ARG, "viterbi beam: C must be 1..%d, 1 <= fs <= max_len", VB_MAX_C);
    if (max_hypotheses < 1) return vit_fail(MUCON_E_ARG, "viterbi beam: max_hypotheses must be >= 1 (0 and inf never prune: use mucon_viterbi_decode_host)");
    const int J = max_len / fs;
    if (J > VB_MAX_J) return vit_fail(MUCON_E_ARG, "viterbi beam: max_len / fs = %d length slots, at most %d", J, VB_MAX_J);
    size_t sum_N = 0, ws_bytes = 0;
    for (int v = 0; v < n_videos; ++v) {
        const mucon_viterbi_video &q = videos[v];
        if (!q.lp || !q.transcript || !q.table || q.T < 0 || q.N < 1 || q.N > VB_MAX_N)
            return vit_fail(MUCON_E_ARG, "viterbi beam: video %d: null pointer, T < 0 or N outside 1..%d", v, VB_MAX_N);
        if (max_hypotheses + q.N > VB_MAX_ITEMS)
            return vit_fail(MUCON_E_ARG, "viterbi beam: max_hypotheses + N = %lld, the list holds %d hypotheses", (long long)max_hypotheses + q.N, VB_MAX_ITEMS);
        sum_N += (size_t)q.N;
        const size_t K = (size_t)(q.T / fs);
        ws_bytes += vb_al(K * C * sizeof(float)) + vb_al(K * q.N);
    }
    // one device slab, one pinned slab with the same layout: [jobs | transcripts | tables | score | n_seg | status | seg_len] (+ device only: scratch)
    const size_t o_jobs = 0, o_tr = vb_al(o_jobs + sizeof(VbJob) * (size_t)n_videos), o_tab = vb_al(o_tr + 4 * sum_N);
    const size_t o_score = vb_al(o_tab + 8 * (size_t)J * sum_N), o_nseg = vb_al(o_score + 8 * (size_t)n_videos), o_stat = vb_al(o_nseg + 4 * (size_t)n_videos);
    const size_t o_seg = vb_al(o_stat + 4 * (size_t)n_videos), o_ws = vb_al(o_seg + 4 * sum_N), total = o_ws + ws_bytes;
    std::lock_guard<std::mutex> lock(g_vb.mu);
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess) return vit_fail(MUCON_E_HIP, "viterbi beam: hipGetDevice failed");
    if (cur_dev != g_vb.device) {          // another device than last time: its own scratch, its own kernel attribute
        if (g_vb.dev) {
            const int old_dev = g_vb.device;
            (void)hipSetDevice(old_dev);
            (void)hipFree(g_vb.dev);
            (void)hipSetDevice(cur_dev);
        }
        g_vb.dev = nullptr;
        g_vb.dev_cap = 0;
        g_vb.attr_set = false;
        g_vb.device = cur_dev;
    }
    int rc = vb_grow(&g_vb.dev, &g_vb.dev_cap, total, false);
    if (rc != MUCON_OK) return rc;
    rc = vb_grow(&g_vb.pin, &g_vb.pin_cap, o_ws, true);
    if (rc != MUCON_OK) return rc;
    VbJob *jobs = reinterpret_cast<VbJob *>(g_vb.pin + o_jobs);
    int32_t *tr = reinterpret_cast<int32_t *>(g_vb.pin + o_tr);
    double *tab = reinterpret_cast<double *>(g_vb.pin + o_tab);
    size_t at_n = 0, at_ws = o_ws;
    for (int v = 0; v < n_videos; ++v) {
        const mucon_viterbi_video &q = videos[v];
        const size_t K = (size_t)(q.T / fs);
        VbJob j;
        j.lp = q.lp;
        j.T = q.T;
        j.N = q.N;
        j.tr_off = (int64_t)at_n;
        j.p_off = (int64_t)((size_t)J * at_n);
        j.f_off = (int64_t)at_ws;
        at_ws += vb_al(K * C * sizeof(float));
        j.bp_off = (int64_t)at_ws;
        at_ws += vb_al(K * q.N);
        j.seg_off = (int64_t)at_n;
        jobs[v] = j;
        memcpy(tr + at_n, q.transcript, 4 * (size_t)q.N);
        memcpy(tab + (size_t)J * at_n, q.table, 8 * (size_t)J * q.N);
        at_n += (size_t)q.N;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(g_vb.dev, g_vb.pin, o_score, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return vit_fail(MUCON_E_HIP, "viterbi beam: upload failed: %s", hipGetErrorString(e));
    constexpr size_t dyn = 2 * (size_t)VB_MAX_ITEMS * 10 + VB_MAX_ITEMS;
    if (!g_vb.attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(vb_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return vit_fail(MUCON_E_HIP, "viterbi beam: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        g_vb.attr_set = true;
    }
    const VbJob *d_jobs = reinterpret_cast<const VbJob *>(g_vb.dev + o_jobs);
    hipLaunchKernelGGL(vb_framescore_kernel, dim3(n_videos), dim3(64), 0, s, d_jobs, g_vb.dev, (int)C, (int)fs);
    hipLaunchKernelGGL(vb_decode_kernel, dim3(n_videos), dim3(VB_T), dyn, s, d_jobs, reinterpret_cast<const int32_t *>(g_vb.dev + o_tr),
                       reinterpret_cast<const double *>(g_vb.dev + o_tab), g_vb.dev, (int)C, (int)fs, J, (long long)max_hypotheses,
                       reinterpret_cast<double *>(g_vb.dev + o_score), reinterpret_cast<int32_t *>(g_vb.dev + o_nseg),
                       reinterpret_cast<int32_t *>(g_vb.dev + o_stat), reinterpret_cast<int32_t *>(g_vb.dev + o_seg));
    e = hipGetLastError();
    if (e != hipSuccess) return vit_fail(MUCON_E_HIP, "viterbi beam: launch failed: %s", hipGetErrorString(e));
    e = hipMemcpyAsync(g_vb.pin + o_score, g_vb.dev + o_score, o_ws - o_score, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return vit_fail(MUCON_E_HIP, "viterbi beam: download failed: %s", hipGetErrorString(e));
    memcpy(score, g_vb.pin + o_score, 8 * (size_t)n_videos);
    memcpy(n_seg, g_vb.pin + o_nseg, 4 * (size_t)n_videos);
    memcpy(status, g_vb.pin + o_stat, 4 * (size_t)n_videos);
    memcpy(seg_len, g_vb.pin + o_seg, 4 * sum_N);
    return MUCON_OK;
}
