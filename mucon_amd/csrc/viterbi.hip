// Transcript-constrained Viterbi decode on gfx950 -- bit-exact against the reference's
// Viterbi.decode (src/core/viterbi/viterbi.py:49-158) with SingleTranscriptGrammar
// (src/core/viterbi/grammar.py:196-217) and an f64 length table (PoissonModel,
// src/core/viterbi/length_model.py:76-80).  Compiled with -ffp-contract=off: every score is a
// chain of single IEEE adds in the reference's order and dtype.
//
// Formulation.  A hypothesis is (n, k0): transcript position n entered at column k0 (column k
// <-> frame (k+1)*fs-1); its segment length at column k is (k-k0+1)*fs.  Instead of shifting a
// [N x J] table every column (J = max_len/fs length slots, 66 by default), slot r = k0 mod J of
// state n holds the hypothesis for its whole life: a slot is re-used exactly when its old tenant
// reaches the maximum length.  Which slots are alive is a closed form of (n, k0, k):
// entries into state n >= 1 exist for n <= k0 <= J*n, state 0 is entered at k0 = 0 only.
//
//   phase 1  viterbi_framescore_kernel: one wave per video, one lane per class: the sequential
//            float32 cumsum of the emissions (np.cumsum, viterbi.py:51) and its fs-strided
//            differences (frame_score, viterbi.py:68-72) -> F[K][C].
//   phase 2  viterbi_dp_kernel: one workgroup per video, one wave per transcript state (looping
//            when N > 16), lanes = length slots (lane and lane+64).  The previous time column
//            S_old[N][J] (f64) lives in LDS, double buffered, one barrier per column.  The wave of
//            state m also evaluates the "advance" candidates of state m-1 (a max-reduction with
//            the reference's tie rule: `<=` in HypDict.update keeps the LAST, i.e. longest, of
//            equal candidates) so entries need no second barrier.  Back-pointers go to HBM
//            scratch; thread 0 walks them (viterbi.py:140-158) and all threads expand labels.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mucon_hip.h"

namespace {

constexpr int VIT_THREADS = 1024;
constexpr int VIT_FCHUNK = 16;  // columns of frame scores staged in LDS at a time

__global__ __launch_bounds__(64) void viterbi_framescore_kernel(const mucon_viterbi_job *jobs, const float *lp,
                                                                char *ws, int C, int fs) {
    const mucon_viterbi_job job = jobs[blockIdx.x];
    const int K = job.T / fs;
    if (K < 1) return;
    float *F = reinterpret_cast<float *>(ws + job.ws_off);
    const int n = K * fs;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float *p = lp + job.lp_off + c;
        float run = 0.f, prev = 0.f;
        int until = fs;  // frames left in the current column
        int k = 0;
        for (int t0 = 0; t0 < n; t0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (t0 + u < n) ? p[(long)(t0 + u) * C] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < n) {
                    run = (t0 + u == 0) ? v[u] : run + v[u];  // sequential float32 chain
                    if (--until == 0) {
                        F[(long)k * C + c] = (k == 0) ? run : run - prev;
                        prev = run;
                        until = fs;
                        ++k;
                    }
                }
            }
        }
    }
}

struct Cand {
    double v;
    int j;
};
__device__ __forceinline__ Cand better(Cand a, Cand b) {
    // the later (longer) of equal candidates wins: HypDict.update uses `<=` (viterbi.py:27)
    const bool take = (b.v > a.v) || (b.v == a.v && b.j > a.j);
    return take ? b : a;
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Cand d;
        d.v = __shfl_xor(c.v, o);
        d.j = __shfl_xor(c.j, o);
        c = better(c, d);
    }
    return c;
}
__device__ __forceinline__ bool alive_at(int n, int k0, int J) {
    return k0 >= 0 && (n == 0 ? (k0 == 0) : (k0 >= n && k0 <= J * n));
}
// hyp.score + frame_score (viterbi.py:99,113): float32 + float32 while in the first transcript
// state, float64 + float32 afterwards (NumPy 2 promotion; the length table makes later states f64)
__device__ __forceinline__ double add_frame(double s, float f, int n) {
    if (n == 0) {
        const float t = (float)s + f;
        return (double)t;
    }
    return s + (double)f;
}

__global__ __launch_bounds__(VIT_THREADS) void viterbi_dp_kernel(
    const mucon_viterbi_job *jobs, const int32_t *transcripts, const double *tables, int32_t *labels,
    int32_t *seg_len, int32_t *n_seg, double *score, int32_t *status, char *ws, int C, int fs, int J) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const mucon_viterbi_job job = jobs[blockIdx.x];
    const int T = job.T, N = job.N;
    const int K = T / fs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = VIT_THREADS / 64;
    const int vid = blockIdx.x;

    if (K < 1) {  // frame_scores[fs-1] does not exist: IndexError in the reference (viterbi.py:87)
        if (tid == 0) {
            status[vid] = MUCON_VIT_INDEX_ERROR;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }
    if (job.force_n < 0 && K > J * N) {  // every hypothesis has outlived max_length: empty set
        if (tid == 0) {
            status[vid] = MUCON_VIT_NO_HYPOTHESIS;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }

    // LDS carve (all dynamic, 16-byte aligned base):
    //   fin[16 B] | S[2][N][J] f64 | P[N][J] f64 | Fs[2][16][C] f32 | a[N] i32 | pre[N+1] i32
    double &fin_score = *reinterpret_cast<double *>(smem_raw);
    int &fin_n = *reinterpret_cast<int *>(smem_raw + 8);
    int &fin_j = *reinterpret_cast<int *>(smem_raw + 12);
    double *S = reinterpret_cast<double *>(smem_raw + 16);
    double *Pl = S + 2 * (size_t)N * J;
    float *Fs = reinterpret_cast<float *>(Pl + (size_t)N * J);
    int *a = reinterpret_cast<int *>(Fs + 2 * VIT_FCHUNK * C);
    int *pre = a + N;

    const float *F = reinterpret_cast<const float *>(ws + job.ws_off);
    const size_t f_bytes = ((size_t)K * C * sizeof(float) + 15) & ~(size_t)15;
    uint8_t *bp = reinterpret_cast<uint8_t *>(ws + job.ws_off + f_bytes);  // [K][N]

    for (int e = tid; e < N; e += VIT_THREADS) a[e] = transcripts[job.tr_off + e];
    for (int e = tid; e < N * J; e += VIT_THREADS) {
        const int n = e / J, j = e - n * J;
        Pl[e] = tables[job.p_off + (size_t)j * N + n];
    }
    // frame-score chunks 0 and 1
    for (int e = tid; e < 2 * VIT_FCHUNK * C; e += VIT_THREADS) {
        const int col = e / C;
        Fs[e] = (col < K) ? F[e] : 0.f;
    }
    __syncthreads();

    // init_decoding (viterbi.py:81-90): score = 0.0 + frame_score(fs-1, a_0), float32
    if (tid == 0) {
        const float s0 = 0.0f + Fs[a[0]];
        S[0] = (double)s0;  // buffer 0, state 0, slot 0
    }
    __syncthreads();

    // slot -> current length index: at old column c = k-1 slot r holds j = (c - r) mod J
    int jr0 = (J - (lane % J)) % J;                    // c = 0, r = lane       (lane < J assumed for r)
    int jr1 = (J - ((lane + 64) % J)) % J;             // c = 0, r = lane + 64
    const bool has0 = lane < J, has1 = lane + 64 < J;  // J <= 128 slots per state
    float fpre = 0.f;

    for (int k = 1; k < K; ++k) {
        double *So = S + (size_t)((k - 1) & 1) * N * J;
        double *Sn = S + (size_t)(k & 1) * N * J;
        // stage frame scores: chunk q+1 is fetched at the start of chunk q and stored half way
        const int kin = k & (VIT_FCHUNK - 1);
        const int q = k / VIT_FCHUNK;
        if (kin == 0 && k >= VIT_FCHUNK) {
            const int col = (q + 1) * VIT_FCHUNK + tid / C;
            fpre = (tid < VIT_FCHUNK * C && col < K) ? F[(size_t)(q + 1) * VIT_FCHUNK * C + tid] : 0.f;
        }
        if (kin == VIT_FCHUNK / 2 && k >= VIT_FCHUNK) {
            if (tid < VIT_FCHUNK * C) Fs[((q + 1) & 1) * VIT_FCHUNK * C + tid] = fpre;
        }
        const float *Fk = Fs + (q & 1) * VIT_FCHUNK * C + kin * C;
        const int c_old = k - 1;
        const int kslot = k % J;
        const int k00 = c_old - jr0, k01 = c_old - jr1;  // entry columns of this lane's two slots

        for (int m = wave; m < N; m += nwaves) {
            // (1) stay in state m (viterbi.py:96-104)
            const float fm = Fk[a[m]];
            if (has0 && alive_at(m, k00, J) && jr0 + 1 < J) Sn[m * J + lane] = add_frame(So[m * J + lane], fm, m);
            if (has1 && alive_at(m, k01, J) && jr1 + 1 < J)
                Sn[m * J + lane + 64] = add_frame(So[m * J + lane + 64], fm, m);
            // (2) enter state m from state m-1 (viterbi.py:105-121): the frame score is the OLD label's
            if (m >= 1 && k >= m && k <= J * m) {
                const int pm = m - 1;
                const float fp = Fk[a[pm]];
                Cand best;
                best.v = -INFINITY;
                best.j = -1;
                if (has0 && alive_at(pm, k00, J)) {
                    Cand c;
                    c.v = (add_frame(So[pm * J + lane], fp, pm) + Pl[pm * J + jr0]) + 0.0;
                    c.j = jr0;
                    best = better(best, c);
                }
                if (has1 && alive_at(pm, k01, J)) {
                    Cand c;
                    c.v = (add_frame(So[pm * J + lane + 64], fp, pm) + Pl[pm * J + jr1]) + 0.0;
                    c.j = jr1;
                    best = better(best, c);
                }
                best = wave_best(best);
                if (lane == 0) {
                    Sn[m * J + kslot] = best.v;
                    bp[(size_t)k * N + m] = (uint8_t)best.j;
                }
            }
        }
        jr0 = (jr0 + 1 == J) ? 0 : jr0 + 1;
        jr1 = (jr1 + 1 == J) ? 0 : jr1 + 1;
        __syncthreads();
    }

    // finalize_decoding (viterbi.py:125-138)
    const double *Sf = S + (size_t)((K - 1) & 1) * N * J;
    const bool forced = job.force_n >= 0 || K < N;
    if (forced) {
        // Degenerate outcomes of the reference: no hypothesis of the last transcript state has a
        // comparable score, every final score is -inf and the LAST hypothesis in dictionary order
        // wins (`>=`, viterbi.py:135).  For K < N that is (K-1, just entered); the NaN-length-model
        // cases are resolved on the host (mucon_amd/core/viterbi/viterbi.py) and passed in force_*.
        if (tid == 0) {
            fin_n = job.force_n >= 0 ? job.force_n : K - 1;
            fin_j = job.force_n >= 0 ? job.force_j : 0;
            fin_score = -INFINITY;
        }
    } else if (wave == 0) {
        const int nf = N - 1;
        const int c = K - 1;
        Cand best;
        best.v = -INFINITY;
        best.j = -1;
        // jr0/jr1 now describe column c = K-1
        if (has0 && alive_at(nf, c - jr0, J)) {
            Cand d;
            d.v = (Sf[nf * J + lane] + Pl[nf * J + jr0]) + 0.0;
            d.j = jr0;
            best = better(best, d);
        }
        if (has1 && alive_at(nf, c - jr1, J)) {
            Cand d;
            d.v = (Sf[nf * J + lane + 64] + Pl[nf * J + jr1]) + 0.0;
            d.j = jr1;
            best = better(best, d);
        }
        best = wave_best(best);
        if (lane == 0) {
            fin_n = nf;
            fin_j = best.j;
            fin_score = best.v;
        }
    }
    __syncthreads();
    if (fin_j < 0) {  // no comparable final hypothesis (NaN scores): traceback is None in the reference
        if (tid == 0) {
            status[vid] = MUCON_VIT_NO_HYPOTHESIS;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }

    // traceback (viterbi.py:140-158)
    const int nseg = fin_n + 1;
    const int missing = T - K * fs;
    if (tid == 0) {
        int n = fin_n, j = fin_j, k = K - 1;
        int32_t *sl = seg_len + job.seg_off;
        for (int s = nseg - 1; s >= 0; --s) {
            sl[s] = (j + 1) * fs;
            const int k0 = k - j;
            if (n > 0) {
                j = bp[(size_t)k0 * N + n];
                k = k0 - 1;
                --n;
            }
        }
        int acc = 0;
        for (int s = 0; s < nseg; ++s) {
            pre[s] = acc;
            acc += sl[s];
        }
        pre[nseg] = acc;
        sl[nseg - 1] += missing;  // leftover frames are added to the last segment's length
        for (int s = nseg; s < N; ++s) sl[s] = 0;
        n_seg[vid] = nseg;
        score[vid] = fin_score;
        status[vid] = forced ? MUCON_VIT_TRUNCATED : MUCON_VIT_OK;
    }
    __syncthreads();
    // ... and labelled, at the START of the video, with the last segment's label
    int32_t *lab = labels + job.label_off;
    for (int t = tid; t < T; t += VIT_THREADS) {
        int l;
        if (t < missing) {
            l = a[nseg - 1];
        } else {
            const int u = t - missing;
            int lo = 0, hi = nseg - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pre[mid] <= u) lo = mid;
                else hi = mid - 1;
            }
            l = a[lo];
        }
        lab[t] = l;
    }
}

thread_local char g_err[256];

}  // namespace

void mucon_internal_set_error(const char *msg);  // mucon_hip.hip: feeds mucon_last_error()
#define VIT_FAIL(code)                    \
    do {                                  \
        mucon_internal_set_error(g_err);  \
        return (code);                    \
    } while (0)

extern "C" size_t mucon_viterbi_job_workspace_bytes(int32_t T, int32_t C, int32_t N, int32_t fs) {
    const size_t K = fs > 0 ? (size_t)(T / fs) : 0;
    const size_t f_bytes = (K * (size_t)C * sizeof(float) + 15) & ~(size_t)15;
    const size_t bp_bytes = (K * (size_t)N + 15) & ~(size_t)15;
    return f_bytes + bp_bytes + 16;
}

extern "C" int mucon_viterbi_decode_batch(int32_t n_videos, const mucon_viterbi_job *jobs, int32_t C, int32_t fs,
                                          int32_t max_len, int32_t max_N, const float *lp,
                                          const int32_t *transcripts, const double *length_tables,
                                          int32_t *labels, int32_t *seg_len, int32_t *n_seg, double *score,
                                          int32_t *status, void *workspace, void *stream) {
    if (n_videos <= 0) return MUCON_OK;
    if (fs <= 0 || max_len < fs || C <= 0 || C > 64 || max_N <= 0) {
        snprintf(g_err, sizeof(g_err), "viterbi: unsupported arguments (C=%d must be <= 64, fs=%d, max_len=%d, max_N=%d)",
                 C, fs, max_len, max_N);
        VIT_FAIL(MUCON_E_ARG);
    }
    const int J = max_len / fs;
    if (J > 128) {
        snprintf(g_err, sizeof(g_err), "viterbi: max_len/fs = %d length slots > 128 not supported", J);
        VIT_FAIL(MUCON_E_ARG);
    }
    const size_t smem = 16 + (size_t)3 * max_N * J * sizeof(double) + (size_t)2 * VIT_FCHUNK * C * sizeof(float) +
                        (size_t)(2 * max_N + 1) * sizeof(int) + 16;
    if (smem > 160 * 1024 - 64) {
        snprintf(g_err, sizeof(g_err), "viterbi: transcript of %d states x %d slots needs %zu B of LDS (> 160 KiB)",
                 max_N, J, smem);
        VIT_FAIL(MUCON_E_ARG);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_dp_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess) {
            snprintf(g_err, sizeof(g_err), "viterbi: hipFuncSetAttribute failed");
            VIT_FAIL(MUCON_E_HIP);
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(viterbi_framescore_kernel, dim3(n_videos), dim3(64), 0, s, jobs, lp,
                       static_cast<char *>(workspace), C, fs);
    hipLaunchKernelGGL(viterbi_dp_kernel, dim3(n_videos), dim3(VIT_THREADS), smem, s, jobs, transcripts,
                       length_tables, labels, seg_len, n_seg, score, status, static_cast<char *>(workspace), C,
                       fs, J);
    if (hipGetLastError() != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "viterbi: kernel launch failed");
        VIT_FAIL(MUCON_E_HIP);
    }
    return MUCON_OK;
}
