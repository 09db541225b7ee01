// The s-head's attention decoder as two persistent kernels (reference
// src/mucon/models.py:585-744: sequence_generation_forward + _calculate_attention).
//
// Per decoding step the reference runs ~30 small torch ops on 128-wide vectors (embedding, additive
// attention over the Tz encoder states, attn_combine, one LSTM cell, the transcript MLP, the length MLP,
// log-softmax, arg-max feedback): ~600 launches per training step, 11.5 ms of pure launch latency.  Here ONE
// workgroup (1024 threads) walks all steps: the recurrent state stays in LDS, weights stream from L2
// (~0.8 MB per step: what a step costs is this ONE CU's L1 path and its instruction issue), the memory [Tz x 2E] is read
// once per step; attention_l2's weight and (Tz permitting) the memory projection are LDS-resident; whatever does not feed
// the recurrence -- the transcript / length heads under teacher forcing, all of their backward -- runs outside the serial
// loops for all steps at once.
//   dec_memproj_kernel   mp = memory @ W1                                        (all Tz rows, many workgroups)
//   decoder_fwd_kernel   the step loop; saves every activation the backward needs
//   decoder_bwd_kernel   back-propagation through the steps; per-step "delta" vectors go to the workspace
//   dec_outer_kernel     every weight gradient = sum over steps of delta (x) input: one batched launch
//   dec_attn_grad_kernel d_mp and d_memory for all encoder states, from the per-step d_score / d_ctx the backward saved
// Vector width D = 128 (embedding = hidden = attention size: the reference's only configuration).
#pragma once
#include "common.hpp"

// -DDEC_TIMING (MUCON_HIPCC_FLAGS of mucon_amd/build.py): thread 0 stamps the phases of one decoding step and prints them
#ifdef DEC_TIMING
#define DEC_TICK(k) do { if (threadIdx.x == 0 && s == DEC_TIMING) s_tk[k] = clock64(); } while (0)
#else
#define DEC_TICK(k) do { } while (0)
#endif
constexpr int DEC_D = 128;
constexpr int DEC_THREADS = 1024;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int DEC_MAXNC = 128;   // transcript classes + 1 (EOS)
constexpr int DEC_MAXME = 256;   // memory width (2E)
constexpr int DEC_NL = 64;       // hidden width of the length MLP (D / 2)
constexpr int DEC_MAX_DYN_LDS = 134 * 1024;  // dynamic LDS decoder_fwd_kernel may ask for (160 KB - its static arrays, < 26 KB)
// decoder_fwd_kernel's dynamic LDS: attention_l2's weight, scores + weights, and the memory projection if it fits
static inline size_t dec_bwd_lds_bytes(int Tz) { return sizeof(float) * ((size_t)128 * 128 + (size_t)Tz); }
constexpr int DEC_MAX_DYN_LDS_BWD = 96 * 1024;   // decoder_bwd_kernel: 64 KB + 4 Tz bytes (its static arrays take 43 KB)
static inline size_t dec_fwd_lds_bytes(int Tz, int *mp_lds) {
    const size_t base = sizeof(float) * ((size_t)128 * 128 + 2 * (size_t)Tz), with_mp = base + sizeof(float) * (size_t)Tz * 128;
    *mp_lds = with_mp <= (size_t)DEC_MAX_DYN_LDS;
    return *mp_lds ? with_mp : base;
}

struct DecParams {  // torch layouts: Linear weight [out][in]
    const float *ho_w, *ho_b;    // fs_encoder_hidden_out  [D][ME]
    const float *co_w, *co_b;    // fs_encoder_cn_out      [D][ME]
    const float *w1;             // fs_decoder_attention_W1 [ME][D]
    const float *l2_w, *l2_b;    // fs_decoder_attention_l2 [D][D]
    const float *v;              // fs_decoder_attention_V  [D]
    const float *emb;            // fs_decoder_embedding    [n_emb][D]
    const float *cmb_w, *cmb_b;  // fs_decoder_attn_combine [D][D+ME]
    const float *w_ih, *w_hh, *b_ih, *b_hh;  // fs_decoder_lstm [4D][D] x2, [4D] x2
    const float *t1_w, *t1_b;    // fs_decoder_transcript[0] [D][D]
    const float *t2_w, *t2_b;    // fs_decoder_transcript[2] [NC][D]
    const float *n1_w, *n1_b;    // fs_decoder_length[0]     [D/2][D+NC]
    const float *n2_w, *n2_b;    // fs_decoder_length[2]     [1][D/2]
};
constexpr int DEC_NPARAMS = 23;

struct DecDims {
    int Tz, ME, NC, S, n_emb;
    int mp_lds;           // forward kernel: the memory projection [Tz][D] is kept in LDS (host: dec_fwd_lds_bytes)
    int teacher_forcing;  // 1: step input = tf_input[step]; 0: previous arg-max (tf_input[0] first)
    int stop_on_eos;      // 1: stop after the step whose arg-max is `eos` (evaluation without teacher forcing)
    int eos;
};

struct DecSaved {   // forward activations (workspace)
    float *mp;      // [Tz][D]       memory @ W1
    float *h, *c;   // [S+1][D]      row 0 = initial state
    float *q;       // [S][D]
    float *cat;     // [S][D+ME]     dropout(relu(emb)) | context
    float *attn;    // [S][Tz]
    float *mixed;   // [S][D]
    float *gates;   // [S][4D]       post-activation i,f,g,o
    float *t1;      // [S][D]
    float *lencat;  // [S][D+NC]     relu(cat(mixed, logits))
    float *l1;      // [S][D/2]
    int *toks;      // [S]           the token each step consumed
};
struct DecDeltas {  // backward: gradients at the pre-activations (workspace)
    float *ctx;     // [S][ME]   d context, per step   } consumed by dec_attn_grad_kernel, which turns them into
    float *score;   // [S][Tz]   d attention score     } d_memory and d_mp for all encoder states in parallel
    float *mp;      // [Tz][D]
    float *q, *mixed, *gates, *t1, *logits, *l1, *len;  // [S][...]
    float *h0, *c0; // [D]
};

// Wave-wide reductions on the VALU: DPP for the steps inside a row of 16 lanes, gfx950's v_permlane16_swap / v_permlane32_swap
// across rows.  (__shfl_xor is ds_bpermute: every step a trip through the LDS pipe, ~100 cycles of latency and 2 clocks of
// LDS issue per wave -- the 16 waves' matvec reductions of one decoding step were ~1,000 of them.)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {   // v of the lane DPP control CTRL names (all lanes valid for the controls used)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;
// lanes [32:63] of a <-> lanes [0:31] of b, then a + b: the lower half holds a[l] + a[l + 32], the upper half b[l] + b[l + 32]
__device__ __forceinline__ float swap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// odd rows of a <-> even rows of b, then a + b: even rows hold a[l] + a[l + 16], odd rows b[l - 16] + b[l]
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_sum(float v) {   // the same bits in every lane
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    v = swap16_add(v, v);
    v = swap32_add(v, v);
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));   // (the builtin is an int one)
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));   // (the builtin is an int one)
}
__device__ __forceinline__ float block_sum(int tix, float v, float *red) {  // red: DEC_WAVES + 1 floats of LDS
    v = wave_sum(v);
    __syncthreads();
    if ((tix & 63) == 0) red[tix >> 6] = v;
    __syncthreads();
    if (tix < 64) {
        float t = tix < DEC_WAVES ? red[tix] : 0.f;
        t = wave_sum(t);
        if (tix == 0) red[DEC_WAVES] = t;
    }
    __syncthreads();
    return red[DEC_WAVES];
}
// Sum R (8 or 4) per-lane values across the wave with 10 (7) shuffles instead of R x 6: after the call, the lanes
// with (lane >> 3) & 7 == r  (R = 8)  or  (lane >> 4) & 3 == r  (R = 4)  hold the wave-wide sum of v[r].
template <int R>
__device__ __forceinline__ float wave_sum_rows(int lane, float *v) {
    float c;
    if (R == 8) {
        float a[4], b2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = swap32_add(v[i], v[i + 4]);
#pragma unroll
        for (int i = 0; i < 2; ++i) b2[i] = swap16_add(a[i], a[i + 2]);
        const bool h3 = lane & 8;
        c = (h3 ? b2[1] : b2[0]) + dpp_f<DPP_ROR8>(h3 ? b2[0] : b2[1]);
    } else {
        float a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = swap32_add(v[i], v[i + 2]);
        c = swap16_add(a[0], a[1]);
        c += dpp_f<DPP_ROR8>(c);
    }
    c += dpp_f<DPP_HALF_MIRROR>(c);
    c += dpp_f<DPP_XOR2>(c);
    c += dpp_f<DPP_XOR1>(c);
    return c;
}

// out[r] = act(b[r] + W[r][:] . x  [+ b2[r] + W2[r][:] . x2]) for r < rows: R rows per wave at a time, lanes across the
// columns -- all of an iteration's loads (R rows x COLS/64, both matrices) are issued before the first use, so an
// iteration costs about one L2 round trip -- then one tree reduction for the R rows.  COLS is a multiple of 64, or 0 for
// a run-time column count (`cols`, one round trip per 64 columns).
template <int ACT, int R, int COLS, bool DUAL>
__device__ __forceinline__ void matvec_rows(int tix, const float *__restrict__ W, const float *__restrict__ b, int rows, int cols,
                                            const float *x, float *out, const float *__restrict__ W2 = nullptr,
                                            const float *__restrict__ b2 = nullptr, const float *x2 = nullptr) {
    const int lane = tix & 63, wave = tix >> 6;
    const int nc = COLS ? COLS : cols;
    for (int r0 = wave * R; r0 < rows; r0 += DEC_WAVES * R) {
        float acc[R];
        if (COLS) {
            constexpr int NJ = COLS ? COLS / 64 : 1;
            float w[R][NJ], w2[R][DUAL ? NJ : 1];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = min(r0 + i, rows - 1);
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    w[i][jj] = W[(long)r * COLS + jj * 64 + lane];
                    if (DUAL) w2[i][jj] = W2[(long)r * COLS + jj * 64 + lane];
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                float a = 0.f;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    a += w[i][jj] * x[jj * 64 + lane];
                    if (DUAL) a += w2[i][jj] * x2[jj * 64 + lane];
                }
                acc[i] = a;
            }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] = 0.f;
            for (int j = lane; j < nc; j += 64) {
                const float xv = x[j];
#pragma unroll
                for (int i = 0; i < R; ++i) acc[i] += W[(long)min(r0 + i, rows - 1) * nc + j] * xv;
            }
        }
        const float sum = wave_sum_rows<R>(lane, acc);
        const int r = r0 + (R == 8 ? (lane >> 3) & 7 : (lane >> 4) & 3);
        if ((lane & (R == 8 ? 7 : 15)) == 0 && r < rows) {
            float v = sum + (b ? b[r] : 0.f);
            if (DUAL && b2) v += b2[r];
            out[r] = ACT ? fmaxf(v, 0.f) : v;
        }
    }
}

// out[j] = (ACC ? out[j] : 0) + sum_i W[i][j] d[i] for j < cols (W^T d): thread groups split the rows,
// lanes run along a row (coalesced), partial sums meet in `scratch` (DEC_THREADS floats).  Two barriers
// inside; the caller synchronises before reading `out`.
template <bool ACC>
__device__ __forceinline__ void matvec_cols(int tix, const float *__restrict__ W, int rows, int cols, const float *d, float *out,
                                            float *scratch) {
    const int cp = cols <= 128 ? 128 : cols <= 256 ? 256 : 512;
    const int ng = DEC_THREADS / cp;
    const int g = tix / cp, j = tix - g * cp;
    float acc = 0.f;
    if (j < cols) {
#pragma unroll 8
        for (int i = g; i < rows; i += ng) acc += W[(long)i * cols + j] * d[i];
    }
    __syncthreads();  // scratch may still be read by the previous user
    scratch[tix] = acc;
    __syncthreads();
    if (tix < cols) {
        float s = 0.f;
        for (int gg = 0; gg < ng; ++gg) s += scratch[gg * cp + tix];
        out[tix] = ACC ? out[tix] + s : s;
    }
}

// The same for a column count that is a multiple of 4: a thread owns FOUR adjacent columns (one 16-byte load per row), so
// cols/4 threads cover a row and the 1024 threads form up to 32 row groups -- 512 x 128 weights are 16 dependent steps per
// thread instead of 64.  PAIR: two matrices with the same d in one pass (out_a += Wa^T d, out_b = Wb^T d).
// scratch: groups * cols (* 2) floats -- DEC_SCR floats cover every use below.
constexpr int DEC_SCR = 2 * 32 * DEC_D;
template <bool ACC, bool PAIR, int UNR = 8>
__device__ __forceinline__ void matvec_cols4(int tix, const float *__restrict__ Wa, const float *__restrict__ Wb, int rows, int cols,
                                             const float *d, float *out_a, float *out_b, float *scratch) {
    const int tpg = cols >> 2;                 // threads per row group
    const int ng = DEC_THREADS / tpg;          // row groups
    const int g = tix / tpg, j4 = tix - g * tpg;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (g < ng) {
#pragma unroll UNR
        for (int i = g; i < rows; i += ng) {
            const float dv = d[i];
            const f32x4 wa = *reinterpret_cast<const f32x4 *>(Wa + (long)i * cols + 4 * j4);
            a[0] += wa[0] * dv;
            a[1] += wa[1] * dv;
            a[2] += wa[2] * dv;
            a[3] += wa[3] * dv;
            if (PAIR) {
                const f32x4 wb = *reinterpret_cast<const f32x4 *>(Wb + (long)i * cols + 4 * j4);
                b[0] += wb[0] * dv;
                b[1] += wb[1] * dv;
                b[2] += wb[2] * dv;
                b[3] += wb[3] * dv;
            }
        }
    }
    __syncthreads();  // scratch may still be read by the previous user
    if (g < ng) {
        *reinterpret_cast<f32x4 *>(scratch + g * cols + 4 * j4) = a;
        if (PAIR) *reinterpret_cast<f32x4 *>(scratch + ng * cols + g * cols + 4 * j4) = b;
    }
    __syncthreads();
    const int nout = PAIR ? 2 * cols : cols;
    if (tix < nout) {
        const int which = tix >= cols, jj = tix - which * cols;
        float sum = 0.f;
#pragma unroll 8
        for (int gg = 0; gg < ng; ++gg) sum += scratch[which * ng * cols + gg * cols + jj];   // (8 LDS reads in flight)
        if (which == 0) out_a[jj] = ACC ? out_a[jj] + sum : sum;
        else out_b[jj] = sum;
    }
}

// The heads of the decoder do not feed the recurrence when the step inputs are given (teacher forcing) and never do in the
// backward pass, so they leave the serial loop: the two functions below run one weight matrix against the vectors of ALL
// steps at once -- the matrix is read once per launch instead of once per step, and a phase (one L2 round trip + a barrier)
// is paid once instead of S times.
//
// OUT[s][r] = act(b[r] + W[r][:] . X[s][:]) for r < rows, s < S: a wave keeps R rows of W in registers (cols <= 64 NJ) and
// walks the steps; the per-row lane order of the sum is matvec_rows' (same bits as the in-loop path).
template <int ACT, int R, int NJ>
__device__ __forceinline__ void matvec_rows_steps(int tix, const float *__restrict__ W, const float *__restrict__ b, int rows, int cols,
                                                  const float *__restrict__ X, int ldx, int S, float *__restrict__ OUT, int ldo) {
    const int lane = tix & 63, wave = tix >> 6;
    for (int r0 = wave * R; r0 < rows; r0 += DEC_WAVES * R) {
        float w[R][NJ];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = min(r0 + i, rows - 1);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int c = jj * 64 + lane;
                w[i][jj] = c < cols ? W[(long)r * cols + c] : 0.f;
            }
        }
        const int rr = r0 + (R == 8 ? (lane >> 3) & 7 : (lane >> 4) & 3);
        const bool writer = (lane & (R == 8 ? 7 : 15)) == 0 && rr < rows;
        const float bias = (b && rr < rows) ? b[rr] : 0.f;
#pragma unroll 4
        for (int s = 0; s < S; ++s) {
            float x[NJ], acc[R];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int c = jj * 64 + lane;
                x[jj] = c < cols ? X[(long)s * ldx + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                float a = 0.f;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) a += w[i][jj] * x[jj];
                acc[i] = a;
            }
            const float v = wave_sum_rows<R>(lane, acc) + bias;
            if (writer) OUT[(long)s * ldo + rr] = ACT ? fmaxf(v, 0.f) : v;
        }
    }
}

// epi(s, j, sum_i W[i][j] D[s][i]) for s < ns <= DEC_SB, j < cols <= 256, rows <= DEC_D: the ns vectors are staged in LDS
// (sD: DEC_SB x DEC_D floats), a thread owns one column of a row group and DEC_SB accumulators, the row groups meet in
// `scratch` (DEC_SCR floats) and are summed in group order.  Barriers inside (the first one covers whoever wrote D).
constexpr int DEC_SB = 8;
template <typename Epi>
__device__ __forceinline__ void matvec_cols_steps(int tix, const float *__restrict__ W, int rows, int cols, const float *D, int ldd, int ns,
                                                  float *sD, float *scratch, Epi epi) {
    const int tid = tix;
    __syncthreads();
    {
        const int s = tid >> 7, i = tid & 127;   // DEC_SB * DEC_D == DEC_THREADS
        sD[tid] = (s < ns && i < rows) ? D[(long)s * ldd + i] : 0.f;
    }
    __syncthreads();
    const int cp = cols <= 128 ? 128 : 256, ng = DEC_THREADS / cp;
    const int g = tid / cp, j = tid - g * cp;
    float acc[DEC_SB];
#pragma unroll
    for (int s = 0; s < DEC_SB; ++s) acc[s] = 0.f;
    if (j < cols) {
#pragma unroll 4
        for (int i = g; i < rows; i += ng) {
            const float w = W[(long)i * cols + j];
#pragma unroll
            for (int s = 0; s < DEC_SB; ++s) acc[s] += w * sD[s * DEC_D + i];
        }
    }
#pragma unroll
    for (int s = 0; s < DEC_SB; ++s) scratch[(g * DEC_SB + s) * cp + j] = acc[s];
    __syncthreads();
    for (int e = tid; e < ns * cols; e += DEC_THREADS) {
        const int s = e / cols, jj = e - s * cols;
        float sum = 0.f;
        for (int gg = 0; gg < ng; ++gg) sum += scratch[(gg * DEC_SB + s) * cp + jj];
        epi(s, jj, sum);
    }
}
static_assert(DEC_SB * DEC_D == DEC_THREADS && DEC_SB * DEC_THREADS <= DEC_SCR, "matvec_cols_steps staging");

// mp[t][k] = sum_j memory[t][j] W1[j][k]; grid (ceil(Tz/4)), 512 threads = 4 quarters of j x 128 columns, every thread all 4 rows.
// (A thread per (row, column) walking all ME values of j was a chain of dependent L2 round trips: 11 us at Tz = 125.  Here a
// thread loads ME/4 weights, 16 in flight, each used for four rows, and the quarters meet in LDS in order.)
// (r4: also clears what the step kernel behind it expects cleared: `zero_words` 8-byte words at `zero` -- the exchange granules of
// decoder_fwd_mw_kernel and its step counter -- so that no stale tag can be met)
__global__ __launch_bounds__(512) void dec_memproj_kernel(const float *memory, const float *w1, float *mp, int Tz, int ME,
                                                          unsigned long long *zero, int zero_words, int *zero_int) {
    __shared__ float ms[4][DEC_MAXME];
    for (int e = blockIdx.x * 512 + threadIdx.x; e < zero_words; e += gridDim.x * 512) zero[e] = 0ull;
    if (zero_int && blockIdx.x == 0 && threadIdx.x == 0) *zero_int = 0;
    __shared__ float part[4][4][DEC_D];
    const int t0 = blockIdx.x * 4;
    for (int e = threadIdx.x; e < 4 * ME; e += 512) {
        const int t = t0 + e / ME;
        ms[e / ME][e % ME] = t < Tz ? memory[(long)t * ME + e % ME] : 0.f;
    }
    __syncthreads();
    const int qj = threadIdx.x >> 7, k = threadIdx.x & 127;
    const int jq = (ME + 3) / 4, j0 = qj * jq, j1 = min(ME, j0 + jq);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int jb = j0; jb < j1; jb += 16) {
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = w1[(long)min(jb + i, j1 - 1) * DEC_D + k];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (jb + i < j1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += ms[r][jb + i] * w[i];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[qj][r][k] = acc[r];
    __syncthreads();
    const int r = threadIdx.x >> 7;
    if (t0 + r < Tz) mp[(long)(t0 + r) * DEC_D + k] = (part[0][r][k] + part[1][r][k]) + (part[2][r][k] + part[3][r][k]);
}

// dynamic LDS: dec_fwd_lds_bytes()
__global__ __launch_bounds__(DEC_THREADS) void decoder_fwd_kernel(DecDims dm, DecParams p, DecSaved sv, const float *memory,
                                                                  const float *hn, const float *cn, const long *tf_input,
                                                                  const float *dropmask, float *logp_out, float *len_out,
                                                                  int *nsteps_out) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float *s_l2w = s_dyn;                         // [D][D] attention_l2's weight, resident for all steps
    float *s_score = s_dyn + DEC_D * DEC_D;       // [Tz] scores, [Tz] attention weights
    float *s_attn = s_score + dm.Tz;
    float *s_mp = s_attn + dm.Tz;                 // [Tz][D] the memory projection, when dm.mp_lds says it fits
    __shared__ float s_h[DEC_D], s_c[DEC_D], s_q[DEC_D], s_cat[DEC_D + DEC_MAXME], s_mixed[DEC_D], s_gates[4 * DEC_D];
    __shared__ float s_t1[DEC_D], s_logits[DEC_MAXNC], s_lencat[DEC_D + DEC_MAXNC], s_l1[DEC_NL];
    __shared__ float s_hc[2 * DEC_MAXME];
    __shared__ __attribute__((aligned(16))) float s_part[DEC_WAVES * DEC_MAXME];
    __shared__ int s_tok, s_stop;
#ifdef DEC_TIMING
    __shared__ long long s_tk[16];
#endif
    const int tid0 = threadIdx.x, tid = tid0, lane = tid & 63, wave = tid >> 6;
    const int Tz = dm.Tz, ME = dm.ME, NC = dm.NC, CW = DEC_D + ME, LW = DEC_D + NC;

    // initial state: dec_h = hidden_out(h_n.view(1,-1)), dec_c = cn_out(c_n.view(1,-1))   (models.py:612-617)
    if (tid < ME) {
        s_hc[tid] = hn[tid];
        s_hc[DEC_MAXME + tid] = cn[tid];
    }
    if (tid == 0) {
        s_tok = (int)tf_input[0];
        s_stop = 0;
    }
    __syncthreads();
    matvec_rows<0, 4, 0, false>(tid, p.ho_w, p.ho_b, DEC_D, ME, s_hc, s_h);
    matvec_rows<0, 4, 0, false>(tid, p.co_w, p.co_b, DEC_D, ME, s_hc + DEC_MAXME, s_c);
    __syncthreads();
    if (tid < DEC_D) {
        sv.h[tid] = s_h[tid];
        sv.c[tid] = s_c[tid];
    }
    // Residents of the step loop.  One CU streams ~45-64 B/clk from L2 and the ~1 MB of weights a step touches are what a
    // step costs: attention_l2's weight (needed first in every step) and, Tz permitting, the memory projection stay in LDS.
    // (W_hh in registers -- 64 VGPRs per lane -- was tried: with 128 VGPRs per lane at 16 waves the streams of the other
    // phases then spill load by load behind vmcnt(0) waits.)
    const int sub = (lane >> 3) & 7;              // the row (of 8) whose sum wave_sum_rows<8> leaves in this lane
    const float l2b = p.l2_b[wave * 8 + sub];
    for (int e = tid; e < DEC_D * DEC_D / 4; e += DEC_THREADS)
        reinterpret_cast<f32x4 *>(s_l2w)[e] = reinterpret_cast<const f32x4 *>(p.l2_w)[e];
    if (dm.mp_lds)
        for (int e = tid; e < Tz * DEC_D / 4; e += DEC_THREADS) reinterpret_cast<f32x4 *>(s_mp)[e] = reinterpret_cast<const f32x4 *>(sv.mp)[e];
    __syncthreads();
    // given step inputs and no early stop: nothing of the transcript / length heads feeds back, they run after the loop for
    // all steps at once
    const bool defer_heads = dm.teacher_forcing && !dm.stop_on_eos;
    int s = 0;
    for (; s < dm.S; ++s) {
        DEC_TICK(0);
        // the thread index, opaque per step: LLVM otherwise hoists every per-lane row address of every matrix out of the loop
        // (~100 VGPRs of loop invariants), spills them and reloads one in front of each phase's loads
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        int tok = dm.teacher_forcing ? (int)tf_input[s] : s_tok;
        tok = tok < 0 ? 0 : tok >= dm.n_emb ? dm.n_emb - 1 : tok;  // host validated; keeps a bad arg-max in range
        // q = attention_l2(dec_h): weight and state in LDS, no global access in this phase
        {
            const float h0 = s_h[lane], h1 = s_h[lane + 64];
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = s_l2w[(wave * 8 + i) * DEC_D + lane] * h0 + s_l2w[(wave * 8 + i) * DEC_D + 64 + lane] * h1;
            const float sum = wave_sum_rows<8>(lane, acc);
            if ((lane & 7) == 0) s_q[wave * 8 + sub] = sum + l2b;
        }
        __syncthreads();
        DEC_TICK(1);
        if (tid < DEC_D) sv.q[s * DEC_D + tid] = s_q[tid];
        // embedded = dropout(relu(embedding(input))): the last wave requests it now and parks it in LDS two barriers later
        // (attn_combine is its first reader) -- no phase waits for this L2 round trip
        float e0 = 0.f, e1 = 0.f;
        // score[t] = V . tanh(mp[t] + q): 8 encoder states per wave at a time (16 loads in flight)
        {
            if (wave == DEC_WAVES - 1) {
                e0 = fmaxf(p.emb[(long)tok * DEC_D + lane], 0.f);
                e1 = fmaxf(p.emb[(long)tok * DEC_D + 64 + lane], 0.f);
                if (dropmask) {
                    e0 *= dropmask[s * DEC_D + lane];
                    e1 *= dropmask[s * DEC_D + 64 + lane];
                }
                if (lane == 0) sv.toks[s] = tok;
            }
            const float q0 = s_q[lane], q1 = s_q[lane + 64], v0 = p.v[lane], v1 = p.v[lane + 64];
            for (int t0 = wave * 8; t0 < Tz; t0 += DEC_WAVES * 8) {
                float m0[8], m1[8], a[8];
                if (dm.mp_lds) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float *m = s_mp + min(t0 + i, Tz - 1) * DEC_D;
                        m0[i] = m[lane];
                        m1[i] = m[lane + 64];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float *m = sv.mp + (long)min(t0 + i, Tz - 1) * DEC_D;
                        m0[i] = m[lane];
                        m1[i] = m[lane + 64];
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = v0 * tanh_f(m0[i] + q0) + v1 * tanh_f(m1[i] + q1);
                const float sum = wave_sum_rows<8>(lane, a);
                const int t = t0 + ((lane >> 3) & 7);
                if ((lane & 7) == 0 && t < Tz) s_score[t] = sum;
            }
        }
        __syncthreads();
        DEC_TICK(2);
        // attention weights = softmax(score): every wave reduces ALL scores itself (same order, same bits in every wave) --
        // no cross-wave reduction, no barrier until the weights are complete
        {
            float mx = -INFINITY;
            for (int t = lane; t < Tz; t += 64) mx = fmaxf(mx, s_score[t]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int t = lane; t < Tz; t += 64) sum += expf(s_score[t] - mx);
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            for (int t = tid; t < Tz; t += DEC_THREADS) {
                const float a = expf(s_score[t] - mx) * inv;
                s_attn[t] = a;
                sv.attn[(long)s * Tz + t] = a;
            }
        }
        __syncthreads();
        DEC_TICK(3);
        if (wave == DEC_WAVES - 1) {
            s_cat[lane] = e0;
            s_cat[lane + 64] = e1;
        }
        // context = sum_t attn[t] memory[t]: wave w takes the states t = w (mod 16), a lane four adjacent columns, 8 rows in
        // flight at a time (one round trip for Tz <= 128); the 16 partial rows meet in LDS
        if ((ME & 3) == 0) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (4 * lane < ME) {
                for (int tb = wave; tb < Tz; tb += 8 * DEC_WAVES) {
                    f32x4 m[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        m[i] = *reinterpret_cast<const f32x4 *>(memory + (long)min(tb + DEC_WAVES * i, Tz - 1) * ME + 4 * lane);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = tb + DEC_WAVES * i < Tz ? s_attn[tb + DEC_WAVES * i] : 0.f;
                        acc[0] += a * m[i][0];
                        acc[1] += a * m[i][1];
                        acc[2] += a * m[i][2];
                        acc[3] += a * m[i][3];
                    }
                }
            }
            *reinterpret_cast<f32x4 *>(&s_part[wave * DEC_MAXME + 4 * lane]) = acc;
            __syncthreads();
            if (tid < ME) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < DEC_WAVES; ++w) v += s_part[w * DEC_MAXME + tid];
                s_cat[DEC_D + tid] = v;
            }
        } else {
            const int gq = tid >> 8, j = tid & 255;
            float acc = 0.f;
            if (j < ME) {
#pragma unroll 8
                for (int t = gq; t < Tz; t += 4) acc += s_attn[t] * memory[(long)t * ME + j];
            }
            s_part[tid] = acc;
            __syncthreads();
            if (tid < ME) s_cat[DEC_D + tid] = (s_part[tid] + s_part[256 + tid]) + (s_part[512 + tid] + s_part[768 + tid]);
        }
        __syncthreads();
        DEC_TICK(4);
        if (tid < CW) sv.cat[(long)s * CW + tid] = s_cat[tid];
        // mixed = relu(attn_combine(cat(embedded, context))): all 128 rows in ONE round trip (8 rows per wave, 6 loads per row
        // and lane in flight for the bidirectional encoder)
        if (ME == 256) matvec_rows<1, 8, DEC_D + 256, false>(tid, p.cmb_w, p.cmb_b, DEC_D, CW, s_cat, s_mixed);
        else if (ME == 128) matvec_rows<1, 8, DEC_D + 128, false>(tid, p.cmb_w, p.cmb_b, DEC_D, CW, s_cat, s_mixed);
        else matvec_rows<1, 4, 0, false>(tid, p.cmb_w, p.cmb_b, DEC_D, CW, s_cat, s_mixed);
        __syncthreads();
        DEC_TICK(5);
        // one LSTM cell: 512 KB of weights, the step's big stream.  A wave owns gate rows 32w..32w+31, eight at a time, and
        // requests the next eight rows (32 loads per lane) before it reduces the current ones.
        {
            const float x0 = s_mixed[lane], x1 = s_mixed[lane + 64], h0 = s_h[lane], h1 = s_h[lane + 64];
            const float *wi = p.w_ih + (long)(wave * 32) * DEC_D + lane, *wh = p.w_hh + (long)(wave * 32) * DEC_D + lane;
            float w[2][8][4];
            auto request = [&](int c, float (&d)[8][4]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    d[i][0] = wi[(c * 8 + i) * DEC_D];
                    d[i][1] = wi[(c * 8 + i) * DEC_D + 64];
                    d[i][2] = wh[(c * 8 + i) * DEC_D];
                    d[i][3] = wh[(c * 8 + i) * DEC_D + 64];
                }
            };
            request(0, w[0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int r = wave * 32 + c * 8 + sub;
                const float bias_i = p.b_ih[r], bias_h = p.b_hh[r];
                if (c < 3) request(c + 1, w[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // two batches in flight, not three: 128 VGPRs
                float acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {   // matvec_rows' order: W_ih x and W_hh h interleaved per 64-column half
                    float a = 0.f;
                    a += w[c & 1][i][0] * x0;
                    a += w[c & 1][i][2] * h0;
                    a += w[c & 1][i][1] * x1;
                    a += w[c & 1][i][3] * h1;
                    acc[i] = a;
                }
                const float sum = wave_sum_rows<8>(lane, acc);
                if ((lane & 7) == 0) s_gates[r] = (sum + bias_i) + bias_h;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        DEC_TICK(6);
        if (tid < DEC_D) {
            const float gi = sigmoid_f(s_gates[tid]), gf = sigmoid_f(s_gates[DEC_D + tid]);
            const float gg = tanh_f(s_gates[2 * DEC_D + tid]), go = sigmoid_f(s_gates[3 * DEC_D + tid]);
            const float c = gf * s_c[tid] + gi * gg;
            const float h = go * tanh_f(c);
            s_c[tid] = c;
            s_h[tid] = h;
            float *gs = sv.gates + (long)s * 4 * DEC_D;
            gs[tid] = gi;
            gs[DEC_D + tid] = gf;
            gs[2 * DEC_D + tid] = gg;
            gs[3 * DEC_D + tid] = go;
            sv.c[(s + 1) * DEC_D + tid] = c;
            sv.h[(s + 1) * DEC_D + tid] = h;
            sv.mixed[s * DEC_D + tid] = s_mixed[tid];
        }
        __syncthreads();
        DEC_TICK(7);
        if (defer_heads) continue;
        // word logits = transcript MLP(dec_out)
        matvec_rows<1, 8, DEC_D, false>(tid, p.t1_w, p.t1_b, DEC_D, DEC_D, s_h, s_t1);
        __syncthreads();
        if (tid < DEC_D) sv.t1[s * DEC_D + tid] = s_t1[tid];
        matvec_rows<0, 4, DEC_D, false>(tid, p.t2_w, p.t2_b, NC, DEC_D, s_t1, s_logits);
        __syncthreads();
        // length = length MLP(relu(cat(mixed, word logits)))
        if (tid < LW) {
            const float v = tid < DEC_D ? s_mixed[tid] : fmaxf(s_logits[tid - DEC_D], 0.f);
            s_lencat[tid] = v;
            sv.lencat[(long)s * LW + tid] = v;
        }
        __syncthreads();
        matvec_rows<1, 4, 0, false>(tid, p.n1_w, p.n1_b, DEC_NL, LW, s_lencat, s_l1);
        __syncthreads();
        if (wave == 0) {
            sv.l1[s * DEC_NL + lane] = s_l1[lane];
            const float a = wave_sum(p.n2_w[lane] * s_l1[lane]);
            if (lane == 0) len_out[s] = a + p.n2_b[0];
        } else if (wave == 1) {  // log-softmax + arg-max (lowest index on ties)
            const float x0 = lane < NC ? s_logits[lane] : -INFINITY, x1 = lane + 64 < NC ? s_logits[lane + 64] : -INFINITY;
            const float mx = wave_max(fmaxf(x0, x1));
            const float se = wave_sum((lane < NC ? expf(x0 - mx) : 0.f) + (lane + 64 < NC ? expf(x1 - mx) : 0.f));
            const float lse = mx + logf(se);
            if (lane < NC) logp_out[(long)s * NC + lane] = x0 - lse;
            if (lane + 64 < NC) logp_out[(long)s * NC + lane + 64] = x1 - lse;
            int cand = x0 == mx ? lane : x1 == mx ? lane + 64 : 1 << 20;
#pragma unroll
            for (int o = 32; o; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
            if (lane == 0) {
                s_tok = cand;
                if (dm.stop_on_eos && cand == dm.eos) s_stop = 1;
            }
        }
        __syncthreads();
        if (s_stop) {
            ++s;
            break;
        }
    }
    if (tid == 0) *nsteps_out = s;
#ifdef DEC_TIMING
    if (tid == 0) {
        s_tk[8] = clock64();
        printf("decoder_fwd step %d: q %lld  score %lld  softmax %lld  context %lld  combine %lld  lstm %lld  cell %lld   (cycles)\n", DEC_TIMING,
               s_tk[1] - s_tk[0], s_tk[2] - s_tk[1], s_tk[3] - s_tk[2], s_tk[4] - s_tk[3], s_tk[5] - s_tk[4], s_tk[6] - s_tk[5], s_tk[7] - s_tk[6]);
    }
#endif
    if (!defer_heads) return;
    const int S = dm.S;
    matvec_rows_steps<1, 8, 2>(tid, p.t1_w, p.t1_b, DEC_D, DEC_D, sv.h + DEC_D, DEC_D, S, sv.t1, DEC_D);
    __syncthreads();
    matvec_rows_steps<0, 4, 2>(tid, p.t2_w, p.t2_b, NC, DEC_D, sv.t1, DEC_D, S, logp_out, NC);   // the logits, for now
    __syncthreads();
    for (int e = tid; e < S * LW; e += DEC_THREADS) {
        const int st = e / LW, k = e - st * LW;
        sv.lencat[e] = k < DEC_D ? sv.mixed[st * DEC_D + k] : fmaxf(logp_out[(long)st * NC + k - DEC_D], 0.f);
    }
    __syncthreads();
    matvec_rows_steps<1, 4, 4>(tid, p.n1_w, p.n1_b, DEC_NL, LW, sv.lencat, LW, S, sv.l1, DEC_NL);
    for (int st = wave; st < S; st += DEC_WAVES) {   // log-softmax in place
        float *row = logp_out + (long)st * NC;
        const float x0 = lane < NC ? row[lane] : -INFINITY, x1 = lane + 64 < NC ? row[lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(x0, x1));
        const float se = wave_sum((lane < NC ? expf(x0 - mx) : 0.f) + (lane + 64 < NC ? expf(x1 - mx) : 0.f));
        const float lse = mx + logf(se);
        if (lane < NC) row[lane] = x0 - lse;
        if (lane + 64 < NC) row[lane + 64] = x1 - lse;
    }
    __syncthreads();
    for (int st = wave; st < S; st += DEC_WAVES) {
        const float a = wave_sum(p.n2_w[lane] * sv.l1[st * DEC_NL + lane]);
        if (lane == 0) len_out[st] = a + p.n2_b[0];
    }
#ifdef DEC_TIMING
    if (tid == 0) printf("decoder_fwd heads for %d steps: %lld cycles\n", S, (long long)clock64() - s_tk[8]);
#endif
}

// dynamic LDS: dec_bwd_lds_bytes().  dm.S = the number of steps the forward ran.  d_logp [S][NC] / d_len [S] may be null.
// d_emb [n_emb][D] is zeroed here; d_v [D]; d_hn / d_cn [ME]; d_memory and dl.mp are written by dec_attn_grad_kernel.
__global__ __launch_bounds__(DEC_THREADS) void decoder_bwd_kernel(DecDims dm, DecParams p, DecSaved sv, DecDeltas dl,
                                                                  const float *memory, const float *logp, const float *d_logp,
                                                                  const float *d_len, const float *dropmask, float *d_memory,
                                                                  float *d_emb, float *d_v, float *d_hn, float *d_cn) {
    extern __shared__ __attribute__((aligned(16))) float s_dynb[];
    float *s_l2w = s_dynb;                   // [D][D] attention_l2's weight, resident: the last matrix of every step's chain
    float *s_ds = s_dynb + DEC_D * DEC_D;    // [Tz] d_attn, then d_score
    __shared__ float s_dh[DEC_D], s_dc[DEC_D];
    __shared__ float s_dgates[4 * DEC_D], s_dmixed[DEC_D], s_dcat[DEC_D + DEC_MAXME], s_dq[DEC_D];
    __shared__ __attribute__((aligned(16))) float s_scr[DEC_SCR];
    __shared__ float s_red[DEC_WAVES + 1], s_out[DEC_MAXME], s_sd[DEC_SB * DEC_D];
#ifdef DEC_TIMING
    __shared__ long long s_tk[16];
#endif
    const int tid0 = threadIdx.x, tid = tid0, lane = tid & 63, wave = tid >> 6;
    const int Tz = dm.Tz, ME = dm.ME, NC = dm.NC, CW = DEC_D + ME, LW = DEC_D + NC;

    for (long e = tid; e < (long)dm.n_emb * DEC_D; e += DEC_THREADS) d_emb[e] = 0.f;
    for (int e = tid; e < DEC_D * DEC_D / 4; e += DEC_THREADS)
        reinterpret_cast<f32x4 *>(s_l2w)[e] = reinterpret_cast<const f32x4 *>(p.l2_w)[e];
    if (tid < DEC_D) {
        s_dh[tid] = 0.f;
        s_dc[tid] = 0.f;
    }
    float dv_acc = 0.f;  // thread (g = tid >> 7, k = tid & 127): partial dV[k]
    __syncthreads();

    // The heads first, for all steps (DEC_SB at a time): d_logits, d_t1 and the length MLP's deltas go to the workspace as
    // before; what reaches the recurrence -- d dec_out from the transcript MLP, d mixed from the length MLP -- waits in
    // dl.q[s] / dl.mixed[s], which the loop below overwrites with their final contents at step s after reading them.
    for (int s0 = 0; s0 < dm.S; s0 += DEC_SB) {
        const int ns = min(DEC_SB, dm.S - s0);
        if (wave < DEC_SB) {            // log-softmax backward
            const int st = s0 + wave;
            if (wave < ns) {
                const float g0 = (d_logp && lane < NC) ? d_logp[(long)st * NC + lane] : 0.f;
                const float g1 = (d_logp && lane + 64 < NC) ? d_logp[(long)st * NC + lane + 64] : 0.f;
                const float tot = wave_sum(g0 + g1);
                if (lane < NC) dl.logits[(long)st * NC + lane] = g0 - expf(logp[(long)st * NC + lane]) * tot;
                if (lane + 64 < NC) dl.logits[(long)st * NC + lane + 64] = g1 - expf(logp[(long)st * NC + lane + 64]) * tot;
            }
        } else {                        // length MLP output layer backward
            const int st = s0 + wave - DEC_SB;
            if (wave - DEC_SB < ns) {
                const float dlen = d_len ? d_len[st] : 0.f;
                dl.l1[st * DEC_NL + lane] = sv.l1[st * DEC_NL + lane] > 0.f ? dlen * p.n2_w[lane] : 0.f;
                if (lane == 0) dl.len[st] = dlen;
            }
        }
        matvec_cols_steps(tid, p.n1_w, DEC_NL, LW, dl.l1 + s0 * DEC_NL, DEC_NL, ns, s_sd, s_scr, [&](int st, int j, float sum) {
            st += s0;
            const float v = sv.lencat[(long)st * LW + j] > 0.f ? sum : 0.f;
            if (j < DEC_D) dl.mixed[st * DEC_D + j] = v;
            else dl.logits[(long)st * NC + j - DEC_D] += v;
        });
        // transcript MLP backward -> d dec_out
        matvec_cols_steps(tid, p.t2_w, NC, DEC_D, dl.logits + (long)s0 * NC, NC, ns, s_sd, s_scr, [&](int st, int j, float sum) {
            st += s0;
            dl.t1[st * DEC_D + j] = sv.t1[st * DEC_D + j] > 0.f ? sum : 0.f;
        });
        matvec_cols_steps(tid, p.t1_w, DEC_D, DEC_D, dl.t1 + s0 * DEC_D, DEC_D, ns, s_sd, s_scr,
                          [&](int st, int j, float sum) { dl.q[(s0 + st) * DEC_D + j] = sum; });
    }
    __syncthreads();

    // what the cell backward of a step needs from the workspace -- gates, cell states, the heads' shares parked above --
    // travels one step ahead of the loop: eight values per thread (tid < D), requested behind the previous cell backward's barrier
    float nx[8];
    auto fetch_step = [&](int st, int t) {
        const float *gs = sv.gates + (long)st * 4 * DEC_D;
        nx[0] = gs[t];
        nx[1] = gs[DEC_D + t];
        nx[2] = gs[2 * DEC_D + t];
        nx[3] = gs[3 * DEC_D + t];
        nx[4] = sv.c[(st + 1) * DEC_D + t];
        nx[5] = sv.c[st * DEC_D + t];
        nx[6] = dl.q[st * DEC_D + t];
        nx[7] = dl.mixed[st * DEC_D + t];
    };
    if (tid < DEC_D) fetch_step(dm.S - 1, tid);
    for (int s = dm.S - 1; s >= 0; --s) {
        int tid = tid0;     // opaque per step: see decoder_fwd_kernel
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        DEC_TICK(0);
        // LSTM cell backward (its saved activations were requested a step ago)
        if (tid < DEC_D) {
            const float gi = nx[0], gf = nx[1], gg = nx[2], go = nx[3], ct = nx[4], cp = nx[5];
            const float dh = s_dh[tid] + nx[6], th = tanh_f(ct);   // recurrent + transcript head
            s_dmixed[tid] = nx[7];                                  // the length head's share
            const float dct = s_dc[tid] + dh * go * (1.f - th * th);
            const float dpi = dct * gg * gi * (1.f - gi), dpf = dct * cp * gf * (1.f - gf);
            const float dpg = dct * gi * (1.f - gg * gg), dpo = dh * th * go * (1.f - go);
            s_dc[tid] = dct * gf;
            s_dgates[tid] = dpi;
            s_dgates[DEC_D + tid] = dpf;
            s_dgates[2 * DEC_D + tid] = dpg;
            s_dgates[3 * DEC_D + tid] = dpo;
            float *o = dl.gates + (long)s * 4 * DEC_D;
            o[tid] = dpi;
            o[DEC_D + tid] = dpf;
            o[2 * DEC_D + tid] = dpg;
            o[3 * DEC_D + tid] = dpo;
        }
        __syncthreads();
        DEC_TICK(1);
        if (tid < DEC_D && s > 0) fetch_step(s - 1, tid);   // lands under the weight stream below
        // d mixed += W_ih^T dgates;  dh w.r.t. the previous hidden state = W_hh^T dgates
        matvec_cols4<true, true>(tid, p.w_ih, p.w_hh, 4 * DEC_D, DEC_D, s_dgates, s_dmixed, s_dh, s_scr);
        __syncthreads();
        if (tid < DEC_D) {
            const float v = sv.mixed[s * DEC_D + tid] > 0.f ? s_dmixed[tid] : 0.f;
            s_dmixed[tid] = v;
            dl.mixed[s * DEC_D + tid] = v;
        }
        __syncthreads();
        DEC_TICK(2);
        if ((CW & 3) == 0) matvec_cols4<false, false>(tid, p.cmb_w, nullptr, DEC_D, CW, s_dmixed, s_dcat, nullptr, s_scr);
        else matvec_cols<false>(tid, p.cmb_w, DEC_D, CW, s_dmixed, s_dcat, s_scr);
        __syncthreads();
        DEC_TICK(3);
        // embedding row gradient (this workgroup is the only writer; thread tid owns column tid)
        if (tid < DEC_D) {
            const int tok = sv.toks[s];
            float g = p.emb[(long)tok * DEC_D + tid] > 0.f ? s_dcat[tid] : 0.f;
            if (dropmask) g *= dropmask[s * DEC_D + tid];
            d_emb[(long)tok * DEC_D + tid] += g;
        }
        // context backward: d_attn[t] = memory[t] . d_ctx   (d_memory += attn (x) d_ctx is summed over the steps later, in
        // dec_attn_grad_kernel: a read-modify-write of [Tz][ME] per step does not belong in this serial loop)
        if (tid < ME) dl.ctx[(long)s * ME + tid] = s_dcat[DEC_D + tid];
        if ((ME & 3) == 0) {   // wave w takes the states t = w (mod 16), 8 at a time -- all in flight --, a lane four adjacent columns
            f32x4 dc = {0.f, 0.f, 0.f, 0.f};
            if (4 * lane < ME) dc = *reinterpret_cast<const f32x4 *>(&s_dcat[DEC_D + 4 * lane]);
            for (int tb = wave; tb < Tz; tb += 8 * DEC_WAVES) {
                f32x4 m[8];
                float acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    m[i] = 4 * lane < ME ? *reinterpret_cast<const f32x4 *>(memory + (long)min(tb + DEC_WAVES * i, Tz - 1) * ME + 4 * lane)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = (m[i][0] * dc[0] + m[i][1] * dc[1]) + (m[i][2] * dc[2] + m[i][3] * dc[3]);
                const float sum = wave_sum_rows<8>(lane, acc);
                const int t = tb + DEC_WAVES * ((lane >> 3) & 7);
                if ((lane & 7) == 0 && t < Tz) s_ds[t] = sum;
            }
        } else {
            const float *dctx = s_dcat + DEC_D;
            for (int t0 = wave * 4; t0 < Tz; t0 += DEC_WAVES * 4) {   // 4 encoder states per wave at a time
                float acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float *m = memory + (long)min(t0 + i, Tz - 1) * ME;
                    float v = 0.f;
                    for (int j = lane; j < ME; j += 64) v += m[j] * dctx[j];
                    acc[i] = v;
                }
                const float sum = wave_sum_rows<4>(lane, acc);
                const int t = t0 + ((lane >> 4) & 3);
                if ((lane & 15) == 0 && t < Tz) s_ds[t] = sum;
            }
        }
        __syncthreads();
        DEC_TICK(4);
        {
            float part = 0.f;
            for (int t = tid; t < Tz; t += DEC_THREADS) part += sv.attn[(long)s * Tz + t] * s_ds[t];
            const float dot = block_sum(tid, part, s_red);
            for (int t = tid; t < Tz; t += DEC_THREADS) {
                const float v = sv.attn[(long)s * Tz + t] * (s_ds[t] - dot);
                s_ds[t] = v;
                dl.score[(long)s * Tz + t] = v;
            }
        }
        __syncthreads();
        DEC_TICK(5);
        // score backward through tanh: d_q, dV   (d_mp is rebuilt from the saved d_score in dec_attn_grad_kernel)
        {
            const int g = tid >> 7, k = tid & 127;
            const float qk = sv.q[s * DEC_D + k], vk = p.v[k];
            float dq = 0.f;
            for (int tb = g; tb < Tz; tb += 16 * (DEC_THREADS / DEC_D)) {   // 16 rows of the projection in flight per thread
                float m[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) m[i] = sv.mp[(long)min(tb + 8 * i, Tz - 1) * DEC_D + k];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int t = tb + 8 * i;
                    const float u = tanh_f(m[i] + qk);
                    const float ds = t < Tz ? s_ds[t] : 0.f;
                    dv_acc += ds * u;
                    dq += ds * vk * (1.f - u * u);
                }
            }
            s_scr[tid] = dq;
            __syncthreads();
            if (tid < DEC_D) {
                float v = 0.f;
                for (int gg = 0; gg < DEC_THREADS / DEC_D; ++gg) v += s_scr[gg * DEC_D + tid];
                s_dq[tid] = v;
                dl.q[s * DEC_D + tid] = v;
            }
        }
        __syncthreads();
        DEC_TICK(6);
        matvec_cols4<true, false>(tid, s_l2w, nullptr, DEC_D, DEC_D, s_dq, s_dh, nullptr, s_scr);
        __syncthreads();
        DEC_TICK(7);
    }
#ifdef DEC_TIMING
    if (tid == 0)
        printf("decoder_bwd step %d: cell %lld  lstm^T %lld  cmb^T %lld  d_attn %lld  softmax' %lld  d_q %lld  l2^T %lld   (cycles)\n", DEC_TIMING,
               s_tk[1] - s_tk[0], s_tk[2] - s_tk[1], s_tk[3] - s_tk[2], s_tk[4] - s_tk[3], s_tk[5] - s_tk[4], s_tk[6] - s_tk[5], s_tk[7] - s_tk[6]);
#endif
    // initial state -> h_n / c_n through hidden_out / cn_out
    if (tid < DEC_D) {
        dl.h0[tid] = s_dh[tid];
        dl.c0[tid] = s_dc[tid];
    }
    matvec_cols<false>(tid, p.ho_w, DEC_D, ME, s_dh, s_out, s_scr);
    __syncthreads();
    if (tid < ME) d_hn[tid] = s_out[tid];
    matvec_cols<false>(tid, p.co_w, DEC_D, ME, s_dc, s_out, s_scr);
    __syncthreads();
    if (tid < ME) d_cn[tid] = s_out[tid];
    __syncthreads();
    s_scr[tid] = dv_acc;
    __syncthreads();
    if (tid < DEC_D) {
        float v = 0.f;
        for (int gg = 0; gg < DEC_THREADS / DEC_D; ++gg) v += s_scr[gg * DEC_D + tid];
        d_v[tid] = v;
    }
}

// (OuterJob / OuterBatch / dec_outer_body: lstm.hpp -- the batch can ride in the LSTM's backward recurrence launch, mucon_decoder_bwd_defer)
__global__ __launch_bounds__(256) void dec_outer_kernel(OuterBatch ob) { dec_outer_body(ob, (int)blockIdx.x, (int)threadIdx.x); }

// Per encoder state t (grid (Tz), 256 threads), summed over the S decoding steps in step order:
//   d_mp[t][k]     = sum_s d_score[s][t] v[k] (1 - tanh^2(mp[t][k] + q[s][k]))        -> dl.mp (input of dW1's outer product)
//   d_memory[t][j] = sum_s attn[s][t] d_ctx[s][j]  +  sum_k d_mp[t][k] W1[j][k]
__global__ __launch_bounds__(256) void dec_attn_grad_kernel(DecSaved sv, DecDeltas dl, const float *w1, const float *v,
                                                            float *d_memory, int S, int Tz, int ME) {
    __shared__ __attribute__((aligned(16))) float ds[DEC_D];
    const int t = blockIdx.x, j = threadIdx.x;
    if (j < DEC_D) {
        const float m = sv.mp[(long)t * DEC_D + j], vk = v[j];
        float acc = 0.f;
        for (int s = 0; s < S; ++s) {
            const float u = tanh_f(m + sv.q[s * DEC_D + j]);
            acc += dl.score[(long)s * Tz + t] * vk * (1.f - u * u);
        }
        ds[j] = acc;
        dl.mp[(long)t * DEC_D + j] = acc;
    }
    __syncthreads();
    if (j >= ME) return;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += sv.attn[(long)s * Tz + t] * dl.ctx[(long)s * ME + j];
    const f32x4 *w = reinterpret_cast<const f32x4 *>(w1 + (long)j * DEC_D);
#pragma unroll 8
    for (int k4 = 0; k4 < DEC_D / 4; ++k4) {
        const f32x4 wv = w[k4];
        const f32x4 dv = *reinterpret_cast<const f32x4 *>(&ds[k4 * 4]);
        acc += wv[0] * dv[0] + wv[1] * dv[1] + wv[2] * dv[2] + wv[3] * dv[3];
    }
    d_memory[(long)t * ME + j] = acc;
}
