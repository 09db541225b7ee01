// The s-head's attention decoder as two persistent kernels (reference
// src/mucon/models.py:585-744: sequence_generation_forward + _calculate_attention).
//
// Per decoding step the reference runs ~30 small torch ops on 128-wide vectors (embedding, additive
// attention over the Tz encoder states, attn_combine, one LSTM cell, the transcript MLP, the length MLP,
// log-softmax, arg-max feedback): ~600 launches per training step, 11.5 ms of pure launch latency.  Here ONE
// workgroup (1024 threads) walks all steps: the recurrent state stays in LDS, weights stream from L2
// (~1 MB per step), the memory [Tz x 2E] and its projection are read once per step.
//   dec_memproj_kernel   mp = memory @ W1                                        (all Tz rows, many workgroups)
//   decoder_fwd_kernel   the step loop; saves every activation the backward needs
//   decoder_bwd_kernel   back-propagation through the steps; per-step "delta" vectors go to the workspace
//   dec_outer_kernel     every weight gradient = sum over steps of delta (x) input: one batched launch
//   dec_attn_grad_kernel d_mp and d_memory for all encoder states, from the per-step d_score / d_ctx the backward saved
// Vector width D = 128 (embedding = hidden = attention size: the reference's only configuration).
#pragma once
#include "common.hpp"

constexpr int DEC_D = 128;
constexpr int DEC_THREADS = 1024;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int DEC_MAXNC = 128;   // transcript classes + 1 (EOS)
constexpr int DEC_MAXME = 256;   // memory width (2E)
constexpr int DEC_NL = 64;       // hidden width of the length MLP (D / 2)

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float block_sum(float v, float *red) {  // red: DEC_WAVES + 1 floats of LDS
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = threadIdx.x < DEC_WAVES ? red[threadIdx.x] : 0.f;
        t = wave_sum(t);
        if (threadIdx.x == 0) red[DEC_WAVES] = t;
    }
    __syncthreads();
    return red[DEC_WAVES];
}
__device__ __forceinline__ float block_max(float v, float *red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = threadIdx.x < DEC_WAVES ? red[threadIdx.x] : -INFINITY;
        t = wave_max(t);
        if (threadIdx.x == 0) red[DEC_WAVES] = t;
    }
    __syncthreads();
    return red[DEC_WAVES];
}

// Sum R (8 or 4) per-lane values across the wave with 10 (7) shuffles instead of R x 6: after the call, the lanes
// with (lane >> 3) & 7 == r  (R = 8)  or  (lane >> 4) & 3 == r  (R = 4)  hold the wave-wide sum of v[r].
template <int R>
__device__ __forceinline__ float wave_sum_rows(float *v) {
    const int lane = threadIdx.x & 63;
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
    float c;
    if (R == 8) {
        float a[4], b2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (h5 ? v[i + 4] : v[i]) + __shfl_xor(h5 ? v[i] : v[i + 4], 32);
#pragma unroll
        for (int i = 0; i < 2; ++i) b2[i] = (h4 ? a[i + 2] : a[i]) + __shfl_xor(h4 ? a[i] : a[i + 2], 16);
        c = (h3 ? b2[1] : b2[0]) + __shfl_xor(h3 ? b2[0] : b2[1], 8);
    } else {
        float a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = (h5 ? v[i + 2] : v[i]) + __shfl_xor(h5 ? v[i] : v[i + 2], 32);
        c = (h4 ? a[1] : a[0]) + __shfl_xor(h4 ? a[0] : a[1], 16);
        c += __shfl_xor(c, 8);
    }
    c += __shfl_xor(c, 4);
    c += __shfl_xor(c, 2);
    c += __shfl_xor(c, 1);
    return c;
}

// out[r] = act(b[r] + W[r][:] . x  [+ b2[r] + W2[r][:] . x2]) for r < rows: R rows per wave at a time, lanes across the
// columns -- all of an iteration's loads (R rows x COLS/64, both matrices) are issued before the first use, so an
// iteration costs about one L2 round trip -- then one tree reduction for the R rows.  COLS is a multiple of 64, or 0 for
// a run-time column count (`cols`, one round trip per 64 columns).
template <int ACT, int R, int COLS, bool DUAL>
__device__ __forceinline__ void matvec_rows(const float *__restrict__ W, const float *__restrict__ b, int rows, int cols,
                                            const float *x, float *out, const float *__restrict__ W2 = nullptr,
                                            const float *__restrict__ b2 = nullptr, const float *x2 = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        const float sum = wave_sum_rows<R>(acc);
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
__device__ __forceinline__ void matvec_cols(const float *__restrict__ W, int rows, int cols, const float *d, float *out,
                                            float *scratch) {
    const int cp = cols <= 128 ? 128 : cols <= 256 ? 256 : 512;
    const int ng = DEC_THREADS / cp;
    const int g = threadIdx.x / cp, j = threadIdx.x - g * cp;
    float acc = 0.f;
    if (j < cols) {
#pragma unroll 8
        for (int i = g; i < rows; i += ng) acc += W[(long)i * cols + j] * d[i];
    }
    __syncthreads();  // scratch may still be read by the previous user
    scratch[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < cols) {
        float s = 0.f;
        for (int gg = 0; gg < ng; ++gg) s += scratch[gg * cp + threadIdx.x];
        out[threadIdx.x] = ACC ? out[threadIdx.x] + s : s;
    }
}

// The same for a column count that is a multiple of 4: a thread owns FOUR adjacent columns (one 16-byte load per row), so
// cols/4 threads cover a row and the 1024 threads form up to 32 row groups -- 512 x 128 weights are 16 dependent steps per
// thread instead of 64.  PAIR: two matrices with the same d in one pass (out_a += Wa^T d, out_b = Wb^T d).
// scratch: groups * cols (* 2) floats -- DEC_SCR floats cover every use below.
constexpr int DEC_SCR = 2 * 32 * DEC_D;
template <bool ACC, bool PAIR>
__device__ __forceinline__ void matvec_cols4(const float *__restrict__ Wa, const float *__restrict__ Wb, int rows, int cols,
                                             const float *d, float *out_a, float *out_b, float *scratch) {
    const int tpg = cols >> 2;                 // threads per row group
    const int ng = DEC_THREADS / tpg;          // row groups
    const int g = threadIdx.x / tpg, j4 = threadIdx.x - g * tpg;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (g < ng) {
#pragma unroll 4
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
    if ((int)threadIdx.x < nout) {
        const int which = threadIdx.x >= cols, jj = threadIdx.x - which * cols;
        float sum = 0.f;
        for (int gg = 0; gg < ng; ++gg) sum += scratch[which * ng * cols + gg * cols + jj];
        if (which == 0) out_a[jj] = ACC ? out_a[jj] + sum : sum;
        else out_b[jj] = sum;
    }
}

// mp[t][k] = sum_j memory[t][j] W1[j][k]; grid (ceil(Tz/4)), 512 threads = 4 rows x 128 columns
__global__ __launch_bounds__(512) void dec_memproj_kernel(const float *memory, const float *w1, float *mp, int Tz, int ME) {
    __shared__ float ms[4][DEC_MAXME];
    const int t0 = blockIdx.x * 4;
    for (int e = threadIdx.x; e < 4 * ME; e += 512) {
        const int t = t0 + e / ME;
        ms[e / ME][e % ME] = t < Tz ? memory[(long)t * ME + e % ME] : 0.f;
    }
    __syncthreads();
    const int r = threadIdx.x >> 7, k = threadIdx.x & 127;
    if (t0 + r >= Tz) return;
    float acc = 0.f;
    for (int j = 0; j < ME; ++j) acc += ms[r][j] * w1[(long)j * DEC_D + k];
    mp[(long)(t0 + r) * DEC_D + k] = acc;
}

// dynamic LDS: Tz floats (attention scores / weights)
__global__ __launch_bounds__(DEC_THREADS) void decoder_fwd_kernel(DecDims dm, DecParams p, DecSaved sv, const float *memory,
                                                                  const float *hn, const float *cn, const long *tf_input,
                                                                  const float *dropmask, float *logp_out, float *len_out,
                                                                  int *nsteps_out) {
    extern __shared__ float s_score[];
    __shared__ float s_h[DEC_D], s_c[DEC_D], s_q[DEC_D], s_cat[DEC_D + DEC_MAXME], s_mixed[DEC_D], s_gates[4 * DEC_D];
    __shared__ float s_t1[DEC_D], s_logits[DEC_MAXNC], s_lencat[DEC_D + DEC_MAXNC], s_l1[DEC_NL], s_scr[DEC_THREADS];
    __shared__ float s_hc[2 * DEC_MAXME], s_red[DEC_WAVES + 1];
    __shared__ int s_tok, s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    matvec_rows<0, 4, 0, false>(p.ho_w, p.ho_b, DEC_D, ME, s_hc, s_h);
    matvec_rows<0, 4, 0, false>(p.co_w, p.co_b, DEC_D, ME, s_hc + DEC_MAXME, s_c);
    __syncthreads();
    if (tid < DEC_D) {
        sv.h[tid] = s_h[tid];
        sv.c[tid] = s_c[tid];
    }
    int s = 0;
    for (; s < dm.S; ++s) {
        int tok = dm.teacher_forcing ? (int)tf_input[s] : s_tok;
        tok = tok < 0 ? 0 : tok >= dm.n_emb ? dm.n_emb - 1 : tok;  // host validated; keeps a bad arg-max in range
        // embedded = dropout(relu(embedding(input)));  q = attention_l2(dec_h)
        if (tid < DEC_D) {
            float e = fmaxf(p.emb[(long)tok * DEC_D + tid], 0.f);
            if (dropmask) e *= dropmask[s * DEC_D + tid];
            s_cat[tid] = e;
        }
        if (tid == 0) sv.toks[s] = tok;
        matvec_rows<0, 8, DEC_D, false>(p.l2_w, p.l2_b, DEC_D, DEC_D, s_h, s_q);
        __syncthreads();
        if (tid < DEC_D) sv.q[s * DEC_D + tid] = s_q[tid];
        // score[t] = V . tanh(mp[t] + q): 8 encoder states per wave at a time (16 loads in flight)
        {
            const float q0 = s_q[lane], q1 = s_q[lane + 64], v0 = p.v[lane], v1 = p.v[lane + 64];
            for (int t0 = wave * 8; t0 < Tz; t0 += DEC_WAVES * 8) {
                float m0[8], m1[8], a[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float *m = sv.mp + (long)min(t0 + i, Tz - 1) * DEC_D;
                    m0[i] = m[lane];
                    m1[i] = m[lane + 64];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = v0 * tanh_f(m0[i] + q0) + v1 * tanh_f(m1[i] + q1);
                const float sum = wave_sum_rows<8>(a);
                const int t = t0 + ((lane >> 3) & 7);
                if ((lane & 7) == 0 && t < Tz) s_score[t] = sum;
            }
        }
        __syncthreads();
        // attention weights = softmax(score)
        {
            float mx = -INFINITY;
            for (int t = tid; t < Tz; t += DEC_THREADS) mx = fmaxf(mx, s_score[t]);
            mx = block_max(mx, s_red);
            float sum = 0.f;
            for (int t = tid; t < Tz; t += DEC_THREADS) {
                const float e = expf(s_score[t] - mx);
                s_score[t] = e;
                sum += e;
            }
            sum = block_sum(sum, s_red);
            const float inv = 1.f / sum;
            for (int t = tid; t < Tz; t += DEC_THREADS) {
                const float a = s_score[t] * inv;
                s_score[t] = a;
                sv.attn[(long)s * Tz + t] = a;
            }
        }
        __syncthreads();
        // context = sum_t attn[t] memory[t]
        {
            const int g = tid >> 8, j = tid & 255;
            float acc = 0.f;
            if (j < ME) {
#pragma unroll 8
                for (int t = g; t < Tz; t += 4) acc += s_score[t] * memory[(long)t * ME + j];
            }
            s_scr[tid] = acc;
            __syncthreads();
            if (tid < ME) s_cat[DEC_D + tid] = (s_scr[tid] + s_scr[256 + tid]) + (s_scr[512 + tid] + s_scr[768 + tid]);
        }
        __syncthreads();
        if (tid < CW) sv.cat[(long)s * CW + tid] = s_cat[tid];
        // mixed = relu(attn_combine(cat(embedded, context)))
        if (ME == 256) matvec_rows<1, 4, DEC_D + 256, false>(p.cmb_w, p.cmb_b, DEC_D, CW, s_cat, s_mixed);
        else if (ME == 128) matvec_rows<1, 4, DEC_D + 128, false>(p.cmb_w, p.cmb_b, DEC_D, CW, s_cat, s_mixed);
        else matvec_rows<1, 4, 0, false>(p.cmb_w, p.cmb_b, DEC_D, CW, s_cat, s_mixed);
        __syncthreads();
        // one LSTM cell
        matvec_rows<0, 8, DEC_D, true>(p.w_ih, p.b_ih, 4 * DEC_D, DEC_D, s_mixed, s_gates, p.w_hh, p.b_hh, s_h);
        __syncthreads();
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
        // word logits = transcript MLP(dec_out)
        matvec_rows<1, 8, DEC_D, false>(p.t1_w, p.t1_b, DEC_D, DEC_D, s_h, s_t1);
        __syncthreads();
        if (tid < DEC_D) sv.t1[s * DEC_D + tid] = s_t1[tid];
        matvec_rows<0, 4, DEC_D, false>(p.t2_w, p.t2_b, NC, DEC_D, s_t1, s_logits);
        __syncthreads();
        // length = length MLP(relu(cat(mixed, word logits)))
        if (tid < LW) {
            const float v = tid < DEC_D ? s_mixed[tid] : fmaxf(s_logits[tid - DEC_D], 0.f);
            s_lencat[tid] = v;
            sv.lencat[(long)s * LW + tid] = v;
        }
        __syncthreads();
        matvec_rows<1, 4, 0, false>(p.n1_w, p.n1_b, DEC_NL, LW, s_lencat, s_l1);
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
}

// dynamic LDS: Tz floats.  dm.S = the number of steps the forward ran.  d_logp [S][NC] / d_len [S] may be null.
// d_emb [n_emb][D] is zeroed here; d_v [D]; d_hn / d_cn [ME]; d_memory and dl.mp are written by dec_attn_grad_kernel.
__global__ __launch_bounds__(DEC_THREADS) void decoder_bwd_kernel(DecDims dm, DecParams p, DecSaved sv, DecDeltas dl,
                                                                  const float *memory, const float *logp, const float *d_logp,
                                                                  const float *d_len, const float *dropmask, float *d_memory,
                                                                  float *d_emb, float *d_v, float *d_hn, float *d_cn) {
    extern __shared__ float s_ds[];  // d_attn, then d_score
    __shared__ float s_dh[DEC_D], s_dc[DEC_D], s_dlogits[DEC_MAXNC], s_dl1[DEC_NL], s_dlencat[DEC_D + DEC_MAXNC];
    __shared__ float s_dt1[DEC_D], s_dgates[4 * DEC_D], s_dmixed[DEC_D], s_dcat[DEC_D + DEC_MAXME], s_dq[DEC_D];
    __shared__ __attribute__((aligned(16))) float s_scr[DEC_SCR];
    __shared__ float s_red[DEC_WAVES + 1], s_out[DEC_MAXME];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tz = dm.Tz, ME = dm.ME, NC = dm.NC, CW = DEC_D + ME, LW = DEC_D + NC;

    for (long e = tid; e < (long)dm.n_emb * DEC_D; e += DEC_THREADS) d_emb[e] = 0.f;
    if (tid < DEC_D) {
        s_dh[tid] = 0.f;
        s_dc[tid] = 0.f;
    }
    float dv_acc = 0.f;  // thread (g = tid >> 7, k = tid & 127): partial dV[k]
    __syncthreads();

    for (int s = dm.S - 1; s >= 0; --s) {
        // log-softmax backward; length MLP output layer backward
        if (wave == 0) {
            const float g0 = (d_logp && lane < NC) ? d_logp[(long)s * NC + lane] : 0.f;
            const float g1 = (d_logp && lane + 64 < NC) ? d_logp[(long)s * NC + lane + 64] : 0.f;
            const float tot = wave_sum(g0 + g1);
            if (lane < NC) s_dlogits[lane] = g0 - expf(logp[(long)s * NC + lane]) * tot;
            if (lane + 64 < NC) s_dlogits[lane + 64] = g1 - expf(logp[(long)s * NC + lane + 64]) * tot;
        } else if (wave == 1) {
            const float dlen = d_len ? d_len[s] : 0.f;
            const float v = sv.l1[s * DEC_NL + lane] > 0.f ? dlen * p.n2_w[lane] : 0.f;
            s_dl1[lane] = v;
            dl.l1[s * DEC_NL + lane] = v;
            if (lane == 0) dl.len[s] = dlen;
        }
        __syncthreads();
        matvec_cols<false>(p.n1_w, DEC_NL, LW, s_dl1, s_dlencat, s_scr);
        __syncthreads();
        if (tid < DEC_D) s_dmixed[tid] = sv.lencat[(long)s * LW + tid] > 0.f ? s_dlencat[tid] : 0.f;
        if (tid < NC) {
            const float v = s_dlogits[tid] + (sv.lencat[(long)s * LW + DEC_D + tid] > 0.f ? s_dlencat[DEC_D + tid] : 0.f);
            s_dlogits[tid] = v;
            dl.logits[(long)s * NC + tid] = v;
        }
        __syncthreads();
        // transcript MLP backward -> d dec_out (added to the recurrent dh)
        matvec_cols4<false, false>(p.t2_w, nullptr, NC, DEC_D, s_dlogits, s_dt1, nullptr, s_scr);
        __syncthreads();
        if (tid < DEC_D) {
            const float v = sv.t1[s * DEC_D + tid] > 0.f ? s_dt1[tid] : 0.f;
            s_dt1[tid] = v;
            dl.t1[s * DEC_D + tid] = v;
        }
        __syncthreads();
        matvec_cols4<true, false>(p.t1_w, nullptr, DEC_D, DEC_D, s_dt1, s_dh, nullptr, s_scr);
        __syncthreads();
        // LSTM cell backward
        if (tid < DEC_D) {
            const float *gs = sv.gates + (long)s * 4 * DEC_D;
            const float gi = gs[tid], gf = gs[DEC_D + tid], gg = gs[2 * DEC_D + tid], go = gs[3 * DEC_D + tid];
            const float ct = sv.c[(s + 1) * DEC_D + tid], cp = sv.c[s * DEC_D + tid];
            const float dh = s_dh[tid], th = tanh_f(ct);
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
        // d mixed += W_ih^T dgates;  dh w.r.t. the previous hidden state = W_hh^T dgates
        matvec_cols4<true, true>(p.w_ih, p.w_hh, 4 * DEC_D, DEC_D, s_dgates, s_dmixed, s_dh, s_scr);
        __syncthreads();
        if (tid < DEC_D) {
            const float v = sv.mixed[s * DEC_D + tid] > 0.f ? s_dmixed[tid] : 0.f;
            s_dmixed[tid] = v;
            dl.mixed[s * DEC_D + tid] = v;
        }
        __syncthreads();
        if ((CW & 3) == 0) matvec_cols4<false, false>(p.cmb_w, nullptr, DEC_D, CW, s_dmixed, s_dcat, nullptr, s_scr);
        else matvec_cols<false>(p.cmb_w, DEC_D, CW, s_dmixed, s_dcat, s_scr);
        __syncthreads();
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
        {
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
                const float sum = wave_sum_rows<4>(acc);
                const int t = t0 + ((lane >> 4) & 3);
                if ((lane & 15) == 0 && t < Tz) s_ds[t] = sum;
            }
        }
        __syncthreads();
        {
            float part = 0.f;
            for (int t = tid; t < Tz; t += DEC_THREADS) part += sv.attn[(long)s * Tz + t] * s_ds[t];
            const float dot = block_sum(part, s_red);
            for (int t = tid; t < Tz; t += DEC_THREADS) {
                const float v = sv.attn[(long)s * Tz + t] * (s_ds[t] - dot);
                s_ds[t] = v;
                dl.score[(long)s * Tz + t] = v;
            }
        }
        __syncthreads();
        // score backward through tanh: d_q, dV   (d_mp is rebuilt from the saved d_score in dec_attn_grad_kernel)
        {
            const int g = tid >> 7, k = tid & 127;
            const float qk = sv.q[s * DEC_D + k], vk = p.v[k];
            float dq = 0.f;
#pragma unroll 4
            for (int t = g; t < Tz; t += DEC_THREADS / DEC_D) {
                const float u = tanh_f(sv.mp[(long)t * DEC_D + k] + qk);
                const float ds = s_ds[t];
                dv_acc += ds * u;
                dq += ds * vk * (1.f - u * u);
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
        matvec_cols4<true, false>(p.l2_w, nullptr, DEC_D, DEC_D, s_dq, s_dh, nullptr, s_scr);
        __syncthreads();
    }
    // initial state -> h_n / c_n through hidden_out / cn_out
    if (tid < DEC_D) {
        dl.h0[tid] = s_dh[tid];
        dl.c0[tid] = s_dc[tid];
    }
    matvec_cols<false>(p.ho_w, DEC_D, ME, s_dh, s_out, s_scr);
    __syncthreads();
    if (tid < ME) d_hn[tid] = s_out[tid];
    matvec_cols<false>(p.co_w, DEC_D, ME, s_dc, s_out, s_scr);
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

// out[i][j] = sum_n A[n*lda + i] B[n*ldb + j];  bias[i] = sum_n A[n*lda + i]
struct OuterJob {
    const float *A, *B;
    float *out, *bias, *bias2;
    int lda, ldb, ra, cb, n, block0;
};
constexpr int DEC_MAXJOBS = 12;
struct OuterBatch {
    OuterJob job[DEC_MAXJOBS];
    int njobs;
};
__global__ __launch_bounds__(256) void dec_outer_kernel(OuterBatch ob) {
    int ji = 0;
    while (ji + 1 < ob.njobs && (int)blockIdx.x >= ob.job[ji + 1].block0) ++ji;
    const OuterJob &jb = ob.job[ji];
    const long e = (long)(blockIdx.x - jb.block0) * 256 + threadIdx.x;
    if (e >= (long)jb.ra * jb.cb) return;
    const int i = (int)(e / jb.cb), j = (int)(e - (long)i * jb.cb);
    float acc = 0.f, accb = 0.f;
    // eight terms' loads in flight at a time, summed in order (dW1 sums over the Tz encoder states: a term-at-a-time loop is a
    // chain of Tz memory round trips)
    for (int n0 = 0; n0 < jb.n; n0 += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = min(n0 + q, jb.n - 1);
            av[q] = jb.A[(long)n * jb.lda + i];
            bv[q] = jb.B[(long)n * jb.ldb + j];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (n0 + q < jb.n) {
                acc += av[q] * bv[q];
                accb += av[q];
            }
        }
    }
    jb.out[e] = acc;
    if (j == 0) {
        if (jb.bias) jb.bias[i] = accb;
        if (jb.bias2) jb.bias2[i] = accb;
    }
}

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
