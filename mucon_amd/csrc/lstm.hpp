// Persistent bidirectional LSTM for the s-head's sequence encoder (reference src/mucon/models.py:195-201,
// 605-611: nn.LSTM(128 -> 128, batch_first, bidirectional) over the temporally encoded video [1 x Tz x 128]).
// SURVEY.md 8f row 1.  Batch 1, input = hidden = 128 (the reference's only configuration); gate order and
// arithmetic are torch.nn.LSTM's:  gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh,  i,f,o = sigmoid, g = tanh,
// c_t = f c_{t-1} + i g,  h_t = o tanh(c_t),  h_0 = c_0 = 0;  the reverse direction walks t = T-1 .. 0.
//
// MIOpen needs 5.9 ms for this forward and 11.9 ms with the backward (Tz = 125): the work is a chain of Tz
// dependent 512x128 mat-vecs, i.e. pure latency.  Here ONE workgroup per direction stays resident for the
// whole sequence: 512 threads, thread r keeps row r of W_hh (128 floats) in registers, h_{t-1} is broadcast
// from LDS, the four gates of a hidden unit meet through LDS; two barriers per time step.
//   lstm_inproj_kernel      Gx[d][t][r] = W_ih[d][r] . x[t] + b_ih[d][r] + b_hh[d][r]     (all t at once)
//   lstm_recur_fwd_kernel   the recurrence; saves gate activations and cell states for the backward
//   lstm_recur_bwd_kernel   BPTT: thread (q, j) keeps column j of gate block q of W_hh, dh_{t-1} = W_hh^T dgates_t
//   lstm_wgrad_kernel       dW_ih, dW_hh, db from the saved pre-activation gradients dG (reductions over t)
//   lstm_dx_kernel          dx[t] = sum_d W_ih[d]^T dG[d][t]
#pragma once
#include "common.hpp"

constexpr int LSTM_H = 128;
constexpr int LSTM_G = 4 * LSTM_H;  // 512 gate rows

struct LstmWeights {
    const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];  // [512][128], [512][128], [512], [512] per direction
};
struct LstmGrads {
    float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// grid (ceil(T/8), ndir), 512 threads: thread r holds W_ih[d][r] in registers, 8 time steps per workgroup
__global__ __launch_bounds__(512) void lstm_inproj_kernel(const float *x, LstmWeights w, float *Gx, int T) {
    __shared__ __attribute__((aligned(16))) float xs[8][LSTM_H];
    const int d = blockIdx.y, r = threadIdx.x, t0 = blockIdx.x * 8;
    float wr[LSTM_H];
#pragma unroll
    for (int c4 = 0; c4 < LSTM_H / 4; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w.w_ih[d] + (long)r * LSTM_H + c4 * 4);
        wr[c4 * 4 + 0] = v[0];
        wr[c4 * 4 + 1] = v[1];
        wr[c4 * 4 + 2] = v[2];
        wr[c4 * 4 + 3] = v[3];
    }
    for (int e = threadIdx.x; e < 8 * LSTM_H; e += 512) {
        const int tt = t0 + e / LSTM_H;
        xs[e / LSTM_H][e % LSTM_H] = tt < T ? x[(long)tt * LSTM_H + e % LSTM_H] : 0.f;
    }
    __syncthreads();
    const float bias = w.b_ih[d][r] + w.b_hh[d][r];
    for (int i = 0; i < 8; ++i) {
        const int t = t0 + i;
        if (t >= T) break;
        float acc = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < LSTM_H / 4; ++c4) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xs[i][c4 * 4]);
            acc += wr[c4 * 4 + 0] * xv[0] + wr[c4 * 4 + 1] * xv[1] + wr[c4 * 4 + 2] * xv[2] + wr[c4 * 4 + 3] * xv[3];
        }
        Gx[((long)d * T + t) * LSTM_G + r] = acc + bias;
    }
}

// grid (ndir), 512 threads.  out [T][ndir*128]; saves: gates [ndir][T][4][128] (post-activation i,f,g,o),
// cells [ndir][T][128]; hn / cn [ndir][128].
__global__ __launch_bounds__(512) void lstm_recur_fwd_kernel(const float *Gx, LstmWeights w, float *out, float *gates,
                                                             float *cells, float *hn, float *cn, int T, int ndir) {
    __shared__ __attribute__((aligned(16))) float hs[2][LSTM_H];
    __shared__ float pre[LSTM_G];
    const int d = blockIdx.x, r = threadIdx.x;
    float wr[LSTM_H];
#pragma unroll
    for (int c4 = 0; c4 < LSTM_H / 4; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w.w_hh[d] + (long)r * LSTM_H + c4 * 4);
        wr[c4 * 4 + 0] = v[0];
        wr[c4 * 4 + 1] = v[1];
        wr[c4 * 4 + 2] = v[2];
        wr[c4 * 4 + 3] = v[3];
    }
    if (r < LSTM_H) hs[0][r] = 0.f;
    float c = 0.f;  // cell state of hidden unit r (threads r < 128)
    const float *gx = Gx + (long)d * T * LSTM_G + r;
    float gnext = gx[(long)(d == 0 ? 0 : T - 1) * LSTM_G];
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int cur = s & 1;
        float acc = gnext;
        if (s + 1 < T) gnext = gx[(long)(d == 0 ? s + 1 : T - 2 - s) * LSTM_G];  // next step's input projection: in flight
#pragma unroll
        for (int c4 = 0; c4 < LSTM_H / 4; ++c4) {
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(&hs[cur][c4 * 4]);  // broadcast read
            acc += wr[c4 * 4 + 0] * hv[0] + wr[c4 * 4 + 1] * hv[1] + wr[c4 * 4 + 2] * hv[2] + wr[c4 * 4 + 3] * hv[3];
        }
        pre[r] = acc;
        __syncthreads();
        if (r < LSTM_H) {
            const float gi = sigmoid_f(pre[r]), gf = sigmoid_f(pre[LSTM_H + r]);
            const float gg = tanhf(pre[2 * LSTM_H + r]), go = sigmoid_f(pre[3 * LSTM_H + r]);
            c = gf * c + gi * gg;
            const float h = go * tanhf(c);
            float *gs = gates + ((long)d * T + t) * LSTM_G;
            gs[r] = gi;
            gs[LSTM_H + r] = gf;
            gs[2 * LSTM_H + r] = gg;
            gs[3 * LSTM_H + r] = go;
            cells[((long)d * T + t) * LSTM_H + r] = c;
            out[(long)t * (ndir * LSTM_H) + d * LSTM_H + r] = h;
            hs[cur ^ 1][r] = h;
            if (s == T - 1) {
                hn[d * LSTM_H + r] = h;
                cn[d * LSTM_H + r] = c;
            }
        }
        __syncthreads();
    }
}

// grid (ndir), 512 threads: thread (q = tid >> 7, j = tid & 127) keeps W_hh[d][q*128 + r'][j], r' = 0..127.
// dG [ndir][T][512]: gradient at the gate pre-activations (input of the weight / input gradients).
__global__ __launch_bounds__(512) void lstm_recur_bwd_kernel(LstmWeights w, const float *out, const float *gates,
                                                             const float *cells, const float *d_out, const float *d_hn,
                                                             const float *d_cn, float *dG, int T, int ndir) {
    __shared__ __attribute__((aligned(16))) float dgs[LSTM_G];
    __shared__ float part[4][LSTM_H];
    __shared__ float dhrec[LSTM_H];
    const int d = blockIdx.x, tid = threadIdx.x;
    const int q = tid >> 7, j = tid & 127;
    float wt[LSTM_H];
#pragma unroll
    for (int rr = 0; rr < LSTM_H; ++rr) wt[rr] = w.w_hh[d][(long)(q * LSTM_H + rr) * LSTM_H + j];
    float dc = 0.f;
    if (tid < LSTM_H) {
        dhrec[tid] = d_hn ? d_hn[d * LSTM_H + tid] : 0.f;
        dc = d_cn ? d_cn[d * LSTM_H + tid] : 0.f;
    }
    __syncthreads();
    for (int s = T - 1; s >= 0; --s) {           // reverse of the processing order
        const int t = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? t - 1 : t + 1;   // previous step in processing order (s - 1)
        if (tid < LSTM_H) {
            const float *gs = gates + ((long)d * T + t) * LSTM_G;
            const float gi = gs[tid], gf = gs[LSTM_H + tid], gg = gs[2 * LSTM_H + tid], go = gs[3 * LSTM_H + tid];
            const float ct = cells[((long)d * T + t) * LSTM_H + tid];
            const float cp = s > 0 ? cells[((long)d * T + tp) * LSTM_H + tid] : 0.f;
            const float dh = (d_out ? d_out[(long)t * (ndir * LSTM_H) + d * LSTM_H + tid] : 0.f) + dhrec[tid];
            const float th = tanhf(ct);
            const float dct = dc + dh * go * (1.f - th * th);
            const float dpi = dct * gg * gi * (1.f - gi);
            const float dpf = dct * cp * gf * (1.f - gf);
            const float dpg = dct * gi * (1.f - gg * gg);
            const float dpo = dh * th * go * (1.f - go);
            dc = dct * gf;
            dgs[tid] = dpi;
            dgs[LSTM_H + tid] = dpf;
            dgs[2 * LSTM_H + tid] = dpg;
            dgs[3 * LSTM_H + tid] = dpo;
            float *o = dG + ((long)d * T + t) * LSTM_G;
            o[tid] = dpi;
            o[LSTM_H + tid] = dpf;
            o[2 * LSTM_H + tid] = dpg;
            o[3 * LSTM_H + tid] = dpo;
        }
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int r4 = 0; r4 < LSTM_H / 4; ++r4) {
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(&dgs[q * LSTM_H + r4 * 4]);  // broadcast within a wave pair
            acc += wt[r4 * 4 + 0] * gv[0] + wt[r4 * 4 + 1] * gv[1] + wt[r4 * 4 + 2] * gv[2] + wt[r4 * 4 + 3] * gv[3];
        }
        part[q][j] = acc;
        __syncthreads();
        if (tid < LSTM_H) dhrec[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        __syncthreads();
    }
}

// dW_ih[d][r][c] = sum_t dG[d][t][r] x[t][c];  dW_hh[d][r][c] = sum_t dG[d][t][r] hprev[d][t][c];  db = sum_t dG[d][t][r]
// grid (512/4, ndir), 512 threads = 4 gate rows x 128 columns.
__global__ __launch_bounds__(512) void lstm_wgrad_kernel(const float *dG, const float *x, const float *out, LstmGrads g,
                                                         int T, int ndir) {
    const int d = blockIdx.y;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 7), c = threadIdx.x & 127;
    const float *dg = dG + (long)d * T * LSTM_G + r;
    float ai = 0.f, ah = 0.f, ab = 0.f;
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? t - 1 : t + 1;
        const float gv = dg[(long)t * LSTM_G];
        ai += gv * x[(long)t * LSTM_H + c];
        if (s > 0) ah += gv * out[(long)tp * (ndir * LSTM_H) + d * LSTM_H + c];
        ab += gv;
    }
    g.w_ih[d][(long)r * LSTM_H + c] = ai;
    g.w_hh[d][(long)r * LSTM_H + c] = ah;
    if (c == 0) {
        g.b_ih[d][r] = ab;
        g.b_hh[d][r] = ab;
    }
}

// dx[t][c] = sum_d sum_r dG[d][t][r] W_ih[d][r][c]; grid (T), 128 threads
__global__ __launch_bounds__(128) void lstm_dx_kernel(const float *dG, LstmWeights w, float *dx, int T, int ndir) {
    __shared__ float gsm[2 * LSTM_G];
    const int t = blockIdx.x, c = threadIdx.x;
    for (int e = c; e < ndir * LSTM_G; e += 128) gsm[e] = dG[((long)(e / LSTM_G) * T + t) * LSTM_G + e % LSTM_G];
    __syncthreads();
    float acc = 0.f;
    for (int d = 0; d < ndir; ++d)
        for (int r = 0; r < LSTM_G; ++r) acc += gsm[d * LSTM_G + r] * w.w_ih[d][(long)r * LSTM_H + c];
    dx[(long)t * LSTM_H + c] = acc;
}
