// Persistent bidirectional LSTM for the s-head's sequence encoder (reference src/mucon/models.py:195-201,
// 605-611: nn.LSTM(128 -> 128, batch_first, bidirectional) over the temporally encoded video [1 x Tz x 128]).
// SURVEY.md 8f row 1.  Batch 1, input = hidden = 128 (the reference's only configuration); gate order and
// arithmetic are torch.nn.LSTM's:  gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh,  i,f,o = sigmoid, g = tanh,
// c_t = f c_{t-1} + i g,  h_t = o tanh(c_t),  h_0 = c_0 = 0;  the reverse direction walks t = T-1 .. 0.
//
// MIOpen needs 5.9 ms for this forward and 11.9 ms with the backward (Tz = 125): the work is a chain of Tz
// dependent 512x128 mat-vecs, i.e. pure latency.  Here ONE workgroup per direction stays resident for the
// whole sequence: 512 threads, each keeps one row of W_hh (128 floats) in registers, h_{t-1} is broadcast
// from LDS, the four gates of a hidden unit sit in one wave and meet through shuffles; one barrier per time step.
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

// exp through the hardware exp2 and the hardware reciprocal (1 ulp each) instead of libm's expf / tanhf and an IEEE division:
// the recurrences are chains of dependent vector instructions, and tanhf alone was ~30 of them.  |error| < 3e-7 absolute
// (tanh as 1 - 2 / (1 + e^{2x}) cancels for tiny |x|: absolute, not relative accuracy -- what a gate needs).
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

constexpr int LSTM_IP_T = 4;   // time steps per workgroup of the input projection (8 left 32 workgroups for T = 125: 15 us)
// grid (ceil(T/LSTM_IP_T), ndir), 512 threads: thread r holds W_ih[d][r] in registers
__global__ __launch_bounds__(512) void lstm_inproj_kernel(const float *x, LstmWeights w, float *Gx, int T) {
    __shared__ __attribute__((aligned(16))) float xs[LSTM_IP_T][LSTM_H];
    const int d = blockIdx.y, r = threadIdx.x, t0 = blockIdx.x * LSTM_IP_T;
    float wr[LSTM_H];
#pragma unroll
    for (int c4 = 0; c4 < LSTM_H / 4; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w.w_ih[d] + (long)r * LSTM_H + c4 * 4);
        wr[c4 * 4 + 0] = v[0];
        wr[c4 * 4 + 1] = v[1];
        wr[c4 * 4 + 2] = v[2];
        wr[c4 * 4 + 3] = v[3];
    }
    for (int e = threadIdx.x; e < LSTM_IP_T * LSTM_H; e += 512) {
        const int tt = t0 + e / LSTM_H;
        xs[e / LSTM_H][e % LSTM_H] = tt < T ? x[(long)tt * LSTM_H + e % LSTM_H] : 0.f;
    }
    __syncthreads();
    const float bias = w.b_ih[d][r] + w.b_hh[d][r];
    for (int i = 0; i < LSTM_IP_T; ++i) {
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
// Thread layout: wave w owns hidden units 16w .. 16w+15; lane (u = lane & 15, q = lane >> 4) holds row q*128 + 16w + u of
// W_hh, i.e. the four gates of a unit sit in ONE wave: they meet through three wave shuffles instead of an LDS round trip
// and a barrier, every lane applies its own gate's non-linearity (tanh x = 2 sigmoid(2x) - 1: branch-free), lanes 0-15
// update the cell.  One barrier per time step (the new h for everybody).
__global__ __launch_bounds__(512) void lstm_recur_fwd_kernel(const float *Gx, LstmWeights w, float *out, float *gates,
                                                             float *cells, float *hn, float *cn, int T, int ndir) {
    __shared__ __attribute__((aligned(16))) float hs[2][LSTM_H];
    const int d = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, unit = wave * 16 + (lane & 15);
    const int r = q * LSTM_H + unit;
    float wr[LSTM_H];
#pragma unroll
    for (int c4 = 0; c4 < LSTM_H / 4; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w.w_hh[d] + (long)r * LSTM_H + c4 * 4);
        wr[c4 * 4 + 0] = v[0];
        wr[c4 * 4 + 1] = v[1];
        wr[c4 * 4 + 2] = v[2];
        wr[c4 * 4 + 3] = v[3];
    }
    if (tid < LSTM_H) hs[0][tid] = 0.f;
    float c = 0.f;  // cell state of `unit` (lanes with q == 0)
    const float *gx = Gx + (long)d * T * LSTM_G + r;
    float gnext = gx[(long)(d == 0 ? 0 : T - 1) * LSTM_G];
    const float asc = q == 2 ? 2.f : 1.f;   // gate g: tanh through the sigmoid
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int cur = s & 1;
        const float g0 = gnext;
        if (s + 1 < T) gnext = gx[(long)(d == 0 ? s + 1 : T - 2 - s) * LSTM_G];  // next step's input projection: in flight
        // the step is bound by this wave's vector issue (two waves per SIMD, 128 multiply-adds each): packed FMAs, two
        // accumulator pairs (v_pk_fma_f32: two per lane and instruction) -- 64 instead of 128 instructions
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int c4 = 0; c4 < LSTM_H / 4; ++c4) {
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(&hs[cur][c4 * 4]);  // broadcast read
            a0 = f32x2{wr[c4 * 4 + 0], wr[c4 * 4 + 1]} * f32x2{hv[0], hv[1]} + a0;
            a1 = f32x2{wr[c4 * 4 + 2], wr[c4 * 4 + 3]} * f32x2{hv[2], hv[3]} + a1;
        }
        const float acc = g0 + ((a0[0] + a0[1]) + (a1[0] + a1[1]));
        const float sg = sigmoid_f(asc * acc);
        const float act = q == 2 ? 2.f * sg - 1.f : sg;
        gates[((long)d * T + t) * LSTM_G + r] = act;
        const float gf = __shfl(act, (lane & 15) + 16), gg = __shfl(act, (lane & 15) + 32), go = __shfl(act, (lane & 15) + 48);
        if (q == 0) {
            c = gf * c + act * gg;
            const float h = go * tanh_f(c);
            cells[((long)d * T + t) * LSTM_H + unit] = c;
            out[(long)t * (ndir * LSTM_H) + d * LSTM_H + unit] = h;
            hs[cur ^ 1][unit] = h;
            if (s == T - 1) {
                hn[d * LSTM_H + unit] = h;
                cn[d * LSTM_H + unit] = c;
            }
        }
        __syncthreads();
    }
}

// grid (ndir), 512 threads.  Same ownership as the forward for the gate gradients (lane (u, q) -> gate q of unit 16w + u:
// every lane derives its own pre-activation gradient, the four lanes of a unit carry identical copies of dc); for
// dh_{t-1} = W_hh^T dgates thread (qq = tid >> 7, j = tid & 127) keeps W_hh[qq*128 + r'][j], r' = 0..127, and the four
// partial sums of a unit are added by the lanes that need them: two barriers per time step.
// dG [ndir][T][512]: gradient at the gate pre-activations (input of the weight / input gradients).
__global__ __launch_bounds__(512) void lstm_recur_bwd_kernel(LstmWeights w, const float *out, const float *gates,
                                                             const float *cells, const float *d_out, const float *d_hn,
                                                             const float *d_cn, float *dG, int T, int ndir) {
    __shared__ __attribute__((aligned(16))) float dgs[LSTM_G];
    __shared__ float part[4][LSTM_H];
    const int d = blockIdx.x, tid = threadIdx.x;
    const int qq = tid >> 7, j = tid & 127;                 // mat-vec role
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, unit = wave * 16 + (lane & 15);   // gate role
    float wt[LSTM_H];
#pragma unroll
    for (int rr = 0; rr < LSTM_H; ++rr) wt[rr] = w.w_hh[d][(long)(qq * LSTM_H + rr) * LSTM_H + j];
    float dc = d_cn ? d_cn[d * LSTM_H + unit] : 0.f;
    if (tid < LSTM_H) {   // the recurrent dh of the first processed step = d_hn: park it in part[0], zeros in part[1..3]
        part[0][tid] = d_hn ? d_hn[d * LSTM_H + tid] : 0.f;
        part[1][tid] = 0.f;
        part[2][tid] = 0.f;
        part[3][tid] = 0.f;
    }
    __syncthreads();
    // the saved activations / upstream gradient of a step do not depend on the recurrence: step s-1's are requested while
    // step s is processed (one L2 round trip per step would otherwise sit on the critical path)
    struct StepIn {
        float gi, gf, gg, go, ct, cp, dout;
    };
    auto load_step = [&](int s) {
        StepIn v;
        const int t = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? t - 1 : t + 1;   // previous step in processing order (s - 1)
        const float *gs = gates + ((long)d * T + t) * LSTM_G;
        v.gi = gs[unit];
        v.gf = gs[LSTM_H + unit];
        v.gg = gs[2 * LSTM_H + unit];
        v.go = gs[3 * LSTM_H + unit];
        v.ct = cells[((long)d * T + t) * LSTM_H + unit];
        v.cp = s > 0 ? cells[((long)d * T + tp) * LSTM_H + unit] : 0.f;
        v.dout = d_out ? d_out[(long)t * (ndir * LSTM_H) + d * LSTM_H + unit] : 0.f;
        return v;
    };
    StepIn nx = load_step(T - 1);
    for (int s = T - 1; s >= 0; --s) {           // reverse of the processing order
        const int t = d == 0 ? s : T - 1 - s;
        const StepIn in = nx;
        if (s > 0) nx = load_step(s - 1);
        {
            const float gi = in.gi, gf = in.gf, gg = in.gg, go = in.go, ct = in.ct, cp = in.cp;
            const float dh = in.dout + ((part[0][unit] + part[1][unit]) + (part[2][unit] + part[3][unit]));
            const float th = tanh_f(ct);
            const float dct = dc + dh * go * (1.f - th * th);
            dc = dct * gf;
            float dp;
            if (q == 0) dp = dct * gg * gi * (1.f - gi);
            else if (q == 1) dp = dct * cp * gf * (1.f - gf);
            else if (q == 2) dp = dct * gi * (1.f - gg * gg);
            else dp = dh * th * go * (1.f - go);
            // (dgs was last read before the barrier that ended the previous step; part[] is rewritten only after the next one)
            dgs[q * LSTM_H + unit] = dp;
            dG[((long)d * T + t) * LSTM_G + q * LSTM_H + unit] = dp;
        }
        __syncthreads();
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};     // packed FMAs, as in the forward
#pragma unroll
        for (int r4 = 0; r4 < LSTM_H / 4; ++r4) {
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(&dgs[qq * LSTM_H + r4 * 4]);  // broadcast within a wave pair
            a0 = f32x2{wt[r4 * 4 + 0], wt[r4 * 4 + 1]} * f32x2{gv[0], gv[1]} + a0;
            a1 = f32x2{wt[r4 * 4 + 2], wt[r4 * 4 + 3]} * f32x2{gv[2], gv[3]} + a1;
        }
        part[qq][j] = (a0[0] + a0[1]) + (a1[0] + a1[1]);
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
    // 32 time steps' loads in flight at a time, summed in time order (a step-at-a-time loop is a chain of T memory round trips;
    // eight at a time still 16 of them: 15.7 us at T = 125)
    constexpr int WG_B = 32;
    for (int s0 = 0; s0 < T; s0 += WG_B) {
        float gv[WG_B], xv[WG_B], hv[WG_B];
#pragma unroll
        for (int j = 0; j < WG_B; ++j) {
            const int s = min(s0 + j, T - 1);
            const int t = d == 0 ? s : T - 1 - s;
            const int tp = d == 0 ? max(t - 1, 0) : min(t + 1, T - 1);
            gv[j] = dg[(long)t * LSTM_G];
            xv[j] = x[(long)t * LSTM_H + c];
            hv[j] = out[(long)tp * (ndir * LSTM_H) + d * LSTM_H + c];
        }
#pragma unroll
        for (int j = 0; j < WG_B; ++j) {
            if (s0 + j < T) {
                ai += gv[j] * xv[j];
                if (s0 + j > 0) ah += gv[j] * hv[j];
                ab += gv[j];
            }
        }
    }
    g.w_ih[d][(long)r * LSTM_H + c] = ai;
    g.w_hh[d][(long)r * LSTM_H + c] = ah;
    if (c == 0) {
        g.b_ih[d][r] = ab;
        g.b_hh[d][r] = ab;
    }
}

// dx[t][c] = (dx_add[t][c]) + sum_d sum_r dG[d][t][r] W_ih[d][r][c]; grid (T), 512 threads = 4 row quarters x 128 columns.  What a workgroup costs is
// its chain of weight loads (W_ih comes from L2): every thread walks a quarter of the rows with 32 loads in flight, the four
// partial sums meet in LDS in quarter order.  (128 threads with 16 loads in flight: 64 dependent round trips, 18.6 us at T = 125.)
__global__ __launch_bounds__(512) void lstm_dx_kernel(const float *dG, LstmWeights w, float *dx, const float *dx_add, int T, int ndir) {
    __shared__ float gsm[2 * LSTM_G];
    __shared__ float part[4][LSTM_H];
    const int t = blockIdx.x, c = threadIdx.x & 127, k = threadIdx.x >> 7;
    const int R = ndir * LSTM_G;
    for (int e = threadIdx.x; e < R; e += 512) gsm[e] = dG[((long)(e / LSTM_G) * T + t) * LSTM_G + e % LSTM_G];
    __syncthreads();
    float acc = 0.f;
    const int rows = R / 4;
    for (int r0 = k * rows; r0 < (k + 1) * rows; r0 += 32) {   // (a quarter never straddles the two directions: rows = 128 or 256)
        const float *wd = w.w_ih[r0 / LSTM_G] + (long)(r0 % LSTM_G) * LSTM_H + c;
        float wv[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) wv[j] = wd[(long)j * LSTM_H];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc += gsm[r0 + j] * wv[j];
    }
    part[k][c] = acc;
    __syncthreads();
    const float base = dx_add ? dx_add[(long)t * LSTM_H + c] : 0.f;   // (requested before the row walk would be nicer; it is one load)
    if (k == 0) dx[(long)t * LSTM_H + c] = base + ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c]));
}
