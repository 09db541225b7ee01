// Persistent bidirectional LSTM for the s-head's sequence encoder (reference src/mucon/models.py:195-201,
// 605-611: nn.LSTM(128 -> 128, batch_first, bidirectional) over the temporally encoded video [1 x Tz x 128]).
// SURVEY.md 8f row 1.  Batch 1, input = hidden = 128 (the reference's only configuration); gate order and
// arithmetic are torch.nn.LSTM's:  gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh,  i,f,o = sigmoid, g = tanh,
// c_t = f c_{t-1} + i g,  h_t = o tanh(c_t),  h_0 = c_0 = 0;  the reverse direction walks t = T-1 .. 0.
//
// MIOpen needs 5.9 ms for this forward and 11.9 ms with the backward (Tz = 125): the work is a chain of Tz
// dependent 512x128 mat-vecs, i.e. pure latency.  Here ONE workgroup per direction stays resident for the
// whole sequence: 512 threads, each keeps 128 weights of W_hh in registers (forward: a quarter of each of a unit's four
// gate rows; backward: a 32 x 4 tile), h_{t-1} / the gate gradients are read from LDS, partial sums and the four gates of a
// hidden unit meet through DPP inside a quad / a row of lanes; one barrier per time step.
//   lstm_inproj_kernel      Gx[d][t][r] = W_ih[d][r] . x[t] + b_ih[d][r] + b_hh[d][r]     (all t at once)
//   lstm_recur_fwd_kernel   the recurrence; saves gate activations and cell states for the backward
//   lstm_recur_bwd_kernel   BPTT: a lane keeps a 32 x 4 tile of W_hh, dh_{t-1} = W_hh^T dgates_t stays in registers (DPP row sums)
//   lstm_wgrad_dx_kernel    (r6: one launch) dW_ih, dW_hh, db from the saved pre-activation gradients dG (reductions over t: lstm_wgrad_body)
//                           and dx[t] = sum_d W_ih[d]^T dG[d][t] (lstm_dx_body)
#pragma once
#include "common.hpp"
#include "head_body.hpp"

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

// DPP moves inside a quad / a row of 16 lanes (VALU, no LDS trip)
template <int CTRL>
__device__ __forceinline__ float lstm_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v) {   // the same bits in all four lanes (float addition commutes)
    v += lstm_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += lstm_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
// acc += w * g.lo / acc += w * g.hi for both halves of a packed pair: the op_sel modifiers broadcast one half of the second
// operand, which a C expression f32x2{g, g} pays a v_mov for
__device__ __forceinline__ void pk_fma_lo(f32x2 &acc, f32x2 w, f32x2 g) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(g));
}
__device__ __forceinline__ void pk_fma_hi(f32x2 &acc, f32x2 w, f32x2 g) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(g));
}

constexpr int LSTM_IP_T = 4;   // time steps per workgroup of the input projection (8 left 32 workgroups for T = 125: 15 us)
// grid (ceil(T/LSTM_IP_T), ndir), 512 threads.  The recurrence's ownership: a quad of lanes keeps the four gate rows of one
// hidden unit, lane p the columns {16 k + 4 p .. + 3 : k = 0..7} of each -- the four lanes of a quad load 64 contiguous bytes
// of a row (a lane per whole row touched 64 cache lines per load instruction: 8 us of the kernel's 13), the partial sums meet
// through DPP, lane p writes gate p.
__global__ __launch_bounds__(512) void lstm_inproj_kernel(const float *x, LstmWeights w, float *Gx, int T) {
    __shared__ __attribute__((aligned(16))) float xs[LSTM_IP_T][LSTM_H];
    const int d = blockIdx.y, t0 = blockIdx.x * LSTM_IP_T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 3, unit = wave * 16 + (lane >> 2);
    float wr[4][32];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(w.w_ih[d] + (long)(q * LSTM_H + unit) * LSTM_H + 16 * k + 4 * p);
            wr[q][k * 4 + 0] = v[0];
            wr[q][k * 4 + 1] = v[1];
            wr[q][k * 4 + 2] = v[2];
            wr[q][k * 4 + 3] = v[3];
        }
    for (int e = threadIdx.x; e < LSTM_IP_T * LSTM_H; e += 512) {
        const int tt = t0 + e / LSTM_H;
        xs[e / LSTM_H][e % LSTM_H] = tt < T ? x[(long)tt * LSTM_H + e % LSTM_H] : 0.f;
    }
    __syncthreads();
    const int r = p * LSTM_H + unit;
    const float bias = w.b_ih[d][r] + w.b_hh[d][r];
    for (int i = 0; i < LSTM_IP_T; ++i) {
        const int t = t0 + i;
        if (t >= T) break;
        f32x2 a[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xs[i][16 * k + 4 * p]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = f32x2{wr[q][k * 4 + 0], wr[q][k * 4 + 1]} * f32x2{xv[0], xv[1]} + a[q];
                a[q] = f32x2{wr[q][k * 4 + 2], wr[q][k * 4 + 3]} * f32x2{xv[2], xv[3]} + a[q];
            }
        }
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = quad_sum(a[q][0] + a[q][1]);
        const float mine = p == 0 ? pre[0] : p == 1 ? pre[1] : p == 2 ? pre[2] : pre[3];
        Gx[((long)d * T + t) * LSTM_G + r] = mine + bias;
    }
}

// grid (ndir), 512 threads.  out [T][ndir*128]; saves: gates [ndir][T][4][128] (post-activation i,f,g,o),
// cells [ndir][T][128]; hn / cn [ndir][128].
// Thread layout: a QUAD of lanes owns one hidden unit (wave w: units 16w .. 16w+15); lane p of the quad keeps the columns
// {16 k + 4 p .. + 3 : k = 0..7} of the unit's four gate rows of W_hh (4 x 32 = 128 registers; the four lanes of a quad load 64
// contiguous bytes of a row).  A step is bound by what LDS returns to the registers (128 B/clk per CU, broadcast or not): with a
// whole row per lane every lane read all 128 values of h -- 256 KB per step, ~2,000 clocks; a quarter row per lane reads 32 of
// them (64 KB; a quad reads 64 contiguous bytes, every quad the same: conflict-free).  The four partial sums of a gate meet
// through two DPP adds inside the quad, lane p applies the non-linearity of gate p (tanh x = 2 sigmoid(2x) - 1: branch-free),
// the four activations are passed round the quad by DPP and all four lanes update the cell.  One barrier per time step (the
// new h for everybody).
constexpr int LSTM_SEG = 36;   // backward: pitch of the sixteen 32-float segments of the gate gradients in LDS
__global__ __launch_bounds__(512) void lstm_recur_fwd_kernel(const float *Gx, LstmWeights w, float *out, float *gates,
                                                             float *cells, float *hn, float *cn, int T, int ndir, const HeadFwdArgs ha, const int hgx, const int hblocks) {
    if ((int)blockIdx.x >= ndir) {   // (r6) workgroups behind the directions: a deferred y-head forward (head_body.hpp, H = 128), two of its 256-thread blocks each
        __shared__ __attribute__((aligned(16))) float hsm[2][HF_Z * 128 + 2 * HF_Z * HEAD_MAXC];
        const int half = (int)(threadIdx.x >> 8);
        const int blk = ((int)blockIdx.x - ndir) * 2 + half;
        const bool active = blk < hblocks;
        head_fwd_z_body(ha, hsm[half], active ? blk % hgx : 0, active ? blk / hgx : 0, (int)(threadIdx.x & 255), active);
        return;
    }
    __shared__ __attribute__((aligned(16))) float hs[2][LSTM_H];
    const int d = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 3, unit = wave * 16 + (lane >> 2);
    float wr[4][32];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(w.w_hh[d] + (long)(q * LSTM_H + unit) * LSTM_H + 16 * c4 + 4 * p);
            wr[q][c4 * 4 + 0] = v[0];
            wr[q][c4 * 4 + 1] = v[1];
            wr[q][c4 * 4 + 2] = v[2];
            wr[q][c4 * 4 + 3] = v[3];
        }
    if (tid < LSTM_H) hs[0][tid] = 0.f;
    float c = 0.f;  // cell state of `unit` (identical in the four lanes)
    const int r = p * LSTM_H + unit;   // the gate row whose input projection / activation this lane handles
    const float *gx = Gx + (long)d * T * LSTM_G + r;
    float gnext = gx[(long)(d == 0 ? 0 : T - 1) * LSTM_G];
    const float asc = p == 2 ? 2.f : 1.f;   // gate g: tanh through the sigmoid
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int cur = s & 1;
        const float g0 = gnext;
        gnext = gx[(long)(d == 0 ? min(s + 1, T - 1) : max(T - 2 - s, 0)) * LSTM_G];  // next step's input projection: in flight (unconditional: see the backward's load_step)
        // packed FMAs (v_pk_fma_f32: two per lane and instruction), one accumulator pair per gate
        f32x2 a[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        // all eight reads of h first (the compiler pairs them with their FMAs otherwise, two in flight: four LDS latencies per step; 80.6 -> 76 us at Tz = 125)
        f32x4 hv[8];
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) hv[c4] = *reinterpret_cast<const f32x4 *>(&hs[cur][16 * c4 + 4 * p]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = f32x2{wr[q][c4 * 4 + 0], wr[q][c4 * 4 + 1]} * f32x2{hv[c4][0], hv[c4][1]} + a[q];
                a[q] = f32x2{wr[q][c4 * 4 + 2], wr[q][c4 * 4 + 3]} * f32x2{hv[c4][2], hv[c4][3]} + a[q];
            }
        }
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = quad_sum(a[q][0] + a[q][1]);
        const float mine = p == 0 ? pre[0] : p == 1 ? pre[1] : p == 2 ? pre[2] : pre[3];
        const float sg = sigmoid_f(asc * (g0 + mine));
        const float act = p == 2 ? 2.f * sg - 1.f : sg;
        gates[((long)d * T + t) * LSTM_G + r] = act;
        const float gi = lstm_dpp<0x00>(act), gf = lstm_dpp<0x55>(act), gg = lstm_dpp<0xAA>(act), go = lstm_dpp<0xFF>(act);
        c = gf * c + gi * gg;
        const float h = go * tanh_f(c);
        if (p == 0) {
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

// ------------------------------------------------------------------------------------------------------------------------
// The decoder's weight gradients as a batch of outer products (decoder.hpp: dec_outer_kernel).  Defined here because the batch
// can ride in THIS header's backward recurrence launch (r6: mucon_decoder_bwd_defer): nobody waits for these sums before the
// optimizer step, and that launch keeps two workgroups busy for ~89 us while the rest of the chip idles.
//   out[i][j] = sum_n A[n*lda + i] B[n*ldb + j];  bias[i] = sum_n A[n*lda + i]
// One thread per output element, no barriers: `block` / `tid` = a 256-thread block of the stand-alone launch.
struct OuterJob {
    const float *A, *B;
    float *out, *bias, *bias2;
    int lda, ldb, ra, cb, n, block0;
};
constexpr int DEC_MAXJOBS = 12;
struct OuterBatch {
    OuterJob job[DEC_MAXJOBS];
    int njobs;
    int nblocks;     // 256-thread blocks of the whole batch (0: nothing)
};
__device__ __forceinline__ void dec_outer_body(const OuterBatch &ob, const int block, const int tid) {
#pragma clang fp contract(off)   // (explicit fmaf only: both launches that inline this body must produce the same bits)
    int ji = 0;
    while (ji + 1 < ob.njobs && block >= ob.job[ji + 1].block0) ++ji;
    const OuterJob &jb = ob.job[ji];
    const long e = (long)(block - jb.block0) * 256 + tid;
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
                acc = __builtin_fmaf(av[q], bv[q], acc);
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


// grid (ndir), 512 threads.  dG [ndir][T][512]: gradient at the gate pre-activations (input of the weight / input gradients).
// A ROW of 16 lanes (the unit DPP row operations work on) owns four hidden units j0 .. j0+3 (wave w: units 16w .. 16w+15)
// in both roles of a step:
//   * gate role: lane rho = 4 (unit - j0) + q derives the pre-activation gradient of gate q of its unit (the four lanes of a
//     unit carry identical copies of dc) and posts it in LDS;
//   * mat-vec role, dh_{t-1} = W_hh^T dgates_t: lane rho keeps W_hh[32 rho .. 32 rho + 31][j0 .. j0+3] (128 registers), reads
//     its 32 gate gradients from LDS (64 KB per step for the workgroup instead of 256 KB with a whole column block per
//     thread -- what LDS returns to the registers bounds the step), and the 16 partial sums of a column meet through four
//     DPP adds inside the row -- after which every lane of the row holds dh of all four units, its own included.
// So the recurrent dh never leaves the registers and a step needs ONE barrier (the gate gradients for everybody).
// The gate gradients sit in LDS as sixteen 32-float segments 36 floats apart: the 16 lanes of a row read 16 different
// 16-byte bank groups.
__global__ __launch_bounds__(512) void lstm_recur_bwd_kernel(LstmWeights w, const float *out, const float *gates,
                                                             const float *cells, const float *d_out, const float *d_hn,
                                                             const float *d_cn, float *dG, int T, int ndir, const OuterBatch ob) {
    if ((int)blockIdx.x >= ndir) {   // (r6) a deferred batch of the decoder's outer products: two of its 256-thread blocks per workgroup
        const int blk = ((int)blockIdx.x - ndir) * 2 + (int)(threadIdx.x >> 8);
        if (blk < ob.nblocks) dec_outer_body(ob, blk, (int)(threadIdx.x & 255));
        return;
    }
    __shared__ __attribute__((aligned(16))) float dgs[2][16 * LSTM_SEG];
    const int d = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int rho = lane & 15, j0 = wave * 16 + (lane >> 4) * 4;
    const int q = rho & 3, unit = j0 + (rho >> 2);          // gate role
    f32x2 wt[32][2];   // [row][columns (0,1) | (2,3)]
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w.w_hh[d] + (long)(32 * rho + i) * LSTM_H + j0);
        wt[i][0] = f32x2{v[0], v[1]};
        wt[i][1] = f32x2{v[2], v[3]};
    }
    float dc = d_cn ? d_cn[d * LSTM_H + unit] : 0.f;
    float dh_rec = d_hn ? d_hn[d * LSTM_H + unit] : 0.f;    // the recurrent dh of the first processed step = d_hn
    const int grow = q * LSTM_H + unit;                      // this lane's gate row ...
    const int gslot = (grow >> 5) * LSTM_SEG + (grow & 31);  // ... and its place in the padded LDS vector
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
        // every load unconditional (clamped row, the value dropped by a select): a load behind a branch makes the number of loads in flight depend on
        // the path, and the compiler then waits for (nearly) all of them in front of the step that needs only the older set -- the read-ahead would
        // be waited for one step early
        const float cpv = cells[((long)d * T + (s > 0 ? tp : t)) * LSTM_H + unit];
        const float dov = (d_out ? d_out : out)[(long)t * (ndir * LSTM_H) + d * LSTM_H + unit];
        v.cp = s > 0 ? cpv : 0.f;
        v.dout = d_out ? dov : 0.f;
        return v;
    };
    // The gate role without divergent paths: which gate a lane derives is fixed for the whole launch (q = rho & 3), so the four formulas
    //   q = 0: dct gg gi (1 - gi)    q = 1: dct cp gf (1 - gf)    q = 2: dct gi (1 - gg^2)    q = 3: dh th go (1 - go)
    // are ONE, dp = X Y (Z' - Z^2) with lane-constant selections X in {dct, dh}, Y in {gg, cp, gi, th}, Z in {gi, gf, gg, go}, Z' = q == 2 ? 1 : Z
    // (four if-else paths cost the wave the sum of all four plus the exec bookkeeping: the step is bound by vector-instruction issue).
    const bool q0 = q == 0, q1 = q == 1, q2 = q == 2, q3 = q == 3;
    auto step = [&](const int s, const StepIn &in) {
        const int t = d == 0 ? s : T - 1 - s;
        const int cur = s & 1;                   // double-buffered: a fast wave may post step s-1 while a slow one still reads s
        {
            const float dh = in.dout + dh_rec;
            const float th = tanh_f(in.ct);
            const float dct = dc + dh * in.go * (1.f - th * th);
            dc = dct * in.gf;
            const float X = q3 ? dh : dct;
            const float Y = q0 ? in.gg : q1 ? in.cp : q2 ? in.gi : th;
            const float Z = q0 ? in.gi : q1 ? in.gf : q2 ? in.gg : in.go;
            const float dp = X * Y * fmaf(-Z, Z, q2 ? 1.f : Z);
            dgs[cur][gslot] = dp;
            dG[((long)d * T + t) * LSTM_G + grow] = dp;
        }
        __syncthreads();
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};     // packed FMAs: columns (0, 1) and (2, 3)
#pragma unroll
        for (int r4 = 0; r4 < 8; ++r4) {
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(&dgs[cur][rho * LSTM_SEG + r4 * 4]);   // (all eight reads in front, as in the forward: slower, 106 us against 103)
            const f32x2 g01 = {gv[0], gv[1]}, g23 = {gv[2], gv[3]};
            pk_fma_lo(a0, wt[r4 * 4 + 0][0], g01);
            pk_fma_lo(a1, wt[r4 * 4 + 0][1], g01);
            pk_fma_hi(a0, wt[r4 * 4 + 1][0], g01);
            pk_fma_hi(a1, wt[r4 * 4 + 1][1], g01);
            pk_fma_lo(a0, wt[r4 * 4 + 2][0], g23);
            pk_fma_lo(a1, wt[r4 * 4 + 2][1], g23);
            pk_fma_hi(a0, wt[r4 * 4 + 3][0], g23);
            pk_fma_hi(a1, wt[r4 * 4 + 3][1], g23);
        }
        float col[4] = {a0[0], a0[1], a1[0], a1[1]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {               // sum over the 16 lanes of the row: the same bits in all of them
            col[k] += lstm_dpp<0xB1>(col[k]);       // quad_perm [1,0,3,2]
            col[k] += lstm_dpp<0x4E>(col[k]);       // quad_perm [2,3,0,1]
            col[k] += lstm_dpp<0x141>(col[k]);      // row_half_mirror
            col[k] += lstm_dpp<0x140>(col[k]);      // row_mirror
        }
        dh_rec = rho < 4 ? col[0] : rho < 8 ? col[1] : rho < 12 ? col[2] : col[3];
    };
    // two steps per trip with two register sets for the saved activations: the single-set loop copied a set per step (15 v_mov)
    StepIn sa = load_step(T - 1), sb = sa;
    for (int s = T - 1; s >= 0; s -= 2) {        // reverse of the processing order
        sb = load_step(max(s - 1, 0));          // (s = 0: re-loads step 0, nobody uses it)
        step(s, sa);
        if (s == 0) break;
        sa = load_step(max(s - 2, 0));
        step(s - 1, sb);
    }
}

// dW_ih[d][r][c] = sum_t dG[d][t][r] x[t][c];  dW_hh[d][r][c] = sum_t dG[d][t][r] hprev[d][t][c];  db = sum_t dG[d][t][r]
// grid (512/4, ndir), 512 threads = 4 gate rows x 128 columns.
// (r6) x and the previous hidden states of LSTM_WG_T time steps are staged in LDS once per workgroup (coalesced 16-byte loads: ONE memory round trip per 64 steps; the
// workgroup's four gradient rows beside them) and the sums run from LDS in time order -- the per-thread loop over 32 steps' worth of scattered loads was four dependent round
// trips at T = 125 (14.4 us).  Same terms in the same order.
constexpr int LSTM_WG_T = 64;    // (65 KB of LDS: two workgroups of the merged launch per CU -- the launch-wide dynamic allocation is also what the input-gradient workgroups reserve)
constexpr size_t LSTM_WG_LDS_BYTES = sizeof(float) * (2 * LSTM_WG_T * LSTM_H + 4 * LSTM_WG_T);
__device__ __forceinline__ void lstm_wgrad_body(const int bx, const int d, const float *dG, const float *x, const float *out, const LstmGrads &g, int T, int ndir, float *sm) {
    float *xs = sm, *hs = sm + LSTM_WG_T * LSTM_H, *gs = hs + LSTM_WG_T * LSTM_H;   // [step][128], [step][128], [step][4 rows]
    const int tid = threadIdx.x;
    const int rq = tid >> 7, r = bx * 4 + rq, c = tid & 127;
    float ai = 0.f, ah = 0.f, ab = 0.f;
    for (int s0 = 0; s0 < T; s0 += LSTM_WG_T) {
        const int ns = min(LSTM_WG_T, T - s0);
        if (s0) __syncthreads();     // (the previous chunk's readers are done)
        // step s of the sums is time t = s (forward direction) or T - 1 - s (backward); its "previous" state is out[t - 1] / out[t + 1]
        for (int e = tid; e < ns * (LSTM_H / 4); e += 512) {
            const int j = e >> 5, c4 = (e & 31) * 4;
            const int sidx = s0 + j;
            const int t = d == 0 ? sidx : T - 1 - sidx;
            const int tp = d == 0 ? max(t - 1, 0) : min(t + 1, T - 1);
            *reinterpret_cast<f32x4 *>(xs + j * LSTM_H + c4) = *reinterpret_cast<const f32x4 *>(x + (long)t * LSTM_H + c4);
            *reinterpret_cast<f32x4 *>(hs + j * LSTM_H + c4) = *reinterpret_cast<const f32x4 *>(out + (long)tp * (ndir * LSTM_H) + d * LSTM_H + c4);
        }
        for (int e = tid; e < ns * 4; e += 512) {
            const int j = e >> 2, q = e & 3;
            const int sidx = s0 + j;
            const int t = d == 0 ? sidx : T - 1 - sidx;
            gs[j * 4 + q] = dG[((long)d * T + t) * LSTM_G + bx * 4 + q];
        }
        __syncthreads();
        for (int j = 0; j < ns; ++j) {
            const float gv = gs[j * 4 + rq];
            ai += gv * xs[j * LSTM_H + c];
            if (s0 + j > 0) ah += gv * hs[j * LSTM_H + c];
            ab += gv;
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
__device__ __forceinline__ void lstm_dx_body(const int t, const float *dG, const LstmWeights &w, float *dx, const float *dx_add, int T, int ndir) {
    __shared__ float gsm[2 * LSTM_G];
    __shared__ float part[4][LSTM_H];
    const int c = threadIdx.x & 127, k = threadIdx.x >> 7;
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

// (r6) The two launches behind the backward recurrence in ONE: they read the same dG and do not depend on each other.  Workgroups [0, T) take the input
// gradient's rows (what the encoder's backward waits for: dealt first), the next (LSTM_G / 4) x ndir the weight gradients' row quads -- 7.4 + 14.4 us of two
// launches become the longer one's.  Same arithmetic per workgroup as the two kernels it replaces (bitwise).
__global__ __launch_bounds__(512) void lstm_wgrad_dx_kernel(const float *dG, const float *x, const float *out, const LstmGrads g, const LstmWeights w, float *dx,
                                                            const float *dx_add, const int T, const int ndir) {
    extern __shared__ __attribute__((aligned(16))) float wg_sm[];   // (the weight-gradient workgroups' staging tiles: LSTM_WG_LDS_BYTES)
    const int b = blockIdx.x;
    if (b < T) return lstm_dx_body(b, dG, w, dx, dx_add, T, ndir);
    const int q = b - T;
    lstm_wgrad_body(q % (LSTM_G / 4), q / (LSTM_G / 4), dG, x, out, g, T, ndir, wg_sm);
}
