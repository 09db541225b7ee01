// The four MuCon losses and their gradients in three launches (four until r6: loss_mid_kernel is now a function of loss_grad_kernel).
//   reference src/mucon/masks.py:8-74          project_lengths_softmax, create_masks (affine_grid + grid_sample)
//   reference src/mucon/models.py:376-396      loss(): main = sum of multiplier * component
//   reference src/mucon/models.py:398-412      smoothing loss (mse of consecutive rows, right side detached, clamp)
//   reference src/mucon/models.py:414-515      mucon loss: "flint" (mask-averaged logits per segment -> log-softmax ->
//                                              nll) and "arithmetic" (mask-weighted per-frame cross-entropy / T)
//   reference src/mucon/models.py:517-565      length loss (hinge at +-width), transcript loss (nll, optional weights)
// The loss is a scalar, so the forward computes the gradients too (scaled by the multipliers); the autograd backward
// only multiplies by the upstream scalar.
//
// Masks.  mask[n][t] = bilinear sample of a 100-point template at ix, zero outside the template (torch's affine_grid /
// grid_sample; the sampled image has one row, y = 0 hits it exactly), in either convention of those two functions:
//   align_corners = 1 (PyTorch <= 1.2, i.e. the 1.1 the reference pins in docker/pytorch1.1/Dockerfile -- the upstream recipe):
//       x = scale_n * (2t / (T - 1) - 1) + shift_n,   ix = (x + 1) / 2 * 99
//   align_corners = 0 (the default since 1.3, what the reference's unchanged code computes under a current torch):
//       x = scale_n * ((2t + 1) / T - 1) + shift_n,   ix = ((x + 1) * 100 - 1) / 2
// With
//   A = T softmax(lengths), start = cumsum(A) - A, L = A (1 + 2 ov), start' = start - L ov / 2,
//   scale = T / L, shift = (start' + L/2 - T/2) / (-L/2)
// (create_masks rescales the lengths IN PLACE, so the "flint" division uses L, not A).
//
//   loss_acc_kernel    T/32 wgs    : every workgroup derives the segment geometry itself (64 lanes of arithmetic; workgroup 0 also writes it out,
//                                    with the transcript + length losses and their gradients: until r4 a launch of its own, loss_prep_kernel),
//                                    then: masks of its 32 frames, partial windows / arithmetic sums / smoothing sums -> slabs
//   loss_grad_kernel   T/32 wgs    : (loss_mid_body, every workgroup for itself: ordered slab reduction, per-segment log-softmax, the five loss values, d windows), then
//                                    d segmentation, d smoothing input, d mask -> partial d scale / d shift slabs
//   loss_fin_kernel    1 workgroup : ordered slab reduction, chain through the geometry and the softmax -> d lengths
// Every reduction has a fixed order: results are bitwise reproducible.
#pragma once
#include "common.hpp"

constexpr int LOSS_FB = 32;      // frames per workgroup
constexpr int LOSS_MAXN = 64;    // segments
constexpr int LOSS_MAXM = 64;    // classes
constexpr int LOSS_TW = 100;     // template width (masks.py: TEMPLATE_WIDTH)

struct LossDims {
    int T, M, N, S, NC;
    int mucon_type;          // 0 flint, 1 arithmetic
    int smoothing_clamp, transcript_average;
    float overlap, clamp_min, clamp_max, length_width;
    float mul_transcript, mul_length, mul_mucon, mul_smoothing;
    int align_corners;       // convention of affine_grid / grid_sample: 1 = PyTorch <= 1.2 (the reference's pinned 1.1), 0 = the later default
};

struct LossBufs {
    // inputs
    const float *seg;        // [T][M] segmentation logits
    const float *sx;         // [T][M] what the smoothing loss runs on (log-probs or logits)
    const float *tlogp;      // [S][NC] transcript log-probs
    const float *lengths;    // [N] raw length logits
    const long *mtarget;     // [N] mucon target classes
    const long *ttarget;     // [S] transcript targets
    const float *tmpl;       // [100]
    const float *mweight;    // [M] or null
    const float *tweight;    // [NC] or null
    // outputs
    float *losses;           // [5] main, transcript, length, mucon, smoothing
    float *d_seg, *d_sx, *d_tlogp, *d_lengths;
    // workspace
    float *geo;              // [6][LOSS_MAXN]: L, scale, shift, start', p (softmax), w[target]
    double *geod;            // [2][LOSS_MAXN]: scale, shift in double (what the masks are sampled with)
    float *small;            // [8]: transcript loss, length loss, sum of mucon weights
    float *slab;             // [chunks][N*M + 2]  (windows partials | arithmetic partial | smoothing partial)
    float *gwin;             // [N][M] d loss / d (mask-sum), already divided by L and scaled
    float *glwin;            // [N]    d loss / d L through the window division
    float *gsm;              // [1]    smoothing gradient scale
    float *gslab;            // [chunks][N][2]
};

__device__ __forceinline__ float loss_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double loss_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum of p[c * stride] over c = 0 .. n-1 in order, 32 loads in flight at a time (a term-at-a-time loop over the T/32 slab
// chunks is a chain of dependent memory round trips: 25 us of the 29 us loss_mid_kernel took)
__device__ __forceinline__ float loss_ordered_sum(const float *p, long stride, int n) {
    float acc = 0.f;
    constexpr int Q = 32;      // (eight in flight: still eight dependent round trips for the 63 chunks of a T = 2000 video)
    for (int c0 = 0; c0 < n; c0 += Q) {
        float v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = p[(long)min(c0 + q, n - 1) * stride];
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (c0 + q < n) acc += v[q];
    }
    return acc;
}

// mask value and its derivative w.r.t. the pixel coordinate ix
// (align_corners: base grid linspace(-1, 1, T) and ix = (x + 1) / 2 * (W - 1), the corner pixels' CENTRES at -1 and 1; otherwise
// the half-pixel forms.  d ix / d x is mask_dix(ac).)
__device__ __forceinline__ float mask_dix(int ac) { return ac ? (float)(LOSS_TW - 1) * 0.5f : (float)LOSS_TW * 0.5f; }
// The sampling coordinate in DOUBLE (r4).  A segment of L frames maps a frame step to 100 / L template samples, and the gradient of a bilinearly
// sampled template w.r.t. the coordinate is piecewise constant: a frame whose ix sits within float32 rounding of a sample point (the start of
// a segment is a float32 cumsum of lengths up to T ~ 1e4, ulp 1e-3 frames = 4e-3 samples at L = 24) takes the slope of the wrong interval, and
// with two dozen frames per segment one such frame moved d lengths by 1 % (found by tests/test_gpu_fuzz.py: gaussian template, 49 segments,
// T = 10,965; torch's float32 path has the same sensitivity, at other frames).  In double the kernels sample what the float64 oracle samples.
__device__ __forceinline__ void mask_sample(const float *tmpl, double scale, double shift, int t, int T, int ac, float &val, float &dval,
                                            float &xb) {
    const double xbd = ac ? (T > 1 ? 2.0 * (double)t / (double)(T - 1) - 1.0 : 0.0) : (2.0 * (double)t + 1.0) / (double)T - 1.0;
    xb = (float)xbd;
    const double x = scale * xbd + shift;
    const double ixd = ac ? (x + 1.0) * 0.5 * (double)(LOSS_TW - 1) : ((x + 1.0) * (double)LOSS_TW - 1.0) * 0.5;
    const double f0d = floor(ixd);
    const float f0 = (float)fmin(fmax(f0d, -4.0), (double)LOSS_TW + 4.0);
    const float fx = (float)(ixd - f0d);
    // clamp before the int conversion: far-away segments give huge |ix|
    const int i0 = (int)fminf(fmaxf(f0, -2.f), (float)LOSS_TW + 1.f);
    const float t0 = (i0 >= 0 && i0 < LOSS_TW) ? tmpl[i0] : 0.f;
    const float t1 = (i0 + 1 >= 0 && i0 + 1 < LOSS_TW) ? tmpl[i0 + 1] : 0.f;
    val = t0 * (1.f - fx) + t1 * fx;
    dval = t1 - t0;
}

// Wave 0 of every loss_acc_kernel workgroup (64 lanes; lane = segment).  s_geo [6][LOSS_MAXN]: L, scale, shift, start', p (softmax), w[target].
// write_out: workgroup 0 -- the same values to b.geo, the transcript / length losses and the weight sum to b.small, and d_tlogp.
__device__ __forceinline__ void loss_geometry(const LossDims &d, const LossBufs &b, float (*s_geo)[LOSS_MAXN], double (*s_geod)[LOSS_MAXN], bool write_out) {
    const int lane = threadIdx.x;
    const int N = d.N;
    // absolute lengths = T softmax(lengths); start = cumsum(A) - A; scale / shift of the sampling grid -- in double (see mask_sample)
    const float l = lane < N ? b.lengths[lane] : -INFINITY;
    double mxd = (double)l;
#pragma unroll
    for (int o = 32; o; o >>= 1) mxd = fmax(mxd, __shfl_xor(mxd, o));
    const double ed = lane < N ? exp((double)l - mxd) : 0.0;
    const double pd = ed / loss_wave_sum_d(ed);
    const double Ad = (double)d.T * pd;
    double csd = Ad;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(csd, o);
        if (lane >= o) csd += up;
    }
    const double startd = csd - Ad;
    const double Ld = Ad * (1.0 + 2.0 * (double)d.overlap);
    const double startpd = startd - Ld * ((double)d.overlap * 0.5);
    const double scaled = (double)d.T / Ld;
    const double shiftd = (startpd + Ld * 0.5 - (double)d.T * 0.5) / (-(Ld * 0.5));
    const float p = (float)pd, L = (float)Ld, startp = (float)startpd, scale = (float)scaled, shift = (float)shiftd;
    if (lane < N) {
        s_geod[0][lane] = scaled;
        s_geod[1][lane] = shiftd;
        if (write_out) {
            b.geod[lane] = scaled;
            b.geod[LOSS_MAXN + lane] = shiftd;
        }
    }
    float wt = 1.f;
    if (lane < N) {
        const int tg = (int)b.mtarget[lane];
        wt = b.mweight ? b.mweight[tg] : 1.f;
        const float v[6] = {L, scale, shift, startp, p, wt};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            s_geo[k][lane] = v[k];
            if (write_out) b.geo[k * LOSS_MAXN + lane] = v[k];
        }
    }
    if (!write_out) return;
    const float wsum = loss_wave_sum(lane < N ? wt : 0.f);
    // length loss: relu(s - w).sum() + relu(-w - s).sum()
    float ll = 0.f;
    if (lane < N) ll = fmaxf(l - d.length_width, 0.f) + fmaxf(-d.length_width - l, 0.f);
    ll = loss_wave_sum(ll);
    // transcript loss: nll(tlogp, ttarget, weight)
    float num = 0.f, den = 0.f;
    for (int s = lane; s < d.S; s += 64) {
        const int tg = (int)b.ttarget[s];
        const float w = b.tweight ? b.tweight[tg] : 1.f;
        num -= w * b.tlogp[(long)s * d.NC + tg];
        den += w;
    }
    num = loss_wave_sum(num);
    den = loss_wave_sum(den);
    const float tl = d.transcript_average ? num / den : num;
    for (int e2 = lane; e2 < d.S * d.NC; e2 += 64) {
        const int s = e2 / d.NC, c = e2 - s * d.NC;
        const int tg = (int)b.ttarget[s];
        float g = 0.f;
        if (c == tg) {
            const float w = b.tweight ? b.tweight[tg] : 1.f;
            g = -(d.transcript_average ? w / den : w) * d.mul_transcript;
        }
        b.d_tlogp[e2] = g;
    }
    if (lane == 0) {
        b.small[0] = tl;
        b.small[1] = ll;
        b.small[2] = wsum;
    }
}

// grid (chunks), 256 threads.  LDS: seg chunk [FB][M], masks [N][FB]
__global__ __launch_bounds__(256) void loss_acc_kernel(LossDims d, LossBufs b) {
    __shared__ float s_geo[6][LOSS_MAXN];
    __shared__ double s_geod[2][LOSS_MAXN];
    __shared__ float s_seg[LOSS_FB][LOSS_MAXM + 1];
    __shared__ float s_mask[LOSS_MAXN][LOSS_FB + 1];
    __shared__ float s_lse[LOSS_FB];
    __shared__ float s_red[4];
    const int tid = threadIdx.x, t0 = blockIdx.x * LOSS_FB;
    const int T = d.T, M = d.M, N = d.N;
    const int nf = min(LOSS_FB, T - t0);
    if (tid < 64) loss_geometry(d, b, s_geo, s_geod, blockIdx.x == 0);
    __syncthreads();
    for (int e = tid; e < LOSS_FB * M; e += 256) {
        const int f = e / M, m = e - f * M;
        s_seg[f][m] = f < nf ? b.seg[(long)(t0 + f) * M + m] : 0.f;
    }
    for (int e = tid; e < N * LOSS_FB; e += 256) {
        const int n = e / LOSS_FB, f = e - n * LOSS_FB;
        float v = 0.f, dv, xb;
        if (f < nf) mask_sample(b.tmpl, s_geod[0][n], s_geod[1][n], t0 + f, T, d.align_corners, v, dv, xb);
        s_mask[n][f] = v;
    }
    // smoothing partial: sum over the chunk's frames t (t + 1 < T) of (x[t+1] - x[t])^2
    float sq = 0.f;
    for (int e = tid; e < nf * M; e += 256) {
        const int f = e / M, m = e - f * M;
        const int t = t0 + f;
        if (t + 1 < T) {
            const float df = b.sx[(long)(t + 1) * M + m] - b.sx[(long)t * M + m];
            sq += df * df;
        }
    }
    __syncthreads();
    float *slab = b.slab + (long)blockIdx.x * (N * M + 2);
    if (d.mucon_type == 0) {
        for (int e = tid; e < N * M; e += 256) {
            const int n = e / M, m = e - n * M;
            float acc = 0.f;
#pragma unroll 8
            for (int f = 0; f < LOSS_FB; ++f) acc += s_mask[n][f] * s_seg[f][m];
            slab[e] = acc;
        }
    } else {
        // per-frame log-sum-exp, then sum_n sum_f mask[n][f] * (-w_n * logp[f][target_n])
        if (tid < LOSS_FB) {
            float mx = -INFINITY;
            for (int m = 0; m < M; ++m) mx = fmaxf(mx, s_seg[tid][m]);
            float se = 0.f;
            for (int m = 0; m < M; ++m) se += expf(s_seg[tid][m] - mx);
            s_lse[tid] = mx + logf(se);
        }
        __syncthreads();
        float acc = 0.f;
        for (int e = tid; e < N * LOSS_FB; e += 256) {
            const int n = e / LOSS_FB, f = e - n * LOSS_FB;
            if (f < nf) {
                const int tg = (int)b.mtarget[n];
                acc += s_mask[n][f] * (-s_geo[5][n] * (s_seg[f][tg] - s_lse[f]));
            }
        }
        acc = loss_wave_sum(acc);
        if ((tid & 63) == 0) s_red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) slab[N * M] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        __syncthreads();
    }
    sq = loss_wave_sum(sq);
    if ((tid & 63) == 0) s_red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) slab[N * M + 1] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// The step between the two passes over the frames -- the chunk partials summed in order, the per-segment windows' log-softmax and its gradient, the loss scalars -- as a
// function every workgroup of loss_grad_kernel runs for itself (r6: it was a one-workgroup launch of its own, loss_mid_kernel, 10.8 us on the one-video step's critical path;
// the sums are over T / 32 chunks of N x M + 2 floats: recomputing them per workgroup costs less than the launch).  Same arithmetic in the same order in every workgroup;
// workgroup 0 (`write`) also leaves the results the later launch and the caller read (gwin, glwin, gsm, losses).  -> s_gwin (LDS) and the smoothing gradient's scale.
__device__ __forceinline__ float loss_mid_body(const LossDims &d, const LossBufs &b, const int chunks, float (*s_win)[LOSS_MAXM + 1], float (*s_gwin)[LOSS_MAXM + 1],
                                               float *s_seg_loss, float *s_scal, const bool write) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = d.N, M = d.M, stride = N * M + 2;
    // a thread's elements (tid, tid + 256, ... of the N x M windows; the two scalars ride as elements N M, N M + 1) are summed TOGETHER, two at a time: their loads
    // are independent, the chain of round trips is one element's (each element still adds its chunks in order)
    const int nel = (d.mucon_type == 0 ? N * M : 0), first_scal = N * M;
    for (int e0 = tid; e0 < N * M + 2; e0 += 512) {
        const int e1 = e0 + 256;
        const bool a0 = e0 < nel || e0 >= first_scal, a1 = e1 < N * M + 2 && (e1 < nel || e1 >= first_scal);
        float acc0 = 0.f, acc1 = 0.f;
        const float *p0 = b.slab + (a0 ? e0 : first_scal), *p1 = b.slab + (a1 ? e1 : first_scal);
        constexpr int Q = 16;
        for (int c0 = 0; c0 < chunks; c0 += Q) {
            float v0[Q], v1[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const long o = (long)min(c0 + q, chunks - 1) * stride;
                v0[q] = p0[o];
                v1[q] = p1[o];
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (c0 + q < chunks) {
                    acc0 += v0[q];
                    acc1 += v1[q];
                }
        }
        if (a0) {
            if (e0 >= first_scal) s_scal[e0 - first_scal] = acc0;
            else s_win[e0 / M][e0 % M] = acc0 / b.geo[0 * LOSS_MAXN + e0 / M];
        }
        if (a1) {
            if (e1 >= first_scal) s_scal[e1 - first_scal] = acc1;
            else s_win[e1 / M][e1 % M] = acc1 / b.geo[0 * LOSS_MAXN + e1 / M];
        }
    }
    __syncthreads();
    const float wsum = b.small[2];
    if (d.mucon_type == 0) {
        // a wave per segment: log-softmax of the window, nll with class weights (mean over segments)
        for (int n = wave; n < N; n += 4) {
            const float x = lane < M ? s_win[n][lane] : -INFINITY;
            float mx = x;
#pragma unroll
            for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float ex = lane < M ? expf(x - mx) : 0.f;
            const float se = loss_wave_sum(ex);
            const float lse = mx + logf(se);
            const int tg = (int)b.mtarget[n];
            const float wt = b.geo[5 * LOSS_MAXN + n];
            const float L = b.geo[0 * LOSS_MAXN + n];
            // d loss / d window
            float gw = 0.f;
            if (lane < M) gw = (ex / se - (lane == tg ? 1.f : 0.f)) * (wt / wsum) * d.mul_mucon;
            const float gl = loss_wave_sum(lane < M ? gw * x : 0.f);
            if (lane < M) {
                s_gwin[n][lane] = gw / L;
                if (write) b.gwin[n * M + lane] = gw / L;
            }
            if (lane == 0) {
                s_seg_loss[n] = -wt * (s_win[n][tg] - lse);
                if (write) b.glwin[n] = -gl / L;
            }
        }
    } else if (write) {
        for (int e = tid; e < N; e += 256) b.glwin[e] = 0.f;
    }
    __syncthreads();
    // (every thread: a handful of scalar operations on LDS values -- the smoothing gradient's scale is what the frames' pass needs)
    const float cnt = (float)(d.T - 1) * (float)M;
    float sm = s_scal[1] / cnt;
    float gs = 2.f / cnt;
    if (d.smoothing_clamp) {
        if (sm < d.clamp_min) {
            sm = d.clamp_min;
            gs = 0.f;
        } else if (sm > d.clamp_max) {
            sm = d.clamp_max;
            gs = 0.f;
        }
    }
    if (write && tid == 0) {
        float mu;
        if (d.mucon_type == 0) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc += s_seg_loss[n];
            mu = acc / wsum;
        } else {
            mu = s_scal[0] / (float)d.T;
        }
        b.gsm[0] = gs * d.mul_smoothing;
        const float tl = b.small[0], ll = b.small[1];
        b.losses[0] = d.mul_transcript * tl + d.mul_length * ll + d.mul_mucon * mu + d.mul_smoothing * sm;
        b.losses[1] = tl;
        b.losses[2] = ll;
        b.losses[3] = mu;
        b.losses[4] = sm;
    }
    return gs * d.mul_smoothing;
}

// grid (chunks), 256 threads
__global__ __launch_bounds__(256) void loss_grad_kernel(LossDims d, LossBufs b, int chunks) {
    __shared__ float s_seg[LOSS_FB][LOSS_MAXM + 1];
    __shared__ float s_mask[LOSS_MAXN][LOSS_FB + 1];
    __shared__ float s_dmask[LOSS_MAXN][LOSS_FB + 1];   // d mask / d ix
    __shared__ float s_xb[LOSS_FB];
    __shared__ float s_lse[LOSS_FB];
    __shared__ float s_gwin[LOSS_MAXN][LOSS_MAXM + 1];
    __shared__ float s_win[LOSS_MAXN][LOSS_MAXM + 1];
    __shared__ float s_seg_loss[LOSS_MAXN];
    __shared__ float s_scal[2];
    const int tid = threadIdx.x, t0 = blockIdx.x * LOSS_FB;
    const int T = d.T, M = d.M, N = d.N;
    const float gs = loss_mid_body(d, b, chunks, s_win, s_gwin, s_seg_loss, s_scal, blockIdx.x == 0);   // (ends behind a barrier: s_gwin is complete)
    const int nf = min(LOSS_FB, T - t0);
    for (int e = tid; e < LOSS_FB * M; e += 256) {
        const int f = e / M, m = e - f * M;
        s_seg[f][m] = f < nf ? b.seg[(long)(t0 + f) * M + m] : 0.f;
    }
    for (int e = tid; e < N * LOSS_FB; e += 256) {
        const int n = e / LOSS_FB, f = e - n * LOSS_FB;
        float v = 0.f, dv = 0.f, xb = 0.f;
        if (f < nf) mask_sample(b.tmpl, b.geod[n], b.geod[LOSS_MAXN + n], t0 + f, T, d.align_corners, v, dv, xb);
        s_mask[n][f] = v;
        s_dmask[n][f] = dv;
        if (n == 0) s_xb[f] = xb;
    }
    __syncthreads();
    if (d.mucon_type == 1) {
        if (tid < LOSS_FB) {
            float mx = -INFINITY;
            for (int m = 0; m < M; ++m) mx = fmaxf(mx, s_seg[tid][m]);
            float se = 0.f;
            for (int m = 0; m < M; ++m) se += expf(s_seg[tid][m] - mx);
            s_lse[tid] = mx + logf(se);
        }
        __syncthreads();
    }
    // d segmentation (+ the smoothing gradient on its own tensor)
    const float invT = d.mul_mucon / (float)T;
    for (int e = tid; e < nf * M; e += 256) {
        const int f = e / M, m = e - f * M;
        const int t = t0 + f;
        float acc = 0.f;
        if (d.mucon_type == 0) {
            for (int n = 0; n < N; ++n) acc += s_mask[n][f] * s_gwin[n][m];
        } else {
            const float sp = expf(s_seg[f][m] - s_lse[f]);
            for (int n = 0; n < N; ++n) {
                const int tg = (int)b.mtarget[n];
                acc += s_mask[n][f] * b.geo[5 * LOSS_MAXN + n] * (sp - (m == tg ? 1.f : 0.f));
            }
            acc *= invT;
        }
        b.d_seg[(long)t * M + m] = acc;
        b.d_sx[(long)t * M + m] = t >= 1 ? gs * (b.sx[(long)t * M + m] - b.sx[(long)(t - 1) * M + m]) : 0.f;
    }
    // d mask -> d x -> partial d scale, d shift: a wave per segment, lanes = frames (FB = 32: two halves idle)
    const int lane = tid & 63, wave = tid >> 6;
    for (int n = wave; n < N; n += 4) {
        float gx = 0.f;
        if (lane < nf) {
            float gm = 0.f;
            if (d.mucon_type == 0) {
                for (int m = 0; m < M; ++m) gm += s_seg[lane][m] * s_gwin[n][m];
            } else {
                const int tg = (int)b.mtarget[n];
                gm = -b.geo[5 * LOSS_MAXN + n] * (s_seg[lane][tg] - s_lse[lane]) * invT;
            }
            gx = gm * s_dmask[n][lane] * mask_dix(d.align_corners);
        }
        const float gsc = loss_wave_sum(lane < nf ? gx * s_xb[lane] : 0.f);
        const float gsh = loss_wave_sum(gx);
        if (lane == 0) {
            b.gslab[((long)blockIdx.x * N + n) * 2 + 0] = gsc;
            b.gslab[((long)blockIdx.x * N + n) * 2 + 1] = gsh;
        }
    }
}

// 1 workgroup, 64 threads (lane = segment)
__global__ __launch_bounds__(64) void loss_fin_kernel(LossDims d, LossBufs b, int chunks) {
    const int lane = threadIdx.x, N = d.N;
    float gsc = 0.f, gsh = 0.f;
    if (lane < N) {
        gsc = loss_ordered_sum(b.gslab + lane * 2 + 0, (long)N * 2, chunks);
        gsh = loss_ordered_sum(b.gslab + lane * 2 + 1, (long)N * 2, chunks);
    }
    const float L = lane < N ? b.geo[0 * LOSS_MAXN + lane] : 1.f;
    const float startp = lane < N ? b.geo[3 * LOSS_MAXN + lane] : 0.f;
    const float p = lane < N ? b.geo[4 * LOSS_MAXN + lane] : 0.f;
    const float Tf = (float)d.T;
    const float g_startp = gsh * (-2.f / L);
    float gL = (lane < N ? b.glwin[lane] : 0.f) + gsc * (-Tf / (L * L)) + gsh * (2.f * startp / (L * L) - Tf / (L * L)) +
               g_startp * (-d.overlap * 0.5f);
    if (lane >= N) gL = 0.f;
    // g_A[k] = (1 + 2 ov) gL[k] + sum_{n > k} g_startp[n]   (exclusive suffix sum)
    float suf = lane < N ? g_startp : 0.f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float dn = __shfl_down(suf, o);
        if (lane + o < 64) suf += dn;
    }
    const float gA = (1.f + 2.f * d.overlap) * gL + (suf - (lane < N ? g_startp : 0.f));
    // A = T softmax(l)
    const float gp = Tf * gA;
    const float dot = loss_wave_sum(lane < N ? p * gp : 0.f);
    if (lane < N) {
        const float l = b.lengths[lane];
        float g = p * (gp - dot);
        g += d.mul_length * ((l > d.length_width ? 1.f : 0.f) - (l < -d.length_width ? 1.f : 0.f));
        b.d_lengths[lane] = g;
    }
    if (lane == 0) b.d_lengths[N] = 0.f;   // entry N: the decoder's EOS step, whose length no loss reads (d_lengths has N + 1 entries)
}
