// Bandwidth-bound helpers around the MFMA cores: weight re-packing, max-pool un-routing,
// GroupNorm(+ReLU+Dropout) forward/backward, and the y-head (classifier + nearest upsample +
// log-softmax) forward/backward.  All activations are time-major [B][T][128] float32.
#pragma once
#include "common.hpp"

// ------------------------------------------------------------------------------------------
// Weight packing (per forward call; ~2 MB total, L2 resident afterwards).
//   dilated_conv.weight [o][i][tap]  ->  W1f [o][tap*128 + i]   (forward NT operand)
//                                    ->  W1b [i][tap*128 + o]   (data-gradient NT operand)
//   conv_1x1.weight     [o][i]       ->  W2t [i][o]
// grid = (ceil(49152/256), n_mats)
// ------------------------------------------------------------------------------------------
struct PackArgs {
    const float *dil_w[16];
    const float *pw_w[16];
    const float *last_w;
    float *W1f, *W1b, *W2t, *Wlt;  // [L][128][384], [L][128][384], [L][128][128], [128][128]
    const float *first_w;          // [128][D]; split into first_planes when that is non-null (grid row L + 1)
    uint16_t *first_planes;        // hi / mid / lo bf16 of first_conv.weight in the fragment order of gemm_split.hpp (3*128*D values)
    uint16_t *dgrad0_planes;       // same for layer 0's data-gradient operand W1b[i][tap*128 + o] (3*128*384 values), or null
    int L, D;
};
__global__ void pack_weights_kernel(const PackArgs a) {
    const int l = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < a.L) {
        if (e < 128 * 384) {
            const int o = e / 384, r = e - o * 384;
            const int tap = r >> 7, i = r & 127;
            const float w = a.dil_w[l][(o * 128 + i) * 3 + tap];
            a.W1f[(long)l * 49152 + e] = w;
            a.W1b[(long)l * 49152 + i * 384 + tap * 128 + o] = w;
        }
        if (e < 128 * 128) {
            const int o = e >> 7, i = e & 127;
            a.W2t[(long)l * 16384 + i * 128 + o] = a.pw_w[l][e];
        }
    } else if (l == a.L) {
        if (e < 128 * 128) {
            const int o = e >> 7, i = e & 127;
            a.Wlt[i * 128 + o] = a.last_w[e];
        }
    } else if (l == a.L + 1) {
        if (a.first_planes) sp_split_weights(a.first_w, a.first_planes, a.D, e, (long)gridDim.x * blockDim.x);
    } else if (a.dgrad0_planes) {
        const float *w0 = a.dil_w[0];
        sp_split_weights_fn([=](int i, int k) { return w0[((k & 127) * 128 + i) * 3 + (k >> 7)]; }, a.dgrad0_planes, 384, e,
                            (long)gridDim.x * blockDim.x);
    }
}

// ------------------------------------------------------------------------------------------
// Backward of F.max_pool1d(k=2) / avg_pool1d(k=2)*2 (temporal.py:137-142):
//   dyd[t] = dy[t/2] if t is the arg-max of its pair (first wins ties), or for sum pooling always;
//   an odd trailing step gets 0.    ypre: un-pooled forward rows.
// ------------------------------------------------------------------------------------------
__global__ void unpool_kernel(const float *dy, const float *ypre, float *dyd, int B, int Trows, int pool_type) {
    const long n4 = (long)B * Trows * 32;  // float4 elements
    const int Th = Trows >> 1;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e & 31);
        const long bt = e >> 5;
        const int b = (int)(bt / Trows), t = (int)(bt - (long)b * Trows);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        const int tp = t >> 1;
        if (tp < Th) {
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dy + ((long)b * Th + tp) * 128 + c4 * 4);
            if (pool_type == 1) {
                g = d;
            } else {
                const f32x4 y0 = *reinterpret_cast<const f32x4 *>(ypre + ((long)b * Trows + 2 * tp) * 128 + c4 * 4);
                const f32x4 y1 = *reinterpret_cast<const f32x4 *>(ypre + ((long)b * Trows + 2 * tp + 1) * 128 + c4 * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool second = y1[k] > y0[k];
                    g[k] = ((t & 1) == (second ? 1 : 0)) ? d[k] : 0.f;
                }
            }
        }
        *reinterpret_cast<f32x4 *>(dyd + bt * 128 + c4 * 4) = g;
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm(G groups over [channels/G x Tz]) + ReLU + Dropout  (models.py:759-768), one workgroup
// per video.  128 channels / 32 groups = 4 channels per group = one float4 per thread column.
// Generic in G only through cpg = 128/G in {1,2,4,8,...}: thread column c4 covers channels
// 4*c4..4*c4+3, group id = (4*c4)/cpg; cpg >= 4 supported (cpg multiple of 4), which covers the
// reference default (32 groups).  Statistics are two-pass (mean, then centred sum of squares).
// stats: [B][G][2] = (mean, rstd).
// ------------------------------------------------------------------------------------------
struct GnArgs {
    const float *z;      // [B][Tz][128]
    float *enc;          // [B][Tz][128]
    const float *gamma, *beta;
    float *stats;
    int Tz, G;
    float eps;
    int use_gn, use_relu;
    DropCfg drop;
};

constexpr int GN_ROWS = 32;               // row slots per workgroup
constexpr int GN_THREADS = 32 * GN_ROWS;   // x 32 float4 columns

__device__ __forceinline__ float block_group_sum(float v, float *red, int c4, int trow, int lanes_per_group) {
    // sum over the float4 columns of one group (adjacent c4) and over the row slots
    for (int o = 1; o < lanes_per_group; o <<= 1) v += __shfl_xor(v, o);
    __syncthreads();
    red[trow * 32 + c4] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < GN_ROWS; ++r) s += red[r * 32 + c4];
    return s;
}

__global__ __launch_bounds__(GN_THREADS) void gn_fwd_kernel(const GnArgs a) {
    __shared__ float red[GN_THREADS];
    const int b = blockIdx.x;
    const int c4 = threadIdx.x & 31, trow = threadIdx.x >> 5;
    const int cpg = 128 / a.G;           // channels per group (>= 4)
    const int lpg = cpg >> 2;            // float4 columns per group
    const int g = (c4 * 4) / cpg;
    const float *zb = a.z + (long)b * a.Tz * 128 + c4 * 4;
    float mean = 0.f, rstd = 1.f;
    if (a.use_gn) {
        const float n = (float)a.Tz * (float)cpg;
        float s = 0.f;
        for (int t = trow; t < a.Tz; t += GN_ROWS) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(zb + (long)t * 128);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        mean = block_group_sum(s, red, c4, trow, lpg) / n;
        float q = 0.f;
        for (int t = trow; t < a.Tz; t += GN_ROWS) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(zb + (long)t * 128);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[k] - mean;
                q += d * d;
            }
        }
        const float var = block_group_sum(q, red, c4, trow, lpg) / n;
        rstd = rsqrtf(var + a.eps);
        // rsqrtf is approximate on AMD (1 ulp); refine once so rstd matches 1/sqrt to rounding
        rstd = rstd * (1.5f - 0.5f * (var + a.eps) * rstd * rstd);
        if (trow == 0 && (c4 % lpg) == 0) {
            a.stats[((long)b * a.G + g) * 2 + 0] = mean;
            a.stats[((long)b * a.G + g) * 2 + 1] = rstd;
        }
    }
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (a.use_gn) {
        ga = *reinterpret_cast<const f32x4 *>(a.gamma + c4 * 4);
        be = *reinterpret_cast<const f32x4 *>(a.beta + c4 * 4);
    }
    float *eb = a.enc + (long)b * a.Tz * 128 + c4 * 4;
    for (int t = trow; t < a.Tz; t += GN_ROWS) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(zb + (long)t * 128);
        const uint32_t idx = (uint32_t)(b * a.Tz + t) * 128u + (uint32_t)(c4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y = a.use_gn ? (v[k] - mean) * rstd * ga[k] + be[k] : v[k];
            if (a.use_relu) y = fmaxf(y, 0.f);
            if (a.drop.thresh) y *= drop_mul(a.drop, idx + k);
            v[k] = y;
        }
        *reinterpret_cast<f32x4 *>(eb + (long)t * 128) = v;
    }
}

struct GnBwdArgs {
    const float *z, *denc;  // [B][Tz][128]
    float *dz;              // [B][Tz][128]
    const float *gamma, *beta, *stats;
    float *part;            // [B][2][128]: per-video d_gamma, d_beta partials
    int Tz, G;
    int use_gn, use_relu;
    DropCfg drop;
};

__global__ __launch_bounds__(GN_THREADS) void gn_bwd_kernel(const GnBwdArgs a) {
    __shared__ float red[GN_THREADS];
    __shared__ float cred[2][GN_ROWS][128];
    const int b = blockIdx.x;
    const int c4 = threadIdx.x & 31, trow = threadIdx.x >> 5;
    const int cpg = 128 / a.G, lpg = cpg >> 2;
    const int g = (c4 * 4) / cpg;
    const float *zb = a.z + (long)b * a.Tz * 128 + c4 * 4;
    const float *db = a.denc + (long)b * a.Tz * 128 + c4 * 4;
    float *ob = a.dz + (long)b * a.Tz * 128 + c4 * 4;
    float mean = 0.f, rstd = 1.f;
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (a.use_gn) {
        mean = a.stats[((long)b * a.G + g) * 2 + 0];
        rstd = a.stats[((long)b * a.G + g) * 2 + 1];
        ga = *reinterpret_cast<const f32x4 *>(a.gamma + c4 * 4);
        be = *reinterpret_cast<const f32x4 *>(a.beta + c4 * 4);
    }
    // gradient at the GroupNorm output: through dropout and ReLU
    auto dgn_at = [&](int t, f32x4 &xh, f32x4 &dg) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(zb + (long)t * 128);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(db + (long)t * 128);
        const uint32_t idx = (uint32_t)(b * a.Tz + t) * 128u + (uint32_t)(c4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xh[k] = (v[k] - mean) * rstd;
            const float y = a.use_gn ? xh[k] * ga[k] + be[k] : v[k];
            float gk = d[k];
            if (a.drop.thresh) gk *= drop_mul(a.drop, idx + k);
            if (a.use_relu && !(y > 0.f)) gk = 0.f;
            dg[k] = gk;
        }
    };
    if (!a.use_gn) {
        for (int t = trow; t < a.Tz; t += GN_ROWS) {
            f32x4 xh, dg;
            dgn_at(t, xh, dg);
            *reinterpret_cast<f32x4 *>(ob + (long)t * 128) = dg;
        }
        return;
    }
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    for (int t = trow; t < a.Tz; t += GN_ROWS) {
        f32x4 xh, dg;
        dgn_at(t, xh, dg);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sg[k] += dg[k] * xh[k];
            sb[k] += dg[k];
            const float dx = dg[k] * ga[k];
            s1 += dx;
            s2 += dx * xh[k];
        }
    }
    const float n = (float)a.Tz * (float)cpg;
    const float m1 = block_group_sum(s1, red, c4, trow, lpg) / n;
    const float m2 = block_group_sum(s2, red, c4, trow, lpg) / n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cred[0][trow][c4 * 4 + k] = sg[k];
        cred[1][trow][c4 * 4 + k] = sb[k];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int c = threadIdx.x & 127, which = threadIdx.x >> 7;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < GN_ROWS; ++r) s += cred[which][r][c];
        a.part[((long)b * 2 + which) * 128 + c] = s;
    }
    for (int t = trow; t < a.Tz; t += GN_ROWS) {
        f32x4 xh, dg, o;
        dgn_at(t, xh, dg);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = rstd * (dg[k] * ga[k] - m1 - xh[k] * m2);
        *reinterpret_cast<f32x4 *>(ob + (long)t * 128) = o;
    }
}

// ------------------------------------------------------------------------------------------
// y-head.  F.interpolate(mode="nearest") source index (models.py:574), computed as torch does:
// scale = float(Tz)/float(Tf); src = min(int(floorf(i * scale)), Tz - 1).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int zmap(int i, float scale, int Tz) {
    const int z = (int)floorf((float)i * scale);
    return z < Tz - 1 ? z : Tz - 1;
}
// first frame i in [0, Tf] with zmap(i) >= z  (Tf when none)
__device__ __forceinline__ int first_frame(int z, float scale, int Tz, int Tf) {
    if (z <= 0) return 0;
    if (z >= Tz) return Tf;
    long guess = ((long)z * Tf) / Tz - 2;
    int i = guess < 0 ? 0 : (int)guess;
    while (i > 0 && zmap(i - 1, scale, Tz) >= z) --i;
    while (i < Tf && zmap(i, scale, Tz) < z) ++i;
    return i;
}

constexpr int HEAD_FB = 128;  // frames per workgroup (forward)
constexpr int HEAD_ZC = 16;   // z rows per pass
constexpr int HEAD_MAXC = 64; // classes supported by the LDS carve

struct HeadFwdArgs {
    const float *enc;   // [B][Tz][H]
    const float *w, *b; // [C][H], [C]
    float *logits, *logp;  // [B][Tf][C] or null
    float *logp_z;      // [B][Tz][C]
    int Tz, Tf, H, C;
    float scale;
};

__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, C = a.C;
    float *Ws = smem;                          // [C][H+1]
    float *Es = Ws + C * (H + 1);              // [ZC][H]
    float *Ls = Es + HEAD_ZC * H;              // [ZC][C] logits
    float *Ps = Ls + HEAD_ZC * HEAD_MAXC;      // [ZC][C] log-probs
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * HEAD_FB;
    const int i1 = min(i0 + HEAD_FB, a.Tf);
    for (int e = tid; e < C * H; e += 256) Ws[(e / H) * (H + 1) + (e % H)] = a.w[e];
    const int z_lo = zmap(i0, a.scale, a.Tz), z_hi = zmap(i1 - 1, a.scale, a.Tz);
    for (int zc0 = z_lo; zc0 <= z_hi; zc0 += HEAD_ZC) {
        const int nz = min(HEAD_ZC, z_hi - zc0 + 1);
        __syncthreads();
        for (int e = tid; e < nz * H; e += 256) Es[e] = a.enc[((long)b * a.Tz + zc0) * H + e];
        __syncthreads();
        for (int o = tid; o < nz * C; o += 256) {
            const int zi = o / C, c = o - zi * C;
            float s = a.b[c];
            const float *er = Es + zi * H, *wr = Ws + c * (H + 1);
            for (int k = 0; k < H; ++k) s += er[k] * wr[k];
            Ls[zi * HEAD_MAXC + c] = s;
        }
        __syncthreads();
        // log-softmax over the classes: a wave per z row, lane = class (C <= 64); the sum runs up a fixed shuffle tree
        for (int zi = tid >> 6; zi < nz; zi += 4) {
            const int c = tid & 63;
            const float x = c < C ? Ls[zi * HEAD_MAXC + c] : -INFINITY;
            float m = x;
#pragma unroll
            for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            float s = c < C ? expf(x - m) : 0.f;
#pragma unroll
            for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
            if (c < C) Ps[zi * HEAD_MAXC + c] = x - (m + logf(s));
        }
        __syncthreads();
        for (int e = tid; e < (i1 - i0) * C; e += 256) {
            const int fi = e / C, c = e - fi * C;
            const int i = i0 + fi;
            const int z = zmap(i, a.scale, a.Tz);
            const int zi = z - zc0;
            if (zi >= 0 && zi < nz) {
                const long g = ((long)b * a.Tf + i) * C + c;
                if (a.logits) a.logits[g] = Ls[zi * HEAD_MAXC + c];
                if (a.logp) a.logp[g] = Ps[zi * HEAD_MAXC + c];
                if (i == 0 || zmap(i - 1, a.scale, a.Tz) != z)  // first frame of its bin owns the save
                    a.logp_z[((long)b * a.Tz + z) * C + c] = Ps[zi * HEAD_MAXC + c];
            }
        }
    }
}

struct HeadBwdArgs {
    const float *enc, *w;          // [B][Tz][H], [C][H]
    const float *dlogits, *dlogp;  // [B][Tf][C] or null
    const float *logp_z;           // [B][Tz][C]
    float *denc;                   // [B][Tz][H]
    float *w_slabs, *b_slabs;      // [nblk][C][H], [nblk][C]
    int Tz, Tf, H, C;
    float scale;
};

__global__ __launch_bounds__(256) void head_bwd_kernel(const HeadBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, C = a.C;
    float *Ws = smem;                       // [C][H+1]
    float *Es = Ws + C * (H + 1);           // [ZC][H]
    float *G1 = Es + HEAD_ZC * H;           // [ZC][MAXC]: sum dlogits, then dlogit_z
    float *G2 = G1 + HEAD_ZC * HEAD_MAXC;   // [ZC][MAXC]: sum dlogp
    float *S2 = G2 + HEAD_ZC * HEAD_MAXC;   // [ZC]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int z0 = blockIdx.x * HEAD_ZC;
    const int nz = min(HEAD_ZC, a.Tz - z0);
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    for (int e = tid; e < C * H; e += 256) Ws[(e / H) * (H + 1) + (e % H)] = a.w[e];
    for (int e = tid; e < nz * H; e += 256) Es[e] = a.enc[((long)b * a.Tz + z0) * H + e];
    for (int o = tid; o < nz * C; o += 256) {
        const int zi = o / C, c = o - zi * C;
        const int fa = first_frame(z0 + zi, a.scale, a.Tz, a.Tf);
        const int fb = first_frame(z0 + zi + 1, a.scale, a.Tz, a.Tf);
        // frames of the bin in order (same sums as a plain loop), eight loads in flight at a time: a bin has ~Tf/Tz = 16
        // frames and a dependent-looking loop made this kernel a chain of 16 x 3 memory round trips
        float g1 = 0.f, g2 = 0.f;
        const int nfr = fb - fa;
        for (int base = 0; base < nfr; base += 8) {
            float v1[8], v2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long g = ((long)b * a.Tf + fa + min(base + j, nfr - 1)) * C + c;
                v1[j] = a.dlogits ? a.dlogits[g] : 0.f;
                v2[j] = a.dlogp ? a.dlogp[g] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (base + j < nfr) {
                    g1 += v1[j];
                    g2 += v2[j];
                }
            }
        }
        G1[zi * HEAD_MAXC + c] = g1;
        G2[zi * HEAD_MAXC + c] = g2;
    }
    __syncthreads();
    if (tid < nz) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += G2[tid * HEAD_MAXC + c];
        S2[tid] = s;
    }
    __syncthreads();
    for (int o = tid; o < nz * C; o += 256) {
        const int zi = o / C, c = o - zi * C;
        float d = G1[zi * HEAD_MAXC + c] + G2[zi * HEAD_MAXC + c];
        if (S2[zi] != 0.f) d -= expf(a.logp_z[((long)b * a.Tz + z0 + zi) * C + c]) * S2[zi];
        G1[zi * HEAD_MAXC + c] = d;
    }
    __syncthreads();
    // d_enc[z][k] = sum_c dlogit[z][c] * W[c][k]
    for (int o = tid; o < nz * H; o += 256) {
        const int zi = o / H, k = o - zi * H;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += G1[zi * HEAD_MAXC + c] * Ws[c * (H + 1) + k];
        a.denc[((long)b * a.Tz + z0) * H + o] = s;
    }
    // partial dW[c][k] = sum_z dlogit[z][c] * enc[z][k];  partial db[c] = sum_z dlogit[z][c]
    for (int o = tid; o < C * H; o += 256) {
        const int c = o / H, k = o - c * H;
        float s = 0.f;
        for (int zi = 0; zi < nz; ++zi) s += G1[zi * HEAD_MAXC + c] * Es[zi * H + k];
        a.w_slabs[(long)blk * C * H + o] = s;
    }
    if (tid < C) {
        float s = 0.f;
        for (int zi = 0; zi < nz; ++zi) s += G1[zi * HEAD_MAXC + tid];
        a.b_slabs[(long)blk * C + tid] = s;
    }
}

static inline size_t head_smem_bytes(int H, int C) {
    return sizeof(float) * ((size_t)C * (H + 1) + (size_t)HEAD_ZC * H + 2 * HEAD_ZC * HEAD_MAXC + HEAD_ZC);
}
