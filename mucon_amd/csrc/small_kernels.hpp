// Bandwidth-bound helpers around the MFMA cores: weight re-packing, max-pool un-routing,
// GroupNorm(+ReLU+Dropout) forward/backward, and the y-head (classifier + nearest upsample +
// log-softmax) forward/backward.  All activations are time-major [B][T][128] float32.
#pragma once
#include "common.hpp"
#include "head_body.hpp"

// ------------------------------------------------------------------------------------------
// Weight packing (per forward call; ~2 MB total, L2 resident afterwards).
//   dilated_conv.weight [o][i][tap]  ->  W1f [o][tap*128 + i]   (forward NT operand)
//                                    ->  W1b [i][tap*128 + o]   (data-gradient NT operand)
//   conv_1x1.weight     [o][i]       ->  W2t [i][o]           last_conv.weight -> Wlt likewise
//   first_conv.weight   [o][D]       ->  pre-split bf16 fragment image of gemm_split.hpp (and layer 0's W1b likewise)
// Every transposition goes through LDS, so that both the reads and the writes of a workgroup are whole contiguous runs
// (the element-per-thread version issued 1.1 M scattered 4-byte stores: 13 us in front of every forward pass).
// grid = (32, L + 3), 1024 threads:
//   y <  L      x < 4: layer y, output channels 32x .. 32x+31 (W1f, W1b, W2t)
//   y == L      x < 4: last_conv
//   y == L + 1  first_conv image, 64-deep k-tiles x, x + 32, ...      y == L + 2  x < 6: layer 0's W1b image, k-tile x
// ------------------------------------------------------------------------------------------
struct PackArgs {
    const float *dil_w[16];
    const float *pw_w[16];
    const float *last_w;
    float *W1f, *W1b, *W2t, *Wlt;  // [L][128][384], [L][128][384], [L][128][128], [128][128]; W1f null: none of the four is written
    const float *first_w;          // [128][D]; split into first_planes when that is non-null
    uint16_t *first_planes;        // hi / mid / lo bf16 of first_conv.weight in the fragment order of gemm_split.hpp (3*128*D values)
    uint16_t *dgrad0_planes;       // same for layer 0's data-gradient operand W1b[i][tap*128 + o] (3*128*384 values), or null
    int L, D;
    int img16;                     // the two images in the order nt_split16_kernel reads (v_mfma_f32_16x16x32_bf16: MUCON_MFMA16 bit 0)
};
constexpr int PACK_LDS_FLOATS = 32 * 385;   // 32 rows of dilated_conv.weight, padded; also 128 x 65 for an image k-tile

constexpr int PACK_THREADS = 1024;   // few workgroups (86 at L = 11), so each gets many threads: its loops are 12 steps, not 48
// one 64-deep k-tile of a [128][ld] fp32 matrix held in LDS as tile[n][65] -> the 48 KB fragment image (common.hpp order)
__device__ __forceinline__ void pack_image_tile(const float *tile, uint16_t *img_tile) {
    uint32_t *P = reinterpret_cast<uint32_t *>(img_tile);
    // image pair index p = ((s*3 + pl)*2 + half)*512 + n*4 + sp; consecutive threads -> consecutive p of one plane
    for (int q = threadIdx.x; q < 4 * 2 * 512; q += PACK_THREADS) {
        const int sp = q & 3, n = (q >> 2) & 127, half = (q >> 9) & 1, s = q >> 10;
        const int k = 16 * s + (sp < 2 ? 4 * half + 2 * sp : 8 + 4 * half + 2 * (sp - 2));
        uint32_t h, m, l;
        sp_split2(tile[n * 65 + k], tile[n * 65 + k + 1], h, m, l);
        const int base = ((s * 3) * 2 + half) * 512 + n * 4 + sp;
        P[base] = h;
        P[base + 1024] = m;
        P[base + 2048] = l;
    }
}

// the same k-tile in the order of nt_split16_kernel: [k-step 2][plane 3][h 4][channel 128][slot 8] (common.hpp: sp_split_weights16_fn)
__device__ __forceinline__ void pack_image_tile16(const float *tile, uint16_t *img_tile) {
    uint32_t *P = reinterpret_cast<uint32_t *>(img_tile);
    for (int q = threadIdx.x; q < 2 * 4 * 512; q += PACK_THREADS) {
        const int sp = q & 3, n = (q >> 2) & 127, hh = (q >> 9) & 3, ks = q >> 11;
        const int k = 32 * ks + (sp < 2 ? 4 * hh + 2 * sp : 16 + 4 * hh + 2 * (sp - 2));
        uint32_t h, m, l;
        sp_split2(tile[n * 65 + k], tile[n * 65 + k + 1], h, m, l);
        const int base = ((ks * 3) * 4 + hh) * 512 + n * 4 + sp;
        P[base] = h;
        P[base + 2048] = m;
        P[base + 4096] = l;
    }
}

// (y, x): the block's place in the logical grid of the comment above; nx = blocks of a row that loop over k-tiles (32)
__device__ __forceinline__ void pack_weights_body(const PackArgs &a, float *lds, const int y, const int x, const int nx) {
    const int tid = threadIdx.x;
    if (y <= a.L) {
        if (x >= 4 || !a.W1f) return;       // (W1f null: no launch of this pass reads the f32 layouts)
        const int o0 = 32 * x;
        if (y < a.L) {
            // 32 output channels of dilated_conv.weight: 32 x 384 contiguous floats -> lds[o][385]
            const float *src = a.dil_w[y] + (long)o0 * 384;
            const int ro = tid >> 5, rc = tid & 31;   // 32 rows x 32 lanes: division-free index math, 12 independent steps per loop
#pragma unroll
            for (int j = 0; j < 12; ++j) lds[ro * 385 + rc + 32 * j] = src[ro * 384 + rc + 32 * j];
            __syncthreads();
            float *f = a.W1f + (long)y * 49152 + (long)o0 * 384;
#pragma unroll
            for (int j = 0; j < 12; ++j) {       // W1f[o][tap*128 + i] = w[o][i][tap]: contiguous per o
                const int r = rc + 32 * j;
                f[ro * 384 + r] = lds[ro * 385 + (r & 127) * 3 + (r >> 7)];
            }
            float *bk = a.W1b + (long)y * 49152;
#pragma unroll
            for (int j = 0; j < 4; ++j) {         // W1b[i][tap*128 + o]: runs of 32 consecutive o (lane = o)
                const int i = ro + 32 * j;
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) bk[i * 384 + tap * 128 + o0 + rc] = lds[rc * 385 + i * 3 + tap];
            }
            __syncthreads();
        }
        // conv_1x1 / last_conv: 32 rows [o][128] -> [i][o]
        const float *src = (y < a.L ? a.pw_w[y] : a.last_w) + (long)o0 * 128;
        float *dst = y < a.L ? a.W2t + (long)y * 16384 : a.Wlt;
#pragma unroll
        for (int j = 0; j < 32 * 128 / PACK_THREADS; ++j) {
            const int e = tid + PACK_THREADS * j;
            lds[(e >> 7) * 129 + (e & 127)] = src[e];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32 * 128 / PACK_THREADS; ++j) {
            const int e = tid + PACK_THREADS * j;
            const int o = e & 31, i = e >> 5;
            dst[i * 128 + o0 + o] = lds[o * 129 + i];
        }
    } else if (y == a.L + 1) {
        if (!a.first_planes) return;
        for (int S = x; S < (a.D >> 6); S += nx) {
            __syncthreads();
            for (int e = tid; e < 128 * 64; e += PACK_THREADS) lds[(e >> 6) * 65 + (e & 63)] = a.first_w[(long)(e >> 6) * a.D + 64 * S + (e & 63)];
            __syncthreads();
            if (a.img16) pack_image_tile16(lds, a.first_planes + (long)S * 24576);
            else pack_image_tile(lds, a.first_planes + (long)S * 24576);
        }
    } else {
        if (!a.dgrad0_planes || x >= 6) return;
        // k-tile x of W1b(layer 0)[i][tap*128 + o]: tap = x / 2, o = 64 (x & 1) + kk;  source w[o][i][tap]
        const float *w0 = a.dil_w[0];
        const int tap = x >> 1, ob = 64 * (x & 1);
        for (int e = tid; e < 128 * 64; e += PACK_THREADS) {
            const int i = e & 127, kk = e >> 7;   // consecutive threads: consecutive i (stride 3 floats in the source)
            lds[i * 65 + kk] = w0[((long)(ob + kk) * 128 + i) * 3 + tap];
        }
        __syncthreads();
        if (a.img16) pack_image_tile16(lds, a.dgrad0_planes + (long)x * 24576);
        else pack_image_tile(lds, a.dgrad0_planes + (long)x * 24576);
    }
}

// ------------------------------------------------------------------------------------------
// Backward of an MS-TCN++ layer's tail  y = pool(f + dropout(relu(u)))  (reference temporal.py:196-201) in one pass:
//   dyd[t] = the gradient w.r.t. the un-pooled sum: dy[t / 2] on the arg-max row of the forward pair (first wins ties, as
//            torch's max_pool1d backward) when pooled, dy[t] otherwise; an odd trailing row gets 0;
//   gu[t]  = dyd[t] * scale where the branch value x = dropout(relu(u)) is positive (the element passed the ReLU and was
//            kept by the dropout, whose factor is `scale`), else 0 -- the gradient w.r.t. u, the fusion convolution's output.
// ------------------------------------------------------------------------------------------
__global__ void mstcn_tail_bwd_kernel(const float *dy, const float *ypre, const float *xact, float *dyd, float *gu, int B, int Trows,
                                      int pooled, float scale) {
    const long n4 = (long)B * Trows * 32;  // float4 elements
    const int Th = Trows >> 1;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e & 31);
        const long bt = e >> 5;
        const int b = (int)(bt / Trows), t = (int)(bt - (long)b * Trows);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (!pooled) {
            g = *reinterpret_cast<const f32x4 *>(dy + e * 4);
        } else if ((t >> 1) < Th) {
            const int tp = t >> 1;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dy + ((long)b * Th + tp) * 128 + c4 * 4);
            const f32x4 y0 = *reinterpret_cast<const f32x4 *>(ypre + ((long)b * Trows + 2 * tp) * 128 + c4 * 4);
            const f32x4 y1 = *reinterpret_cast<const f32x4 *>(ypre + ((long)b * Trows + 2 * tp + 1) * 128 + c4 * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool second = y1[k] > y0[k];
                g[k] = ((t & 1) == (second ? 1 : 0)) ? d[k] : 0.f;
            }
        }
        const f32x4 x = *reinterpret_cast<const f32x4 *>(xact + e * 4);
        f32x4 u;
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = x[k] > 0.f ? g[k] * scale : 0.f;
        *reinterpret_cast<f32x4 *>(dyd + e * 4) = g;
        *reinterpret_cast<f32x4 *>(gu + e * 4) = u;
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

// One workgroup per (video, group): 256 threads walk the Tz x (channels of the group) elements as float4; sums run up a
// fixed shuffle tree per wave and a fixed 4-wave order (bitwise reproducible, independent of the batch).  (One 1024-thread
// workgroup per video -- 8 workgroups on 256 CUs at B = 8 -- took 12 / 15 us forward / backward: three dependent passes.)
constexpr int GN_THREADS = 256;

__device__ __forceinline__ float gn_block_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// (r6: the pointers and the row count once more as LEADING SCALAR arguments -- preloaded into SGPRs with the wave, the first loads do not wait for the argument block;
// the local copy makes every use below read them)
__global__ __launch_bounds__(GN_THREADS) void gn_fwd_kernel(const float *z_, float *enc_, const float *gamma_, const float *beta_, float *stats_, const int Tz_, const int G_,
                                                            const GnArgs a_) {
    GnArgs a = a_;
    a.z = z_;
    a.enc = enc_;
    a.gamma = gamma_;
    a.beta = beta_;
    a.stats = stats_;
    a.Tz = Tz_;
    a.G = G_;
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = 128 / a.G;           // channels per group (a multiple of 4)
    const int lpg = cpg >> 2;            // float4 columns per group
    const int nel = a.Tz * lpg;
    const float *zb = a.z + (long)b * a.Tz * 128 + g * cpg;
    float mean = 0.f, rstd = 1.f;
    // (r6) a thread's first GN_KEEP elements stay in registers across the three passes (at Tz <= 1024 with one float4 column per group: all of them) -- each
    // pass re-loading them was a memory round trip of its own in front of a block-wide sum; the same values in the same order
    constexpr int GN_KEEP = 4;
    f32x4 keep[GN_KEEP];
#pragma unroll
    for (int k = 0; k < GN_KEEP; ++k) {
        const int e = min((int)threadIdx.x + k * GN_THREADS, nel - 1);
        const int t = e / lpg, c = e - t * lpg;
        keep[k] = *reinterpret_cast<const f32x4 *>(zb + (long)t * 128 + c * 4);
    }
    auto elem = [&](int e) {     // (elements beyond the kept ones: from memory, pass by pass)
        const int t = e / lpg, c = e - t * lpg;
        return *reinterpret_cast<const f32x4 *>(zb + (long)t * 128 + c * 4);
    };
    if (a.use_gn) {
        const float n = (float)a.Tz * (float)cpg;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < GN_KEEP; ++k)
            if ((int)threadIdx.x + k * GN_THREADS < nel) s += (keep[k][0] + keep[k][1]) + (keep[k][2] + keep[k][3]);
        for (int e = threadIdx.x + GN_KEEP * GN_THREADS; e < nel; e += GN_THREADS) {
            const f32x4 v = elem(e);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        mean = gn_block_sum(s, red) / n;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < GN_KEEP; ++k)
            if ((int)threadIdx.x + k * GN_THREADS < nel) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = keep[k][j] - mean;
                    q += d * d;
                }
            }
        for (int e = threadIdx.x + GN_KEEP * GN_THREADS; e < nel; e += GN_THREADS) {
            const f32x4 v = elem(e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[k] - mean;
                q += d * d;
            }
        }
        const float var = gn_block_sum(q, red) / n;
        rstd = rsqrtf(var + a.eps);
        // rsqrtf is approximate on AMD (1 ulp); refine once so rstd matches 1/sqrt to rounding
        rstd = rstd * (1.5f - 0.5f * (var + a.eps) * rstd * rstd);
        if (threadIdx.x == 0) {
            a.stats[((long)b * a.G + g) * 2 + 0] = mean;
            a.stats[((long)b * a.G + g) * 2 + 1] = rstd;
        }
    }
    float *eb = a.enc + (long)b * a.Tz * 128 + g * cpg;
    int kk = 0;
    for (int e = threadIdx.x; e < nel; e += GN_THREADS, ++kk) {
        const int t = e / lpg, c = e - t * lpg;
        f32x4 v = kk == 0 ? keep[0] : kk == 1 ? keep[1] : kk == 2 ? keep[2] : kk == 3 ? keep[3] : *reinterpret_cast<const f32x4 *>(zb + (long)t * 128 + c * 4);
        f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
        if (a.use_gn) {
            ga = *reinterpret_cast<const f32x4 *>(a.gamma + g * cpg + c * 4);
            be = *reinterpret_cast<const f32x4 *>(a.beta + g * cpg + c * 4);
        }
        const uint32_t idx = (uint32_t)(b * a.Tz + t) * 128u + (uint32_t)(g * cpg + c * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y = a.use_gn ? (v[k] - mean) * rstd * ga[k] + be[k] : v[k];
            if (a.use_relu) y = fmaxf(y, 0.f);
            if (a.drop.thresh) y *= drop_mul(a.drop, idx + k);
            v[k] = y;
        }
        *reinterpret_cast<f32x4 *>(eb + (long)t * 128 + c * 4) = v;
    }
}

struct GnBwdArgs {
    const float *z, *denc;  // [B][Tz][128]
    float *dz;              // [B][Tz][128]
    const float *gamma, *beta, *stats;
    float *part;            // [B][2][128]: per-video d_gamma, d_beta partials
    unsigned *zero;         // 64 words workgroup (0, 0) zeroes (encoder_bwd's counters: this is the pass's first kernel), or null
    int Tz, G;
    int use_gn, use_relu;
    DropCfg drop;
};

// (r6) The y-head's DEFERRED slab reduction (mucon_head_bwd_defer): the sums reduce_batch_kernel<16> would take in a launch of its own behind
// head_bwd_z_kernel, taken by extra workgroups of the backward pass's FIRST launch instead -- rows blockIdx.y >= B of gn_bwd_kernel's grid, which run
// beside the GroupNorm workgroups on CUs that launch leaves idle.  Two jobs (d_w [C * H], d_b [C]); a workgroup owns 256 consecutive elements.
// Same sums in the same order as reduce_batch_kernel<16>: slab lane g (of 16) adds slabs g, g + 16, ... in order, the sixteen lanes then add up in order --
// here wave w runs lanes w, w + 4, w + 8, w + 12 one after the other (all their loads independent): bitwise the results of the undeferred call.
struct HeadReduceTail {
    const float *slabs[2];
    float *out[2];
    int stride[2], n_elems[2];
    int nslabs;
    int nblocks0;    // workgroups of job 0
    int nblocks;     // of both jobs (0: nothing deferred)
};
__device__ __forceinline__ void head_reduce_tail(const HeadReduceTail &t, const int block) {
#pragma clang fp contract(off)
    __shared__ f32x4 hpart[16][64];
    const int job = block >= t.nblocks0 ? 1 : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = ((block - (job ? t.nblocks0 : 0)) * 64 + lane) * 4;   // (n_elems, stride: multiples of 4 -- checked by the host)
    const bool live = e < t.n_elems[job];
    const float *p = t.slabs[job] + (live ? e : 0);
    const long stride = t.stride[job];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = wave + 4 * q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int i = g;
        for (; i + 15 * 16 < t.nslabs; i += 16 * 16) {
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (long)(i + 16 * u) * stride);
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; i + 3 * 16 < t.nslabs; i += 4 * 16) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (long)(i + 16 * u) * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; i < t.nslabs; i += 16) s += *reinterpret_cast<const f32x4 *>(p + (long)i * stride);
        hpart[g][lane] = s;
    }
    __syncthreads();
    if (wave == 0 && live) {
        f32x4 r = hpart[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) r += hpart[k][lane];
        *reinterpret_cast<f32x4 *>(t.out[job] + e) = r;
    }
}

__global__ __launch_bounds__(GN_THREADS) void gn_bwd_kernel(const float *z_, const float *denc_, float *dz_, const float *stats_, const int Tz_, const int G_, const int B,
                                                            const GnBwdArgs a_, const HeadReduceTail tail) {
    GnBwdArgs a = a_;   // (leading scalars: see gn_fwd_kernel)
    a.z = z_;
    a.denc = denc_;
    a.dz = dz_;
    a.stats = stats_;
    a.Tz = Tz_;
    a.G = G_;
    if ((int)blockIdx.y >= B) {   // (only with a deferred y-head reduction queued: the grid has no such rows otherwise)
        const int block = ((int)blockIdx.y - B) * (int)gridDim.x + (int)blockIdx.x;
        if (block < tail.nblocks) head_reduce_tail(tail, block);
        return;
    }
    __shared__ float red[4];
    __shared__ float cred[2][4][128];   // per-channel partials [d_gamma | d_beta][wave][channel of the group]
    const int g = blockIdx.x, b = blockIdx.y;
    if (a.zero && g == 0 && b == 0 && threadIdx.x < 64) a.zero[threadIdx.x] = 0u;
    const int cpg = 128 / a.G, lpg = cpg >> 2;
    const int nel = a.Tz * lpg;
    const float *zb = a.z + (long)b * a.Tz * 128 + g * cpg;
    const float *db = a.denc + (long)b * a.Tz * 128 + g * cpg;
    float *ob = a.dz + (long)b * a.Tz * 128 + g * cpg;
    float mean = 0.f, rstd = 1.f;
    if (a.use_gn) {
        mean = a.stats[((long)b * a.G + g) * 2 + 0];
        rstd = a.stats[((long)b * a.G + g) * 2 + 1];
    }
    // gradient at the GroupNorm output: through dropout and ReLU
    auto dgn_at = [&](int t, int c, f32x4 &xh, f32x4 &dg, f32x4 &ga) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(zb + (long)t * 128 + c * 4);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(db + (long)t * 128 + c * 4);
        f32x4 be = {0.f, 0.f, 0.f, 0.f};
        ga = f32x4{1.f, 1.f, 1.f, 1.f};
        if (a.use_gn) {
            ga = *reinterpret_cast<const f32x4 *>(a.gamma + g * cpg + c * 4);
            be = *reinterpret_cast<const f32x4 *>(a.beta + g * cpg + c * 4);
        }
        const uint32_t idx = (uint32_t)(b * a.Tz + t) * 128u + (uint32_t)(g * cpg + c * 4);
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
        for (int e = threadIdx.x; e < nel; e += GN_THREADS) {
            const int t = e / lpg, c = e - t * lpg;
            f32x4 xh, dg, ga;
            dgn_at(t, c, xh, dg, ga);
            *reinterpret_cast<f32x4 *>(ob + (long)t * 128 + c * 4) = dg;
        }
        return;
    }
    // a thread keeps ONE float4 column (threads per column = 256 / lpg row slots): its d_gamma / d_beta partials are per channel
    const int col = threadIdx.x % lpg, slot = threadIdx.x / lpg, nslots = GN_THREADS / lpg;
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    // (r6) the first GB_KEEP rows of a thread keep their (x_hat, gradient at the GroupNorm output) in registers for the second pass (at Tz <= 512 with one
    // float4 column per group: all of them): re-deriving them was two loads and a dropout hash per element behind the block-wide sums
    constexpr int GB_KEEP = 2;
    f32x4 kxh[GB_KEEP], kdg[GB_KEEP], kga[GB_KEEP];
    int it = 0;
    for (int t = slot; t < a.Tz; t += nslots, ++it) {
        f32x4 xh, dg, ga;
        dgn_at(t, col, xh, dg, ga);
        if (it == 0) {
            kxh[0] = xh;
            kdg[0] = dg;
            kga[0] = ga;
        } else if (it == 1) {
            kxh[1] = xh;
            kdg[1] = dg;
            kga[1] = ga;
        }
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
    const float m1 = gn_block_sum(s1, red) / n;
    const float m2 = gn_block_sum(s2, red) / n;
    // per-channel sums: up a fixed shuffle tree over the lanes that hold the same float4 column (lane offsets >= lpg keep the
    // column), then the four waves in order
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        for (int o = 32; o >= lpg; o >>= 1) {
            sg[k] += __shfl_xor(sg[k], o);
            sb[k] += __shfl_xor(sb[k], o);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < lpg) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cred[0][wave][lane * 4 + k] = sg[k];   // lane < lpg: lane == its column (64 % lpg == 0)
            cred[1][wave][lane * 4 + k] = sb[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * cpg) {
        const int which = threadIdx.x / cpg, c = threadIdx.x - which * cpg;
        a.part[((long)b * 2 + which) * 128 + g * cpg + c] =
            (cred[which][0][c] + cred[which][1][c]) + (cred[which][2][c] + cred[which][3][c]);
    }
    it = 0;
    for (int t = slot; t < a.Tz; t += nslots, ++it) {
        f32x4 xh, dg, ga, o;
        if (it == 0) {
            xh = kxh[0];
            dg = kdg[0];
            ga = kga[0];
        } else if (it == 1) {
            xh = kxh[1];
            dg = kdg[1];
            ga = kga[1];
        } else {
            dgn_at(t, col, xh, dg, ga);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = rstd * (dg[k] * ga[k] - m1 - xh[k] * m2);
        *reinterpret_cast<f32x4 *>(ob + (long)t * 128 + col * 4) = o;
    }
}

constexpr int HEAD_FB = 128;  // frames per workgroup (forward)
constexpr int HEAD_ZC = 16;   // z rows per pass


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

// Forward organised by z rows (H a multiple of 16): a workgroup owns HF_Z rows of the encoding and the frames that map onto
// them.  Thread (class c = tid >> 2, k-quarter q = tid & 3) keeps its share of W[c] in registers straight from global (the four
// lanes of a class read 64 contiguous bytes per step) and multiplies it with all HF_Z rows from LDS; the quarters meet by two
// shuffles.  No W staging pass, no divisions: 22 -> ~8 us at B=8, T=4096 against the frame-organised kernel above.
__global__ __launch_bounds__(256) void head_fwd_z_kernel(const float *enc_, const float *w_, const float *b_, float *logp_z_, const int Tz_, const int H_, const int C_,
                                                         const HeadFwdArgs a_) {
    HeadFwdArgs a = a_;   // (leading scalars: see gn_fwd_kernel)
    a.enc = enc_;
    a.w = w_;
    a.b = b_;
    a.logp_z = logp_z_;
    a.Tz = Tz_;
    a.H = H_;
    a.C = C_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    head_fwd_z_body(a, smem, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, true);   // (head_body.hpp)
}
static inline size_t head_fwd_z_smem_bytes(int H) { return sizeof(float) * ((size_t)HF_Z * H + 2 * HF_Z * HEAD_MAXC); }


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

// Backward for H = 128 with HB_Z rows per workgroup (twice the workgroups of the kernel above), nothing staged but the rows
// themselves: thread (k = tid & 127, half = tid >> 7) reads its column of W straight from global (coalesced over k) for
// d_enc, keeps its column of the encoding rows in registers for the weight-gradient partials, and the per-row class sums run
// up shuffle trees.  30 -> ~14 us at B=8, T=4096.
__global__ __launch_bounds__(256) void head_bwd_z_kernel(const float *enc_, const float *w_, const float *dlogits_, const float *dlogp_, const float *logp_z_, const int Tz_,
                                                         const int Tf_, const int C_, const float scale_, const HeadBwdArgs a_) {
    HeadBwdArgs a = a_;   // (leading scalars: see gn_fwd_kernel)
    a.enc = enc_;
    a.w = w_;
    a.dlogits = dlogits_;
    a.dlogp = dlogp_;
    a.logp_z = logp_z_;
    a.Tz = Tz_;
    a.Tf = Tf_;
    a.C = C_;
    a.scale = scale_;
    head_bwd_z_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)threadIdx.x);   // (head_body.hpp)
}

static inline size_t head_smem_bytes(int H, int C) {
    return sizeof(float) * ((size_t)C * (H + 1) + (size_t)HEAD_ZC * H + 2 * HEAD_ZC * HEAD_MAXC + HEAD_ZC);
}
