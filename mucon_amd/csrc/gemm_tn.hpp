// TN GEMM core (weight gradients) on the f32-input MFMA:
//
//     dW[n][k] = sum_{b,t} Y[b][t][n] * X'[b][t][k]        n in [0,128), k in one 128-wide chunk
//
// i.e. the reduction runs over time (rows of both operands), which is the autograd of every
// nn.Conv1d of the reference's encoder (src/core/modules/temporal.py:23-32,108-114) w.r.t. its
// weight.  X' is the layer input gathered with the conv's tap offset (zero padded), the tape for
// first_conv, or the activated input of last_conv.  grid = (k-chunks, time-chunks); every
// workgroup writes its partial 128x128 tile to a slab [time-chunk][128][Ktot]; reduce_slabs_kernel
// then sums the slabs in a fixed order (bitwise reproducible -- no float atomics) and writes the
// gradient in the reference's [out][in][k] layout.
//
// LDS tiles are [32 time steps][128 channels] exactly as in HBM (512-byte rows, no padding
// needed: the MFMA operand read is ds_read_b32 with 32 consecutive lanes on consecutive floats).
#pragma once
#include "common.hpp"

constexpr int TN_SMEM_BYTES = 2 * 2 * 32 * 128 * 4;  // Y and X tiles, double buffered = 64 KiB

struct TnParams {
    const float *Y;   // [B][Trows][128]
    int Trows;
    const float *X;   // [B][Tx][ldx]
    long x_bstride;
    int ldx, Tx;
    int taps;         // 3: k-chunk kc is tap kc (row offset (kc-1)*tap_step), columns 0..127
    int tap_step;     // 1: k-chunk kc is columns kc*128.. of X
    int Ktot;         // number of k columns overall (multiple of 128)
    float *slabs;     // [n_time_chunks][128][Ktot]
    float *bias_slabs;  // [n_time_chunks][128] column sums of Y, or null
    int MC;           // time steps per chunk (multiple of 32)
    int chunks_per_video;
    float slope;
    DropCfg drop;     // Y_DROP: Y element index (b*Trows + t)*128 + n
};

template <bool Y_DROP, bool X_ACT>
__global__ __launch_bounds__(256) void tn_gemm_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ys = smem;
    float *Xs = smem + 2 * 32 * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kc = blockIdx.x;
    const int mc = blockIdx.y;
    const int b = mc / p.chunks_per_video;
    const int tbeg = (mc - b * p.chunks_per_video) * p.MC;
    const int tend = min(tbeg + p.MC, p.Trows);
    const int ntiles = (tend - tbeg + 31) >> 5;
    const int xoff = (p.taps == 3) ? (kc - 1) * p.tap_step : 0;
    const int xcol = (p.taps == 3) ? 0 : kc * 128;
    const float *Yb = p.Y + (long)b * p.Trows * 128;
    const float *Xb = p.X + (long)b * p.x_bstride + xcol;

    f32x4 ry[4], rx[4];
    auto gload = [&](int mtile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = tid + 256 * q;
            const int row = f >> 5, c4 = (f & 31) * 4;
            const int t = tbeg + mtile * 32 + row;
            const int ts = t + xoff;
            f32x4 y = {0.f, 0.f, 0.f, 0.f}, x = {0.f, 0.f, 0.f, 0.f};
            if (t < tend) {
                y = *reinterpret_cast<const f32x4 *>(Yb + (long)t * 128 + c4);
                if (Y_DROP) {
                    if (p.drop.thresh) {
                        const uint32_t idx = (uint32_t)(b * p.Trows + t) * 128u + (uint32_t)c4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] *= drop_mul(p.drop, idx + e);
                    }
                }
                if (ts >= 0 && ts < p.Tx) {
                    x = *reinterpret_cast<const f32x4 *>(Xb + (long)ts * p.ldx + c4);
                    if (X_ACT) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = act_f(x[e], p.slope);
                    }
                }
            }
            ry[q] = y;
            rx[q] = x;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = tid + 256 * q;
            *reinterpret_cast<f32x4 *>(Ys + buf * 4096 + f * 4) = ry[q];
            *reinterpret_cast<f32x4 *>(Xs + buf * 4096 + f * 4) = rx[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (p.bias_slabs != nullptr) && (kc == 0) && (tid < 128);

    // lane (i = lane&31, h = lane>>5): MFMA step s consumes time step 16h + s of the tile.
    const int y_off = (lane >> 5) * 16 * 128 + wr * 64 + (lane & 31);
    const int x_off = (lane >> 5) * 16 * 128 + wc * 64 + (lane & 31);

    if (ntiles > 0) {
        gload(0);
        sstore(0);
    }
    __syncthreads();
    for (int mt = 0; mt < ntiles; ++mt) {
        const int cur = mt & 1;
        if (mt + 1 < ntiles) gload(mt + 1);
        const float *Yw = Ys + cur * 4096 + y_off;
        const float *Xw = Xs + cur * 4096 + x_off;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a0 = Yw[s * 128], a1 = Yw[s * 128 + 32];
            const float b0 = Xw[s * 128], b1 = Xw[s * 128 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (do_bias) {
            const float *Yc = Ys + cur * 4096 + tid;
#pragma unroll 8
            for (int m = 0; m < 32; ++m) bsum += Yc[m * 128];
        }
        if (mt + 1 < ntiles) sstore(cur ^ 1);
        __syncthreads();
    }

    float *slab = p.slabs + (long)mc * 128 * p.Ktot + kc * 128;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = wc * 64 + nt * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = wr * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                slab[(long)row * p.Ktot + col] = acc[mt][nt][reg];
            }
        }
    if (do_bias) p.bias_slabs[(long)mc * 128 + tid] = bsum;
}

template <bool Y_DROP, bool X_ACT>
static hipError_t launch_tn(const TnParams &p, int B, hipStream_t s) {
    auto k = tn_gemm_kernel<Y_DROP, X_ACT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(p.Ktot / 128, B * p.chunks_per_video);
    hipLaunchKernelGGL(k, grid, dim3(256), TN_SMEM_BYTES, s, p);
    return hipGetLastError();
}

// out = sum over slabs, in a fixed order (bitwise reproducible).  1024 threads = 16 slab lanes x 64
// elements: lane group g sums slabs g, g+16, g+32, ... (eight independent loads in flight per
// thread), then the 16 partials are combined through LDS in lane-group order.
//   mode 0: out[e] = sum_s slabs[s*stride + e]                               (e < n_elems)
//   mode 1: conv k=3 weight: slab element (o, tap*128 + i) -> out[(o*128 + i)*3 + tap]
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float *slabs, int nslabs, long stride, float *out,
                                                            int n_elems, int mode) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (e < n_elems) {
        const float *p = slabs + e;
        int i = g;
        for (; i + 7 * 16 < nslabs; i += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(i + 16 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < nslabs; i += 16) s += p[(long)i * stride];
    }
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && e < n_elems) {
        float t = part[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][lane];
        if (mode == 0) {
            out[e] = t;
        } else {
            const int o = e / 384, r = e - o * 384;
            const int tap = r >> 7, i = r & 127;
            out[(o * 128 + i) * 3 + tap] = t;
        }
    }
}
