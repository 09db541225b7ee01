// NT GEMM core on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact f32 = an fmaf chain):
//
//     out[t][n] = epilogue( sum_{tap, c} A[t + (tap - taps/2) * tap_step][c] * W[n][tap*Kc + c] )
//
// One workgroup = 256 threads = 4 waves computes 128 time steps x 128 output channels of one
// video; each wave owns a 64x64 quadrant as 2x2 MFMA tiles (64 accumulator VGPRs).  A and W
// tiles (128 rows x 32 k) are staged global -> registers -> LDS, double buffered, one barrier
// per k-tile; the global loads of tile k+1 are in flight under the 64 MFMAs of tile k.
// LDS rows are padded to 36 floats so that the 16-lane groups of ds_read_b128 hit 64 distinct
// banks (rows r, r+1.. map to bank offsets 36r mod 64: a permutation of the multiples of 4).
//
// It serves every convolution of the reference's encoder (src/core/modules/temporal.py):
//   first_conv (:133)  taps=1 Kc=2048        dilated_conv (:48) taps=3 Kc=128 (zero pad = dilation)
//   conv_1x1 (:50)     taps=1 Kc=128         last_conv (:145)   taps=1 Kc=128, ReLU on the input
// and, with transposed / re-packed weights, their data gradients.
#pragma once
#include "common.hpp"

constexpr int NT_BM = 128;
constexpr int NT_BK = 32;
constexpr int NT_LDS = 36;  // padded row length (floats)
constexpr int NT_SMEM_BYTES = 2 * 2 * NT_BM * NT_LDS * 4;  // A and W tiles, double buffered

struct NtParams {
    const float *A;     // [B][Ta][lda] source rows
    long a_bstride;     // floats between videos in A
    int lda, Ta;
    int Trows;          // output rows per video (before pooling)
    int taps, tap_step; // 1 or 3 taps; signed row offset between taps
    int Kc;             // channels per tap (multiple of 32)
    const float *W;     // [128][taps*Kc]
    const float *bias;  // [128] or null
    float *out;         // [B][Tout][128]; Tout = Trows (POOL 0) or Trows/2 (POOL 1,2)
    float *out_pre;     // POOL 1: un-pooled rows [B][Trows][128], kept for the max-pool backward
    const float *res;   // EPI_RES : [B][Trows][128] added after bias/act/dropout
    const float *mask;  // EPI_MASK: [B][Trows][128]; result *= act'(mask) (skipped when null)
    float slope;        // 0 = ReLU, 0.01 = leaky ReLU
    DropCfg drop;       // element index (b*Trows + t)*128 + c
};

template <bool PRO_ACT, bool PRO_DROP, bool EPI_ACT, bool EPI_DROP, bool EPI_RES, bool EPI_MASK, int POOL>
__global__ __launch_bounds__(256) void nt_gemm_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 2 * NT_BM * NT_LDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * NT_BM;
    const int ktiles_per_tap = p.Kc >> 5;
    const int nkt = p.taps * ktiles_per_tap;
    const int Ktot = p.taps * p.Kc;
    const int lrow = tid >> 3;
    const int lc4 = (tid & 7) * 4;
    const float *Ab = p.A + (long)b * p.a_bstride;

    f32x4 ra[4], rb[4];

    auto gload = [&](int kt) {
        const int tap = kt / ktiles_per_tap;
        const int kk = (kt - tap * ktiles_per_tap) * 32;
        const int off = (tap - (p.taps >> 1)) * p.tap_step;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lrow + 32 * q;
            const int t = t0 + r;
            const int ts = t + off;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < p.Trows && ts >= 0 && ts < p.Ta) {
                v = *reinterpret_cast<const f32x4 *>(Ab + (long)ts * p.lda + kk + lc4);
                if (PRO_ACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.slope);
                }
                if (PRO_DROP) {
                    if (p.drop.thresh) {
                        const uint32_t idx = (uint32_t)(b * p.Trows + t) * 128u + (uint32_t)(kk + lc4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= drop_mul(p.drop, idx + e);
                    }
                }
            }
            ra[q] = v;
            rb[q] = *reinterpret_cast<const f32x4 *>(p.W + (long)r * Ktot + kt * 32 + lc4);
        }
    };
    auto sstore = [&](int buf) {
        float *a = As + buf * NT_BM * NT_LDS;
        float *w = Bs + buf * NT_BM * NT_LDS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f32x4 *>(a + (lrow + 32 * q) * NT_LDS + lc4) = ra[q];
            *reinterpret_cast<f32x4 *>(w + (lrow + 32 * q) * NT_LDS + lc4) = rb[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // lane (i = lane&31, h = lane>>5) feeds row i of its tile and k = 16h + s at MFMA step s:
    // both operands use the same k permutation inside the 32-wide tile, so the sum is unchanged.
    const int a_off = (wr * 64 + (lane & 31)) * NT_LDS + (lane >> 5) * 16;
    const int b_off = (wc * 64 + (lane & 31)) * NT_LDS + (lane >> 5) * 16;

    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const float *Aw = As + cur * NT_BM * NT_LDS + a_off;
        const float *Bw = Bs + cur * NT_BM * NT_LDS + b_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 av[2][2], bv[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                av[m][0] = *reinterpret_cast<const f32x4 *>(Aw + m * 32 * NT_LDS + ks * 8);
                av[m][1] = *reinterpret_cast<const f32x4 *>(Aw + m * 32 * NT_LDS + ks * 8 + 4);
                bv[m][0] = *reinterpret_cast<const f32x4 *>(Bw + m * 32 * NT_LDS + ks * 8);
                bv[m][1] = *reinterpret_cast<const f32x4 *>(Bw + m * 32 * NT_LDS + ks * 8 + 4);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float a0 = av[0][s >> 2][s & 3], a1 = av[1][s >> 2][s & 3];
                const float b0 = bv[0][s >> 2][s & 3], b1 = bv[1][s >> 2][s & 3];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) sstore(cur ^ 1);
        __syncthreads();
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5);
    // registers (4q, 4q+1) and (4q+2, 4q+3) hold time steps (2i, 2i+1): max_pool1d(2) pairs.
    const long vbase = (long)b * p.Trows;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = wc * 64 + nt * 32 + (lane & 31);
            const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                float v[2];
                int tt[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int reg = rp * 2 + u;
                    const int row = wr * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    const int t = t0 + row;
                    tt[u] = t;
                    float x = acc[mt][nt][reg] + bias;
                    if (t < p.Trows) {
                        const long g = (vbase + t) * 128 + col;
                        if (EPI_ACT) x = act_f(x, p.slope);
                        if (EPI_DROP) {
                            if (p.drop.thresh) x *= drop_mul(p.drop, (uint32_t)g);
                        }
                        if (EPI_RES) x += p.res[g];
                        if (EPI_MASK) {
                            if (p.mask) x *= act_grad(p.mask[g], p.slope);
                        }
                        if (POOL == 0) p.out[g] = x;
                        if (POOL == 1) p.out_pre[g] = x;
                    }
                    v[u] = x;
                }
                if (POOL != 0) {
                    if (tt[1] < p.Trows) {  // floor pooling drops an odd last step
                        const long g = ((long)b * (p.Trows >> 1) + (tt[0] >> 1)) * 128 + col;
                        p.out[g] = (POOL == 1) ? fmaxf(v[0], v[1]) : (v[0] + v[1]);
                    }
                }
            }
        }
    }
}

template <bool PRO_ACT, bool PRO_DROP, bool EPI_ACT, bool EPI_DROP, bool EPI_RES, bool EPI_MASK, int POOL>
static hipError_t launch_nt(const NtParams &p, int B, hipStream_t s) {
    auto k = nt_gemm_kernel<PRO_ACT, PRO_DROP, EPI_ACT, EPI_DROP, EPI_RES, EPI_MASK, POOL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NT_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((p.Trows + NT_BM - 1) / NT_BM, B);
    hipLaunchKernelGGL(k, grid, dim3(256), NT_SMEM_BYTES, s, p);
    return hipGetLastError();
}
