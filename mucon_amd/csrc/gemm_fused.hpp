// Two-stage fused NT kernel: two back-to-back GEMMs on one time tile, the intermediate tile handed over
// through LDS instead of HBM and a second launch.  Same MFMA core and staging as gemm_nt.hpp.
//
//   FWD  (one residual layer, reference src/core/modules/temporal.py:43-53 + pooling :137-142)
//        stage 1: h = act(dilated_conv(x) + b1)                        -> out1 (saved for the backward) and LDS
//        stage 2: y = x + dropout(conv_1x1(h) + b2) [-> max/sum pool]  -> out2 (and out_pre)
//   BWD  (data-gradient chain across a layer boundary: layer l+1's dilated conv, then layer l's conv_1x1)
//        stage 1: g = (dgrad_dilated_conv(dpre_{l+1}) + res1) * act'(mask1) -> out1 (gradient at layer l's output) and,
//                 multiplied by layer l's dropout mask, LDS.  With taps = 1 stage 1 is last_conv's data gradient.
//        stage 2: dpre_l = (g*drop . W2^T) * act'(h_l)                  -> out2
//
// Why: the coarse levels (T/8, T/16: <= 64 workgroups) are latency-bound -- every launch costs its own
// prologue, k-loop ramp and epilogue, and the K = 128 GEMMs alone reach < 25 % MFMA utilisation.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "gemm_nt.hpp"

constexpr int FUSED_HS = 132;  // padded row length of the intermediate tile (floats): 132 mod 64 = 4 -> conflict-free b128 reads
constexpr int fused_smem_bytes(int BM) { return nt_smem_bytes(BM) + BM * FUSED_HS * 4; }

struct FusedParams {
    int Trows;          // rows per video (same for both stages)
    // stage 1
    const float *A;     // [B][Trows][128]
    int taps, tap_step;
    const float *W1;    // [128][ldw1]
    int ldw1;
    const float *bias1; // FWD
    const float *res1;  // BWD: [B][Trows][128] or null
    const float *mask1; // BWD: [B][Trows][128] or null
    float *out1;        // [B][Trows][128]
    // stage 2
    const float *W2;    // [128][128]
    const float *bias2; // FWD
    const float *res2;  // FWD: x
    const float *mask2; // BWD: h
    float *out2;        // FWD: [B][Tout][128]; BWD: [B][Trows][128]
    float *out_pre;     // FWD POOL 1
    float slope;
    DropCfg drop;       // element index (b*Trows + t)*128 + c
};

template <int WM, int WAVES_M, bool BWD, int POOL>
__global__ __launch_bounds__(256) void nt_fused_kernel(const FusedParams p) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WN = 4 / WAVES_N;
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int NQA = BM / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 2 * BM * NT_LDS;
    float *Hs = Bs + 2 * 128 * NT_LDS;   // [BM][FUSED_HS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * BM;
    const int lrow = tid >> 3;
    const int lc4 = (tid & 7) * 4;
    const float *Ab = p.A + (long)b * p.Trows * 128;
    const long vbase = (long)b * p.Trows;

    f32x4 ra[NQA], rb[4];
    bool ra_ok[NQA];  // padding rows are zeroed at the LDS store, so that the loads stay in flight (see gemm_nt.hpp)
    f32x16 acc[WM][WN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    auto loadW = [&](const float *W, int ldw, int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rb[q] = *reinterpret_cast<const f32x4 *>(W + (long)(lrow + 32 * q) * ldw + kt * 32 + lc4);
    };
    auto storeW = [&](int buf) {
        float *w = Bs + buf * 128 * NT_LDS;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(w + (lrow + 32 * q) * NT_LDS + lc4) = rb[q];
    };
    auto loadA = [&](int kt) {
        const int tap = kt >> 2;  // 4 k-tiles of 32 per 128-channel tap
        const int kk = (kt & 3) * 32;
        const int off = (tap - (p.taps >> 1)) * p.tap_step;
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            const int t = t0 + lrow + 32 * q;
            const int ts = t + off;
            ra_ok[q] = (t < p.Trows) && (ts >= 0) && (ts < p.Trows);   // branch-free load from a clamped row
            const int tc = ts < 0 ? 0 : (ts >= p.Trows ? p.Trows - 1 : ts);
            ra[q] = *reinterpret_cast<const f32x4 *>(Ab + (long)tc * 128 + kk + lc4);
        }
    };
    auto storeA = [&](int buf) {
        float *a = As + buf * BM * NT_LDS;
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            f32x4 v = ra[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ra_ok[q] ? v[e] : 0.f;
            *reinterpret_cast<f32x4 *>(a + (lrow + 32 * q) * NT_LDS + lc4) = v;
        }
    };
    // 32-deep k-tile of MFMAs: A fragments from `Aw` (row stride lda floats), W fragments from the staging buffer
    auto mfma_tile = [&](const float *Aw, int lda, const float *Bw) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 av[WM][2], bv[WN][2];
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                av[m][0] = *reinterpret_cast<const f32x4 *>(Aw + m * 32 * lda + ks * 8);
                av[m][1] = *reinterpret_cast<const f32x4 *>(Aw + m * 32 * lda + ks * 8 + 4);
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                bv[n][0] = *reinterpret_cast<const f32x4 *>(Bw + n * 32 * NT_LDS + ks * 8);
                bv[n][1] = *reinterpret_cast<const f32x4 *>(Bw + n * 32 * NT_LDS + ks * 8 + 4);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][s >> 2][s & 3], bv[n][s >> 2][s & 3],
                                                                         acc[m][n], 0, 0, 0);
        }
    };
    const int a_row = wr * WM * 32 + (lane & 31);
    const int k_half = (lane >> 5) * 16;
    const int b_off = (wc * WN * 32 + (lane & 31)) * NT_LDS + k_half;

    // ---------------------------------------------------------------- stage 1
    zero_acc();
    const int nkt1 = p.taps * 4;
    loadA(0);
    loadW(p.W1, p.ldw1, 0);
    storeA(0);
    storeW(0);
    __syncthreads();
    for (int kt = 0; kt < nkt1; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt1) {
            loadA(kt + 1);
            loadW(p.W1, p.ldw1, kt + 1);
        } else {
            loadW(p.W2, 128, 0);  // first W2 tile rides under the last stage-1 k-tile
        }
        mfma_tile(As + cur * BM * NT_LDS + a_row * NT_LDS + k_half, NT_LDS, Bs + cur * 128 * NT_LDS + b_off);
        if (kt + 1 < nkt1) storeA(cur ^ 1);
        storeW(cur ^ 1);
        __syncthreads();
    }
    const int w2buf0 = nkt1 & 1;  // staging buffer that now holds W2 k-tile 0

    // stage-1 epilogue: global copy (saved / consumed later) + LDS copy (stage-2 A operand).  FULL tiles run
    // straight-line code (loads, math, stores batched per 32x32 tile; see gemm_nt.hpp).
    const bool full_tile = t0 + BM <= p.Trows;
    auto epilogue1 = [&](auto FULLT) {
        constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int col = (wc * WN + nt) * 32 + (lane & 31);
                const float bias = (!BWD && p.bias1) ? p.bias1[col] : 0.f;
                const int rbase = (wr * WM + mt) * 32 + 4 * (lane >> 5);
                float rres[16], rmask[16];
                if (BWD) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int t = t0 + rbase + (reg & 3) + 8 * (reg >> 2);
                        const long g = (vbase + (FULL ? t : min(t, p.Trows - 1))) * 128 + col;
                        rres[reg] = p.res1 ? p.res1[g] : 0.f;
                        rmask[reg] = p.mask1 ? p.mask1[g] : 1.f;
                    }
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = rbase + (reg & 3) + 8 * (reg >> 2);
                    const int t = t0 + row;
                    const long g = (vbase + t) * 128 + col;
                    float x = acc[mt][nt][reg] + bias;
                    float xl;
                    if (!BWD) {
                        x = act_f(x, p.slope);
                        xl = x;
                    } else {
                        x += rres[reg];
                        if (p.mask1) x *= act_grad(rmask[reg], p.slope);
                        xl = p.drop.thresh ? x * drop_mul(p.drop, (uint32_t)g) : x;
                    }
                    if (FULL || t < p.Trows) p.out1[g] = x;
                    Hs[row * FUSED_HS + col] = (FULL || t < p.Trows) ? xl : 0.f;
                }
            }
    };
    if (full_tile) epilogue1(std::true_type{});
    else epilogue1(std::false_type{});
    __syncthreads();

    // ---------------------------------------------------------------- stage 2 (K = 128: 4 k-tiles, A from Hs)
    zero_acc();
    for (int kt = 0; kt < 4; ++kt) {
        const int cur = (w2buf0 + kt) & 1;
        if (kt + 1 < 4) loadW(p.W2, 128, kt + 1);
        mfma_tile(Hs + a_row * FUSED_HS + kt * 32 + k_half, FUSED_HS, Bs + cur * 128 * NT_LDS + b_off);
        if (kt + 1 < 4) storeW(cur ^ 1);
        __syncthreads();
    }

    // stage-2 epilogue
    auto epilogue2 = [&](auto FULLT) {
        constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int col = (wc * WN + nt) * 32 + (lane & 31);
                const float bias = (!BWD && p.bias2) ? p.bias2[col] : 0.f;
                const int rbase = (wr * WM + mt) * 32 + 4 * (lane >> 5);
                float raux[16];   // FWD: residual x; BWD: h (mask)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int t = t0 + rbase + (reg & 3) + 8 * (reg >> 2);
                    const long g = (vbase + (FULL ? t : min(t, p.Trows - 1))) * 128 + col;
                    raux[reg] = BWD ? p.mask2[g] : p.res2[g];
                }
                float v[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int t = t0 + rbase + (reg & 3) + 8 * (reg >> 2);
                    const long g = (vbase + t) * 128 + col;
                    float x = acc[mt][nt][reg] + bias;
                    if (!BWD) {
                        if (p.drop.thresh) x *= drop_mul(p.drop, (uint32_t)g);
                        x += raux[reg];
                    } else {
                        x *= act_grad(raux[reg], p.slope);
                    }
                    v[reg] = x;
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int t = t0 + rbase + (reg & 3) + 8 * (reg >> 2);
                    const long g = (vbase + t) * 128 + col;
                    if (FULL || t < p.Trows) {
                        if (BWD || POOL == 0) p.out2[g] = v[reg];
                        if (!BWD && POOL == 1) p.out_pre[g] = v[reg];
                    }
                }
                if (!BWD && POOL != 0) {
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp) {
                        const int te = t0 + rbase + ((2 * rp) & 3) + 8 * ((2 * rp) >> 2);
                        if (FULL || te + 1 < p.Trows) {
                            const long g = ((long)b * (p.Trows >> 1) + (te >> 1)) * 128 + col;
                            p.out2[g] = (POOL == 1) ? fmaxf(v[2 * rp], v[2 * rp + 1]) : (v[2 * rp] + v[2 * rp + 1]);
                        }
                    }
                }
            }
    };
    if (full_tile) epilogue2(std::true_type{});
    else epilogue2(std::false_type{});
}

template <int WM, int WAVES_M, bool BWD, int POOL>
static hipError_t launch_fused_cfg(const FusedParams &p, int B, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * 32;
    auto k = nt_fused_kernel<WM, WAVES_M, BWD, POOL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, fused_smem_bytes(BM));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((p.Trows + BM - 1) / BM, B);
    hipLaunchKernelGGL(k, grid, dim3(256), fused_smem_bytes(BM), s, p);
    return hipGetLastError();
}

extern int g_fused_bm;  // 0 = automatic (tuning hook: MUCON_FUSED_BM = 32 / 64)
template <bool BWD, int POOL>
static hipError_t launch_fused(const FusedParams &p, int B, hipStream_t s) {
    int bm = g_fused_bm ? g_fused_bm : (((long)B * p.Trows >= 512L * 64) ? 64 : 32);
    if (bm == 64) return launch_fused_cfg<1, 2, BWD, POOL>(p, B, s);
    return launch_fused_cfg<1, 1, BWD, POOL>(p, B, s);
}
