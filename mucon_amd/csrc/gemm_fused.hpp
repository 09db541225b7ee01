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
//        With POOL = 3 / 4 the boundary is a pooled one: stage 1 works on the coarse level's rows, its epilogue routes every
//        value onto the arg-max row of the forward pair (or both rows, sum pooling) of the finer level -- in global memory
//        (out1: the un-pooled gradient, needed by the weight gradients and as the next residual) and in LDS -- and stage 2
//        multiplies twice as many rows.
//
// Why: the coarse levels (T/8, T/16: <= 128 workgroups) are latency-bound -- every launch costs its own
// prologue, k-loop ramp and epilogue, and the K = 128 GEMMs alone reach < 25 % MFMA utilisation.
//
// Pipeline: the k-tiles of both stages form ONE sequence v = 0 .. nkt1+3 (stage-1 tiles of W1 with their A rows,
// then the four k-tiles of W2), staged exactly like gemm_nt.hpp: two register sets, tile v+2 requested while v
// multiplies, the three phases load / MFMA / LDS-store pinned with sched_barrier, no guards around loads.  The first
// two W2 tiles are therefore already in flight while stage 1 finishes and its epilogue runs.
//
// KS = 2 (opt-in for the BM = 32 variant, MUCON_FUSED_KS=2): 512 threads = 8 waves; waves 4-7 take the second half
// of every 32-deep k-tile (same output tiles) and the two partial accumulators meet through the intermediate-tile LDS
// buffer before each epilogue.  Measured on MI355X: no gain at B=8 x T=4096 (1.386 vs 1.381 ms per step) and a loss at
// batch 1 (2.65 vs 2.29 ms per video), and it changes the summation order (a video alone is no longer bitwise equal
// to the same video inside a batch) -- so KS = 1 is the default.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "dispatch.hpp"
#include "gemm_nt.hpp"

constexpr int FUSED_HS = 132;  // padded row length of the intermediate tile (floats): 132 mod 64 = 4 -> conflict-free b128 reads
// LDS: [intermediate tile Hs, BM x 132] [W staging, 2 x 128 x 36].  The A staging buffers (2 x BM x 36) OVERLAY the
// start of Hs: they are dead once stage 1's k-loop has passed its last barrier, which is before the stage-1 epilogue
// writes Hs.  BM = 64: 70.7 KB instead of 89 KB, i.e. TWO workgroups per CU instead of one (160 KB LDS).
constexpr int fused_smem_bytes(int BM, int R2 = 1) { return (R2 * BM * FUSED_HS + 2 * 128 * NT_LDS) * 4; }
static_assert(FUSED_HS >= 2 * NT_LDS, "the A staging buffers must fit inside the intermediate tile");

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
    // BWD across a pooled boundary (POOL 3 max / 4 sum): stage 1 runs on the Trows rows of the coarse level, its epilogue
    // un-pools onto the 2 x rows of the fine level (out1, mask2, out2 are [B][Tfine][128]), stage 2 runs on those
    const float *ypre;  // POOL 3: the forward's un-pooled rows [B][Tfine][128]
    int Tfine;
    float slope;
    DropCfg drop;       // element index (b*Trows + t)*128 + c
};

template <int WM, int WAVES_M, int KS, bool BWD, int POOL, int MT = 32>
__global__ __launch_bounds__(256 * KS) void nt_fused_kernel(const FusedParams p) {
    using TL = NtTile<MT>;
    constexpr int NREG = TL::NREG;
    static_assert(MT == 32 || KS == 1, "the k-split variant exists for 32x32 tiles only");
    constexpr int NTHR = 256 * KS;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WN = (128 / WAVES_N) / MT;
    constexpr int BM = WAVES_M * WM * MT;
    constexpr int LROWS = NTHR / 8;                       // rows one pass of the cooperative loader covers
    constexpr int NQA = BM > LROWS ? BM / LROWS : 1;      // A float4 loads per thread (BM < LROWS: rows wrap, duplicates)
    constexpr int NQW = 128 / LROWS;                      // W float4 loads per thread
    constexpr bool UNPOOL = BWD && POOL >= 3;             // pooled boundary: stage 2 on 2 x BM rows of the finer level
    constexpr int R2 = UNPOOL ? 2 : 1;
    static_assert(!UNPOOL || KS == 1, "no k-split across a pooled boundary");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Hs = smem;                    // [R2 * BM][FUSED_HS]
    float *As = smem;                    // [2][BM][NT_LDS], dead before Hs is written
    float *Bs = smem + R2 * BM * FUSED_HS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgrp = wave_all >> 2;      // k-half this wave multiplies (KS = 2), 0 otherwise
    const int wave = wave_all & 3;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * BM;
    const int lrow = tid >> 3;
    const int lc4 = (tid & 7) * 4;
    const float *Ab = p.A + (long)b * p.Trows * 128;
    const long vbase = (long)b * p.Trows;
    const int nkt1 = p.taps * 4;         // stage-1 k-tiles (4 per 128-channel tap)
    const int vlast = nkt1 + 3;

    f32x4 ra[2][NQA], rb[2][NQW];
    bool ra_ok[2][NQA];  // padding rows are zeroed at the LDS store, so that the loads stay in flight (see gemm_nt.hpp)
    typename TL::Acc acc[WM][WN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int e = 0; e < NREG; ++e) acc[i][j][e] = 0.f;
    };
    // tile v of the unified sequence: only ISSUES loads (W1 / W2 chosen by select, A rows clamped)
    auto gload = [&](int v, auto SET, auto WITH_A) {
        constexpr int S = decltype(SET)::value;
        const bool s2 = v >= nkt1;
        const float *Wp = s2 ? p.W2 : p.W1;
        const int ldw = s2 ? 128 : p.ldw1;
        const int kcol = (s2 ? v - nkt1 : v) * 32 + lc4;
#pragma unroll
        for (int q = 0; q < NQW; ++q) rb[S][q] = *reinterpret_cast<const f32x4 *>(Wp + (long)(lrow + LROWS * q) * ldw + kcol);
        if (decltype(WITH_A)::value) {
            const int kt = min(v, nkt1 - 1);
            const int tap = kt >> 2;
            const int kk = (kt & 3) * 32;
            const int off = (tap - (p.taps >> 1)) * p.tap_step;
#pragma unroll
            for (int q = 0; q < NQA; ++q) {
                const int t = t0 + ((lrow + LROWS * q) & (BM - 1));
                const int ts = t + off;
                ra_ok[S][q] = (t < p.Trows) && (ts >= 0) && (ts < p.Trows);
                const int tc = ts < 0 ? 0 : (ts >= p.Trows ? p.Trows - 1 : ts);
                ra[S][q] = *reinterpret_cast<const f32x4 *>(Ab + (long)tc * 128 + kk + lc4);
            }
        }
    };
    auto sstore = [&](int buf, auto SET, auto WITH_A) {
        constexpr int S = decltype(SET)::value;
        float *w = Bs + buf * 128 * NT_LDS;
#pragma unroll
        for (int q = 0; q < NQW; ++q) *reinterpret_cast<f32x4 *>(w + (lrow + LROWS * q) * NT_LDS + lc4) = rb[S][q];
        if (decltype(WITH_A)::value) {
            float *a = As + buf * BM * NT_LDS;
#pragma unroll
            for (int q = 0; q < NQA; ++q) {
                f32x4 v = ra[S][q];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ra_ok[S][q] ? v[e] : 0.f;
                *reinterpret_cast<f32x4 *>(a + ((lrow + LROWS * q) & (BM - 1)) * NT_LDS + lc4) = v;   // duplicates store equal values
            }
        }
    };
    // 32-deep k-tile of MFMAs (this wave's half of it when KS = 2): A fragments from `Aw` (row stride lda floats),
    // W fragments from the staging buffer
    typename TL::Acc acc2[R2 * WM][WN];   // stage-2 accumulators (aliases nothing: stage 1's are dead by then)
    auto mfma_tile_n = [&](auto &accs, auto MC, const float *Aw, int lda, const float *Bw) {
        constexpr int NM = decltype(MC)::value;
        if (MT == 32) {
#pragma unroll
            for (int kq = 0; kq < 2 / KS; ++kq) {
                const int ks = KS == 2 ? kgrp : kq;
                f32x4 av[NM][2], bv[WN][2];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    av[m][0] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * lda + ks * 8);
                    av[m][1] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * lda + ks * 8 + 4);
                }
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    bv[n][0] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS + ks * 8);
                    bv[n][1] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS + ks * 8 + 4);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int m = 0; m < NM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n) nt_mfma<MT>(accs[m][n], av[m][s >> 2][s & 3], bv[n][s >> 2][s & 3]);
            }
        } else {   // 16x16x4 tiles: the lane's eight k of the 32-deep tile in two 16-byte reads
            f32x4 av[NM][2], bv[WN][2];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                av[m][0] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * lda);
                av[m][1] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * lda + 4);
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                bv[n][0] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS);
                bv[n][1] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS + 4);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) nt_mfma<MT>(accs[m][n], av[m][s >> 2][s & 3], bv[n][s >> 2][s & 3]);
        }
    };
    auto mfma_tile = [&](const float *Aw, int lda, const float *Bw) {
        mfma_tile_n(acc, std::integral_constant<int, WM>{}, Aw, lda, Bw);
    };
    // KS = 2: the second k-half's partial sums cross to the first through Hs (each element is written and read by
    // the same lane position of the wave pair, so no barrier is needed between this read and the epilogue's write)
    auto merge_halves = [&]() {
        if (KS == 2) {
            if (kgrp == 1) {
#pragma unroll
                for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                        for (int reg = 0; reg < NREG; ++reg) {
                            const int row = (wr * WM + mt) * MT + TL::row0(lane) + TL::rowr(reg);
                            Hs[row * FUSED_HS + (wc * WN + nt) * MT + (lane & (MT - 1))] = acc[mt][nt][reg];
                        }
            }
            __syncthreads();
            if (kgrp == 0) {
#pragma unroll
                for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                        for (int reg = 0; reg < NREG; ++reg) {
                            const int row = (wr * WM + mt) * MT + TL::row0(lane) + TL::rowr(reg);
                            acc[mt][nt][reg] += Hs[row * FUSED_HS + (wc * WN + nt) * MT + (lane & (MT - 1))];
                        }
            }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using YES = std::true_type;
    using NO = std::false_type;
    const int a_row = wr * WM * MT + (lane & (MT - 1));
    const int k_half = TL::koff(lane);
    const int b_off = (wc * WN * MT + (lane & (MT - 1))) * NT_LDS + k_half;

    // ---------------------------------------------------------------- stage 1
    zero_acc();
    gload(0, S0{}, YES{});
    gload(1, S1{}, YES{});
    sstore(0, S0{}, YES{});
    __syncthreads();
    for (int kt = 0; kt < nkt1; kt += 2) {   // nkt1 is 4 or 12; tiles >= nkt1 are W2's
        gload(kt + 2, S0{}, YES{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile(As + a_row * NT_LDS + k_half, NT_LDS, Bs + b_off);
        __builtin_amdgcn_sched_barrier(0);
        sstore(1, S1{}, YES{});
        __syncthreads();
        gload(kt + 3, S1{}, YES{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile(As + BM * NT_LDS + a_row * NT_LDS + k_half, NT_LDS, Bs + 128 * NT_LDS + b_off);
        __builtin_amdgcn_sched_barrier(0);
        sstore(0, S0{}, YES{});
        __syncthreads();
    }
    // now: staging buffer 0 holds W2 k-tile 0, register set 1 holds W2 k-tile 1 (in flight)

    // stage-1 epilogue: global copy (saved / consumed later) + LDS copy (stage-2 A operand).  FULL tiles run
    // straight-line code (loads, math, stores batched per 32x32 tile; see gemm_nt.hpp).
    const bool full_tile = t0 + BM <= p.Trows;
    auto epilogue1 = [&](auto FULLT) {
        constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int col = (wc * WN + nt) * MT + (lane & (MT - 1));
                const float bias = (!BWD && p.bias1) ? p.bias1[col] : 0.f;
                const int rbase = (wr * WM + mt) * MT + TL::row0(lane);
                float rres[NREG], rmask[NREG];
                if (BWD) {
#pragma unroll
                    for (int reg = 0; reg < NREG; ++reg) {
                        const int t = t0 + rbase + TL::rowr(reg);
                        const long g = (vbase + (FULL ? t : min(t, p.Trows - 1))) * 128 + col;
                        rres[reg] = p.res1 ? p.res1[g] : 0.f;
                        rmask[reg] = p.mask1 ? p.mask1[g] : 1.f;
                    }
                }
#pragma unroll
                for (int reg = 0; reg < NREG; ++reg) {
                    const int row = rbase + TL::rowr(reg);
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
                    if (!UNPOOL) {
                        if (FULL || t < p.Trows) p.out1[g] = x;
                        Hs[row * FUSED_HS + col] = (FULL || t < p.Trows) ? xl : 0.f;
                    } else {
                        // max-pool backward (first wins ties, as torch) / sum-pool backward: rows 2t, 2t+1 of the fine level
                        const bool valid = FULL || t < p.Trows;
                        const long gf = ((long)b * p.Tfine + 2 * (valid ? t : 0)) * 128 + col;
                        bool second = false;
                        if (POOL == 3) second = p.ypre[gf + 128] > p.ypre[gf];
                        const float u0 = (POOL == 4 || !second) ? x : 0.f, u1 = (POOL == 4 || second) ? x : 0.f;
                        float h0 = u0, h1 = u1;
                        if (p.drop.thresh) {
                            h0 *= drop_mul(p.drop, (uint32_t)gf);
                            h1 *= drop_mul(p.drop, (uint32_t)(gf + 128));
                        }
                        if (valid) {
                            p.out1[gf] = u0;
                            p.out1[gf + 128] = u1;
                            if (t == p.Trows - 1 && 2 * p.Trows < p.Tfine) {   // odd trailing row of the fine level: no gradient
                                p.out1[gf + 256] = 0.f;
                                p.out2[gf + 256] = 0.f;
                            }
                        }
                        Hs[(2 * row) * FUSED_HS + col] = valid ? h0 : 0.f;
                        Hs[(2 * row + 1) * FUSED_HS + col] = valid ? h1 : 0.f;
                    }
                }
            }
    };
    merge_halves();
    if (kgrp == 0) {
        if (full_tile) epilogue1(std::true_type{});
        else epilogue1(std::false_type{});
    }
    __syncthreads();

    // ---------------------------------------------------------------- stage 2 (K = 128: 4 k-tiles, A from Hs)
    constexpr int WM2 = R2 * WM;
    using M2 = std::integral_constant<int, WM2>;
    const int a_row2 = wr * WM2 * MT + (lane & (MT - 1));
#pragma unroll
    for (int i = 0; i < WM2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < NREG; ++e) acc2[i][j][e] = 0.f;
    for (int kt = 0; kt < 4; kt += 2) {
        gload(min(nkt1 + kt + 2, vlast), S0{}, NO{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile_n(acc2, M2{}, Hs + a_row2 * FUSED_HS + kt * 32 + k_half, FUSED_HS, Bs + b_off);
        __builtin_amdgcn_sched_barrier(0);
        sstore(1, S1{}, NO{});
        __syncthreads();
        gload(min(nkt1 + kt + 3, vlast), S1{}, NO{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile_n(acc2, M2{}, Hs + a_row2 * FUSED_HS + (kt + 1) * 32 + k_half, FUSED_HS, Bs + 128 * NT_LDS + b_off);
        __builtin_amdgcn_sched_barrier(0);
        sstore(0, S0{}, NO{});
        __syncthreads();
    }

    // stage-2 epilogue (rows of the stage-2 level: the finer one across a pooled boundary)
    const int rows2 = UNPOOL ? min(2 * p.Trows, p.Tfine) : p.Trows;   // rows with a gradient
    const long vbase2 = UNPOOL ? (long)b * p.Tfine : vbase;
    const int t02 = R2 * t0;
    auto epilogue2 = [&](auto FULLT) {
        constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
        for (int mt = 0; mt < WM2; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int col = (wc * WN + nt) * MT + (lane & (MT - 1));
                const float bias = (!BWD && p.bias2) ? p.bias2[col] : 0.f;
                const int rbase = (wr * WM2 + mt) * MT + TL::row0(lane);
                float raux[NREG];   // FWD: residual x; BWD: h (mask)
#pragma unroll
                for (int reg = 0; reg < NREG; ++reg) {
                    const int t = t02 + rbase + TL::rowr(reg);
                    const long g = (vbase2 + (FULL ? t : min(t, rows2 - 1))) * 128 + col;
                    raux[reg] = BWD ? p.mask2[g] : p.res2[g];
                }
                float v[NREG];
#pragma unroll
                for (int reg = 0; reg < NREG; ++reg) {
                    const int t = t02 + rbase + TL::rowr(reg);
                    const long g = (vbase2 + t) * 128 + col;
                    float x = acc2[mt][nt][reg] + bias;
                    if (!BWD) {
                        if (p.drop.thresh) x *= drop_mul(p.drop, (uint32_t)g);
                        x += raux[reg];
                    } else {
                        x *= act_grad(raux[reg], p.slope);
                    }
                    v[reg] = x;
                }
#pragma unroll
                for (int reg = 0; reg < NREG; ++reg) {
                    const int t = t02 + rbase + TL::rowr(reg);
                    const long g = (vbase2 + t) * 128 + col;
                    if (FULL || t < rows2) {
                        if (BWD || POOL == 0) p.out2[g] = v[reg];
                        if (!BWD && POOL == 1) p.out_pre[g] = v[reg];
                    }
                }
                if (!BWD && POOL != 0) {
#pragma unroll
                    for (int rp = 0; rp < NREG / 2; ++rp) {
                        const int te = t0 + rbase + TL::rowr(2 * rp);
                        if (FULL || te + 1 < p.Trows) {
                            const long g = ((long)b * (p.Trows >> 1) + (te >> 1)) * 128 + col;
                            p.out2[g] = (POOL == 1) ? fmaxf(v[2 * rp], v[2 * rp + 1]) : (v[2 * rp] + v[2 * rp + 1]);
                        }
                    }
                }
            }
    };
    if (KS == 2) {   // (never with UNPOOL) hand the second k-half's stage-2 sums over
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = acc2[i][j];
        merge_halves();
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc2[i][j] = acc[i][j];
    }
    if (kgrp == 0) {
        if (full_tile) epilogue2(std::true_type{});
        else epilogue2(std::false_type{});
    }
}

template <int WM, int WAVES_M, int KS, bool BWD, int POOL, int MT = 32>
static hipError_t launch_fused_cfg(const FusedParams &p, int B, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * MT;
    constexpr int R2 = (BWD && POOL >= 3) ? 2 : 1;
    auto k = nt_fused_kernel<WM, WAVES_M, KS, BWD, POOL, MT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, fused_smem_bytes(BM, R2));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((p.Trows + BM - 1) / BM, B);
    hipLaunchKernelGGL(k, grid, dim3(256 * KS), fused_smem_bytes(BM, R2), s, p);
    return hipGetLastError();
}

template <bool BWD, int POOL>
static hipError_t launch_fused(const FusedParams &p, int B, hipStream_t s) {
    int bm = kFusedBm ? kFusedBm : (((long)B * p.Trows >= 512L * 64) ? 64 : ((long)B * p.Trows < g_nt_bm16_rows ? 16 : 32));
    if (bm == 64) return launch_fused_cfg<1, 2, 1, BWD, POOL>(p, B, s);
    if (bm == 16) return launch_fused_cfg<1, 1, 1, BWD, POOL, 16>(p, B, s);
    if constexpr (!(BWD && POOL >= 3))
        if (kFusedKs == 2) return launch_fused_cfg<1, 1, 2, BWD, POOL>(p, B, s);
    return launch_fused_cfg<1, 1, 1, BWD, POOL>(p, B, s);
}
