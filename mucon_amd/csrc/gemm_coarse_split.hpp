// The two-stage residual-layer launches of the COARSE levels (< 16,384 rows in the batch: T/4 ... T/16, and every launch of a
// batch-1 step) on the bf16 MFMA with exact three-way operand splitting -- same arithmetic, same weight images and the same
// register-resident hand-over between the two products as gemm_fused_split.hpp, organised for latency instead of throughput.
//
// These launches are chains: with 16 rows per workgroup a level has 128 ... 512 workgroups, and what a workgroup costs is the
// length of its dependent sequence (the f32-MFMA kernels: 16 k-tiles of LDS store / barrier / LDS read / MFMA, 8 - 20 us a
// launch at 5 - 25 % MFMA utilisation).  Here the four waves of a workgroup SPLIT K:
//   stage 1  wave w multiplies channels 32 w .. 32 w + 31 of every tap (one 32-deep step per tap) -- 48 MFMAs per tap instead of
//            192 -- and loads exactly the activation values and weight fragments it multiplies: every weight fragment of the
//            layer is used by ONE wave of the workgroup, once, so it goes straight from L2 into operand registers (the images of
//            fs_pack_body are lane-linear: one coalesced 1 KB load per fragment), no LDS staging, no barrier in the loop;
//   reduce   the four partial tiles meet in LDS (one barrier); wave w finishes channel blocks 2 w, 2 w + 1: bias / non-linearity
//            (FWD) or residual / mask / max-pool un-routing / dropout replay (BWD) -- which are exactly the 32 channels that
//            form 32-deep step w of stage 2's reduction, in accumulator order;
//   stage 2  wave w multiplies that step (48 MFMAs), second reduction, wave w finishes output blocks 2 w, 2 w + 1.
// The dependent sequence is 192 MFMAs and two LDS exchanges instead of 768 MFMAs and 16 barriers; sums are taken in a fixed
// order (wave 0 .. 3): bitwise reproducible, a video alone == the same video inside a batch.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "gemm_fused.hpp"
#include "gemm_fused_split.hpp"

// Timing builds (MUCON_HIPCC_FLAGS=-DCS_STAMP=1; tools/cs_stamps.py, profiles/r05_cs_kernel_phase_stamps.txt): every cs_kernel launch gets the next
// of 64 slots; wave w of its first workgroup and of a workgroup in the middle of the grid leave nine s_memtime values (kernel entry, loads
// issued, operands split = first activation rows arrived, stage-1 MFMAs issued, first exchange done, stage-1 epilogue done, stage-2 MFMAs
// issued, second exchange done, end) and the s_memrealtime pair of entry / end in a buffer of their own.  No output depends on a stamp.
#ifndef CS_STAMP
#define CS_STAMP 0
#endif
#if CS_STAMP
__device__ long long g_cs_stamps[64][2][4][12];
struct CsStampInfo { int bwd, pool, taps, one, rb, gx, gy, rows; };
static CsStampInfo g_cs_info[64];
static int g_cs_slot = 0;
#define CS_T(k) do { __builtin_amdgcn_sched_barrier(0); st_[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define CS_SLOT_PARAM , const int cs_slot
#else
#define CS_T(k) do { } while (0)
#define CS_SLOT_PARAM
#endif

// TAPS = 1: a 1x1 product (a dilated conv whose dilation reaches past the sequence -- W1img then points at the centre tap's four
// steps -- or last_conv); ONE: stage 1 only; PRO_ACT: the non-linearity on the loaded rows (last_conv's input, temporal.py:144).
template <bool BWD, int POOL, int TAPS, bool ONE, bool PRO_ACT, int RB>
// (r6) The arguments every wave needs in front of its first loads -- the operand pointers, the row count, the tap distance -- are LEADING SCALAR arguments: gfx950's
// kernel-argument preload (-mllvm -amdgpu-kernarg-preload-count=16, mucon_amd/build.py) hands them over in SGPRs with the wave, so the activation and weight loads are
// issued without the ~0.25 us scalar-memory round trip a by-value struct costs (tools/experiments/kernarg_latency_probe.hip); the rest of FusedParams arrives meanwhile.
__global__ __launch_bounds__(256) void cs_kernel(const float *__restrict__ A_, const uint16_t *__restrict__ W1img, const uint16_t *__restrict__ W2img, const int Trows_,
                                                 const int tap_step_, const FusedParams p CS_SLOT_PARAM) {
    constexpr bool UNPOOL = BWD && POOL >= 3;
    constexpr int R2 = UNPOOL ? 2 : 1;
#if CS_STAMP
    long long st_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long real0_ = __builtin_amdgcn_s_memrealtime();
    auto cs_publish = [&](int upto) {
        const int which = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : ((blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2) ? 1 : -1);
        if (which >= 0 && (threadIdx.x & 63) == 0) {
            long long *o = g_cs_stamps[cs_slot & 63][which][threadIdx.x >> 6];
            for (int k = 0; k < 9; ++k) o[k] = k <= upto ? st_[k] : st_[upto];
            o[9] = real0_;
            o[10] = __builtin_amdgcn_s_memrealtime();
        }
    };
#endif
    CS_T(0);
    // one exchange buffer for both reductions ([wave][row set][channel block][lane]): RB * R2 * 32 KB
    __shared__ f32x4 red[4 * RB * R2 * 8 * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const long vbase = (long)b * Trows_;
    // RB row blocks of 16 time steps per workgroup: every weight fragment is loaded once and multiplies all of them
    int trow_raw[RB];
    bool valid[RB];
    long grow[RB], grow2[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        trow_raw[rb] = (blockIdx.x * RB + rb) * 16 + c;
        valid[rb] = trow_raw[rb] < Trows_;
        const int tcl = min(trow_raw[rb], Trows_ - 1);
        grow[rb] = (vbase + tcl) * 128 + 4 * g;
        grow2[rb] = UNPOOL ? ((long)b * p.Tfine + 2 * tcl) * 128 + 4 * g : grow[rb];
    }

    // ---- loads: the wave's activation slices, then its weight fragments in the order they are multiplied
    f32x4 ra[RB][TAPS][2];
    bool rok[RB][TAPS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < TAPS; ++i) {
            const int ts = trow_raw[rb] + (i - TAPS / 2) * tap_step_;
            rok[rb][i] = valid[rb] && ts >= 0 && ts < Trows_;
            const float *src = A_ + (vbase + min(max(ts, 0), Trows_ - 1)) * 128 + 32 * w + 8 * g;
            ra[rb][i][0] = *reinterpret_cast<const f32x4 *>(src);
            ra[rb][i][1] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
    constexpr int NP = TAPS * 8;      // (tap, channel block) pairs of stage 1, three plane fragments each
    constexpr int D = 8;              // pairs in flight (12: neutral, profiles/r05_cs_kernel_phase_stamps.txt)
    bf16x8 wf[D][3];
    const uint16_t *w1 = W1img + (long)w * FS_WSTEP + lane * 8;      // step (tap * 4 + w): + tap * 4 * FS_WSTEP
    auto loadW1 = [&](int i, int slot) {
        const uint16_t *src = w1 + (long)(i >> 3) * (4 * FS_WSTEP) + (i & 7) * 512;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[slot][pl] = *reinterpret_cast<const bf16x8 *>(src + pl * (128 * 32));
    };
#pragma unroll
    for (int i = 0; i < D && i < NP; ++i) loadW1(i, i);

    struct Planes { bf16x8 pl[3]; };
    auto split8 = [&](const float (&x)[8]) {
        u32x4 hh, mm, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t a, bb, cc;
            sp_split2(x[2 * e], x[2 * e + 1], a, bb, cc);
            hh[e] = a;
            mm[e] = bb;
            ll[e] = cc;
        }
        Planes P;
        P.pl[0] = __builtin_bit_cast(bf16x8, hh);
        P.pl[1] = __builtin_bit_cast(bf16x8, mm);
        P.pl[2] = __builtin_bit_cast(bf16x8, ll);
        return P;
    };
    auto mfma6 = [&](f32x4 a, const bf16x8 (&wv)[3], const Planes &X) {
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], X.pl[1], a, 0, 0, 0);   // small terms first
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[2], X.pl[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], X.pl[2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], X.pl[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], X.pl[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], X.pl[0], a, 0, 0, 0);
        return a;
    };

    // epilogue operands of the two channel blocks this wave finishes (2 w, 2 w + 1): requested now, used after the reductions
    f32x4 aux1[RB][2], msk1[RB][2], aux2[RB][R2][2], bia1[2], bia2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nb = 2 * w + j;
        // (unconditional loads from a valid address, the condition on the arithmetic: a load behind a branch makes the compiler wait for
        // EVERYTHING in flight at the join -- here the whole weight ring, in front of the first MFMA)
        if (!BWD) bia1[j] = *reinterpret_cast<const f32x4 *>((p.bias1 ? p.bias1 : A_) + 16 * nb + 4 * g);
        if (!BWD && !ONE) bia2[j] = *reinterpret_cast<const f32x4 *>(p.bias2 + 16 * nb + 4 * g);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (BWD) aux1[rb][j] = *reinterpret_cast<const f32x4 *>((p.res1 ? p.res1 : A_) + grow[rb] + 16 * nb);
            if (BWD) msk1[rb][j] = *reinterpret_cast<const f32x4 *>((p.mask1 ? p.mask1 : A_) + grow[rb] + 16 * nb);
            if (!ONE) {
#pragma unroll
                for (int r = 0; r < R2; ++r)
                    aux2[rb][r][j] = *reinterpret_cast<const f32x4 *>((BWD ? p.mask2 : p.res2) + grow2[rb] + 128 * r + 16 * nb);
            }
        }
    }

    CS_T(1);
    // ---- stage 1
    Planes xa[RB][TAPS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < TAPS; ++i) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = ra[rb][i][0][e];
                x[4 + e] = ra[rb][i][1][e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (PRO_ACT) x[e] = act_f(x[e], p.slope);
                x[e] = rok[rb][i] ? x[e] : 0.f;
            }
            xa[rb][i] = split8(x);
        }
    CS_T(2);
    f32x4 acc[RB][8];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stage 2's fragments (step w of W2) ride behind stage 1's in the same ring
    const uint16_t *w2 = ONE ? nullptr : W2img + (long)w * FS_WSTEP + lane * 8;
    auto loadW2 = [&](int nb, int slot) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[slot][pl] = *reinterpret_cast<const bf16x8 *>(w2 + pl * (128 * 32) + nb * 512);
    };
#pragma unroll
    for (int i = 0; i < NP; ++i) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][i & 7] = mfma6(acc[rb][i & 7], wf[i % D], xa[rb][i >> 3]);
        if (i + D < NP) loadW1(i + D, i % D);
        else if (!ONE) loadW2(i + D - NP, i % D);      // NP is a multiple of D: slots line up
        // (pins every refill behind the MFMAs that free its slot: left to itself the scheduler sinks the loads towards their
        // use, eight pairs later, and the ring degenerates to one or two fragments in flight -- s_waitcnt vmcnt(1..6) in the loop)
        __builtin_amdgcn_sched_barrier(0);
    }

    CS_T(3);
    // ---- first reduction (fixed order), stage-1 epilogue on blocks 2 w, 2 w + 1
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) red[((w * RB + rb) * 8 + nb) * 64 + lane] = acc[rb][nb];
    __syncthreads();
    CS_T(4);
    f32x4 h[RB][R2][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = 2 * w + j;
            f32x4 x = ((red[((0 * RB + rb) * 8 + nb) * 64 + lane] + red[((1 * RB + rb) * 8 + nb) * 64 + lane]) +
                       red[((2 * RB + rb) * 8 + nb) * 64 + lane]) + red[((3 * RB + rb) * 8 + nb) * 64 + lane];
            if (!BWD) {
                if (p.bias1) x += bia1[j];
                if (!ONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = act_f(x[e], p.slope);
                }
            } else {
                if (p.res1) x += aux1[rb][j];
                if (p.mask1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] *= act_grad(msk1[rb][j][e], p.slope);
                }
            }
            if constexpr (!UNPOOL) {
                if (valid[rb]) *reinterpret_cast<f32x4 *>(p.out1 + grow[rb] + 16 * nb) = x;
                if (BWD && p.drop.thresh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] *= drop_mul(p.drop, (uint32_t)(grow[rb] + 16 * nb + e));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) h[rb][0][j][e] = valid[rb] ? x[e] : 0.f;
            } else {
                f32x4 u0 = x, u1 = x;
                if (POOL == 3) {
                    const f32x4 y0 = *reinterpret_cast<const f32x4 *>(p.ypre + grow2[rb] + 16 * nb);
                    const f32x4 y1 = *reinterpret_cast<const f32x4 *>(p.ypre + grow2[rb] + 128 + 16 * nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool second = y1[e] > y0[e];
                        u0[e] = second ? 0.f : x[e];
                        u1[e] = second ? x[e] : 0.f;
                    }
                }
                if (valid[rb]) {
                    *reinterpret_cast<f32x4 *>(p.out1 + grow2[rb] + 16 * nb) = u0;
                    *reinterpret_cast<f32x4 *>(p.out1 + grow2[rb] + 128 + 16 * nb) = u1;
                    if (trow_raw[rb] == Trows_ - 1 && 2 * Trows_ < p.Tfine) {   // odd trailing row of the fine level: no gradient
                        *reinterpret_cast<f32x4 *>(p.out1 + grow2[rb] + 256 + 16 * nb) = f32x4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4 *>(p.out2 + grow2[rb] + 256 + 16 * nb) = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (p.drop.thresh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        u0[e] *= drop_mul(p.drop, (uint32_t)(grow2[rb] + 16 * nb + e));
                        u1[e] *= drop_mul(p.drop, (uint32_t)(grow2[rb] + 128 + 16 * nb + e));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[rb][0][j][e] = valid[rb] ? u0[e] : 0.f;
                    h[rb][R2 - 1][j][e] = valid[rb] ? u1[e] : 0.f;
                }
            }
        }
    CS_T(5);
#if CS_STAMP
    if constexpr (ONE) cs_publish(5);
#endif
    if constexpr (ONE) return;

    // ---- stage 2: this wave's 32 channels are step w of the reduction, in accumulator order
    Planes x2[RB][R2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = h[rb][r][0][e];
                x[4 + e] = h[rb][r][1][e];
            }
            x2[rb][r] = split8(x);
        }
    f32x4 acc2[RB][R2][8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < R2; ++r) acc2[rb][r][nb] = mfma6(f32x4{0.f, 0.f, 0.f, 0.f}, wf[nb % D], x2[rb][r]);
    }
    CS_T(6);
    __syncthreads();   // every wave has read its blocks of the first exchange: the buffer is free
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < R2; ++r)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) red[(((w * RB + rb) * R2 + r) * 8 + nb) * 64 + lane] = acc2[rb][r][nb];
    __syncthreads();
    CS_T(7);

    // ---- second reduction, stage-2 epilogue on output blocks 2 w, 2 w + 1
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const long gr = grow2[rb] + 128 * r;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nb = 2 * w + j;
                f32x4 x = ((red[(((0 * RB + rb) * R2 + r) * 8 + nb) * 64 + lane] + red[(((1 * RB + rb) * R2 + r) * 8 + nb) * 64 + lane]) +
                           red[(((2 * RB + rb) * R2 + r) * 8 + nb) * 64 + lane]) + red[(((3 * RB + rb) * R2 + r) * 8 + nb) * 64 + lane];
                if (!BWD) {
                    x += bia2[j];
                    if (p.drop.thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] *= drop_mul(p.drop, (uint32_t)(gr + 16 * nb + e));
                    }
                    x += aux2[rb][r][j];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] *= act_grad(aux2[rb][r][j][e], p.slope);
                }
                if (BWD || POOL == 0) {
                    if (valid[rb]) *reinterpret_cast<f32x4 *>(p.out2 + gr + 16 * nb) = x;
                } else {
                    if (POOL == 1 && valid[rb]) *reinterpret_cast<f32x4 *>(p.out_pre + gr + 16 * nb) = x;
                    f32x4 y;   // rows 2u, 2u + 1 sit on neighbouring lanes
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float o = __shfl_xor(x[e], 1);
                        y[e] = POOL == 1 ? fmaxf(x[e], o) : x[e] + o;
                    }
                    if ((trow_raw[rb] & 1) == 0 && trow_raw[rb] + 1 < Trows_)
                        *reinterpret_cast<f32x4 *>(p.out2 + ((long)b * (Trows_ >> 1) + (trow_raw[rb] >> 1)) * 128 + 4 * g + 16 * nb) = y;
                }
            }
        }
    CS_T(8);
#if CS_STAMP
    cs_publish(8);
#endif
}

// Rows per workgroup: 16, or 32 (two row blocks sharing every weight fragment in registers: half the L2 -> register weight traffic per
// row) where 16-row workgroups would be MORE than one per CU.  r2 measured 32 rows at every level: slower (0.832 vs 0.802 ms per step;
// at <= 256 workgroups a launch costs the latency of its chain, not its weight bytes).  r3: at B = 8 x T = 4096 the T/4 level is 512
// workgroups of 16 rows, two per CU, and IS bound by the 786 KB of fragments its CU streams (15-21 us per launch against 9.6 at T/8):
// 32 rows there and 16 elsewhere is -11.5 us per step (0.7855 -> 0.774 ms, same box, twice).
extern int g_cs_rb;        // 0 = by level size (above), 1 = 16 rows, 2 = 32 rows, 4 = 64 rows forward / 32 backward (MUCON_COARSE_RB; tests force all three)
extern int g_cs_rb4_wgs;   // forward launches take 64 rows where 16-row workgroups would number more than this (MUCON_COARSE_RB4_WGS; 0: never -- the A/B)
template <bool BWD, int POOL, int TAPS, bool ONE = false, bool PRO_ACT = false>
static hipError_t launch_cs(const FusedParams &p, const uint16_t *W1img, const uint16_t *W2img, int B, hipStream_t s) {
    // 64 rows (RB = 4) for the FORWARD launches of a level with more than 512 16-row workgroups (the T/2 level of the bench shape: 1,024): at RB = 2 that level
    // is 512 workgroups, two co-resident per CU, each streaming the layer's 384 KB of fragments through the CU's one path to L2 (~70 GB/s per CU:
    // profiles/r05_cs_kernel_phase_stamps.txt) -- 768 KB per CU, 22 us; one 64-row workgroup per CU streams them once.  (Backward launches stay at 32 rows: the
    // un-pooling variants' second exchange would need 256 KB of LDS at RB = 4.)
    const long wg16 = (long)B * ((p.Trows + 15) / 16);
    int rb = g_cs_rb ? g_cs_rb : (g_cs_rb4_wgs > 0 && wg16 > g_cs_rb4_wgs && !BWD ? 4 : wg16 > kCsRb2Workgroups ? 2 : 1);
    if (BWD && rb > 2) rb = 2;
#if CS_STAMP
    const int slot = g_cs_slot++ & 63;
    g_cs_info[slot] = CsStampInfo{BWD, POOL, TAPS, ONE, rb, (p.Trows + 16 * rb - 1) / (16 * rb), B, p.Trows};
#define CS_SLOT_ARG , slot
#else
#define CS_SLOT_ARG
#endif
    const dim3 grid((p.Trows + 16 * rb - 1) / (16 * rb), B);
    if constexpr (!BWD) {
        if (rb == 4) {
            hipLaunchKernelGGL((cs_kernel<BWD, POOL, TAPS, ONE, PRO_ACT, 4>), grid, dim3(256), 0, s, p.A, W1img, W2img, p.Trows, p.tap_step, p CS_SLOT_ARG);
            return hipGetLastError();
        }
    }
    if (rb == 2) hipLaunchKernelGGL((cs_kernel<BWD, POOL, TAPS, ONE, PRO_ACT, 2>), grid, dim3(256), 0, s, p.A, W1img, W2img, p.Trows, p.tap_step, p CS_SLOT_ARG);
    else hipLaunchKernelGGL((cs_kernel<BWD, POOL, TAPS, ONE, PRO_ACT, 1>), grid, dim3(256), 0, s, p.A, W1img, W2img, p.Trows, p.tap_step, p CS_SLOT_ARG);
    return hipGetLastError();
#undef CS_SLOT_ARG
}

// ---- chained row-local launches ---------------------------------------------------------------------------------------------
// At the coarsest level the dilation of a residual layer reaches past the sequence (d = 512, 1024 at T/16): its dilated conv is the
// centre tap alone, the layer is ROW-LOCAL, and so is last_conv.  A row-local product that follows another needs nothing but the
// rows its own workgroup just produced -- and after cs_kernel's second reduction wave w holds, in accumulator order, exactly the
// 32 channels that are its k-slice of the next product.  ct_kernel therefore walks a LIST of products in one launch: product 0
// reads its operand from memory (natural-order image), every later one takes the previous product's finished values from the
// registers (accumulator-order images: a layer's W2 / W2t, and images 4 / 5 of fs_pack_body for the dilated convs' centre taps
// and last_conv).  Every intermediate the backward pass or the weight gradients need is still written.
//   forward   layers L-2, L-1 and last_conv:  W1c -> ReLU -> W2, dropout, +x -> W1c -> ReLU -> W2, dropout, +x -> ReLU -> W_last
//             (3 launches -> 1; reference temporal.py:43-53, :144-145)
//   backward  last_conv's and layer L-1's data gradients:  W_last^T, x ReLU' -> W2^T, x ReLU' -> W1c^T, + g -> W2^T, x ReLU'  (2 -> 1)
// Arithmetic, reduction order and dropout keys are those of cs_kernel; only the k order inside a 32-deep step of the products that
// used to read memory differs (accumulator order instead of natural).
// What product s of a chain does is fixed by the chain (compile-time): with the descriptors indexed by constants, every pointer is
// an ordinary kernel argument, loaded once up front -- indexed at run time, each product began with a dependent scalar-cache miss
// (descriptor -> pointer -> load), ~0.7 us in front of its loads.
//   CT_ADD_KEEP: + the kept rows;  CT_ACT: non-linearity;  CT_SET_KEEP: keep the result;  CT_ACT_NEXT: hand on act(result)
enum : int { CT_ADD_KEEP = 1, CT_ACT = 2, CT_SET_KEEP = 4, CT_ACT_NEXT = 8, CT_BIAS = 16, CT_MASK = 32, CT_DROP_PRE = 64, CT_DROP_POST = 128 };
constexpr int ct_stages(bool bwd) { return bwd ? 4 : 5; }
constexpr int ct_flags(bool bwd, int s) {
    return bwd ? (s == 0 ? CT_MASK | CT_SET_KEEP | CT_DROP_POST : s == 1 ? CT_MASK : s == 2 ? CT_ADD_KEEP | CT_DROP_POST : CT_MASK)
               : (s == 0 || s == 2 ? CT_BIAS | CT_ACT
                  : s == 1 ? CT_BIAS | CT_DROP_PRE | CT_ADD_KEEP | CT_SET_KEEP
                  : s == 3 ? CT_BIAS | CT_DROP_PRE | CT_ADD_KEEP | CT_ACT_NEXT : CT_BIAS);
}
struct CtStage {
    const uint16_t *img;    // [4 steps][FS_WSTEP]: product 0 natural order, the others accumulator order
    const float *bias;      // CT_BIAS: + bias
    const float *mask;      // CT_MASK: x act'(mask[row][c])
    float *out;             // [B][Trows][128]
    DropCfg drop;           // CT_DROP_PRE: dropout of the sum before the residual (forward conv_1x1);
                            // CT_DROP_POST: dropout of the value handed on (backward: the NEXT product is a conv_1x1^T)
};
constexpr int CT_MAX_STAGES = 5;
struct CtParams {
    const float *A;         // operand of product 0: [B][Trows][128]
    const float *keep0;     // forward: the first layer's input (its residual)
    int Trows;
    float slope;
    CtStage st[CT_MAX_STAGES];
};
template <bool BWD>
// (r6) product 0's operand and image pointers and the row count as LEADING SCALAR arguments: preloaded into SGPRs with the wave (see cs_kernel)
__global__ __launch_bounds__(256) void ct_kernel(const float *__restrict__ A_, const uint16_t *__restrict__ img0_, const int Trows_, const CtParams p) {
    __shared__ f32x4 red[2][4 * 8 * 64];   // the exchange of product s uses buffer s & 1: one barrier per product
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const long vbase = (long)b * Trows_;
    const int trow_raw = blockIdx.x * 16 + c;
    const bool valid = trow_raw < Trows_;
    const int tcl = min(trow_raw, Trows_ - 1);
    const long grow = (vbase + tcl) * 128 + 4 * g;

    // two register sets of weight fragments: product s + 1's 24 KB per wave are requested before product s multiplies, so their
    // trip from L2 (~1 us at this grid size: one wave per SIMD, nothing else to hide it) is covered by a whole product
    bf16x8 wf[2][8][3];
    auto loadW = [&](const uint16_t *img, auto SET) {   // this wave's step (w) of a K = 128 image: 8 channel blocks x 3 planes
        constexpr int Q = decltype(SET)::value;
        const uint16_t *src = img + (long)w * FS_WSTEP + lane * 8;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[Q][nb][pl] = *reinterpret_cast<const bf16x8 *>(src + pl * (128 * 32) + nb * 512);
    };
    struct Planes { bf16x8 pl[3]; };
    auto split8 = [&](const float (&x)[8]) {
        u32x4 hh, mm, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t a, bb, cc;
            sp_split2(x[2 * e], x[2 * e + 1], a, bb, cc);
            hh[e] = a;
            mm[e] = bb;
            ll[e] = cc;
        }
        Planes P;
        P.pl[0] = __builtin_bit_cast(bf16x8, hh);
        P.pl[1] = __builtin_bit_cast(bf16x8, mm);
        P.pl[2] = __builtin_bit_cast(bf16x8, ll);
        return P;
    };
    auto mfma6 = [&](const bf16x8 (&wv)[3], const Planes &X) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], X.pl[1], a, 0, 0, 0);   // small terms first
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[2], X.pl[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], X.pl[2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], X.pl[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], X.pl[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], X.pl[0], a, 0, 0, 0);
        return a;
    };

    // product 0's operand: this wave's 32 channels of its rows, natural order
    loadW(img0_, std::integral_constant<int, 0>{});
    Planes X;
    {
        const float *src = A_ + (vbase + tcl) * 128 + 32 * w + 8 * g;
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(src), r1 = *reinterpret_cast<const f32x4 *>(src + 4);
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[e] = valid ? r0[e] : 0.f;
            x[4 + e] = valid ? r1[e] : 0.f;
        }
        X = split8(x);
    }
    f32x4 keep[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        keep[j] = BWD ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4 *>(p.keep0 + grow + 16 * (2 * w + j));

    constexpr int NS = ct_stages(BWD);
    auto product = [&](auto SI) {
        constexpr int S = decltype(SI)::value, Q = S & 1, F = ct_flags(BWD, S);
        const CtStage &st = p.st[S];
        // the epilogue's operands travel under the MFMAs; they are requested BEFORE the next product's fragments (the memory
        // counter is in order: waiting for them must not mean waiting for the 24 fragment loads behind them)
        f32x4 bia[2], msk[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = 2 * w + j;
            if (F & CT_BIAS) bia[j] = *reinterpret_cast<const f32x4 *>(st.bias + 16 * nb + 4 * g);
            if (F & CT_MASK) msk[j] = *reinterpret_cast<const f32x4 *>(st.mask + grow + 16 * nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S + 1 < NS) loadW(p.st[S + 1 < NS ? S + 1 : S].img, std::integral_constant<int, Q ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[8];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = mfma6(wf[Q][nb], X);
        f32x4 *rd = red[Q];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) rd[(w * 8 + nb) * 64 + lane] = acc[nb];
        __syncthreads();
        float h[8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = 2 * w + j;
            f32x4 x = ((rd[(0 * 8 + nb) * 64 + lane] + rd[(1 * 8 + nb) * 64 + lane]) + rd[(2 * 8 + nb) * 64 + lane]) + rd[(3 * 8 + nb) * 64 + lane];
            if (F & CT_BIAS) x += bia[j];
            if ((F & CT_DROP_PRE) && st.drop.thresh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] *= drop_mul(st.drop, (uint32_t)(grow + 16 * nb + e));
            }
            if (F & CT_ADD_KEEP) x += keep[j];
            if (F & CT_MASK) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] *= act_grad(msk[j][e], p.slope);
            }
            if (F & CT_ACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = act_f(x[e], p.slope);
            }
            if (valid) *reinterpret_cast<f32x4 *>(st.out + grow + 16 * nb) = x;
            if (F & CT_SET_KEEP) keep[j] = x;
            if ((F & CT_DROP_POST) && st.drop.thresh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] *= drop_mul(st.drop, (uint32_t)(grow + 16 * nb + e));
            }
            if (F & CT_ACT_NEXT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = act_f(x[e], p.slope);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) h[4 * j + e] = valid ? x[e] : 0.f;
        }
        if constexpr (S + 1 < NS) X = split8(h);   // channel blocks 2 w, 2 w + 1 in accumulator order = step w of the next product
    };
    product(std::integral_constant<int, 0>{});
    product(std::integral_constant<int, 1>{});
    product(std::integral_constant<int, 2>{});
    product(std::integral_constant<int, 3>{});
    if constexpr (NS > 4) product(std::integral_constant<int, 4>{});
}
template <bool BWD>
static hipError_t launch_ct(const CtParams &p, int B, hipStream_t s) {
    hipLaunchKernelGGL(ct_kernel<BWD>, dim3((p.Trows + 15) / 16, B), dim3(256), 0, s, p.A, p.st[0].img, p.Trows, p);
    return hipGetLastError();
}
