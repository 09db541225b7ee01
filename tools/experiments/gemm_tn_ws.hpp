// EXPERIMENT (round 3), not part of the product: measured against gemm_tn_split.hpp and left out -- see DESIGN.md "What did not pay".
// Correct (tools/ts_ablate.hip TS_CHECK, and the GPU test-suite while it was wired in); 1.4x faster than the product kernel on a
// first_conv-shaped job alone (256 workgroups), 7-10 % SLOWER inside the real batched launch, where the product kernel's 256-column
// workgroups stage every gradient operand once per 256 columns and its long first_conv workgroups share the chip with the short jobs.
// Weight gradients on the bf16 MFMA with exactly split fp32 operands (the arithmetic of gemm_split.hpp / gemm_tn_split.hpp:
// x = hi + mid + lo, six of the nine partial products, fp32 accumulate), WAVE-SPECIALISED:
//
//     dW[n][c] = sum_t G[t][n] * X[t][c]          (autograd of nn.Conv1d w.r.t. its weight: temporal.py:23-32, :133, :145)
//
// gemm_tn_split.hpp gives every wave both jobs -- load, split (5.5 vector instructions per element), stage, multiply -- and weaves
// the split between the MFMAs: two waves per SIMD run the same program in lockstep behind one barrier per tile, and the MFMA pipe
// ends up 41 % busy with the vector pipe 38 % busy beside it (in-kernel stamps: 5,200 cycles per 32-step tile for 3,072 cycles of
// MFMA).  Here the two jobs belong to different waves of a 512-thread workgroup (waves w and w + 4 share a SIMD):
//   * waves 4-7, the STAGERS: 16-byte loads of both operands as they lie in memory (a lane = four adjacent channels / columns
//     of one time step, a wave instruction = two whole 512-byte rows), masks / dropout replay / non-linearity, the exact split,
//     three ds_write_b64 per load into TIME-MAJOR bf16 plane images [plane 3][block of 32 columns 4][time step 32][32] per
//     operand and 32-step tile -- three tiles deep.  Nothing is transposed on the way in;
//   * waves 0-3, the MULTIPLIERS: nothing but LDS reads and v_mfma_f32_32x32x16_bf16.  Both operands want eight consecutive TIME
//     steps of one column per lane: gfx950's transposing read ds_read_b64_tr_b16 delivers exactly that from the time-major
//     images (four time steps x 16 columns per 16-lane group, conflict-free on 64-byte image rows).  Wave cg owns 32 columns x
//     all 128 channels; fragments are requested two 6-MFMA slots ahead of their use (ring of four register sets), also
//     across the tile edge (the stagers run two tiles ahead).
// The matrix pipe and the vector pipe of a SIMD are separate; an MFMA holds the issue port for 8 of its 32 cycles, so a stager's
// instruction stream runs in the gaps of its partner's MFMA stream instead of inside the same program order.
// One workgroup = one 128-column k-chunk of a job over one time chunk (half of gemm_tn_split.hpp's: a workgroup has one G
// operand, so the conv_1x1 chunk of a residual layer needs no second image).  One barrier per tile keeps the roles in step.
// Bias gradients are the exact fp32 column sums of the staged G values (they never see bf16).
#pragma once
#include <type_traits>

#include "../../mucon_amd/csrc/common.hpp"
#include "../../mucon_amd/csrc/gemm_tn.hpp"

#ifndef TW_STAMP
#define TW_STAMP 0   // tools/ts_ablate.hip: s_memtime sums per role of block 0 (timing builds only)
#endif
#if TW_STAMP
__device__ long long g_tw_stamps[12 * 8];
__device__ long long g_tw_blk[4096 * 4];   // per block: s_memrealtime at start / end (100 MHz), s_memtime cycles, XCC id
#endif
#ifndef TW_ABL
#define TW_ABL 0   // tools/ts_ablate.hip (timing only, results are garbage): 1 no global loads, 2 no split, 4 no LDS stores, 8 no MFMAs
#endif

constexpr int TW_CB = 32 * 32 + 32;                        // bf16 elements of a 32-column block of one plane: 32 time steps x 32 columns, + 64 B
                                                           // (the blocks of a row land on different LDS banks for the 8-byte stores)
constexpr int TW_PLANE = 4 * TW_CB;                        // 128 columns
constexpr int TW_IMG = 3 * TW_PLANE;                       // one operand of a 32-step tile: 12,672 elements = 25,344 B
constexpr int TW_NBUF = 3;                                 // tiles in LDS: being multiplied, complete, being written
constexpr int TW_SMEM_BYTES = TW_NBUF * 2 * TW_IMG * 2;    // 152,064 B

typedef short tw_s16x4 __attribute__((ext_vector_type(4)));
typedef short tw_s16x8 __attribute__((ext_vector_type(8)));

// DROP: operand set 1 with its dropout mask replayed; ACT: the non-linearity on X (last_conv's job) -- compile-time, so that the
// stagers' element loops are straight-line code (a run-time flag puts a branch and a wait in front of every element)
// NS: stager waves (4: 512-thread workgroups, 8: 768-thread workgroups -- two stagers beside every multiplier)
template <bool DROP, bool ACT, int NS>
__device__ __forceinline__ void tw_body(const TnParams &p, const int kc, const int mc, const bool dual, uint16_t *smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = mc / p.chunks_per_video;
    const int tbeg = (mc - b * p.chunks_per_video) * p.MC;
    const int tend = min(tbeg + p.MC, p.Trows);
    const int ntiles = (tend - tbeg + 31) >> 5;
    const int last = ntiles - 1;
    const bool second = dual && kc >= p.nk0;               // the conv_1x1 chunk of a residual layer: operand set 1
    const bool bias_wg = p.bias_slabs != nullptr && (kc == 0 || second);   // workgroup-uniform
    float *red = reinterpret_cast<float *>(smem);          // bias partial sums (the images are dead by then)
#if TW_STAMP
    long long st_work = 0, st_wait = 0, st_prev = __builtin_amdgcn_s_memtime();
#define TW_TS(acc) do { const long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - st_prev; st_prev = t_; } while (0)
#else
#define TW_TS(acc) do { } while (0)
#endif

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ stagers
        const int sw = wave - 4;
        const int xoff = (!second && p.taps == 3) ? (kc - 1) * p.tap_step : 0;
        const int xcol = (second || p.taps == 3) ? 0 : kc * 128;
        const int ldx = second ? 128 : p.ldx;
        const int Tx = second ? p.Trows : p.Tx;
        const char *Xu = reinterpret_cast<const char *>(second ? p.X1 + (long)b * p.Trows * 128 : p.X0 + (long)b * p.x_bstride + xcol);
        const char *Yu = reinterpret_cast<const char *>((second ? p.Y1 : p.Y0) + (long)b * p.Trows * 128);   // wave-uniform bases
        const DropCfg dcfg = p.drop;
        // load i of a tile: time steps RPS sw + 2 i (lanes 0-31) and + 1 (lanes 32-63), channels / columns 4 (lane & 31) .. + 3
        constexpr int RPS = 32 / NS, NL = RPS / 2;          // rows / loads per stager, operand and tile
        const int lrow = RPS * sw + (lane >> 5), c4 = (lane & 31) * 4;
        const uint32_t img_lane = (uint32_t)((c4 >> 5) * TW_CB + lrow * 32 + (c4 & 31));   // element offset inside a plane
        f32x4 rg[2][NL], rx[2][NL];                        // [register set][load]
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // A tile is INTERIOR when none of its rows needs a mask (inside the chunk, and for this tap inside the video): its loads
        // take a wave-uniform row base + constant lane offsets and its values go to the split as they are.  Edge tiles (zero
        // padding of a tap, the partial last tile of a video) clamp their rows and mask the registers.
        auto interior = [&](int tile) {
            const int t0 = tbeg + tile * 32;
            return t0 + 32 <= tend && t0 + xoff >= 0 && t0 + 31 + xoff < Tx;
        };
        const uint32_t og_lane = (uint32_t)(lrow * 128 + c4) * 4u;
        uint32_t ox_lane[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) ox_lane[i] = (uint32_t)((lrow + 2 * i) * ldx + c4) * 4u;
        auto gload = [&](int tile, auto SET) {
            constexpr int Q = decltype(SET)::value;
            const int t0 = tbeg + tile * 32;
            if (TW_ABL & 1) {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const float f = __int_as_float(tile + lane + i);
                    rg[Q][i] = rx[Q][i] = f32x4{f, f, f, f};
                }
            } else if (interior(tile)) {
                const char *yb = Yu + (long)t0 * 512, *xb = Xu + (long)(t0 + xoff) * ldx * 4;   // wave-uniform
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    rg[Q][i] = *reinterpret_cast<const f32x4 *>(yb + og_lane + 1024 * i);
                    rx[Q][i] = *reinterpret_cast<const f32x4 *>(xb + ox_lane[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int t = t0 + lrow + 2 * i;
                    const uint32_t og = (uint32_t)(min(t, p.Trows - 1) * 128 + c4) * 4u;
                    const uint32_t ox = (uint32_t)(min(max(t + xoff, 0), Tx - 1) * ldx + c4) * 4u;   // < 2^32: one video's rows
                    rg[Q][i] = *reinterpret_cast<const f32x4 *>(Yu + og);
                    rx[Q][i] = *reinterpret_cast<const f32x4 *>(Xu + ox);
                }
            }
        };
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
        auto put = [&](const f32x4 v, uint16_t *dst) {     // exact split of four values, one 8-byte store per plane
            uint32_t a0, m0, l0, a1, m1, l1;
            if (TW_ABL & 2) {
                a0 = __float_as_uint(v[0]); m0 = __float_as_uint(v[1]); l0 = a0 ^ m0;
                a1 = __float_as_uint(v[2]); m1 = __float_as_uint(v[3]); l1 = a1 ^ m1;
            } else {
                sp_split2(v[0], v[1], a0, m0, l0);
                sp_split2(v[2], v[3], a1, m1, l1);
            }
            if (TW_ABL & 4) {
                asm volatile("" ::"v"(a0), "v"(m0), "v"(l0), "v"(a1), "v"(m1), "v"(l1));
                return;
            }
            *reinterpret_cast<u32x2 *>(dst) = u32x2{a0, a1};
            *reinterpret_cast<u32x2 *>(dst + TW_PLANE) = u32x2{m0, m1};
            *reinterpret_cast<u32x2 *>(dst + 2 * TW_PLANE) = u32x2{l0, l1};
        };
        auto stage = [&](int tile, auto SET) {             // masks, dropout replay, non-linearity, bias sums, split, store
            constexpr int Q = decltype(SET)::value;
            uint16_t *img = smem + (tile % TW_NBUF) * 2 * TW_IMG + img_lane;
            const bool inner = interior(tile);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int t = tbeg + tile * 32 + lrow + 2 * i, ts = t + xoff;
                f32x4 g = rg[Q][i], x = rx[Q][i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (DROP) g[e] *= drop_mul(dcfg, (uint32_t)(b * p.Trows + t) * 128u + (uint32_t)(c4 + e));
                    if (ACT) x[e] = act_f(x[e], p.slope);
                }
                if (!inner) {                              // (wave-uniform)
                    const bool okg = t < tend, okx = okg && ts >= 0 && ts < Tx;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        g[e] = okg ? g[e] : 0.f;
                        x[e] = okx ? x[e] : 0.f;
                    }
                }
                if (bias_wg) bsum += g;
                put(g, img + 2 * i * 32);
                put(x, img + TW_IMG + 2 * i * 32);
            }
        };
        // prologue: tiles 0 and 1 staged, tile 2 in flight
        gload(0, I0{});
        gload(min(1, last), I1{});
        stage(0, I0{});
        gload(min(2, last), I0{});
        if (ntiles > 1) stage(1, I1{});
        __syncthreads();
        // interval mt: the multipliers are on tile mt (and fetch the head of tile mt + 1); tile mt + 2 is written, tile mt + 3 requested
        for (int mt = 0; mt < ntiles; mt += 2) {
            gload(min(mt + 3, last), I1{});
            if (mt + 2 < ntiles) stage(mt + 2, I0{});
            TW_TS(st_work);
            __syncthreads();
            TW_TS(st_wait);
            if (mt + 1 < ntiles) {
                gload(min(mt + 4, last), I0{});
                if (mt + 3 < ntiles) stage(mt + 3, I1{});
                TW_TS(st_work);
                __syncthreads();
                TW_TS(st_wait);
            }
        }
#if TW_STAMP
        if (blockIdx.x == 0 && lane == 0) {
            g_tw_stamps[wave * 8] = st_work;
            g_tw_stamps[wave * 8 + 1] = st_wait;
        }
#endif
        if (bias_wg) {   // column sums of the staged gradient rows: a lane's own time steps, then its row partner, then the four stagers
            *reinterpret_cast<f32x4 *>(red + (sw * 64 + lane) * 4) = bsum;
            __syncthreads();
            if (sw == 0) {
                float *out = p.bias_slabs + (long)mc * 256 + (second ? 128 : 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int n = q * 64 + lane;        // channel n: lanes (n >> 2) and (n >> 2) + 32 of every stager, element n & 3
                    float acc = 0.f;
#pragma unroll
                    for (int w = 0; w < NS; ++w) acc += red[(w * 64 + (n >> 2)) * 4 + (n & 3)] + red[(w * 64 + 32 + (n >> 2)) * 4 + (n & 3)];
                    out[n] = acc;
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- multipliers
#if TW_STAMP
    const long long blk_r0 = __builtin_amdgcn_s_memrealtime(), blk_c0 = __builtin_amdgcn_s_memtime();
#endif
    const int cg = wave;                                    // this wave's 32 columns of the chunk
    const int r = lane & 31, h = lane >> 5;
    // transposing read: lane 4 q + p of a 16-lane group addresses time step q, columns 4 p .. 4 p + 3 of its block of 4 x 16 and
    // receives column (lane & 15) of the four time steps.  Groups 0 / 1 = columns 0-15 / 16-31 of lane half h (time steps 8 h ..).
    const int li = lane & 15;
    const uint32_t tr_lane = (uint32_t)((8 * h + (li >> 2)) * 32 + 16 * ((lane >> 4) & 1) + 4 * (li & 3));   // elements; + 4 * 32 for steps 4-7
    struct Frag { bf16x8 pl[3]; };
    auto rd = [&](Frag &F, const uint16_t *blk) {           // blk: image + block * TW_CB + 16 s * 32 (+ tr_lane)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const uint16_t *a = blk + pl * TW_PLANE;
            const tw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tw_s16x4 *)(a));
            const tw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tw_s16x4 *)(a + 4 * 32));
            const tw_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            F.pl[pl] = __builtin_bit_cast(bf16x8, v);
        }
    };
    auto rdX = [&](Frag &F, int tile, int s) { rd(F, smem + (tile % TW_NBUF) * 2 * TW_IMG + TW_IMG + cg * TW_CB + 16 * s * 32 + tr_lane); };
    auto rdG = [&](Frag &F, int tile, int s, int nb) { rd(F, smem + (tile % TW_NBUF) * 2 * TW_IMG + nb * TW_CB + 16 * s * 32 + tr_lane); };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    auto mm = [&](f32x16 &c, const Frag &G, const Frag &X) {   // small terms first; all six land in the same fp32 accumulator
        if (TW_ABL & 8) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, G.pl[pl])), "v"(__builtin_bit_cast(u32x4, X.pl[pl])));
            return;
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G.pl[1], X.pl[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G.pl[2], X.pl[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G.pl[0], X.pl[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G.pl[1], X.pl[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G.pl[0], X.pl[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G.pl[0], X.pl[0], c, 0, 0, 0);
    };
    // a tile is eight slots (step s, channel block nb) of six MFMAs; the fragments of slot k + 2 are requested in front of slot k's
    // MFMAs (G: ring of four register sets; X: one set per step, step 1's requested at slot 2, the next tile's step 0 at slot 6)
    Frag G0, G1, G2, G3, X0, X1;
    __syncthreads();                 // tiles 0 and 1 are complete
    rdX(X0, 0, 0);
    rdG(G0, 0, 0, 0);
    rdG(G1, 0, 0, 1);
    for (int mt = 0; mt < ntiles; ++mt) {
        const int nx = min(mt + 1, last);   // (the tail re-reads the last tile)
        __builtin_amdgcn_sched_barrier(0);
        rdG(G2, mt, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[0], G0, X0);
        __builtin_amdgcn_sched_barrier(0);
        rdG(G3, mt, 0, 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[1], G1, X0);
        __builtin_amdgcn_sched_barrier(0);
        rdX(X1, mt, 1);
        rdG(G0, mt, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[2], G2, X0);
        __builtin_amdgcn_sched_barrier(0);
        rdG(G1, mt, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[3], G3, X0);
        __builtin_amdgcn_sched_barrier(0);
        rdG(G2, mt, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[0], G0, X1);
        __builtin_amdgcn_sched_barrier(0);
        rdG(G3, mt, 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[1], G1, X1);
        __builtin_amdgcn_sched_barrier(0);
        rdX(X0, nx, 0);               // tile mt + 1 was complete at the last barrier
        rdG(G0, nx, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[2], G2, X1);
        __builtin_amdgcn_sched_barrier(0);
        rdG(G1, nx, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[3], G3, X1);
        __builtin_amdgcn_sched_barrier(0);
        TW_TS(st_work);
        __builtin_amdgcn_s_barrier();   // no fence: the reads in flight are of tile mt + 1, which nobody writes before the next barrier
        TW_TS(st_wait);
    }
#if TW_STAMP
    if (blockIdx.x == 0 && lane == 0) {
        g_tw_stamps[wave * 8] = st_work;
        g_tw_stamps[wave * 8 + 1] = st_wait;
    }
    if (wave == 0 && lane == 0 && blockIdx.x < 4096) {
        g_tw_blk[blockIdx.x * 4] = blk_r0;
        g_tw_blk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        g_tw_blk[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - blk_c0;
        g_tw_blk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(6164);   // HW_REG_XCC_ID (20), offset 0, size 4: ((4-1) << 11) | 20
    }
#endif
    {
        float *slab = p.slabs + (long)mc * 128 * p.Ktot + kc * 128 + cg * 32 + r;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = nb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                slab[(long)row * p.Ktot] = acc[nb][reg];
            }
    }
    if (bias_wg) __syncthreads();    // the stagers' bias exchange
}

// All weight gradients of a backward pass in one launch (the job table of gemm_tn.hpp): a job with n 128-column chunks has n
// workgroups per time chunk.
template <int NS>
__global__ __launch_bounds__(256 + 64 * NS) void tw_batched_kernel(const TnBatch tb) {
    extern __shared__ __attribute__((aligned(16))) uint16_t tw_smem[];
    int ji = 0;
    while (ji + 1 < tb.njobs && (int)blockIdx.x >= tb.j[ji + 1].block0) ++ji;
    const TnJob &job = tb.j[ji];
    const int nkc = job.nkc;
    const int local = blockIdx.x - job.block0;
    if (local >= nkc * job.nmc) return;   // padding block between two jobs
    int mc = local / nkc, kc = local - mc * nkc;
    if (tb.xcd_order) {
        // The nkc workgroups of a time chunk read the same gradient rows.  Workgroups are dealt round-robin over the 8 XCDs
        // (block b and b + 8 share one -- observed, used for speed only), so inside every run of 8 * nkc blocks the chunk is
        // the block index mod 8: the workgroups that share rows share an L2.
        const int nmc = job.nmc, grp = 8 * nkc;
        const int G = local / grp;
        if ((G + 1) * 8 <= nmc) {
            const int in = local - G * grp;
            mc = G * 8 + (in & 7);
            kc = in >> 3;
        }
    }
    const bool second = job.dual && kc >= job.p.nk0;      // workgroup-uniform
    if (second && job.p.drop.thresh) tw_body<true, false, NS>(job.p, kc, mc, true, tw_smem);
    else if (job.x0_act && !second) tw_body<false, true, NS>(job.p, kc, mc, job.dual != 0, tw_smem);
    else tw_body<false, false, NS>(job.p, kc, mc, job.dual != 0, tw_smem);
}

#ifndef TW_NS
#define TW_NS 8
#endif
static hipError_t launch_tw_batch(TnBatch &tb, hipStream_t s) {
    if (tb.njobs == 0) return hipSuccess;
    static int attr_dev = -1;             // the opt-in is per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (attr_dev != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(tw_batched_kernel<TW_NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TW_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_dev = dev;
    }
    // jobs were queued coarse levels first; the fine levels have the longest workgroups: lay them out first
    TnBatch lb;
    lb.njobs = tb.njobs;
    int blocks = 0;
    for (int i = 0; i < tb.njobs; ++i) {
        const TnJob &src = tb.j[tb.njobs - 1 - i];
        lb.j[i] = src;
        lb.j[i].block0 = blocks;
        lb.j[i].nmc = src.block0;                     // block0 carried the time-chunk count while queued
        blocks += src.nkc * src.block0;
        if (kTsXcdOrder) blocks = (blocks + 7) & ~7;     // every job starts on a multiple of 8 (the padding blocks exit at once)
    }
    lb.nblocks = blocks;
    lb.xcd_order = kTsXcdOrder;
    hipLaunchKernelGGL(tw_batched_kernel<TW_NS>, dim3(blocks), dim3(256 + 64 * TW_NS), TW_SMEM_BYTES, s, lb);
    tb.njobs = 0;
    return hipGetLastError();
}
