// EXPERIMENT (round 3), not part of the product: measured against gemm_split.hpp's kernel and left out -- DESIGN.md "What did not pay".
// Correct (tests/test_gpu_split.py while it was wired in), but 112-115 us per launch against 91: with the tape ALSO going through LDS
// (staged by dedicated waves) the four multiplier waves read 60 fragments per 32-deep tile each; reads (960 LDS cycles) + the stagers'
// stores (600) fill the LDS pipe for longer than the tile's 1,536 MFMA cycles -- in-kernel stamps: stagers 2,900 cycles per tile,
// multipliers 2,050 + 1,000 waiting, whatever the prefetch depth.  The product kernel keeps the tape in registers (each element
// belongs to one wave) and sends only the weight image through LDS: that asymmetry is what the exact split needs (three planes per
// operand = three times a plain bf16 GEMM's LDS traffic per MFMA).
// first_conv forward (temporal.py:133: relu(W x + b), 2048 -> 128 channels over every frame of the tape) on the bf16 MFMA with
// exactly split fp32 operands -- the arithmetic of gemm_split.hpp (x = hi + mid + lo, six of the nine partial products, fp32
// accumulate) -- WAVE-SPECIALISED.
//
// gemm_split.hpp's kernel gives every wave both jobs: load its tape rows, split them (5.5 vector instructions per element),
// multiply -- the split woven between the MFMAs, two waves per SIMD in lockstep behind a barrier per k-tile: 88-95 us per launch
// at B = 8 x T = 4096 with the MFMA pipe 51 % busy.  Only the TAPE needs vector arithmetic here (the weight image is pre-split
// by pack_weights), so one stager wave per SIMD has time to spare beside one multiplier wave:
//   * waves 4-7, the STAGERS: the 128 frames x 32 k of a tile as sixteen 1 KB loads (a wave instruction = eight whole 128-byte
//     lines), the exact split, 8-byte stores into a fragment-ordered bf16 image [step 2][plane 3][lane half 2][frame 128][8]
//     (the k order inside a 16-deep step is gemm_split.hpp's: half h, slot j <-> k = 4 h + j, 8 + 4 h + (j - 4): a lane's four
//     floats are four adjacent slots); and a linear copy of the tile's half of the pre-split W image.  Three tiles deep;
//   * waves 0-3, the MULTIPLIERS: nothing but ds_read_b128 and v_mfma_f32_32x32x16_bf16: wave w owns frames 32 w .. 32 w + 31 x
//     all 128 channels for the WHOLE reduction (no k-groups, no exchange at the end); fragments are requested two 6-MFMA slots
//     ahead of their use, also across the tile edge.
// The matrix pipe and the vector pipe of a SIMD are separate; the stager's ~120 instructions per tile run in the gaps of its
// partner's 48 MFMAs (1,536 cycles).  One barrier per 32-deep tile keeps the roles in step.
#pragma once
#include <stdio.h>
#include <type_traits>

#include "../../mucon_amd/csrc/common.hpp"
#include "../../mucon_amd/csrc/gemm_nt.hpp"

constexpr int SW_ABLK = 128 * 8 + 16;                  // bf16 elements of one (step, plane, half) block of the A image: 128 frames x 8
                                                       // slots, + 32 B (the eight blocks a wave's stores hit land on different banks)
constexpr int SW_AIMG = 12 * SW_ABLK;                  // 12,480 elements = 24,960 B
constexpr int SW_WIMG = 2 * 3 * 2 * 128 * 8;           // one 32-deep half of gemm_split.hpp's k-tile image: 12,288 elements = 24,576 B
constexpr int SW_STAGE = SW_AIMG + SW_WIMG;            // one tile in LDS
constexpr int SW_NBUF = 3;                             // tiles in LDS: being multiplied, complete, being written
constexpr int SW_SMEM_BYTES = SW_NBUF * SW_STAGE * 2;  // 148,608 B

#ifndef SW_STAMP
#define SW_STAMP 0
#endif

template <bool EPI_ACT>
__global__ __launch_bounds__(512) void nt_ws_kernel(const NtParams p, const uint16_t *__restrict__ Wimg) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sw_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 128;
    const int ntiles = p.Kc >> 5;                      // 32-deep tiles (Kc is a multiple of 128)
    const int last = ntiles - 1;
#if SW_STAMP
    long long st_work = 0, st_wait = 0, st_prev = __builtin_amdgcn_s_memtime();
    const long long st_r0 = __builtin_amdgcn_s_memrealtime();
#define SW_TS(acc) do { const long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - st_prev; st_prev = t_; } while (0)
#define SW_REPORT(role) do { if (blockIdx.x == 3 && blockIdx.y == 1 && lane == 0) printf("nt_ws wave %d (" role "): cycles per tile: work %lld  barrier wait %lld  | %lld tiles, %.2f GHz\n", wave, st_work / ntiles, st_wait / ntiles, (long long)ntiles, (double)(st_work + st_wait) / ((double)(__builtin_amdgcn_s_memrealtime() - st_r0) * 10.0)); } while (0)
#else
#define SW_TS(acc) do { } while (0)
#define SW_REPORT(role) do { } while (0)
#endif

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ stagers
        const int sw = wave - 4, st = tid - 256;
        // load i of a tile: frames 32 sw + 8 i + (lane >> 3), chunk c = lane & 7 = k 4 c .. 4 c + 3 of the tile
        const int c = lane & 7;
        const int s = c >> 2, h = c & 1, half = (c >> 1) & 1;              // k16 = 4 (c & 3): slots 4 * half .. of lane half h
        const float *a_lane[4];
        uint32_t a_dst[4];                                                 // element offset inside the A image (plane 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 32 * sw + 8 * i + (lane >> 3);
            a_lane[i] = p.A + (long)b * p.a_bstride + (long)min(t0 + row, p.Trows - 1) * p.lda + 4 * c;   // padding rows re-read a valid row
            a_dst[i] = (uint32_t)(((s * 3) * 2 + h) * SW_ABLK + row * 8 + 4 * half);
        }
        const uint16_t *w_src = Wimg + st * 8;
        f32x4 ra[3][4];     // three register sets: a tile's loads are issued three intervals before it is split (one interval is
        u32x4 rw[3][6];     // shorter than an HBM round trip under load: with two sets the stagers' period WAS that round trip)
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        auto gload = [&](int tile, auto SET) {
            constexpr int Q = decltype(SET)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[Q][i] = *reinterpret_cast<const f32x4 *>(a_lane[i] + 32 * tile);
#pragma unroll
            for (int q = 0; q < 6; ++q) rw[Q][q] = *reinterpret_cast<const u32x4 *>(w_src + (long)tile * SW_WIMG + q * 2048);
        };
        auto stage = [&](int tile, auto SET) {
            constexpr int Q = decltype(SET)::value;
            uint16_t *img = sw_smem + (tile % SW_NBUF) * SW_STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t a0, m0, l0, a1, m1, l1;
                sp_split2(ra[Q][i][0], ra[Q][i][1], a0, m0, l0);
                sp_split2(ra[Q][i][2], ra[Q][i][3], a1, m1, l1);
                uint16_t *dst = img + a_dst[i];
                *reinterpret_cast<u32x2 *>(dst) = u32x2{a0, a1};
                *reinterpret_cast<u32x2 *>(dst + 2 * SW_ABLK) = u32x2{m0, m1};
                *reinterpret_cast<u32x2 *>(dst + 4 * SW_ABLK) = u32x2{l0, l1};
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) *reinterpret_cast<u32x4 *>(img + SW_AIMG + st * 8 + q * 2048) = rw[Q][q];
        };
        // prologue: tiles 0 and 1 staged; tiles 2, 3, 4 in flight (sets 2, 0, 1)
        gload(0, I0{});
        gload(min(1, last), I1{});
        gload(min(2, last), I2{});
        stage(0, I0{});
        gload(min(3, last), I0{});
        if (ntiles > 1) stage(1, I1{});
        gload(min(4, last), I1{});
        __syncthreads();
        // interval mt: the multipliers are on tile mt (and fetch the head of tile mt + 1); tile mt + 2 is written, tile mt + 5 requested
        for (int mt = 0; mt < ntiles; mt += 3) {
            if (mt + 2 < ntiles) stage(mt + 2, I2{});
            gload(min(mt + 5, last), I2{});
            SW_TS(st_work);
            __syncthreads();
            SW_TS(st_wait);
            if (mt + 1 < ntiles) {
                if (mt + 3 < ntiles) stage(mt + 3, I0{});
                gload(min(mt + 6, last), I0{});
                SW_TS(st_work);
                __syncthreads();
                SW_TS(st_wait);
            }
            if (mt + 2 < ntiles) {
                if (mt + 4 < ntiles) stage(mt + 4, I1{});
                gload(min(mt + 7, last), I1{});
                SW_TS(st_work);
                __syncthreads();
                SW_TS(st_wait);
            }
        }
        SW_REPORT("stager");
        return;
    }

    // ---------------------------------------------------------------------------------------------------- multipliers
    const int r = lane & 31, h = lane >> 5;
    struct Frag { bf16x8 pl[3]; };
    auto rdA = [&](Frag &F, int tile, int s) {      // this wave's 32 frames, step s
        const uint16_t *base = sw_smem + (tile % SW_NBUF) * SW_STAGE + ((s * 3) * 2 + h) * SW_ABLK + (32 * wave + r) * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) F.pl[pl] = *reinterpret_cast<const bf16x8 *>(base + pl * 2 * SW_ABLK);
    };
    auto rdW = [&](Frag &F, int tile, int s, int nb) {   // channel block nb, step s
        const uint16_t *base = sw_smem + (tile % SW_NBUF) * SW_STAGE + SW_AIMG + ((s * 3) * 2 + h) * (128 * 8) + (nb * 32 + r) * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) F.pl[pl] = *reinterpret_cast<const bf16x8 *>(base + pl * 2 * (128 * 8));
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    auto mm = [&](f32x16 &c, const Frag &A, const Frag &W) {   // small terms first; all six land in the same fp32 accumulator
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[1], W.pl[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[2], W.pl[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[0], W.pl[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[1], W.pl[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[0], W.pl[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[0], W.pl[0], c, 0, 0, 0);
    };
    // a tile is eight slots (step s, channel block nb) of six MFMAs; the fragments of slot k + 2 are requested in front of slot k's
    // MFMAs (W: ring of four register sets; A: one set per step, step 1's requested at slot 2, the next tile's step 0 at slot 6)
    Frag W0, W1, W2, W3, A0, A1;
    __syncthreads();                 // tiles 0 and 1 are complete
    rdA(A0, 0, 0);
    rdW(W0, 0, 0, 0);
    rdW(W1, 0, 0, 1);
    for (int mt = 0; mt < ntiles; ++mt) {
        const int nx = min(mt + 1, last);   // (the tail re-reads the last tile)
        __builtin_amdgcn_sched_barrier(0);
        rdW(W2, mt, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[0], A0, W0);
        __builtin_amdgcn_sched_barrier(0);
        rdW(W3, mt, 0, 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[1], A0, W1);
        __builtin_amdgcn_sched_barrier(0);
        rdA(A1, mt, 1);
        rdW(W0, mt, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[2], A0, W2);
        __builtin_amdgcn_sched_barrier(0);
        rdW(W1, mt, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[3], A0, W3);
        __builtin_amdgcn_sched_barrier(0);
        rdW(W2, mt, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[0], A1, W0);
        __builtin_amdgcn_sched_barrier(0);
        rdW(W3, mt, 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[1], A1, W1);
        __builtin_amdgcn_sched_barrier(0);
        rdA(A0, nx, 0);               // tile mt + 1 was complete at the last barrier
        rdW(W0, nx, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[2], A1, W2);
        __builtin_amdgcn_sched_barrier(0);
        rdW(W1, nx, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(acc[3], A1, W3);
        __builtin_amdgcn_sched_barrier(0);
        SW_TS(st_work);
        __builtin_amdgcn_s_barrier();   // no fence: the reads in flight are of tile mt + 1, which nobody writes before the next barrier
        SW_TS(st_wait);
    }
    SW_REPORT("multiplier");
    // epilogue: bias, non-linearity, store (C layout: column = lane & 31, row of register e = (e & 3) + 8 (e >> 2) + 4 (lane >> 5))
    const long vbase = (long)b * p.Trows;
    const bool full = t0 + 128 <= p.Trows;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int col = nb * 32 + r;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int t = t0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
            float x = acc[nb][e] + bias;
            if (EPI_ACT) x = act_f(x, p.slope);
            if (full || t < p.Trows) p.out[(vbase + t) * 128 + col] = x;
        }
    }
}

template <bool EPI_ACT>
static hipError_t launch_nt_ws(const NtParams &p, const uint16_t *Wimg, int B, hipStream_t s) {
    auto k = nt_ws_kernel<EPI_ACT>;
    static int attr_dev = -1;             // the opt-in is per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (attr_dev != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, SW_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_dev = dev;
    }
    dim3 grid((p.Trows + 127) / 128, B);
    hipLaunchKernelGGL(k, grid, dim3(512), SW_SMEM_BYTES, s, p, Wimg);
    return hipGetLastError();
}
