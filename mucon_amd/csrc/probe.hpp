// Box calibration (bench.py: `box_calibration`): what THIS board sustains on a bare all-CU bf16 MFMA loop with random operands, and the
// shader clock it holds while doing so.  The MFMA kernels of the hot path run against the board's power management, and the boards of a
// pool differ by up to 12 % on exactly this kind of loop (MI355X_MICROARCH.md, DVFS give-back item 5): without a figure for the box a
// bench line cannot tell a code change from a different board.  Not part of the hot path; nothing the path computes depends on it.
//
// One wave per SIMD (256 threads per workgroup, one workgroup per CU and launch wave), operands in registers, four independent
// accumulators per wave, `iters` x 16 MFMAs back to back.  Operands are pseudo-random finite bf16 in (-1, 1) (power follows the bits that
// toggle: zero operands would run at the top clock and calibrate nothing).  s_memtime / s_memrealtime are stamped around the loop and
// left, per workgroup, in a buffer of their own.
#pragma once
#include "common.hpp"

struct ProbeOut {
    long long cycles, ticks;   // shader cycles, 100 MHz ticks spent in the loop (lane 0 of wave 0)
};

template <bool M16>
__global__ __launch_bounds__(256) void mfma_probe_kernel(int iters, ProbeOut *out, float *sink) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // two bf16 per word, exponent field forced into [2^-8, 1): sign and mantissa bits random
            const uint32_t ra = mix32(id * 8u + k * 2u + 0x1234567u * (e + 1)), rb = mix32(id * 8u + k * 2u + 1u + 0x7654321u * (e + 1));
            a[k][e] = (ra & 0x807f807fu) | (((0x77u + (ra >> 28 & 7u)) << 7) * 0x00010001u);
            b[k][e] = (rb & 0x807f807fu) | (((0x77u + (rb >> 28 & 7u)) << 7) * 0x00010001u);
        }
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float total = 0.f;
    if constexpr (M16) {
        f32x4 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k)   // 32 x (16x16x32) = the FLOPs of 16 x (32x32x16)
                acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[k & 3]), __builtin_bit_cast(bf16x8, b[(k >> 2) & 3]),
                                                                     acc[k & 7], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) total += acc[k][0] + acc[k][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[k & 3]), __builtin_bit_cast(bf16x8, b[(k >> 2) & 3]),
                                                                     acc[k & 3], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) total += acc[k][0] + acc[k][15];
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x].cycles = c1 - c0;
        out[blockIdx.x].ticks = r1 - r0;
    }
    if (total == 12345.678f) sink[0] = total;   // keeps the accumulators alive
}
constexpr double kProbeFlopsPerIter = 16.0 * 2.0 * 32 * 32 * 16;   // per wave and loop iteration, either shape

// The same loop with ONE operand of every MFMA re-read from LDS by ds_read_b128 (32x32x16 shape; 128 B/clk per CU: half the LDS peak, about what
// the split-bf16 kernels of the path ask of it).  Devices of one pool differ more on this kind of loop than on the register-only one
// (MI355X_MICROARCH.md, DVFS give-back item 5: 12 % across devices), and it is the kind the path's kernels are.
__global__ __launch_bounds__(256) void mfma_lds_probe_kernel(int iters, ProbeOut *out, float *sink) {
    __shared__ u32x4 img[4][16][64];   // 64 KB: per wave 16 fragments of 1 KB
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 b[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t r = mix32(id * 64u + k * 4u + e + 0x2468aceu);
            v[e] = (r & 0x807f807fu) | (((0x77u + (r >> 28 & 7u)) << 7) * 0x00010001u);
        }
        img[wave][k][lane] = v;
        if (k < 4) b[k] = v;
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const u32x4 a = img[wave][k][lane];
            acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[(k >> 2) & 3]), acc[k & 3], 0, 0, 0);
        }
        asm volatile("" ::: "memory");   // (the fragments are re-read every iteration)
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) total += acc[k][0] + acc[k][15];
    if (threadIdx.x == 0) {
        out[blockIdx.x].cycles = c1 - c0;
        out[blockIdx.x].ticks = r1 - r0;
    }
    if (total == 12345.678f) sink[0] = total;
}
