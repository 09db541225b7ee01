// Device helpers shared by the MuCon hot-path kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- exact three-way bf16 split of fp32 (gemm_split.hpp explains the arithmetic) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// two fp32 -> two bf16 (round to nearest even), packed low = a, high = b: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t sp_pack(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
// exact split of two fp32 values into packed (hi, mid, lo) bf16 pairs
__device__ __forceinline__ void sp_split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = sp_pack(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = sp_pack(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = sp_pack(s0, s1);
}

// W [128][D] fp32 -> the fragment-ordered bf16 image gemm_split.hpp stages (pack_weights writes it once per forward pass): per 64-deep k-tile
// [k-step 4][plane 3][lane half 2][channel 128][slot 8] bf16, with k16 = k mod 16 at half (k16 >> 2) & 1, slot (k16 & 3) + 4 (k16 >> 3)
template <class SRC>   // src(n, k): element k of output channel n's weight row
__device__ __forceinline__ void sp_split_weights_fn(SRC src, uint16_t *img, int D, long first, long stride) {
    uint32_t *P = reinterpret_cast<uint32_t *>(img);
    const long n_pairs = 64L * D;
    for (long e = first; e < n_pairs; e += stride) {
        const int n = (int)(e / (D / 2)), k = (int)(e - (long)n * (D / 2)) * 2;
        uint32_t h, m, l;
        sp_split2(src(n, k), src(n, k + 1), h, m, l);
        const int S = k >> 6, s = (k >> 4) & 3, k16 = k & 15;
        const int half = (k16 >> 2) & 1, slot = (k16 & 3) + 4 * (k16 >> 3);
        // element offset (((s*3 + pl)*2 + half)*128 + n)*8 + slot, in pairs; plane stride 2*128*8/2 = 1024 pairs
        const long base = (long)S * (24576 / 2) + ((long)((s * 3) * 2 + half) * 128 + n) * 4 + (slot >> 1);
        P[base] = h;
        P[base + 1024] = m;
        P[base + 2048] = l;
    }
}
__device__ __forceinline__ void sp_split_weights(const float *W, uint16_t *img, int D, long first, long stride) {
    sp_split_weights_fn([=](int n, int k) { return W[(long)n * D + k]; }, img, D, first, stride);
}

// The image nt_split16_kernel stages (v_mfma_f32_16x16x32_bf16): per 64-deep k-tile [k-step 2][plane 3][h 4][channel 128][slot 8] bf16,
// with k32 = k mod 32 at h = (k32 >> 2) & 3, slot (k32 & 3) + 4 (k32 >> 4)
template <class SRC>
__device__ __forceinline__ void sp_split_weights16_fn(SRC src, uint16_t *img, int D, long first, long stride) {
    uint32_t *P = reinterpret_cast<uint32_t *>(img);
    const long n_pairs = 64L * D;
    for (long e = first; e < n_pairs; e += stride) {
        const int n = (int)(e / (D / 2)), k = (int)(e - (long)n * (D / 2)) * 2;
        uint32_t h, m, l;
        sp_split2(src(n, k), src(n, k + 1), h, m, l);
        const int S = k >> 6, ks = (k >> 5) & 1, k32 = k & 31;
        const int hh = (k32 >> 2) & 3, slot = (k32 & 3) + 4 * (k32 >> 4);
        const long base = (long)S * (24576 / 2) + ((long)((ks * 3) * 4 + hh) * 128 + n) * 4 + (slot >> 1);   // plane stride 4*128*8/2 pairs
        P[base] = h;
        P[base + 2048] = m;
        P[base + 4096] = l;
    }
}
__device__ __forceinline__ void sp_split_weights16(const float *W, uint16_t *img, int D, long first, long stride) {
    sp_split_weights16_fn([=](int n, int k) { return W[(long)n * D + k]; }, img, D, first, stride);
}

// Diagnostic build only (MUCON_HIPCC_FLAGS=-DCLK_STAMP=1): the MFMA kernels of gemm_split.hpp (slot 0) and gemm_tn_split.hpp (slot 1)
// stamp s_memtime (shader cycles) and s_memrealtime (100 MHz) around their main loop; lane 0 of every workgroup's wave 0 leaves the two
// differences in a buffer of their own, which mucon_test_read_clock copies out: the in-kernel clock = delta cycles / delta real x 100 MHz.
// No output value depends on a stamp; in the normal build no stamp executes.
#ifndef CLK_STAMP
#define CLK_STAMP 0
#endif
#if CLK_STAMP
__device__ long long g_clk[2][4096][2];
__device__ long long g_clk_wg[4096][2];   // ts_batched_kernel: s_memrealtime at a workgroup's entry and exit (100 MHz ticks, absolute)
__device__ long long g_clk_ph[4096][2];   // ... behind its job lookup, and at the first tile (absolute)
__device__ long long g_clk_pe[4096][2];   // ... behind its last tile (absolute; second word unused)
#define CLK_BEGIN() const long long clk_c0_ = __builtin_amdgcn_s_memtime(), clk_r0_ = __builtin_amdgcn_s_memrealtime()
#define CLK_END(slot, wg)                                                                          \
    do {                                                                                           \
        const long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0 && (wg) < 4096) {                                                     \
            g_clk[slot][wg][0] = c1_ - clk_c0_;                                                    \
            g_clk[slot][wg][1] = r1_ - clk_r0_;                                                    \
            if ((slot) == 1) {                                                                     \
                g_clk_ph[wg][1] = clk_r0_;                                                         \
                g_clk_pe[wg][0] = r1_;                                                             \
                g_clk_pe[wg][1] = 1;                                                               \
            }                                                                                      \
        }                                                                                          \
    } while (0)
#else
#define CLK_BEGIN() do { } while (0)
#define CLK_END(slot, wg) do { } while (0)
#endif

#define MUCON_H 128  // hidden width the MFMA kernels are specialised for (cfg.model.ft.hidden_size)

// activation of the reference's apply_non_lin (temporal.py:40-41): relu or leaky_relu(0.01);
// slope == 0 gives relu.  d/dx uses the torch convention (x > 0 ? 1 : slope).
__device__ __forceinline__ float act_f(float x, float slope) { return x > 0.f ? x : x * slope; }
__device__ __forceinline__ float act_grad(float y, float slope) { return y > 0.f ? 1.f : slope; }

// Counter-based dropout: element `idx` of dropout site `site` is kept iff hash >= thresh.
// The same function is replayed in the backward kernels (and in tests/, in numpy).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
struct DropCfg {
    uint32_t s0, s1;   // per-site keys derived from the 64-bit seed
    uint32_t thresh;   // p * 2^32 (0 = dropout off)
    float scale;       // 1 / (1 - p)
};
__device__ __forceinline__ float drop_mul(const DropCfg &d, uint32_t idx) {
    return mix32((idx ^ d.s0) * 0x9E3779B1U + d.s1) >= d.thresh ? d.scale : 0.f;
}
static inline uint32_t mix32_host(uint32_t x) {   // mix32 on the host
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// Both element keys come from the mixer applied to the WHOLE 64-bit seed and the site: two steps (seeds that differ in a
// few low bits) or two data-parallel ranks get unrelated (s0, s1) pairs, not masks that are index permutations of each other.
static inline DropCfg make_drop(uint64_t seed, int site, float p, bool training) {
    DropCfg d;
    const uint32_t lo = (uint32_t)(seed & 0xffffffffu), hi = (uint32_t)(seed >> 32);
    const uint32_t k = mix32_host(hi ^ mix32_host(lo + 0x9E3779B9u * (uint32_t)(site + 1)));
    d.s0 = k;
    d.s1 = mix32_host(k ^ 0x85EBCA6Bu) + hi;
    if (!training || p <= 0.f) {
        d.thresh = 0;
        d.scale = 1.f;
    } else {
        double t = (double)p * 4294967296.0;
        d.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
        d.scale = 1.f / (1.f - p);
    }
    return d;
}
