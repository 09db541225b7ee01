// Device helpers shared by the MuCon hot-path kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
static inline DropCfg make_drop(uint64_t seed, int site, float p, bool training) {
    DropCfg d;
    d.s0 = (uint32_t)(seed & 0xffffffffu) ^ (0x85EBCA6Bu * (uint32_t)(site + 1));
    d.s1 = (uint32_t)(seed >> 32) + 0xC2B2AE35u * (uint32_t)(site + 1);
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
