// Stand-alone timing harness for the split-bf16 weight-gradient kernel (mucon_amd/csrc/gemm_tn_split.hpp), run ON THE GPU BOX:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTS_ABL=n] tools/ts_ablate.hip -o /tmp/ts_ablate && /tmp/ts_ablate [M] [K] [MC] [iters]
// One first_conv-shaped job: dW[128][K] = G[M][128]^T X[M][K].  TS_ABL removes parts of the loop (timing only, results are
// garbage): 1 MFMAs, 2 the X split, 4 the G split and its LDS stores, 8 the global loads.  Prints microseconds per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

int g_ts_stagger = 0;   // TS_ST=1 in the environment: the staggered schedule
int g_mfma16 = 0;   // TS_M16=1 in the environment: the 16x16x32 instantiation (bit 1, as MUCON_MFMA16)
#include "../mucon_amd/csrc/gemm_tn_split.hpp"
#include "experiments/gemm_tn_ws.hpp"
#ifndef USE_TW
#define USE_TW 0   // 1: the wave-specialised kernel (gemm_tn_ws.hpp)
#endif
#if USE_TW
#define LAUNCH launch_tw_batch
#else
#define LAUNCH launch_ts_batch
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(float *p, long n, uint32_t seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        p[i] = (float)(int)(mix32((uint32_t)i * 2654435761u + seed) >> 8) * (1.f / 8388608.f) - 1.f;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, K = argc > 2 ? atoi(argv[2]) : 2048;
    const int MC = argc > 3 ? atoi(argv[3]) : 2048, iters = argc > 4 ? atoi(argv[4]) : 20;
    if (getenv("TS_M16") && atoi(getenv("TS_M16"))) g_mfma16 = 2;
    if (getenv("TS_ST") && atoi(getenv("TS_ST"))) g_ts_stagger = 32;
    float *Y, *X, *slabs, *bslabs;
    const int nmc = (M + MC - 1) / MC;
    CK(hipMalloc(&Y, (size_t)M * 128 * 4));
    CK(hipMalloc(&X, (size_t)M * K * 4));
    CK(hipMalloc(&slabs, (size_t)nmc * 128 * K * 4));
    CK(hipMalloc(&bslabs, (size_t)nmc * 256 * 4));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, Y, (long)M * 128, 1u);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, X, (long)M * K, 2u);
    TnBatch tb;
    auto job = [&]() {
        tb.njobs = 1;
        TnParams &t = tb.j[0].p;
        memset(&t, 0, sizeof(t));
        t.Trows = M; t.Y0 = Y; t.X0 = X; t.x_bstride = 0; t.ldx = K; t.Tx = M; t.taps = 1; t.nk0 = K / 128; t.Ktot = K;
        t.slabs = slabs; t.bias_slabs = bslabs; t.MC = MC; t.chunks_per_video = nmc; t.drop.thresh = 0; t.drop.scale = 1.f;
        tb.j[0].nkc = K / 128; tb.j[0].block0 = nmc; tb.j[0].x0_act = 0; tb.j[0].dual = 0;
    };
    if (getenv("TS_CHECK")) {   // one launch against a float64 host product of the slabs' sum (small sizes only)
        job();
        CK(LAUNCH(tb, 0));
        CK(hipDeviceSynchronize());
        double *ref = new double[(size_t)128 * K]();
        float *hy = new float[(size_t)M * 128], *hx = new float[(size_t)M * K], *hs = new float[(size_t)nmc * 128 * K];
        CK(hipMemcpy(hy, Y, (size_t)M * 128 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx, X, (size_t)M * K * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs, slabs, (size_t)nmc * 128 * K * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < M; ++t)
            for (int n = 0; n < 128; ++n) {
                const double g = hy[(size_t)t * 128 + n];
                for (int k = 0; k < K; ++k) ref[(size_t)n * K + k] += g * hx[(size_t)t * K + k];
            }
        double worst = 0, scale = 0;
        for (size_t i = 0; i < (size_t)128 * K; ++i) {
            double got = 0;
            for (int c = 0; c < nmc; ++c) got += hs[(size_t)c * 128 * K + i];
            worst = fmax(worst, fabs(got - ref[i]));
            scale = fmax(scale, fabs(ref[i]));
        }
        printf("check: max abs error %.3e against max |dW| %.3e (relative %.2e)\n", worst, scale, worst / scale);
        float hb[256];
        CK(hipMemcpy(hb, bslabs, sizeof(hb), hipMemcpyDeviceToHost));
        double bs = 0;
        for (int t = 0; t < (MC < M ? MC : M); ++t) bs += hy[(size_t)t * 128 + 5];
        printf("check: bias slab 0 channel 5: %.6f against %.6f\n", hb[5], bs);
    }
    const int wgs = nmc * (K / (USE_TW ? 128 : 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) { job(); CK(LAUNCH(tb, 0)); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) { job(); CK(LAUNCH(tb, 0)); }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, flop = 2.0 * M * 128.0 * K;
#if TW_STAMP
    {   // block 0, last launch: cycles per tile spent working / waiting at the tile barrier, multipliers (waves 0-3) and stagers (4-7)
        long long h[96];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tw_stamps), sizeof(h)));
        const double nt = MC / 32;
        {
            static long long hb[4096 * 4];
            CK(hipMemcpyFromSymbol(hb, HIP_SYMBOL(g_tw_blk), sizeof(hb)));
            const int nb = wgs < 4096 ? wgs : 4096;
            long long t0 = hb[0], t1 = hb[1];
            for (int i = 0; i < nb; ++i) { if (hb[i * 4] < t0) t0 = hb[i * 4]; if (hb[i * 4 + 1] > t1) t1 = hb[i * 4 + 1]; }
            double dmin = 1e30, dmax = 0, dsum = 0, smax = 0, cmin = 1e30, cmax = 0;
            for (int i = 0; i < nb; ++i) {
                const double d = (hb[i * 4 + 1] - hb[i * 4]) / 100.0, st = (hb[i * 4] - t0) / 100.0, ghz = hb[i * 4 + 2] / d / 1e3;
                dmin = fmin(dmin, d); dmax = fmax(dmax, d); dsum += d; smax = fmax(smax, st); cmin = fmin(cmin, ghz); cmax = fmax(cmax, ghz);
            }
            printf("  blocks: span %.1f us; start skew up to %.1f us; per-block duration min %.1f / mean %.1f / max %.1f us; clock %.2f-%.2f GHz\n",
                   (t1 - t0) / 100.0, smax, dmin, dsum / nb, dmax, cmin, cmax);
        }
        for (int w = 0; w < 4 + TW_NS; ++w) printf("  wave %d (%s) cycles/tile: work %.0f  barrier wait %.0f\n", w, w < 4 ? "multiplier" : "stager", h[w * 8] / nt, h[w * 8 + 1] / nt);
    }
#endif
#if TS_STAMP
    {   // per-wave cycle sums of block 0 over the last launch, per tile: pre-phase-0 | phase 0 (24 MFMAs) | pre-phase-1 | phase 1 | barrier | loop edge
        long long h[64];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts_stamps), sizeof(h)));
        const double nt = MC / 32;
        for (int w = 0; w < 8; ++w)
            printf("  wave %d cycles/tile: pre0 %.0f  ph0 %.0f  pre1 %.0f  ph1 %.0f  barrier %.0f  edge %.0f  total %.0f\n", w, h[w * 8] / nt,
                   h[w * 8 + 1] / nt, h[w * 8 + 2] / nt, h[w * 8 + 3] / nt, h[w * 8 + 4] / nt, h[w * 8 + 5] / nt,
                   (h[w * 8] + h[w * 8 + 1] + h[w * 8 + 2] + h[w * 8 + 3] + h[w * 8 + 4] + h[w * 8 + 5]) / nt);
    }
#endif
    printf("TS_ABL=%d xcd=%d M=%d K=%d MC=%d: %d workgroups x %d tiles: %.1f us per launch, %.1f TFLOP/s fp32-equivalent, %.2f us per tile\n",
           TS_ABL, kTsXcdOrder, M, K, MC, wgs, MC / 32, us, flop / us * 1e-6, us / (MC / 32) / ((wgs + 255) / 256));
    return 0;
}
