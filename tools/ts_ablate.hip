// Stand-alone timing harness for the split-bf16 weight-gradient kernel (mucon_amd/csrc/gemm_tn_split.hpp), run ON THE GPU BOX:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTS_ABL=n] tools/ts_ablate.hip -o /tmp/ts_ablate && /tmp/ts_ablate [M] [K] [MC] [iters]
// One first_conv-shaped job: dW[128][K] = G[M][128]^T X[M][K].  TS_ABL removes parts of the loop (timing only, results are
// garbage): 1 MFMAs, 2 the X split, 4 the G split and its LDS stores, 8 the global loads.  Prints microseconds per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../mucon_amd/csrc/gemm_tn_split.hpp"
int g_tn_batch_ks = 2;
int g_ts_xcd = 1;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(float *p, long n, uint32_t seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        p[i] = (float)(int)(mix32((uint32_t)i * 2654435761u + seed) >> 8) * (1.f / 8388608.f) - 1.f;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, K = argc > 2 ? atoi(argv[2]) : 2048;
    const int MC = argc > 3 ? atoi(argv[3]) : 2048, iters = argc > 4 ? atoi(argv[4]) : 20;
    if (argc > 5) g_ts_xcd = atoi(argv[5]);
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
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) { job(); CK(launch_ts_batch(tb, 0)); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) { job(); CK(launch_ts_batch(tb, 0)); }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, flop = 2.0 * M * 128.0 * K;
    const int wgs = nmc * (K / 256);
    printf("TS_ABL=%d xcd=%d M=%d K=%d MC=%d: %d workgroups x %d tiles: %.1f us per launch, %.1f TFLOP/s fp32-equivalent, %.2f us per tile\n",
           TS_ABL, g_ts_xcd, M, K, MC, wgs, MC / 32, us, flop / us * 1e-6, us / (MC / 32) / ((wgs + 255) / 256));
    return 0;
}
