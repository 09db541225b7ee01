// Does a hipGraph shorten the GPU-side cost of a chain of dependent launches?  (The coarse chain of a step is ~30 launches whose boundary -- 3.0 - 3.4 us for an EMPTY
// 256-workgroup launch, tools/experiments/kernarg_latency_probe.hip -- is ~95 us of the 680-us step.)  N dependent empty launches on one stream, plain against the same
// N launches captured once into a graph and replayed:   hipcc --offload-arch=gfx950 -O3 tools/experiments/graph_launch_probe.hip -o tools/build/graph_probe && tools/build/graph_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_empty(float *p, int i) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0 && i < 0) p[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_touch(float *p, int i) {   // a little dependent memory traffic: every workgroup rewrites its own 1 KB
    float *q = p + (long)blockIdx.x * 256 + threadIdx.x;
    *q = *q + (float)i;
}
__global__ __launch_bounds__(256) void k_spin(float *p, int i) {    // ~8 us of work per workgroup (100 MHz s_memrealtime): the HOST is ahead of the GPU, as in the real step
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 800) __builtin_amdgcn_s_sleep(2);
    if (p && threadIdx.x == 0 && blockIdx.x == 0 && i < 0) p[0] = 1.f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int N = 30, REPS = 200;
    float *d;
    CK(hipMalloc(&d, 256 * 256 * 4));
    CK(hipMemset(d, 0, 256 * 256 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 3; ++kind) {
        auto chain = [&]() {
            for (int i = 0; i < N; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, d, i);
                else if (kind == 1) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, d, i);
                else hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, i);
            }
        };
        for (int w = 0; w < 20; ++w) chain();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REPS; ++r) chain();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms_plain;
        CK(hipEventElapsedTime(&ms_plain, e0, e1));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        chain();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms_graph;
        CK(hipEventElapsedTime(&ms_graph, e0, e1));
        printf("%s: %d dependent launches of 256 workgroups: plain %.2f us per launch, as a replayed graph %.2f us per launch\n", kind == 0 ? "empty kernel " : kind == 1 ? "1 KB per WG  " : "8 us of spin ",
               N, ms_plain * 1e3f / (REPS * N), ms_graph * 1e3f / (REPS * N));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
