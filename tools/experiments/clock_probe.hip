// Shader clock of a LIGHT latency-bound kernel (one wave per SIMD, a dependent f64 add chain with DPP moves: the Viterbi DP's
// instruction mix) as a function of how many CUs run it: s_memtime (shader cycles) against s_memrealtime (100 MHz wall clock).
// Question (r4): why does the 256-in-flight DP launch take 1.7x the 9-in-flight one when every video has a CU of its own?
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/build/clock_probe && tools/build/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void chain(double *out, long long *stamps, int iters, double x) {
    double v = x * (threadIdx.x + 1);
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, true);
            v = v + __hiloint2double(__double2hiint(v), lo) * 1e-9;
        }
        if ((i & 15) == 15) __builtin_amdgcn_s_barrier();
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}
int main() {
    double *out;
    long long *st, h[2 * 1024];
    hipMalloc(&out, sizeof(double) * 1024 * 256);
    hipMalloc(&st, sizeof(h));
    for (int rep = 0; rep < 2; ++rep)
        for (int grid : {1, 8, 32, 64, 128, 256, 512}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(chain, dim3(grid), dim3(256), 0, 0, out, st, 2000, 1.0);   // warm
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(chain, dim3(grid), dim3(256), 0, 0, out, st, 20000, 1.0);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, st, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int b = 0; b < grid; ++b) {
                cyc += (double)h[2 * b];
                real += (double)h[2 * b + 1];
            }
            printf("grid %4d: %8.1f us  s_memtime/s_memrealtime = %.3f (x 100 MHz = %.0f MHz if s_memtime counts shader cycles); cycles per chain step %.2f\n", grid,
                   ms * 1e3, cyc / real, cyc / real * 100.0, cyc / grid / (20000.0 * 16));
        }
    return 0;
}
