// How long does a wave wait for its kernel arguments, and does KERNARG PRELOAD (gfx950: the command processor writes the first <= 16 dwords of the kernel
// arguments into SGPRs while it launches the wave; -mllvm -amdgpu-kernarg-preload-count=16, only for leading scalar / pointer arguments) remove that wait?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/kernarg_latency_probe.hip -o tools/build/kp_plain && tools/build/kp_plain
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/experiments/kernarg_latency_probe.hip -o tools/build/kp_pre && tools/build/kp_pre
// Every launch: 256 workgroups x 256 threads; wave 0 of each workgroup stamps s_memtime at entry and again once a value that depends on a kernel argument exists
// (a pointer from the argument block, used for the store).  Two argument forms: a 200-byte struct by value in FRONT of the pointer (what cs_kernel / fs_kernel
// take: never preloaded) and the pointer + three ints in front of the struct.  Launches are issued back to back on one stream, the argument block changes every launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Big { int v[50]; };
__global__ __launch_bounds__(256) void k_struct_first(const Big b, long long *out, int slot) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    long long *o = out + (long)slot * 256 + blockIdx.x;     // needs `out` and `slot`
    asm volatile("" ::"s"(o));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) *o = (t1 - t0) + (b.v[7] == 12345 ? 1 : 0);
}
__global__ __launch_bounds__(256) void k_ptr_first(long long *out, int slot, int x, int y, const Big b) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    long long *o = out + (long)slot * 256 + blockIdx.x;
    asm volatile("" ::"s"(o));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) *o = (t1 - t0) + (b.v[7] == 12345 ? 1 : 0);
}
int main() {
    const int N = 200;
    long long *d;
    hipMalloc(&d, sizeof(long long) * 256 * N);
    std::vector<long long> h(256 * N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int form = 0; form < 2; ++form)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0, sizeof(long long) * 256 * N);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int i = 0; i < N; ++i) {
                Big b;
                for (int j = 0; j < 50; ++j) b.v[j] = i + j;
                if (form == 0) hipLaunchKernelGGL(k_struct_first, dim3(256), dim3(256), 0, 0, b, d, i);
                else hipLaunchKernelGGL(k_ptr_first, dim3(256), dim3(256), 0, 0, d, i, i, i, b);
            }
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, sizeof(long long) * 256 * N, hipMemcpyDeviceToHost);
            std::vector<long long> v(h.begin() + 256 * 20, h.end());
            std::sort(v.begin(), v.end());
            printf("%s: %d launches back to back %.2f us per launch; cycles from entry to an argument-dependent value: min %lld  median %lld  p90 %lld  max %lld\n",
                   form == 0 ? "struct first (no preload possible)" : "pointer + ints first            ", N, ms * 1000.f / N, v[0], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
        }
    return 0;
}
