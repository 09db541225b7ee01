// Does HIP on this stack take kernel arguments larger than 4 KB?  (CUDA's classic limit; the static-runs launch passes its job table and its line of
// work by value.)   hipcc --offload-arch=gfx950 -O2 tools/experiments/kernarg_probe.hip -o /tmp/kernarg_probe && /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
struct Big { unsigned v[N]; };
template <int N>
__global__ void k(const Big<N> b, unsigned *out) {
    unsigned s = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) s += b.v[i];
    atomicAdd(out, s);
}
template <int N>
static void run(unsigned *d) {
    Big<N> b;
    unsigned want = 0;
    for (int i = 0; i < N; ++i) { b.v[i] = i * 7 + 1; want += b.v[i]; }
    hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
    hipError_t e = hipGetLastError();
    unsigned got = 0;
    hipError_t e2 = hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
    printf("kernel argument of %5zu bytes: launch %s, copy %s, sum %s\n", sizeof(b), hipGetErrorString(e), hipGetErrorString(e2), got == want ? "ok" : "WRONG");
}
int main() {
    unsigned *d;
    hipMalloc(&d, 4);
    run<512>(d); run<1000>(d); run<1024>(d); run<1100>(d); run<2048>(d); run<4000>(d);
    return 0;
}
