// ON THE GPU BOX: does gfx950 execute the scalar-memory atomic s_atomic_add (returns through lgkmcnt, not vmcnt)?  1024 workgroups draw one ticket each
// from one counter: all tickets must be distinct and the counter must end at 1024.   hipcc --offload-arch=gfx950 -O3 tools/scalar_atomic_probe.hip -o /tmp/sap && /tmp/sap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned *p, unsigned *out) {
    unsigned v = 1;
    if (threadIdx.x < 64) {   // wave 0 (uniform)
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(p) : "memory");
        if (threadIdx.x == 0) out[blockIdx.x] = v;
    }
}
int main() {
    unsigned *p, *out;
    hipMalloc(&p, 256);
    hipMalloc(&out, 1024 * 4);
    hipMemset(p, 0, 256);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, p, out);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<unsigned> h(1024);
    unsigned c = 0;
    hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(&c, p, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = c == 1024;
    for (int i = 0; i < 1024; ++i) ok = ok && h[i] == (unsigned)i;
    printf("counter %u, tickets distinct 0..1023: %s\n", c, ok ? "yes" : "NO");
    return ok ? 0 : 1;
}
