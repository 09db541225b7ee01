// ON THE GPU BOX: where does the dispatcher put the workgroups of a one-round launch?  grid = 128 / 256 / 512 workgroups of 256 threads with LDS_KB of LDS each,
// every workgroup busy for ~5 us; each records (XCC_ID, HW_ID) and its start / end (s_memrealtime).  Prints how many workgroups each CU received and the launch's
// span -- a launch of 256 four-wave workgroups is "one per CU" only if the dispatcher makes it so.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o /tmp/dp && /tmp/dp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
struct Rec { unsigned xcc, hwid; long long t0, t1; };
template <int LDS_KB>
__global__ __launch_bounds__(256) void k(Rec *out, int spin_ticks) {
    __shared__ float lds[LDS_KB * 256];
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    float a = lds[(threadIdx.x * 7) & 255];
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) a = a * 1.0001f + 0.5f;
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) {
        Rec r{xcc, hwid, t0, (long long)__builtin_amdgcn_s_memrealtime()};
        out[blockIdx.x + gridDim.x * blockIdx.y] = r;
        lds[0] = a;
    }
}
template <int LDS_KB>
static void run(int gx, int gy, Rec *d) {
    const int n = gx * gy;
    std::vector<Rec> h(n);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<LDS_KB>, dim3(gx, gy), dim3(256), 0, 0, d, 500);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, sizeof(Rec) * n, hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> per_cu;
    long long t0 = h[0].t0, t1 = h[0].t1;
    for (auto &r : h) {
        // HW_ID: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se (gfx9 layout); XCC_ID [3:0]
        const unsigned cu = (r.hwid >> 8) & 15, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7, xcc = r.xcc & 15;
        per_cu[((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu]++;
        t0 = std::min(t0, r.t0);
        t1 = std::max(t1, r.t1);
    }
    int hist[8] = {0};
    for (auto &kv : per_cu) hist[std::min(kv.second, 7)]++;
    long long last_start = 0;
    for (auto &r : h) last_start = std::max(last_start, r.t0 - t0);
    printf("LDS %3d KB  grid %3d x %d = %4d workgroups: %3zu distinct CUs; CUs with 1 / 2 / 3 / 4+ workgroups: %d / %d / %d / %d; first start -> last end %.2f us, last start %.2f us\n",
           LDS_KB, gx, gy, n, per_cu.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7], (t1 - t0) / 100.0, last_start / 100.0);
}
int main() {
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * 4096);
    for (int g : {16, 32, 64}) {
        run<1>(g, 8, d);
        run<32>(g, 8, d);
        run<64>(g, 8, d);
        run<96>(g, 8, d);
    }
    run<64>(256, 1, d);
    run<64>(255, 1, d);
    return 0;
}
