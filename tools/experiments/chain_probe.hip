// What one wave's dependent float32 add chain costs on this GPU, cold and right behind a chip-filling kernel: the floor
// of the Viterbi frame-score cumsum (np.cumsum is sequential; lanes = classes).  Prints ns per add for
//   regs    a chain over 16 register operands (no memory)
//   lds128  the same chain fed by ds_read_b128 with four reads in flight (the kernel's inner loop)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_probe.hip -o tools/build/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void chain_regs(float *out, int iters, float x) {
    float v[16];
    for (int u = 0; u < 16; ++u) v[u] = x * (u + 1 + threadIdx.x);
    float run = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) run = run + v[u];
        asm volatile("" : "+v"(run));
    }
    out[threadIdx.x] = run;
}
// mode bit 0: lanes >= 48 alias lane 47's row; bit 1: one ds_write_b32 of the running sum per four reads;
// bit 2: waves 1..7 keep writing transposed words into the other half of the LDS (the staging traffic)
__global__ void chain_lds(float *out, int iters, float x, int pitch, int mode) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * pitch; i += blockDim.x) sm[i] = x * i;
    __syncthreads();
    float *other = sm + 64 * pitch;
    if (tid >= 64) {
        if (mode & 4)
            for (int it = 0; it < iters / 8; ++it) {
                const int e = (it * 448 + tid - 64) % 2880, row = e / 12, c4 = e % 12;
                float *d = other + 4 * c4 * pitch + row;
                d[0] = x; d[pitch] = x; d[2 * pitch] = x; d[3 * pitch] = x;
            }
        return;
    }
    const int lane = (mode & 1) ? (tid < 48 ? tid : 47) : tid;
    const f4 *q = reinterpret_cast<const f4 *>(sm + lane * pitch);
    float *w = other + 62 * pitch + tid;
    float run = 0.f;
    for (int it = 0; it < iters; it += 256) {
        f4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        const f4 *p = q;
        for (int s = 0; s < 64; s += 4) {
            p += 4;
            __builtin_amdgcn_sched_barrier(0);
            run = run + q0.x; run = run + q0.y; run = run + q0.z; run = run + q0.w; q0 = p[0];
            __builtin_amdgcn_sched_barrier(0);
            run = run + q1.x; run = run + q1.y; run = run + q1.z; run = run + q1.w; q1 = p[1];
            __builtin_amdgcn_sched_barrier(0);
            run = run + q2.x; run = run + q2.y; run = run + q2.z; run = run + q2.w; q2 = p[2];
            __builtin_amdgcn_sched_barrier(0);
            run = run + q3.x; run = run + q3.y; run = run + q3.z; run = run + q3.w;
            if (mode & 2) w[0] = run;
            q3 = p[3];
            __builtin_amdgcn_sched_barrier(0);
        }
        run = run + q0.x + q1.x + q2.x + q3.x;
    }
    out[tid] = run;
}
__global__ void burn(float *out, int iters) {
    float a = threadIdx.x, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    float *out, *big;
    hipMalloc(&out, 4096);
    hipMalloc(&big, 2048 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int N = 32768;
    hipFuncSetAttribute(reinterpret_cast<const void *>(chain_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto timeit = [&](const char *name, auto launch) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-44s %7.1f us for %d adds = %.2f ns/add\n", name, ms * 1e3, N, ms * 1e6 / N);
        }
    };
    timeit("regs", [&] { hipLaunchKernelGGL(chain_regs, dim3(1), dim3(64), 0, 0, out, N / 16, 1e-3f); });
    const int pitches[3] = {516, 324, 276};
    for (int pi = 0; pi < 3; ++pi)
        for (int mode = 0; mode < 8; ++mode) {
            char name[96];
            snprintf(name, sizeof(name), "lds128 pitch %d%s%s%s", pitches[pi], mode & 1 ? " alias48" : "", mode & 2 ? " +write" : "",
                     mode & 4 ? " +7 staging waves" : "");
            const int pitch = pitches[pi];
            const size_t smem = (size_t)2 * 64 * pitch * 4 > 160 * 1024 ? 160 * 1024 : (size_t)2 * 64 * pitch * 4;
            if ((size_t)(64 + 63) * pitch * 4 > smem) continue;
            timeit(name, [&] { hipLaunchKernelGGL(chain_lds, dim3(1), dim3(mode & 4 ? 512 : 64), smem, 0, out, N, 1e-3f, pitch, mode); });
        }
    return 0;
}
