// Known-answer check of the wave reductions of csrc/decoder.hpp (DPP row operations, v_permlane16_swap / v_permlane32_swap) on
// non-integer data:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dpp_probe.hip -o tools/build/dpp_probe && gpurun -- ./tools/build/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../mucon_amd/csrc/lstm.hpp"
#include "../mucon_amd/csrc/decoder.hpp"
__global__ void k(const float *in, float *o8, float *o4, float *os, float *om, float *raw) {
    const int lane = threadIdx.x;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[i * 64 + lane];
    o8[lane] = wave_sum_rows<8>(lane, v);
    float w[4];
    for (int i = 0; i < 4; ++i) w[i] = in[i * 64 + lane];
    o4[lane] = wave_sum_rows<4>(lane, w);
    os[lane] = wave_sum(in[lane]);
    om[lane] = wave_max(in[lane]);
    raw[lane] = swap32_add(in[lane], in[64 + lane]);
    raw[64 + lane] = swap16_add(in[lane], in[64 + lane]);
    raw[128 + lane] = dpp_f<DPP_ROR8>(in[lane]);
    raw[192 + lane] = dpp_f<DPP_HALF_MIRROR>(in[lane]);
    raw[256 + lane] = dpp_f<DPP_XOR1>(in[lane]);
    raw[320 + lane] = dpp_f<DPP_XOR2>(in[lane]);
    raw[384 + lane] = dpp_f<DPP_MIRROR>(in[lane]);
}
int main() {
    float h[512], *d, *o;
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 37) % 101) * 0.37f + 1000.f * (i / 64);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4 * 1024);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, o + 64, o + 128, o + 192, o + 256);
    float r[1024];
    hipMemcpy(r, o, 4 * 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        double s8 = 0, s4 = 0, ss = 0, mx = -1e30;
        for (int j = 0; j < 64; ++j) { s8 += h[((l >> 3) & 7) * 64 + j]; s4 += h[((l >> 4) & 3) * 64 + j]; ss += h[j]; mx = fmax(mx, h[j]); }
        if ((l & 7) == 0 && fabs(r[l] - s8) > 1e-2) { printf("rows8 lane %d got %f want %f\n", l, r[l], s8); ++bad; }
        if ((l & 15) == 0 && fabs(r[64 + l] - s4) > 1e-2) { printf("rows4 lane %d got %f want %f\n", l, r[64 + l], s4); ++bad; }
        if (fabs(r[128 + l] - ss) > 1e-2) { printf("sum lane %d got %f want %f\n", l, r[128 + l], ss); ++bad; }
        if (r[192 + l] != mx) { printf("max lane %d got %f want %f\n", l, r[192 + l], mx); ++bad; }
    }
    const float *raw = r + 256;
    printf("swap32_add lane0 %f (a0+a32=%f) lane32 %f (b0+b32=%f)\n", raw[0], h[0] + h[32], raw[32], h[64] + h[96]);
    printf("swap16_add lane0 %f (a0+a16=%f) lane16 %f (b0+b16=%f) lane32 %f (a32+a48=%f)\n", raw[64], h[0] + h[16], raw[64 + 16], h[64] + h[80], raw[64+32], h[32]+h[48]);
    printf("ror8 lane0 %f (in8=%f) lane8 %f (in0=%f)\n", raw[128], h[8], raw[136], h[0]);
    printf("half_mirror lane0 %f (in7=%f) lane1 %f (in6 %f)\n", raw[192], h[7], raw[193], h[6]);
    printf("xor1 lane0 %f (in1=%f); xor2 lane0 %f (in2=%f); mirror lane0 %f (in15=%f)\n", raw[256], h[1], raw[320], h[2], raw[384], h[15]);
    printf("%s\n", bad ? "FAILED" : "all reductions ok");
    return bad != 0;
}
