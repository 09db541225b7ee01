// The y-head's z-level backward as a device function shared by two translation units (round 6): mucon_hip.hip launches it as head_bwd_z_kernel
// (small_kernels.hpp), shead.hip runs it in extra workgroups of the eight-workgroup decoder's backward launch when the caller deferred it
// (include/mucon_hip.h: mucon_head_bwd_defer, bit 1) -- that launch keeps eight CUs busy for ~65 us while the rest of the chip idles.
#pragma once
#include "common.hpp"

constexpr int HEAD_MAXC = 64; // classes supported by the LDS carve

// ------------------------------------------------------------------------------------------
// y-head.  F.interpolate(mode="nearest") source index (models.py:574), computed as torch does:
// scale = float(Tz)/float(Tf); src = min(int(floorf(i * scale)), Tz - 1).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int zmap(int i, float scale, int Tz) {
    const int z = (int)floorf((float)i * scale);
    return z < Tz - 1 ? z : Tz - 1;
}
// first frame i in [0, Tf] with zmap(i) >= z  (Tf when none)
__device__ __forceinline__ int first_frame(int z, float scale, int Tz, int Tf) {
    if (z <= 0) return 0;
    if (z >= Tz) return Tf;
    long guess = ((long)z * Tf) / Tz - 2;
    int i = guess < 0 ? 0 : (int)guess;
    while (i > 0 && zmap(i - 1, scale, Tz) >= z) --i;
    while (i < Tf && zmap(i, scale, Tz) < z) ++i;
    return i;
}


struct HeadBwdArgs {
    const float *enc, *w;          // [B][Tz][H], [C][H]
    const float *dlogits, *dlogp;  // [B][Tf][C] or null
    const float *logp_z;           // [B][Tz][C]
    float *denc;                   // [B][Tz][H]
    float *w_slabs, *b_slabs;      // [nblk][C][H], [nblk][C]
    int Tz, Tf, H, C;
    float scale;
};

// What a deferred mucon_head_bwd leaves behind for the next mucon_decoder_bwd on the same stream (defined in mucon_hip.hip)
struct HeadKernelPending {
    bool armed = false, pending = false;
    HeadBwdArgs a;
    int gx = 0, gy = 0;          // the grid head_bwd_z_kernel would have been launched on
    hipStream_t stream = nullptr;
};
extern HeadKernelPending g_head_kernel_pending;
extern "C" int head_kernel_flush();   // (mucon_hip.hip; not part of the ABI: the two translation units of the library share it) launches a pending kernel on its own, on the stream it was left on; MUCON_OK when nothing is pending

// Backward for H = 128 with HB_Z rows per workgroup (twice the workgroups of head_bwd_kernel), nothing staged but the rows
// themselves: thread (k = tid & 127, half = tid >> 7) reads its column of W straight from global (coalesced over k) for
// d_enc, keeps its column of the encoding rows in registers for the weight-gradient partials, and the per-row class sums run
// up shuffle trees.  30 -> ~14 us at B=8, T=4096.
constexpr int HB_Z = 8;
// (the kernel's body as a function of (block x, block y, blocks in x, thread): head_bwd_z_kernel calls it with its own indices; the eight-workgroup decoder's
// backward launch calls it from extra workgroups of the same 256 threads -- mucon_head_bwd_defer bit 1)
__device__ __forceinline__ void head_bwd_z_body(const HeadBwdArgs &a, const int bx, const int by, const int gx, const int tid) {
    __shared__ __attribute__((aligned(16))) float Es[HB_Z][128];
    __shared__ float G1[HB_Z][HEAD_MAXC], G2[HB_Z][HEAD_MAXC], S2[HB_Z];
    const int C = a.C;
    const int b = by;
    const int z0 = bx * HB_Z;
    const int nz = min(HB_Z, a.Tz - z0);
    const int blk = by * gx + bx;
    {
        const int zi = tid >> 5, k4 = (tid & 31) * 4;   // 8 rows x 32 float4
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (zi < nz) v = *reinterpret_cast<const f32x4 *>(a.enc + ((long)b * a.Tz + z0 + zi) * 128 + k4);
        *reinterpret_cast<f32x4 *>(&Es[zi][k4]) = v;
    }
    for (int o = tid; o < HB_Z * HEAD_MAXC; o += 256) {
        const int zi = o >> 6, c = o & 63;
        float g1 = 0.f, g2 = 0.f;
        if (zi < nz && c < C) {
            const int fa = first_frame(z0 + zi, a.scale, a.Tz, a.Tf);
            const int fb = first_frame(z0 + zi + 1, a.scale, a.Tz, a.Tf);
            // frames of the bin in order, eight loads in flight at a time (a bin has ~Tf/Tz frames)
            const int nfr = fb - fa;
            // (both arrays are read unconditionally, a missing one through the other's pointer with weight 0: a load behind a
            // branch costs a branch and a full wait per element)
            const float *p1 = a.dlogits ? a.dlogits : a.dlogp, *p2 = a.dlogp ? a.dlogp : a.dlogits;
            const float w1 = a.dlogits ? 1.f : 0.f, w2 = a.dlogp ? 1.f : 0.f;
            constexpr int HB_FRAMES = 16;   // (r5) frames of a bin requested at once (8: round 4 -- a bin of Tf / Tz = 16 frames was two dependent round trips)
            for (int base = 0; base < nfr; base += HB_FRAMES) {
                float v1[HB_FRAMES], v2[HB_FRAMES];
#pragma unroll
                for (int j = 0; j < HB_FRAMES; ++j) {
                    const long gi = ((long)b * a.Tf + fa + min(base + j, nfr - 1)) * C + c;
                    v1[j] = p1[gi];
                    v2[j] = p2[gi];
                }
#pragma unroll
                for (int j = 0; j < HB_FRAMES; ++j) {
                    if (base + j < nfr) {
                        g1 += v1[j];
                        g2 += v2[j];
                    }
                }
            }
            g1 *= w1;
            g2 *= w2;
        }
        G1[zi][c] = g1;
        G2[zi][c] = g2;
    }
    // (r5) the saved log-probabilities of this thread's two rows are requested in front of the barrier (they were a round trip behind it)
    float lpz[HB_Z / 4];
#pragma unroll
    for (int i = 0; i < HB_Z / 4; ++i) {
        const int zi = (tid >> 6) + 4 * i, c = tid & 63;
        lpz[i] = a.logp_z[((long)b * a.Tz + z0 + min(zi, nz - 1)) * C + min(c, C - 1)];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < HB_Z / 4; ++i) {   // a wave per row: S2 = sum_c G2, then the log-softmax backward
        const int zi = (tid >> 6) + 4 * i;
        const int c = tid & 63;
        float s = G2[zi][c];
#pragma unroll
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        float d = G1[zi][c] + G2[zi][c];
        if (zi < nz && c < C && s != 0.f) d -= expf(lpz[i]) * s;
        G1[zi][c] = d;
        if (c == 0) S2[zi] = s;
    }
    __syncthreads();
    const int k = tid & 127, half = tid >> 7;
    {   // d_enc[z][k] = sum_c dlogit[z][c] * W[c][k] for rows 4 half .. 4 half + 3
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            const float w = a.w[(long)c * 128 + k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += G1[half * 4 + j][c] * w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (half * 4 + j < nz) a.denc[((long)b * a.Tz + z0 + half * 4 + j) * 128 + k] = acc[j];
    }
    {   // partial dW[c][k] = sum_z dlogit[z][c] * enc[z][k] for classes c = half, half + 2, ...
        float e[HB_Z];
#pragma unroll
        for (int zi = 0; zi < HB_Z; ++zi) e[zi] = Es[zi][k];
        for (int c = half; c < C; c += 2) {
            float sacc = 0.f;
#pragma unroll
            for (int zi = 0; zi < HB_Z; ++zi) sacc += G1[zi][c] * e[zi];
            a.w_slabs[(long)blk * C * 128 + c * 128 + k] = sacc;
        }
    }
    if (tid < C) {
        float sacc = 0.f;
#pragma unroll
        for (int zi = 0; zi < HB_Z; ++zi) sacc += G1[zi][tid];
        a.b_slabs[(long)blk * C + tid] = sacc;
    }
}
