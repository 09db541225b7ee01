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

struct HeadFwdArgs {
    const float *enc;   // [B][Tz][H]
    const float *w, *b; // [C][H], [C]
    float *logits, *logp;  // [B][Tf][C] or null
    float *logp_z;      // [B][Tz][C]
    int Tz, Tf, H, C;
    float scale;
};

// What a deferred mucon_head_fwd leaves behind for the next mucon_lstm_fwd on the same stream (defined in mucon_hip.hip; H = 128 only)
struct HeadFwdPending {
    bool armed = false, pending = false;
    HeadFwdArgs a;
    int gx = 0, gy = 0;          // the grid head_fwd_z_kernel would have been launched on
    hipStream_t stream = nullptr;
};
extern HeadFwdPending g_head_fwd_pending;
extern "C" int head_fwd_flush();   // (mucon_hip.hip; not part of the ABI) launches a pending forward kernel on its own, on the stream it was left on

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
#pragma clang fp contract(off)   // every multiply-add below is an explicit fmaf, nothing else may fuse: the two translation units that inline this body (other flags,
                                 // other neighbours) must produce the same bits -- tests/test_gpu_fused_step.py::test_fused_step_deferrals_change_no_bit
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
        if (zi < nz && c < C && s != 0.f) d = __builtin_fmaf(-expf(lpz[i]), s, d);
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
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(G1[half * 4 + j][c], w, acc[j]);
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
            for (int zi = 0; zi < HB_Z; ++zi) sacc = __builtin_fmaf(G1[zi][c], e[zi], sacc);
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

// Forward organised by z rows (H a multiple of 16): a workgroup owns HF_Z rows of the encoding and the frames that map onto
// them.  Thread (class c = tid >> 2, k-quarter q = tid & 3) keeps its share of W[c] in registers straight from global (the four
// lanes of a class read 64 contiguous bytes per step) and multiplies it with all HF_Z rows from LDS; the quarters meet by two
// shuffles.  No W staging pass, no divisions: 22 -> ~8 us at B=8, T=4096 against the frame-organised kernel.
constexpr int HF_Z = 8;
// (the kernel's body as a function of (LDS carve, block x, block y, thread, active): head_fwd_z_kernel calls it with its own indices; the LSTM's forward recurrence
// launch (512 threads) calls it from extra workgroups, two 256-thread blocks each, when the caller deferred the y-head's forward -- mucon_head_fwd_defer.  An
// inactive half (an odd block count) walks the same barriers with no rows: it loads nothing it stores.)
__device__ __forceinline__ void head_fwd_z_body(const HeadFwdArgs &a, float *smem, const int bx, const int by, const int tid, const bool active) {
#pragma clang fp contract(off)   // (explicit fmaf chains only: see head_bwd_z_body)
    const int H = a.H, C = a.C;
    float *Es = smem;                      // [HF_Z][H]
    float *Ls = Es + HF_Z * H;             // [HF_Z][MAXC] logits
    float *Ps = Ls + HF_Z * HEAD_MAXC;     // [HF_Z][MAXC] log-probs
    const int b = by;
    const int z0 = bx * HF_Z;
    const int nz = active ? min(HF_Z, a.Tz - z0) : 0;
    const int c = tid >> 2, q = tid & 3;
    for (int e = tid * 4; e < nz * H; e += 1024)
        *reinterpret_cast<f32x4 *>(Es + e) = *reinterpret_cast<const f32x4 *>(a.enc + ((long)b * a.Tz + z0) * H + e);
    for (int e = nz * H + tid; e < HF_Z * H; e += 256) Es[e] = 0.f;
    // (r5) H = 128: the thread's eight pieces of W are requested in front of the barrier, beside the rows' own trip (they were a second trip behind it)
    const float *wr = a.w + (long)min(c, C - 1) * H;
    f32x4 wpre[8];
    const bool pre = H == 128;
    if (pre) {
#pragma unroll
        for (int i = 0; i < 8; ++i) wpre[i] = *reinterpret_cast<const f32x4 *>(wr + q * 4 + 16 * i);
    }
    __syncthreads();
    float acc[HF_Z];
#pragma unroll
    for (int zi = 0; zi < HF_Z; ++zi) acc[zi] = 0.f;
    if (c < C && pre) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = q * 4 + 16 * i;
#pragma unroll
            for (int zi = 0; zi < HF_Z; ++zi) {
                const f32x4 ev = *reinterpret_cast<const f32x4 *>(Es + zi * H + k);
                acc[zi] = __builtin_fmaf(wpre[i][3], ev[3], __builtin_fmaf(wpre[i][2], ev[2], __builtin_fmaf(wpre[i][1], ev[1], __builtin_fmaf(wpre[i][0], ev[0], acc[zi]))));
            }
        }
    } else if (c < C) {
        for (int k = q * 4; k < H; k += 16) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wr + k);
#pragma unroll
            for (int zi = 0; zi < HF_Z; ++zi) {
                const f32x4 ev = *reinterpret_cast<const f32x4 *>(Es + zi * H + k);
                acc[zi] = __builtin_fmaf(wv[3], ev[3], __builtin_fmaf(wv[2], ev[2], __builtin_fmaf(wv[1], ev[1], __builtin_fmaf(wv[0], ev[0], acc[zi]))));
            }
        }
    }
#pragma unroll
    for (int zi = 0; zi < HF_Z; ++zi) {
        acc[zi] += __shfl_xor(acc[zi], 1);
        acc[zi] += __shfl_xor(acc[zi], 2);
    }
    if (c < C && q == 0) {
        const float bias = a.b[c];
#pragma unroll
        for (int zi = 0; zi < HF_Z; ++zi) Ls[zi * HEAD_MAXC + c] = acc[zi] + bias;
    }
    __syncthreads();
    // log-softmax over the classes: a wave per z row, lane = class (C <= 64); the sum runs up a fixed shuffle tree
    for (int zi = tid >> 6; zi < nz; zi += 4) {
        const int cc = tid & 63;
        const float x = cc < C ? Ls[zi * HEAD_MAXC + cc] : -INFINITY;
        float m = x;
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float s = cc < C ? expf(x - m) : 0.f;
#pragma unroll
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if (cc < C) {
            const float lp = x - (m + logf(s));
            Ps[zi * HEAD_MAXC + cc] = lp;
            a.logp_z[((long)b * a.Tz + z0 + zi) * C + cc] = lp;
        }
    }
    __syncthreads();
    // the frames of these rows are contiguous: [first frame of z0, first frame of z0 + nz)
    const int fa = first_frame(z0, a.scale, a.Tz, a.Tf), fb = first_frame(z0 + nz, a.scale, a.Tz, a.Tf);
    if ((C & 3) == 0) {
        // (r5) 16-byte pieces: thread (frame slot tid / (C / 4), piece tid % (C / 4)); 256 / 12 = 21 frames per pass at C = 48 -- seven passes
        // for a workgroup's ~128 frames instead of 32 passes of 4-byte stores (a wave per frame, 48 of 64 lanes)
        const int C4 = C >> 2, slots = 256 / C4;
        const int slot = tid / C4, c4 = (tid - slot * C4) * 4;
        if (slot < slots) {
            for (int i = fa + slot; i < fb; i += slots) {
                const int zi = zmap(i, a.scale, a.Tz) - z0;
                const long gi = ((long)b * a.Tf + i) * C + c4;
                if (a.logits) *reinterpret_cast<f32x4 *>(a.logits + gi) = *reinterpret_cast<const f32x4 *>(Ls + zi * HEAD_MAXC + c4);
                if (a.logp) *reinterpret_cast<f32x4 *>(a.logp + gi) = *reinterpret_cast<const f32x4 *>(Ps + zi * HEAD_MAXC + c4);
            }
        }
        return;
    }
    for (int i = fa + (tid >> 6); i < fb; i += 4) {   // a wave per frame, lane = class
        const int cc = tid & 63;
        const int zi = zmap(i, a.scale, a.Tz) - z0;
        if (cc < C) {
            const long gi = ((long)b * a.Tf + i) * C + cc;
            if (a.logits) a.logits[gi] = Ls[zi * HEAD_MAXC + cc];
            if (a.logp) a.logp[gi] = Ps[zi * HEAD_MAXC + cc];
        }
    }
}
