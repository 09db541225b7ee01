// Two-stage residual-layer kernels (the launches of gemm_fused.hpp) on the bf16 MFMA with EXACT three-way operand splitting
// -- the arithmetic of gemm_split.hpp (x = hi + mid + lo, six of nine partial products, fp32 accumulate: fp32-grade) -- for the
// levels that fill the chip (>= 16,384 rows in the batch: the full-resolution layers, where the f32-MFMA kernel spends 50 us a
// launch at half of the f32 peak).
//
//   FWD  h = act(dilated_conv(x) + b1) -> out1;   y = x + dropout(conv_1x1(h) + b2) [-> max / sum pool] -> out2 (, out_pre)
//        (reference src/core/modules/temporal.py:43-53, :137-142)
//   BWD  g = dgrad_dilated_conv(dpre_{l+1}) + res1 [* act'(mask1)] -> out1;   dpre_l = ((g * dropmask) . W2^T) * act'(h_l) -> out2
//
// Orientation.  Both products are computed TRANSPOSED: the weights are the MFMA's A operand (rows = output channels), the
// activations its B operand (columns = time steps), v_mfma_f32_16x16x32_bf16.  A wave owns 16 time steps x all 128 channels:
//   * stage 1's B fragments are the wave's own activation rows -- lane (t, g) loads 8 consecutive channels (32 bytes) of its
//     row per 32-deep step, straight from global memory into operand position, split in registers, once;
//   * the result h^T[n][t] has its time step on the lane and its channels in the accumulator registers, which is exactly the B
//     operand layout of the next product (it sums over the accumulator's ROW index): stage 2 needs no transpose, no LDS, no
//     exchange between waves -- the stage-1 epilogue (bias, non-linearity / residual, mask, dropout replay) runs on the
//     registers, which are then split and fed back.  Only the k order inside a 32-step differs (accumulator order:
//     n = 16 (j >> 2) + 4 g + (j & 3)); the W2 image is packed in that order.
//   * only the weights go through LDS: pre-split fragment-ordered images written by fs_pack_body ([k-step][plane 3]
//     [channel block 8][lane 64][8 bf16]: conflict-free ds_read_b128), 64-deep tiles (48 KB), double buffered, one barrier per
//     tile; the eight tiles of both stages (6 x W1, 2 x W2) form one sequence, so W2 is in flight while stage 1 ends.
// One workgroup = 8 waves = 128 time steps; the waves only share the weight tiles.  16x16x32 rather than 32x32x16: a wave's
// 16 rows need no k-split across waves, and the activation fragment of a step is reused by 48 MFMAs (one vector instruction
// per MFMA instead of three).
#pragma once
#include <type_traits>

#include "common.hpp"
#include "dispatch.hpp"
#include "gemm_fused.hpp"

constexpr int FS_WSTEP = 3 * 128 * 32;          // bf16 elements of one 32-deep step of a [128][K] image: [plane 3][block 8][lane 64][8]
constexpr int FS_WTILE = 2 * FS_WSTEP;          // 64-deep LDS tile: 24,576 elements = 49,152 B
constexpr int FS_SMEM_BYTES = 2 * FS_WTILE * 2; // double buffered: 98,304 B
constexpr int FS_IMG_K384 = 3 * 128 * 384;      // elements of a K = 384 image
constexpr int FS_IMG_K128 = 3 * 128 * 128;
constexpr int FS_LAYER_ELEMS = 2 * FS_IMG_K384 + 4 * FS_IMG_K128;   // W1f, W1b, W2, W2t, W1fc', W1bc' of one layer (983,040 B)

typedef float f32x8 __attribute__((ext_vector_type(8)));

#ifndef FS_STAMP
#define FS_STAMP 0   // timing builds (MUCON_HIPCC_FLAGS=-DFS_STAMP=1): s_memtime sums per phase, block 0 of every variant -> mucon_test_read_stamps
#endif
#if FS_STAMP
__device__ long long g_fs_stamps[64 * 8 * 8];   // [variant: BWD * 32 + POOL * 4 + ONE * 2 + (NW == 8)][wave][phase]
#define FS_T(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define FS_T(k) do { } while (0)
#endif

// ---- weight images ---------------------------------------------------------------------------------------------------
// Fragment (k-step q, channel block nb, lane (r, g)): the 8 k of channel n = 16 nb + r that lane group g feeds the MFMA:
//   natural order      k32 = 8 g + j                      (stage 1: the lane loads 8 consecutive channels of its row)
//   accumulator order  k32 = 16 (j >> 2) + 4 g + (j & 3)  (stage 2: the lane's accumulator registers of two channel blocks)
// Images 4 and 5 are the CENTRE taps of W1f / W1b in accumulator order: what a dilated conv whose dilation reaches past the sequence
// multiplies when its operand is the previous product's accumulator (the chained launches of gemm_coarse_split.hpp: ct_kernel).
struct FsPackArgs {
    const float *dil_w[16];   // [o][i][tap]
    const float *pw_w[16];    // [m][n]
    const float *last_w;      // [o][i]: slot nl (when non-null) holds last_conv's two images in the W2 / W2t positions, NATURAL order,
                              // and last_conv's forward image once more in position 4, accumulator order
    uint16_t *img;            // [slot][FS_LAYER_ELEMS]
    int nl;
};
// (a device function: pack_all_kernel of mucon_hip.hip runs it in the same launch as pack_weights' blocks)
__device__ __forceinline__ void fs_pack_body(const FsPackArgs &a, const int slot, const int f) {   // f: fragment index over the six matrices: 6144 + 6144 + 2048 + 2048 + 2048 + 2048
    if (f >= 20480) return;
    const bool last = slot == a.nl;
    if (last && (f < 12288 || f >= 18432)) return;
    const float *dw = last ? nullptr : a.dil_w[slot], *pw = last ? a.last_w : a.pw_w[slot];
    int mat, q;
    if (f < 6144) { mat = 0; q = f; }
    else if (f < 12288) { mat = 1; q = f - 6144; }
    else { mat = 2 + ((f - 12288) >> 11); q = (f - 12288) & 2047; }
    const int kstep = q >> 9, rem = q & 511;
    const int nb = rem >> 6, g = (rem >> 4) & 3, r = rem & 15;
    const int n = nb * 16 + r;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool natural = mat < 2 || (last && mat < 4);
        const int k32 = natural ? 8 * g + j : 16 * (j >> 2) + 4 * g + (j & 3);
        const int k = kstep * 32 + k32;
        if (mat == 0) v[j] = dw[((long)n * 128 + (k & 127)) * 3 + (k >> 7)];          // W1f[o = n][tap*128 + i]
        else if (mat == 1) v[j] = dw[((long)(k & 127) * 128 + n) * 3 + (k >> 7)];     // W1b[i = n][tap*128 + o]
        else if (mat == 2) v[j] = pw[(long)n * 128 + k];                              // W2[m = n][n']
        else if (mat == 3) v[j] = pw[(long)k * 128 + n];                              // W2t[n][m]
        else if (mat == 4) v[j] = last ? pw[(long)n * 128 + k] : dw[((long)n * 128 + k) * 3 + 1];   // centre tap of W1f (last_conv: W)
        else v[j] = dw[((long)k * 128 + n) * 3 + 1];                                  // centre tap of W1b
    }
    u32x4 hh, mm, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        uint32_t x, y, z;
        sp_split2(v[2 * e], v[2 * e + 1], x, y, z);
        hh[e] = x;
        mm[e] = y;
        ll[e] = z;
    }
    const long moff = mat == 0 ? 0 : (mat == 1 ? FS_IMG_K384 : 2L * FS_IMG_K384 + (long)(mat - 2) * FS_IMG_K128);
    uint16_t *dst = a.img + (long)slot * FS_LAYER_ELEMS + moff + (long)kstep * FS_WSTEP + rem * 8;
    *reinterpret_cast<u32x4 *>(dst) = hh;
    *reinterpret_cast<u32x4 *>(dst + 128 * 32) = mm;
    *reinterpret_cast<u32x4 *>(dst + 2 * 128 * 32) = ll;
}

// ---- the kernel -------------------------------------------------------------------------------------------------------
template <bool BWD, int POOL, bool ONE = false, int NW = 8>
// (r6) operand pointers, row count and tap distance as LEADING SCALAR arguments: preloaded into SGPRs with the wave (see cs_kernel, gemm_coarse_split.hpp)
__global__ __launch_bounds__(64 * NW) void fs_kernel(const float *__restrict__ A_, const uint16_t *__restrict__ W1img, const uint16_t *__restrict__ W2img, const int Trows_,
                                                     const int tap_step_, const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t fs_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    constexpr int NTHR = 64 * NW, NQ = FS_WTILE / (NTHR * 8);   // 16-byte pieces of a W tile per thread: 6 (8 waves) or 12 (4 waves)
    const int t0 = blockIdx.x * (16 * NW);
    const int trow_raw = t0 + wave * 16 + c;
    const bool valid = trow_raw < Trows_;
    const long vbase = (long)b * Trows_;
    const float *Ab = A_ + vbase * 128 + 8 * g;

#if FS_STAMP
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 prologue | 1 step 0 of a tile | 2 step 1 | 3 barrier | 4 epilogue 1 | 5 stage 2 | 6 epilogue 2
    long long st_prev = __builtin_amdgcn_s_memtime();
#endif
    f32x4 ra[2][4];       // two k-tiles of this lane's activation values in flight: [set][2 steps x 2 halves]
    bool rok[2] = {true, true};
    u32x4 rws[2][NQ];     // this thread's share of the next TWO W tiles (a tile is requested two tiles before it is multiplied)
    auto gloadA = [&](int S, auto SET) {      // k-tile S of stage 1: tap S >> 1, channels 64 (S & 1) ..
        constexpr int Q = decltype(SET)::value;
        const int tap = S >> 1;
        const int ts = trow_raw + (tap - 1) * tap_step_;
        rok[Q] = valid && ts >= 0 && ts < Trows_;
        const float *src = Ab + (long)min(max(ts, 0), Trows_ - 1) * 128 + 64 * (S & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[Q][i] = *reinterpret_cast<const f32x4 *>(src + 32 * (i >> 1) + 4 * (i & 1));
    };
    auto gloadW = [&](int v, auto SET) {      // tile v of the unified sequence: 0..5 W1, 6..7 W2
        constexpr int Q = decltype(SET)::value;
        const uint16_t *src = (v < 6 ? W1img + (long)v * FS_WTILE : W2img + (long)(v - 6) * FS_WTILE) + tid * 8;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rws[Q][q] = *reinterpret_cast<const u32x4 *>(src + q * (NTHR * 8));
    };
    auto storeW = [&](int buf, auto SET) {
        constexpr int Q = decltype(SET)::value;
#pragma unroll
        for (int q = 0; q < NQ; ++q) *reinterpret_cast<u32x4 *>(fs_smem + buf * FS_WTILE + tid * 8 + q * (NTHR * 8)) = rws[Q][q];
    };
    struct Planes { bf16x8 pl[3]; };
    auto split8 = [&](const float (&x)[8]) {
        u32x4 hh, mm, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t a, bb, cc;
            sp_split2(x[2 * e], x[2 * e + 1], a, bb, cc);
            hh[e] = a;
            mm[e] = bb;
            ll[e] = cc;
        }
        Planes P;
        P.pl[0] = __builtin_bit_cast(bf16x8, hh);
        P.pl[1] = __builtin_bit_cast(bf16x8, mm);
        P.pl[2] = __builtin_bit_cast(bf16x8, ll);
        return P;
    };
    auto convertA = [&](f32x4 v0, f32x4 v1, bool ok) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[e] = ok ? v0[e] : 0.f;
            x[4 + e] = ok ? v1[e] : 0.f;
        }
        return split8(x);
    };

    f32x4 acc[8];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // 48 MFMAs of one 32-deep step: W fragment (step s of the tile in `buf`, plane pl, channel block nb) x the wave's operand
    auto mfma_step = [&](int buf, int s, const Planes &X) {
        const uint16_t *base = fs_smem + buf * FS_WTILE + s * FS_WSTEP + lane * 8;
        bf16x8 w[8][3];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[nb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (128 * 32) + nb * 512);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {   // small terms first; all six land in the same fp32 accumulator
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][1], X.pl[1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][2], X.pl[0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][0], X.pl[2], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][1], X.pl[0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][0], X.pl[1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][0], X.pl[0], acc[nb], 0, 0, 0);
        }
    };
    // 48 MFMAs, 24 fragment reads (the first channel block's in front, the rest under the MFMAs), one vector instruction per MFMA,
    // the LDS stores of the next tile in the second half
    auto weave = [&](auto WITH_STORES) {
        constexpr bool with_stores = decltype(WITH_STORES)::value;
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 42 && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            if (with_stores && i >= 24 && (i & (NQ == 6 ? 3 : 1)) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };
    auto use = [](const Planes &P) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, P.pl[pl])));
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // lane (t, g) holds channels n = 16 nb + 4 g + e (e = 0..3) of its time step in acc[nb].
    // UNPOOL (BWD across a pooled boundary, POOL 3 max / 4 sum): stage 1 runs on the COARSE level's rows; its epilogue routes every
    // value onto the arg-max row of the forward pair (first wins ties, as torch) or onto both (sum pooling) of the FINE level --
    // rows 2t, 2t + 1 stay with lane t -- writes the un-pooled gradient (out1) and feeds stage 2 twice (R2 = 2 row sets).
    constexpr bool UNPOOL = BWD && POOL >= 3;
    constexpr int R2 = UNPOOL ? 2 : 1;
    const int tcl = min(trow_raw, Trows_ - 1);
    const long grow = (vbase + tcl) * 128 + 4 * g;   // this lane's row of the stage-1 level, its first channel of block 0
    const long grow2 = UNPOOL ? ((long)b * p.Tfine + 2 * tcl) * 128 + 4 * g : grow;   // first of its R2 rows of the stage-2 level
    f32x4 aux1[8], msk1[BWD ? 8 : 1];
    auto prefetch1 = [&]() {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            // (unconditional loads from a valid address, the condition on the arithmetic below: a load behind a branch makes the compiler wait for
            //  everything in flight at the join -- eight branches with a wait each in front of the stage-1 epilogue; measured neutral here, 0.6915 = 0.6918 ms)
            if (!BWD) aux1[nb] = *reinterpret_cast<const f32x4 *>(p.bias1 + 16 * nb + 4 * g);
            else aux1[nb] = *reinterpret_cast<const f32x4 *>((p.res1 ? p.res1 : A_) + grow + 16 * nb);
            if (BWD) msk1[nb] = *reinterpret_cast<const f32x4 *>((p.mask1 ? p.mask1 : A_) + grow + 16 * nb);
        }
    };

    // ---------------------------------------------------------------- stage 1: six 64-deep tiles (3 taps x 128 channels)
    zero_acc();
    gloadW(0, I0{});
    gloadA(0, I0{});
    gloadA(1, I1{});
    gloadW(1, I1{});
    storeW(0, I0{});
    Planes cur = convertA(ra[0][0], ra[0][1], rok[0]);
    __syncthreads();
    FS_T(0);
    auto tile1 = [&](int S, int buf, auto SET, auto OTHER) {
        constexpr int Q = decltype(SET)::value, O = decltype(OTHER)::value;
        if (S + 2 < (ONE ? 6 : 8)) gloadW(S + 2, SET);   // tiles 6, 7: W2
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(buf, 0, cur);
        Planes nxt = convertA(ra[Q][2], ra[Q][3], rok[Q]);
        weave(std::false_type{});
        use(nxt);
        __builtin_amdgcn_sched_barrier(0);
        FS_T(1);
        gloadA(min(S + 2, 5), SET);          // the tail re-loads the last tile; nobody uses it
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(buf, 1, nxt);
        cur = convertA(ra[O][0], ra[O][1], rok[O]);
        if (!ONE || S < 5) storeW(buf ^ 1, OTHER);
        weave(std::true_type{});
        use(cur);
        __builtin_amdgcn_sched_barrier(0);
        if (S == 4) prefetch1();             // the stage-1 epilogue's operands travel under the last tile
        FS_T(2);
        __syncthreads();
        FS_T(3);
    };
    // (r5) fully unrolled: S is a constant in every copy, so `if (S + 2 < 8) gloadW` and `if (S == 4) prefetch1()` are no branches around loads any more
    // (a branch between a load and its use makes the compiler's wait counts conservative: gemm_tn_split.hpp)
#pragma unroll
    for (int S = 0; S < 6; S += 2) {
        tile1(S, 0, I0{}, I1{});
        tile1(S + 1, 1, I1{}, I0{});
    }
    // now: buffer 0 holds W2 tile 0; every wave is past the barrier

    // ---------------------------------------------------------------- stage-1 epilogue, on registers
    f32x4 h[R2][8];
    {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            f32x4 x = acc[nb];
            if (!BWD || p.res1) x += aux1[nb];
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = act_f(x[e], p.slope);
            } else if (p.mask1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] *= act_grad(msk1[nb][e], p.slope);
            }
            if constexpr (!UNPOOL) {
                if (valid) *reinterpret_cast<f32x4 *>(p.out1 + grow + 16 * nb) = x;
                if (BWD && p.drop.thresh) {      // layer l's dropout mask, replayed (element index = (b*T + t)*128 + n)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] *= drop_mul(p.drop, (uint32_t)(grow + 16 * nb + e));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) h[0][nb][e] = valid ? x[e] : 0.f;
            } else {
                f32x4 u0 = x, u1 = x;
                if (POOL == 3) {
                    const f32x4 y0 = *reinterpret_cast<const f32x4 *>(p.ypre + grow2 + 16 * nb);
                    const f32x4 y1 = *reinterpret_cast<const f32x4 *>(p.ypre + grow2 + 128 + 16 * nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool second = y1[e] > y0[e];
                        u0[e] = second ? 0.f : x[e];
                        u1[e] = second ? x[e] : 0.f;
                    }
                }
                if (valid) {
                    *reinterpret_cast<f32x4 *>(p.out1 + grow2 + 16 * nb) = u0;
                    *reinterpret_cast<f32x4 *>(p.out1 + grow2 + 128 + 16 * nb) = u1;
                    if (trow_raw == Trows_ - 1 && 2 * Trows_ < p.Tfine) {   // odd trailing row of the fine level: no gradient
                        *reinterpret_cast<f32x4 *>(p.out1 + grow2 + 256 + 16 * nb) = f32x4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4 *>(p.out2 + grow2 + 256 + 16 * nb) = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (p.drop.thresh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        u0[e] *= drop_mul(p.drop, (uint32_t)(grow2 + 16 * nb + e));
                        u1[e] *= drop_mul(p.drop, (uint32_t)(grow2 + 128 + 16 * nb + e));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[0][nb][e] = valid ? u0[e] : 0.f;
                    h[R2 - 1][nb][e] = valid ? u1[e] : 0.f;
                }
            }
        }
    }

    FS_T(4);
#if FS_STAMP
    if constexpr (ONE) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
            for (int k = 0; k < 8; ++k) g_fs_stamps[((BWD * 32 + POOL * 4 + ONE * 2 + (NW == 8)) * 8 + wave) * 8 + k] = st_acc[k];
    }
#endif
    if constexpr (ONE) return;

    // ---------------------------------------------------------------- stage 2: two tiles of W2, operand(s) from the registers
    // (its epilogue's operands -- residual rows / ReLU masks, biases -- are requested now and arrive under the MFMAs)
    f32x4 aux2[R2][8], bia2[BWD ? 1 : 8];
#pragma unroll
    for (int r = 0; r < R2; ++r)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            aux2[r][nb] = *reinterpret_cast<const f32x4 *>((BWD ? p.mask2 : p.res2) + grow2 + 128 * r + 16 * nb);
            if (!BWD && r == 0) bia2[nb] = *reinterpret_cast<const f32x4 *>(p.bias2 + 16 * nb + 4 * g);
        }
    f32x4 acc2[R2][8];
#pragma unroll
    for (int r = 0; r < R2; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Planes2 { Planes r[R2]; };
    auto hstep = [&](int s2) {   // 32-deep step s2: channel blocks 2 s2, 2 s2 + 1 in accumulator order
        Planes2 out;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = h[r][2 * s2][e];
                x[4 + e] = h[r][2 * s2 + 1][e];
            }
            out.r[r] = split8(x);
        }
        return out;
    };
    // every W2 fragment is read once and multiplies all R2 row sets
    auto mfma_step2 = [&](int buf, int s, const Planes2 &X) {
        const uint16_t *base = fs_smem + buf * FS_WTILE + s * FS_WSTEP + lane * 8;
        bf16x8 w[8][3];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[nb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (128 * 32) + nb * 512);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                acc2[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][1], X.r[r].pl[1], acc2[r][nb], 0, 0, 0);
                acc2[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][2], X.r[r].pl[0], acc2[r][nb], 0, 0, 0);
                acc2[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][0], X.r[r].pl[2], acc2[r][nb], 0, 0, 0);
                acc2[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][1], X.r[r].pl[0], acc2[r][nb], 0, 0, 0);
                acc2[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][0], X.r[r].pl[1], acc2[r][nb], 0, 0, 0);
                acc2[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nb][0], X.r[r].pl[0], acc2[r][nb], 0, 0, 0);
            }
    };
    auto weave2 = [&](auto WITH_STORES) {
        constexpr bool with_stores = decltype(WITH_STORES)::value;
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int i = 0; i < 48 * R2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 42 * R2 && (i % (2 * R2)) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            if (with_stores && i >= 24 * R2 && (i % ((NQ == 6 ? 4 : 2) * R2)) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };
    auto use2 = [&](const Planes2 &P) {
#pragma unroll
        for (int r = 0; r < R2; ++r) use(P.r[r]);
    };
    Planes2 c2 = hstep(0);          // (W2 tile 0 sits in buffer 0, tile 1 is on its way in register set 1: requested at S = 5)
    __builtin_amdgcn_sched_barrier(0);
    mfma_step2(0, 0, c2);
    Planes2 n2 = hstep(1);
    weave2(std::false_type{});
    use2(n2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step2(0, 1, n2);
    c2 = hstep(2);
    storeW(1, I1{});
    weave2(std::true_type{});
    use2(c2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    mfma_step2(1, 0, c2);
    n2 = hstep(3);
    weave2(std::false_type{});
    use2(n2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step2(1, 1, n2);
    weave2(std::false_type{});
    __builtin_amdgcn_sched_barrier(0);

    FS_T(5);
    // ---------------------------------------------------------------- stage-2 epilogue
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const long gr = grow2 + 128 * r;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            f32x4 x = acc2[r][nb];
            if (!BWD) {
                x += bia2[nb];
                if (p.drop.thresh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] *= drop_mul(p.drop, (uint32_t)(gr + 16 * nb + e));
                }
                x += aux2[r][nb];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] *= act_grad(aux2[r][nb][e], p.slope);
            }
            if (BWD || POOL == 0) {
                if (valid) *reinterpret_cast<f32x4 *>(p.out2 + gr + 16 * nb) = x;
            } else {
                if (POOL == 1 && valid) *reinterpret_cast<f32x4 *>(p.out_pre + gr + 16 * nb) = x;
                // rows 2u, 2u + 1 sit on neighbouring lanes (the low bit of the lane is the low bit of the time step)
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = __shfl_xor(x[e], 1);
                    y[e] = POOL == 1 ? fmaxf(x[e], o) : x[e] + o;
                }
                if ((trow_raw & 1) == 0 && trow_raw + 1 < Trows_)
                    *reinterpret_cast<f32x4 *>(p.out2 + ((long)b * (Trows_ >> 1) + (trow_raw >> 1)) * 128 + 4 * g + 16 * nb) = y;
            }
        }
    }
#if FS_STAMP
    FS_T(6);
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
        for (int k = 0; k < 8; ++k) g_fs_stamps[((BWD * 32 + POOL * 4 + ONE * 2 + (NW == 8)) * 8 + wave) * 8 + k] = st_acc[k];
#endif
}

template <bool BWD, int POOL, bool ONE, int NW>
static hipError_t launch_fs_cfg(const FusedParams &p, const uint16_t *W1img, const uint16_t *W2img, int B, hipStream_t s) {
    auto k = fs_kernel<BWD, POOL, ONE, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, FS_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((p.Trows + 16 * NW - 1) / (16 * NW), B);
    hipLaunchKernelGGL(k, grid, dim3(64 * NW), FS_SMEM_BYTES, s, p.A, W1img, W2img, p.Trows, p.tap_step, p);
    return hipGetLastError();
}
// 128 rows per workgroup (8 waves) where that gives at least one workgroup per CU, 64 rows (4 waves, one per SIMD) below:
// the T/2 level and every pooled boundary (whose stage 1 runs on the coarse level's rows) would leave half of the chip idle
template <bool BWD, int POOL, bool ONE = false>
static hipError_t launch_fs(const FusedParams &p, const uint16_t *W1img, const uint16_t *W2img, int B, hipStream_t s) {
    const int nw = kFsNw ? kFsNw : ((long)B * ((p.Trows + 127) / 128) >= 256 ? 8 : 4);
    if (nw == 8) return launch_fs_cfg<BWD, POOL, ONE, 8>(p, W1img, W2img, B, s);
    return launch_fs_cfg<BWD, POOL, ONE, 4>(p, W1img, W2img, B, s);
}
