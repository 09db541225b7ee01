// first_conv forward on the bf16 MFMA with EXACT three-way operand splitting (fp32-grade arithmetic):
//
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
//
// is an identity for every finite fp32 x (3 x 8 significand bits = 24; both subtractions are exact), so
// a*w = sum of nine bf16 x bf16 products, each exact in the MFMA's fp32 accumulator.  Six of them are kept --
// hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- the dropped three (mid*lo, lo*mid, lo*lo) are below 2^-25 |a*w|,
// i.e. under the rounding of the fp32 accumulation itself.  tests/test_gpu_split.py measures the result against the
// float64 oracle next to the f32-MFMA kernel (gemm_nt.hpp): the two errors are of the same size.
//
// Why: first_conv (temporal.py:133, 2048 -> 128 channels over every frame) is 17.2 GFLOP per launch at B=8, T=4096.
// v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s (0.11 ms at 100 %), v_mfma_f32_32x32x16_bf16 at 2.5 PFLOP/s: six bf16
// MFMAs per 32x32x16 block cost 192 cycles against 512 for the eight f32 MFMAs -- the kernel moves from the f32 MFMA
// roof towards the HBM roof (the tape is read once: 268 MB = 34 us at 8 TB/s).
//
// One workgroup = 512 threads = 8 waves (4 along time x 2 along channels, 32 x 64 outputs each) computes 128 time steps
// x 128 channels.  k-tiles of 32: tape rows are loaded coalesced (128 B per row), split in registers by the loading
// thread (once per element), and stored as three bf16 planes in LDS; W arrives pre-split from pack_weights
// (bf16 planes in k-tile order, [D/32][3][128][32]).  LDS rows are 32 bf16 + 8 padding = 80 B so that the 16 lanes of a ds_read_b128 pass
// hit 64 distinct banks (20 r mod 64 is a permutation of the multiples of 4).  Double buffered, two register sets,
// phases pinned with sched_barrier exactly as in gemm_nt.hpp.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "gemm_nt.hpp"

constexpr int SP_BM = 128;
constexpr int SP_ROW = 40;                    // bf16 per LDS row: 32 k + 8 padding
constexpr int SP_PLANE = 128 * SP_ROW;        // one plane of the tape tile or of the W tile (both 128 rows)
constexpr int SP_STAGE = 6 * SP_PLANE;        // planes 0..2: tape hi/mid/lo, 3..5: W hi/mid/lo
constexpr int SP_SMEM_BYTES = 2 * SP_STAGE * 2;   // double buffered: 122,880 B
constexpr int SP_NS = 4;                      // register sets of the global -> LDS staging (k-tiles in flight + 1)

template <bool EPI_ACT, int ABL = 0>
__global__ __launch_bounds__(512) void first_conv_split_kernel(const NtParams p, const uint16_t *__restrict__ Wp, int rot_stride) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sp_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * SP_BM;
    const int K = p.Kc;
    const int nkt = K >> 5;
    const float *Ab = p.A + (long)b * p.a_bstride;

    // staging roles: tape rows (tid >> 3) and (tid >> 3) + 64, four k each; W row tid >> 2, eight k of each plane
    const int arow = tid >> 3, ak = (tid & 7) * 4;
    const int wrow = tid >> 2, wk = (tid & 3) * 8;
    const int tA0 = min(t0 + arow, p.Trows - 1), tA1 = min(t0 + arow + 64, p.Trows - 1);   // padding rows re-read a valid row
    const float *a_src0 = Ab + (long)tA0 * p.lda + ak;
    const float *a_src1 = Ab + (long)tA1 * p.lda + ak;
    const uint16_t *w_src = Wp + tid * 8;   // k-tile image [3][128][32]: thread tid stages 16 consecutive bytes of each plane

    // SP_NS register sets: the loads of k-tile kt+4 are issued while kt is multiplied and kt+1 is split and stored --
    // three k-tiles (48 KB of tape per CU) in flight, which is what ~2 us of HBM latency needs at this kernel's rate
    f32x4 ra[SP_NS][2];
    u32x4 rw[SP_NS][3];
    // workgroups walk k from different starting tiles (the sum is over the same terms): with every workgroup on the same
    // k-tile at the same time, all concurrent tape reads sit at the same offset inside the 8 KB rows
    const int rot = (int)((blockIdx.x * (unsigned)rot_stride) % (unsigned)nkt);
    auto gload = [&](int kt_lin, auto SET) {
        constexpr int S = decltype(SET)::value;
        int kt = kt_lin + rot;
        kt = kt >= nkt ? kt - nkt : kt;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            if (ABL & 0x100) rw[S][pl] = u32x4{(uint32_t)kt, 1u, 2u, 3u};   // no W loads
            else rw[S][pl] = *reinterpret_cast<const u32x4 *>(w_src + ((long)kt * 3 + pl) * 4096);
        }
        if (ABL & 0x200) {   // no tape traffic: every k-tile re-reads tile 0 (L2 / L1 hits)
            ra[S][0] = *reinterpret_cast<const f32x4 *>(a_src0 + (kt & 1) * 32);
            ra[S][1] = *reinterpret_cast<const f32x4 *>(a_src1 + (kt & 1) * 32);
        } else {
            ra[S][0] = *reinterpret_cast<const f32x4 *>(a_src0 + kt * 32);
            ra[S][1] = *reinterpret_cast<const f32x4 *>(a_src1 + kt * 32);
        }
    };
    auto sstore = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        uint16_t *st = sp_smem + buf * SP_STAGE;
        if (ABL & 0x400) {   // no LDS stores: keep the loads alive
            if (ra[S][0][0] == 1.2345f && rw[S][0][0] == 77u) st[tid] = 1;
            return;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(st + (3 + pl) * SP_PLANE + wrow * SP_ROW + wk) = rw[S][pl];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 v = ra[S][q];
            uint32_t h0, m0, l0, h1, m1, l1;
            if (ABL & 0x800) {   // no split arithmetic
                h0 = m0 = l0 = __float_as_uint(v[0]) ^ __float_as_uint(v[1]);
                h1 = m1 = l1 = __float_as_uint(v[2]) ^ __float_as_uint(v[3]);
            } else {
                sp_split2(v[0], v[1], h0, m0, l0);
                sp_split2(v[2], v[3], h1, m1, l1);
            }
            const u32x2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
            uint16_t *row = st + (arow + 64 * q) * SP_ROW + ak;
            *reinterpret_cast<u32x2 *>(row) = h;
            *reinterpret_cast<u32x2 *>(row + SP_PLANE) = m;
            *reinterpret_cast<u32x2 *>(row + 2 * SP_PLANE) = l;
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // MFMA operand lane (r = lane & 31, h = lane >> 5) holds k = 8h .. 8h+7 of the 16-deep step
    const int a_off = (wr * 32 + (lane & 31)) * SP_ROW + 8 * (lane >> 5);
    const int b_off = 3 * SP_PLANE + (wc * 64 + (lane & 31)) * SP_ROW + 8 * (lane >> 5);

    auto compute = [&](int cur) {
        const uint16_t *st = sp_smem + cur * SP_STAGE;
        // all 18 fragment reads of the k-tile first (72 registers), then the 24 MFMAs: with the reads interleaved hipcc
        // puts lgkmcnt(0) waits between the MFMAs and the LDS latency shows three times per k-step
        bf16x8 a[2][3], w[2][3][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if (ABL & 0x4000) {   // no LDS reads
                    a[s][pl] = __builtin_bit_cast(bf16x8, u32x4{(uint32_t)cur, 1u, 2u, 3u});
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) w[s][pl][nb] = __builtin_bit_cast(bf16x8, u32x4{(uint32_t)cur + nb, 5u, 6u, 7u});
                    continue;
                }
                a[s][pl] = *reinterpret_cast<const bf16x8 *>(st + a_off + pl * SP_PLANE + 16 * s);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    w[s][pl][nb] = *reinterpret_cast<const bf16x8 *>(st + b_off + pl * SP_PLANE + nb * 32 * SP_ROW + 16 * s);
            }
        if (!(ABL & 8)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (ABL & 0x1000) {   // no MFMA: keep the LDS reads alive
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[nb][pl] += (float)a[s][pl][0] * (float)w[s][pl][nb][1];
                continue;
            }
            // small terms first; all six land in the same fp32 accumulator
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], w[s][1][nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2], w[s][0][nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], w[s][2][nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], w[s][0][nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], w[s][1][nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], w[s][0][nb], acc[nb], 0, 0, 0);
        }
    };

    // Waves w and w + 4 share a SIMD.  Per k-tile one of them splits and stores the next tile first and multiplies
    // after, the other multiplies first: the VALU / LDS-store phase of each runs in the shadow of the other's MFMAs.
    const bool store_first = wave < 4;
    auto step = [&](int kt, int buf, auto LOADSET, auto STORESET) {
        gload(min(kt + SP_NS, nkt - 1), LOADSET);   // the tail re-loads the last tile; nobody reads it
        __builtin_amdgcn_sched_barrier(0);
        if (ABL & 8) {   // one order; the split arithmetic and the LDS stores of the next tile are woven between the MFMAs
            compute(buf);
            sstore(buf ^ 1, STORESET);
            __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);   // the 18 fragment reads
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two VALU (split arithmetic)
                if (i >= 8 && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // an LDS store now and then
            }
        } else if (store_first) {
            sstore(buf ^ 1, STORESET);
            __builtin_amdgcn_sched_barrier(0);
            compute(buf);
        } else {
            compute(buf);
            __builtin_amdgcn_sched_barrier(0);
            sstore(buf ^ 1, STORESET);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 0x2000)) __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    static_assert(SP_NS == 4, "the k loop is unrolled over four register sets");
    gload(0, I0{});
    gload(1, I1{});
    gload(2, I2{});
    gload(3, I3{});
    sstore(0, I0{});
    __syncthreads();
    // nkt is a multiple of 4 (D is a multiple of 128).  k-tile kt lives in set kt & 3 and LDS buffer kt & 1.
    for (int kt = 0; kt < nkt; kt += 4) {
        step(kt, 0, I0{}, I1{});
        step(kt + 1, 1, I1{}, I2{});
        step(kt + 2, 0, I2{}, I3{});
        step(kt + 3, 1, I3{}, I0{});
    }

    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const long vbase = (long)b * p.Trows;
    auto epilogue = [&](auto FULLT) {
        constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = wc * 64 + nb * 32 + (lane & 31);
            const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int t = t0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                float x = acc[nb][reg] + bias;
                if (EPI_ACT) x = act_f(x, p.slope);
                if (FULL || t < p.Trows) p.out[(vbase + t) * 128 + col] = x;
            }
        }
    };
    if (t0 + SP_BM <= p.Trows) epilogue(std::true_type{});   // full tiles: straight-line stores
    else epilogue(std::false_type{});
}

extern int g_split_rot;   // k-tile rotation stride between neighbouring workgroups (MUCON_SPLIT_ROT)
template <bool EPI_ACT, int ABL = 0>
static hipError_t launch_first_conv_split(const NtParams &p, const uint16_t *Wp, int B, hipStream_t s) {
    auto k = first_conv_split_kernel<EPI_ACT, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           SP_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((p.Trows + SP_BM - 1) / SP_BM, B);
    hipLaunchKernelGGL(k, grid, dim3(512), SP_SMEM_BYTES, s, p, Wp, g_split_rot);
    return hipGetLastError();
}
