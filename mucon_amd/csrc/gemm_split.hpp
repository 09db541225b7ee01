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
// Structure: the tape never touches LDS.  One workgroup = 512 threads = 8 waves computes 128 time steps x 128 channels;
// the waves are 4 row groups x 2 k-groups, each owning 32 time steps x ALL 128 channels for half of every 64-deep k-tile.
// Every tape element is then needed by exactly one wave: its lane loads it straight into the MFMA operand position
// (whole 128-B lines per row and wave, two k-tiles in flight), splits it in registers -- once per element -- and only
// W, shared by the four row groups, goes through LDS: a 48 KB image per k-tile that pack_weights wrote pre-split in
// fragment order [k-step 4][plane 3][lane half 2][channel 128][8 bf16] (conflict-free ds_read_b128, staged by a linear
// copy, double buffered, one barrier per k-tile).  k inside a 16-deep MFMA step is permuted the same way on both
// operands (lane half h, slot j: k = 4h + j for j < 4, 8 + 4h + (j - 4) above) so that a lane's two float4 loads are 32
// contiguous bytes per row and instruction pair.  The ~44 split instructions of an MFMA step are woven between the 24
// MFMAs of the step before it (sched_group_barrier).  The two k-groups' accumulators meet through LDS at the end,
// always summed as group 0 + group 1 (bitwise reproducible, batch independent).
//
// Measured (tools/split_bench.py): 93 us at B=8, T=4096 (one workgroup on every CU) against 153-160 us for the
// f32-MFMA kernel; 60 us when only a quarter of the CUs are busy (B=2) -- a workgroup alone runs its MFMA pipe 68 % busy
// at 2.4 GHz, and with all 256 CUs issuing bf16 MFMAs the clock settles near 2.0 GHz (GRBM cycles / time) and the pipe at
// 52 %.  With the tape served from cache the full launch still takes 89 us: bound by MFMA issue, LDS-read latency and
// power, not yet by HBM (3.1 TB/s).  It wins from 8,192 frames per launch (mucon_hip.hip: g_first_conv_split_rows).
// A first structure that staged both operands as bf16 planes in LDS (tape split by the staging threads, 18 fragment
// reads + 9 VGPR-path stores per wave and 32-deep k-tile) reached 99-108 us: its MFMAs alone took 52 us, its LDS work
// alone 63 us, and the two overlapped badly.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "gemm_nt.hpp"

constexpr int S2_WIMG = 4 * 3 * 2 * 128 * 8;          // bf16 elements of one k-tile's W image (49,152 B)
constexpr int S2_SMEM_BYTES = 2 * S2_WIMG * 2;        // double buffered: 98,304 B (the final exchange reuses it: 64 KB)

// TAPS: the k-tiles walk taps x Kc channels, A rows shifted by (tap - taps/2) * tap_step with zero padding (the dilated
// convolutions and their data gradients); without it the one tap needs no bounds logic (first_conv).  EPI_* as in gemm_nt.hpp.
template <bool EPI_ACT, bool EPI_RES, bool EPI_MASK, bool TAPS>
// (r6: the tape pointer, its strides, the row count and the image pointer once more as LEADING SCALAR arguments -- preloaded into SGPRs with the wave, see cs_kernel)
__global__ __launch_bounds__(512) void nt_split_kernel(const float *__restrict__ A_, const uint16_t *__restrict__ Wimg, const long a_bstride_, const int lda_, const int Trows_,
                                                         const int Kc_, const NtParams p_) {
    NtParams p = p_;
    p.A = A_;
    p.a_bstride = a_bstride_;
    p.lda = lda_;
    p.Trows = Trows_;
    p.Kc = Kc_;
    extern __shared__ __attribute__((aligned(16))) uint16_t s2_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = wave >> 2, wr = wave & 3;   // k-group, row group
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 128;
    const int ktt = p.Kc >> 6;                // 64-deep k-tiles per tap
    const int nS = (TAPS ? p.taps : 1) * ktt; // even (Kc is a multiple of 128)
    const int trow_raw = t0 + wr * 32 + r;
    const int trow = min(trow_raw, p.Trows - 1);   // padding rows re-read a valid row
    const float *a_vid = p.A + (long)b * p.a_bstride + 32 * g + 4 * h;
    const float *a_src = a_vid + (long)trow * p.lda;
    const uint16_t *w_src = Wimg + tid * 8;

    f32x4 ra[2][4];   // two k-tiles of this lane's A values in flight
    bool rok[2] = {true, true};   // TAPS: the staged row exists (else it is zero padding; the load re-read a clamped row)
    u32x4 rws[6];     // this thread's share of the next W image
    auto gloadA = [&](int S, auto SET) {
        constexpr int Q = decltype(SET)::value;
        if (TAPS) {
            const int tap = S / ktt;
            const int ts = trow_raw + (tap - (p.taps >> 1)) * p.tap_step;
            rok[Q] = trow_raw < p.Trows && ts >= 0 && ts < p.Ta;
            const float *src = a_vid + (long)min(max(ts, 0), p.Ta - 1) * p.lda + 64 * (S - tap * ktt);
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[Q][i] = *reinterpret_cast<const f32x4 *>(src + 8 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[Q][i] = *reinterpret_cast<const f32x4 *>(a_src + 64 * S + 8 * i);
        }
    };
    auto gloadW = [&](int S) {
#pragma unroll
        for (int q = 0; q < 6; ++q) rws[q] = *reinterpret_cast<const u32x4 *>(w_src + (long)S * S2_WIMG + q * 4096);
    };
    auto storeW = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 6; ++q) *reinterpret_cast<u32x4 *>(s2_smem + buf * S2_WIMG + tid * 8 + q * 4096) = rws[q];
    };
    struct Planes { bf16x8 pl[3]; };
    // the lane's 8 k of one MFMA step -> (hi, mid, lo) operands
    auto convert = [&](f32x4 v0, f32x4 v1, bool ok) {
        u32x4 hh, mm, ll;
        uint32_t a, bb, c;
        if (TAPS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = ok ? v0[e] : 0.f;
                v1[e] = ok ? v1[e] : 0.f;
            }
        }
        sp_split2(v0[0], v0[1], a, bb, c); hh[0] = a; mm[0] = bb; ll[0] = c;
        sp_split2(v0[2], v0[3], a, bb, c); hh[1] = a; mm[1] = bb; ll[1] = c;
        sp_split2(v1[0], v1[1], a, bb, c); hh[2] = a; mm[2] = bb; ll[2] = c;
        sp_split2(v1[2], v1[3], a, bb, c); hh[3] = a; mm[3] = bb; ll[3] = c;
        Planes P;
        P.pl[0] = __builtin_bit_cast(bf16x8, hh);
        P.pl[1] = __builtin_bit_cast(bf16x8, mm);
        P.pl[2] = __builtin_bit_cast(bf16x8, ll);
        return P;
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // W fragment of (k-step s, plane pl, column block nb): [s][pl][h][n][8]
    const int w_off = (h * 128 + r) * 8;
    auto mfma_step = [&](int buf, int s, const Planes &A) {
        const uint16_t *base = s2_smem + buf * S2_WIMG + s * (3 * 2 * 128 * 8) + w_off;
        bf16x8 w[4][3];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                w[nb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (2 * 128 * 8) + nb * 32 * 8);
            }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {   // small terms first; all six land in the same fp32 accumulator
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[1], w[nb][1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[2], w[nb][0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[0], w[nb][2], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[1], w[nb][0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[0], w[nb][1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.pl[0], w[nb][0], acc[nb], 0, 0, 0);
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    gloadW(0);
    gloadA(0, I0{});
    gloadA(1, I1{});
    storeW(0);
    Planes cur = convert(ra[0][0], ra[0][1], rok[0]);
    __syncthreads();

    // One k-tile: { MFMAs of this group's first step | split of the second step } then { MFMAs of the second step |
    // split of the NEXT tile's first step, W image of the next tile into the other buffer }.  The sched_group_barrier
    // sequences weave the ~44 split instructions of a step between the 24 MFMAs of the step before it.
    auto tile = [&](int S, int buf, auto SET, auto OTHER) {
        constexpr int Q = decltype(SET)::value, O = decltype(OTHER)::value;
        gloadW(min(S + 1, nS - 1));
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(buf, 2 * g, cur);
        Planes nxt = convert(ra[Q][2], ra[Q][3], rok[Q]);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // the first two column blocks' fragments, the rest under their MFMAs
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        gloadA(min(S + 2, nS - 1), SET);   // the tail re-loads the last tile; nobody uses it
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(buf, 2 * g + 1, nxt);
        cur = convert(ra[O][0], ra[O][1], rok[O]);
        storeW(buf ^ 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            if (i >= 12 && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    CLK_BEGIN();
    for (int S = 0; S < nS; S += 2) {
        tile(S, 0, I0{}, I1{});
        tile(S + 1, 1, I1{}, I0{});
    }
    CLK_END(0, blockIdx.y * gridDim.x + blockIdx.x);

    // The k-groups exchange halves: group 0 finishes column blocks 0, 1 and group 1 blocks 2, 3 of their 32 rows;
    // each hands the other two blocks over as [wave][block][reg][lane] floats.  Sum order: group 0 + group 1.
    float *xch = reinterpret_cast<float *>(s2_smem);
    const int partner = wave ^ 4;
    const long vbase = (long)b * p.Trows;
    const bool use_mask = EPI_MASK && (p.mask != nullptr);
    // residual / mask values of the blocks this wave keeps: requested before the exchange, so that their latency runs under it
    float rres[EPI_RES ? 2 : 1][16], rmask[EPI_MASK ? 2 : 1][16];
    if (EPI_RES || EPI_MASK) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = (g == 0 ? j : 2 + j) * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = t0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const long gi = (vbase + min(t, p.Trows - 1)) * 128 + col;
                if (EPI_RES) rres[j][e] = p.res[gi];
                if (EPI_MASK) rmask[j][e] = use_mask ? p.mask[gi] : 1.f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[((wave * 2 + j) * 16 + e) * 64 + lane] = g == 0 ? acc[2 + j][e] : acc[j][e];   // the blocks given away
    }
    __syncthreads();
    auto epilogue = [&](auto FULLT) {
        constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = g == 0 ? j : 2 + j;   // the blocks this wave keeps
            const int col = nb * 32 + r;
            const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float other = xch[((partner * 2 + j) * 16 + e) * 64 + lane];
                const float mine = g == 0 ? acc[j][e] : acc[2 + j][e];
                float x = (g == 0 ? mine + other : other + mine) + bias;
                if (EPI_ACT) x = act_f(x, p.slope);
                if (EPI_RES) x += rres[EPI_RES ? j : 0][e];
                if (EPI_MASK) {
                    if (use_mask) x *= act_grad(rmask[EPI_MASK ? j : 0][e], p.slope);
                }
                const int t = t0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (FULL || t < p.Trows) p.out[(vbase + t) * 128 + col] = x;
            }
        }
    };
    if (t0 + 128 <= p.Trows) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}


// ---------------------------------------------------------------------------------------------------------------------
// The same product on v_mfma_f32_16x16x32_bf16 (round 5: MUCON_MFMA16 bit 0).  Why a second shape: with every CU issuing bf16
// MFMAs on random operands the chip holds its clock down, and the clock it holds depends on the MFMA shape (the 16x16x32 loop
// sustains more FLOP/s than the 32x32x16 loop at equal cycles per FLOP); profiles/r05_mfma_shape.txt has both kernels alternated
// in one process.  Same structure -- the tape never touches LDS, only W is staged, two k-groups of four row groups -- with the
// fragment geometry of the narrower instruction:
//   * lane (r = lane & 15, h = lane >> 4) holds 8 k of row r: k32 = 4h + j (j < 4), 16 + 4h + (j - 4) above, so that the four
//     lanes of a row fetch 64 contiguous bytes per instruction (two float4 per lane and 32-deep step: a whole 128-B line per row);
//   * a wave's 32 rows are two 16-row halves, its half of a 64-deep k-tile is ONE 32-deep step: 2 x 8 x 6 = 96 MFMAs per k-tile
//     (32x32x16: 2 steps x 4 x 6 = 48 of twice the cycles).  A k-tile runs as two phases of four channel blocks each, both row
//     halves in every phase, so each W fragment is read from LDS once (12 per phase, as before); the split of the NEXT k-tile's
//     row half 0 / 1 is woven into phase 0 / 1, i.e. the operand planes of a whole k-tile are complete when it starts;
//   * W is the A operand and the tape the B operand: a lane's four accumulator registers are four consecutive CHANNELS of one
//     frame, so the k-group exchange and the output (and residual / mask reads) move as 16-byte pieces;
//   * W image per k-tile: [k-step 2][plane 3][h 4][channel 128][slot 8] bf16 (sp_split_weights16 / pack_image_tile16), 48 KB as before.
// ---------------------------------------------------------------------------------------------------------------------
template <bool EPI_ACT, bool EPI_RES, bool EPI_MASK, bool TAPS>
// (r6: the tape pointer, its strides, the row count and the image pointer once more as LEADING SCALAR arguments -- preloaded into SGPRs with the wave, see cs_kernel)
__global__ __launch_bounds__(512) void nt_split16_kernel(const float *__restrict__ A_, const uint16_t *__restrict__ Wimg, const long a_bstride_, const int lda_, const int Trows_,
                                                         const int Kc_, const NtParams p_) {
    NtParams p = p_;
    p.A = A_;
    p.a_bstride = a_bstride_;
    p.lda = lda_;
    p.Trows = Trows_;
    p.Kc = Kc_;
    extern __shared__ __attribute__((aligned(16))) uint16_t s2_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = wave >> 2, wr = wave & 3;   // k-group, row group
    const int r = lane & 15, h = lane >> 4;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 128;
    const int ktt = p.Kc >> 6;                // 64-deep k-tiles per tap
    const int nS = (TAPS ? p.taps : 1) * ktt; // even (Kc is a multiple of 128)
    // Every workgroup walks the k-tiles in a ROTATED order, starting at a tile that depends on its place in the grid: with all 256
    // workgroups starting at k = 0 and advancing in step, every tape access of the chip at one moment is a 256-byte piece at the same
    // offset inside an 8 KB frame -- addresses that differ by multiples of 8 KB only.  Depends on blockIdx.x
    // alone among the videos of a batch: a video alone sums in the order it sums inside a batch.
    const int rot = (int)((blockIdx.x * 5u) % (unsigned)nS);
    auto phys = [&](int S) { const int q = S + rot; return q >= nS ? q - nS : q; };   // logical tile S of this workgroup's walk -> k-tile
    const int trow_raw0 = t0 + wr * 32 + r, trow_raw1 = trow_raw0 + 16;
    const float *a_vid = p.A + (long)b * p.a_bstride + 32 * g + 4 * h;
    const float *a_src0 = a_vid + (long)min(trow_raw0, p.Trows - 1) * p.lda;   // padding rows re-read a valid row
    const float *a_src1 = a_vid + (long)min(trow_raw1, p.Trows - 1) * p.lda;
    const uint16_t *w_src = Wimg + tid * 8;

    f32x4 ra[2][4];   // two k-tiles of this lane's A values in flight: [set][2 * row half + piece]
    bool rok[2][2] = {{true, true}, {true, true}};   // TAPS: the staged row exists (else zero padding; the load re-read a clamped row)
    u32x4 rws[6];     // this thread's share of the next W image
    auto gloadA = [&](int Sl, auto SET) {
        constexpr int Q = decltype(SET)::value;
        const int S = phys(Sl);
        if (TAPS) {
            const int tap = S / ktt;
            const int sh = (tap - (p.taps >> 1)) * p.tap_step;
            const int ts0 = trow_raw0 + sh, ts1 = trow_raw1 + sh;
            rok[Q][0] = trow_raw0 < p.Trows && ts0 >= 0 && ts0 < p.Ta;
            rok[Q][1] = trow_raw1 < p.Trows && ts1 >= 0 && ts1 < p.Ta;
            const float *s0 = a_vid + (long)min(max(ts0, 0), p.Ta - 1) * p.lda + 64 * (S - tap * ktt);
            const float *s1 = a_vid + (long)min(max(ts1, 0), p.Ta - 1) * p.lda + 64 * (S - tap * ktt);
            ra[Q][0] = *reinterpret_cast<const f32x4 *>(s0);
            ra[Q][1] = *reinterpret_cast<const f32x4 *>(s0 + 16);
            ra[Q][2] = *reinterpret_cast<const f32x4 *>(s1);
            ra[Q][3] = *reinterpret_cast<const f32x4 *>(s1 + 16);
        } else {
            ra[Q][0] = *reinterpret_cast<const f32x4 *>(a_src0 + 64 * S);
            ra[Q][1] = *reinterpret_cast<const f32x4 *>(a_src0 + 64 * S + 16);
            ra[Q][2] = *reinterpret_cast<const f32x4 *>(a_src1 + 64 * S);
            ra[Q][3] = *reinterpret_cast<const f32x4 *>(a_src1 + 64 * S + 16);
        }
    };
    auto gloadW = [&](int Sl) {
        const int S = phys(Sl);
#pragma unroll
        for (int q = 0; q < 6; ++q) rws[q] = *reinterpret_cast<const u32x4 *>(w_src + (long)S * S2_WIMG + q * 4096);
    };
    auto storeW = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 6; ++q) *reinterpret_cast<u32x4 *>(s2_smem + buf * S2_WIMG + tid * 8 + q * 4096) = rws[q];
    };
    struct Planes { bf16x8 pl[3]; };
    auto convert = [&](f32x4 v0, f32x4 v1, bool ok) {   // the lane's 8 k of one row half -> (hi, mid, lo) operands
        u32x4 hh, mm, ll;
        uint32_t a, bb, c;
        if (TAPS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = ok ? v0[e] : 0.f;
                v1[e] = ok ? v1[e] : 0.f;
            }
        }
        sp_split2(v0[0], v0[1], a, bb, c); hh[0] = a; mm[0] = bb; ll[0] = c;
        sp_split2(v0[2], v0[3], a, bb, c); hh[1] = a; mm[1] = bb; ll[1] = c;
        sp_split2(v1[0], v1[1], a, bb, c); hh[2] = a; mm[2] = bb; ll[2] = c;
        sp_split2(v1[2], v1[3], a, bb, c); hh[3] = a; mm[3] = bb; ll[3] = c;
        Planes P;
        P.pl[0] = __builtin_bit_cast(bf16x8, hh);
        P.pl[1] = __builtin_bit_cast(bf16x8, mm);
        P.pl[2] = __builtin_bit_cast(bf16x8, ll);
        return P;
    };

    f32x4 acc[2][8];   // [row half][channel block]: channels 16 cb + 4 (lane >> 4) + e of frame 16 rho + (lane & 15)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // W fragment of (plane pl, channel block cb) of this group's k-step: [g][pl][h][n][8]
    const int w_off = g * (3 * 4 * 128 * 8) + (h * 128 + r) * 8;
    auto mfma_phase = [&](int buf, auto HALF, const Planes &A0, const Planes &A1) {
        constexpr int CB0 = 4 * decltype(HALF)::value;
        const uint16_t *base = s2_smem + buf * S2_WIMG + w_off + CB0 * 16 * 8;
        bf16x8 w[4][3];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[cb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (4 * 128 * 8) + cb * 16 * 8);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {   // small terms first; all six land in the same fp32 accumulator
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const Planes &A = q ? A1 : A0;
                f32x4 c = acc[q][CB0 + cb];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][1], A.pl[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], A.pl[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][2], A.pl[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], A.pl[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][1], A.pl[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], A.pl[0], c, 0, 0, 0);
                acc[q][CB0 + cb] = c;
            }
        }
    };
    // 48 MFMAs: the first two channel blocks' fragments in front, the other six under the first 12 MFMAs (issued only four 16-cycle
    // MFMAs ahead of its use a fragment read still exposed its latency: 86.2 -> 86.5 ... against 90-92 us for the wide shape; a tile as
    // ONE woven region of 96 MFMAs with the fragments streaming two blocks ahead measured slower, 88.7: r05_mfma_shape.txt), one vector
    // instruction behind every MFMA (the MFMA holds the SIMD's issue for 8 of its 16 cycles: one 4-cycle instruction per gap is what hides)
    auto weave = [&](bool stores) {
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 12 && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            if (stores && i >= 24 && (i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    gloadW(0);
    gloadA(0, I0{});
    gloadA(1, I1{});
    storeW(0);
    Planes cur0 = convert(ra[0][0], ra[0][1], rok[0][0]);
    Planes cur1 = convert(ra[0][2], ra[0][3], rok[0][1]);
    __syncthreads();

    // One k-tile (its operand planes cur0 / cur1 are complete): { tape of tile S + 2 and W image S + 1 requested;
    // MFMAs of channel blocks 0-3 | split of tile S + 1, row half 0 } { MFMAs of channel blocks 4-7 | split of tile S + 1,
    // row half 1; W image S + 1 into the other buffer }.
    auto tile = [&](int S, int buf, auto SET, auto OTHER) {
        constexpr int O = decltype(OTHER)::value;
        gloadW(min(S + 1, nS - 1));
        gloadA(min(S + 2, nS - 1), SET);   // (set SET held tile S: converted already.  The tail re-loads the last tile; nobody uses it)
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(buf, I0{}, cur0, cur1);
        Planes nxt0 = convert(ra[O][0], ra[O][1], rok[O][0]);
        weave(false);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(buf, I1{}, cur0, cur1);
        Planes nxt1 = convert(ra[O][2], ra[O][3], rok[O][1]);
        storeW(buf ^ 1);
        weave(true);
        __builtin_amdgcn_sched_barrier(0);
        cur0 = nxt0;
        cur1 = nxt1;
        __syncthreads();
    };
    CLK_BEGIN();
    for (int S = 0; S < nS; S += 2) {
        tile(S, 0, I0{}, I1{});
        tile(S + 1, 1, I1{}, I0{});
    }
    CLK_END(0, blockIdx.y * gridDim.x + blockIdx.x);

    // The k-groups exchange halves: group 0 finishes channel blocks 0-3 and group 1 blocks 4-7 of their 32 rows; each hands
    // the other four blocks (both row halves) over as [wave][8][lane] float4.  Sum order: group 0 + group 1.
    f32x4 *xch = reinterpret_cast<f32x4 *>(s2_smem);
    const int partner = wave ^ 4;
    const long vbase = (long)b * p.Trows;
    const bool use_mask = EPI_MASK && (p.mask != nullptr);
    const int keep0 = g == 0 ? 0 : 4, give0 = g == 0 ? 4 : 0;
    const int q4 = 4 * h;   // first of this lane's four channels inside a block
    f32x4 rres[EPI_RES ? 8 : 1], rmask[EPI_MASK ? 8 : 1];
    if (EPI_RES || EPI_MASK) {   // residual / mask values of the blocks this wave keeps: requested before the exchange
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + wr * 32 + 16 * q + r;
                const long gi = (vbase + min(t, p.Trows - 1)) * 128 + (keep0 + j) * 16 + q4;
                if (EPI_RES) rres[EPI_RES ? q * 4 + j : 0] = *reinterpret_cast<const f32x4 *>(p.res + gi);
                if (EPI_MASK) rmask[EPI_MASK ? q * 4 + j : 0] = use_mask ? *reinterpret_cast<const f32x4 *>(p.mask + gi) : f32x4{1.f, 1.f, 1.f, 1.f};
            }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) xch[(wave * 8 + q * 4 + j) * 64 + lane] = g == 0 ? acc[q][4 + j] : acc[q][j];   // the blocks given away
    (void)give0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int t = t0 + wr * 32 + 16 * q + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = (keep0 + j) * 16 + q4;
            const f32x4 other = xch[(partner * 8 + q * 4 + j) * 64 + lane];
            const f32x4 mine = g == 0 ? acc[q][j] : acc[q][4 + j];
            f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias = *reinterpret_cast<const f32x4 *>(p.bias + ch);
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = (g == 0 ? mine[e] + other[e] : other[e] + mine[e]) + bias[e];
                if (EPI_ACT) v = act_f(v, p.slope);
                if (EPI_RES) v += rres[EPI_RES ? q * 4 + j : 0][e];
                if (EPI_MASK) {
                    if (use_mask) v *= act_grad(rmask[EPI_MASK ? q * 4 + j : 0][e], p.slope);
                }
                x[e] = v;
            }
            if (t < p.Trows) *reinterpret_cast<f32x4 *>(p.out + (vbase + t) * 128 + ch) = x;
        }
    }
}



extern int g_mfma16;   // mucon_hip.hip (MUCON_MFMA16): bit 0 = this header's launches, bit 1 = the weight gradients on v_mfma_f32_16x16x32_bf16
template <bool EPI_ACT, bool EPI_RES = false, bool EPI_MASK = false, bool TAPS = false>
static hipError_t launch_nt_split(const NtParams &p, const uint16_t *Wimg, int B, hipStream_t s) {
    // (the image at Wimg must be in the order of the shape that reads it: PackArgs::img16 / split_weights_kernel follow g_mfma16 too)
    auto k = (g_mfma16 & 1) ? nt_split16_kernel<EPI_ACT, EPI_RES, EPI_MASK, TAPS> : nt_split_kernel<EPI_ACT, EPI_RES, EPI_MASK, TAPS>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[g_mfma16 & 1]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           S2_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_set[g_mfma16 & 1] = true;
    }
    dim3 grid((p.Trows + 127) / 128, B);
    hipLaunchKernelGGL(k, grid, dim3(512), S2_SMEM_BYTES, s, p.A, Wimg, p.a_bstride, p.lda, p.Trows, p.Kc, p);
    return hipGetLastError();
}
