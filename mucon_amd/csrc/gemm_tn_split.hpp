// Weight gradients (the TN products of gemm_tn.hpp) on the bf16 MFMA with EXACT three-way operand splitting -- the
// arithmetic of gemm_split.hpp (x = hi + mid + lo, six of the nine partial products, fp32 accumulate: fp32-grade) applied to
//
//     dW[n][c] = sum_t G[t][n] * X[t][c]          (autograd of nn.Conv1d w.r.t. its weight: temporal.py:23-32, :133, :145)
//
// Both operands are time-major in HBM and the reduction runs over time, so both MFMA fragments want 8 consecutive TIME steps
// of one column per lane: lane (r, h) of v_mfma_f32_32x32x16_bf16 holds A[row r][k = 8h + j] / B[k = 8h + j][col r].  With
// k = time that is eight 4-byte loads whose 32 lanes sit on 32 adjacent columns (whole 128-B lines per row) -- the transpose
// is done by the load addresses, nothing is shuffled.
//
// One workgroup = 8 waves = 256 weight columns x all 128 output channels over one time chunk.
//   * X (the tape / layer input): wave (half, cg) owns 32 columns.  Every X element belongs to exactly ONE wave, which loads
//     it straight into the MFMA operand position (two 32-step tiles in flight), splits it in registers, once.  No LDS.
//   * G (the gradient rows, shared by all column groups): staged by all threads (thread = (channel, 8 time steps)), split once
//     per workgroup, written to LDS as a fragment-ordered bf16 image [step 2][plane 3][lane half 2][channel 128][8 steps]
//     (conflict-free ds_read_b128 / ds_write_b64), double buffered, one barrier per 32 time steps.
//   * the two 128-column halves of a workgroup are two k-chunks of the job (two taps of a dilated conv, two column blocks
//     of the tape).  Where they need different gradient operands -- the last tap and the conv_1x1 chunk of a residual layer --
//     each half stages its own image (TWO_G); the conv_1x1 gradient's dropout mask is replayed at staging.
// Bias gradients are the exact fp32 column sums of the staged G values (they never see bf16).
//
// Interior tiles run a mask-free body; tiles that touch a video edge (zero padding of a tap, a partial last tile) or need the
// non-linearity on X take the general body.  The split arithmetic of a step is woven between the MFMAs of the step before.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "dispatch.hpp"
#include "gemm_tn.hpp"

extern int g_mfma16;       // mucon_hip.hip (MUCON_MFMA16): bit 1 = this header's launch on v_mfma_f32_16x16x32_bf16
extern int g_ts_stagger;   // mucon_hip.hip (MUCON_TS_STAGGER): single-image jobs with time chunks of at least this many steps take the staggered schedule ts_body_st (0: none)

#ifndef TS_ABL
#define TS_ABL 0   // tools/ts_ablate.hip: 1 no MFMAs, 2 no X split, 4 no G split / LDS stores, 8 no global loads in the loop (timing only)
#endif
#ifndef TS_STAMP
#define TS_STAMP 0   // tools/ts_ablate.hip: 1 = s_memtime stamps around the phases of a tile (block 0 publishes per-wave sums; timing builds only)
#endif
#if TS_STAMP
__device__ long long g_ts_stamps[8 * 8];
#define TS_T(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define TS_T(k) do { } while (0)
#endif
constexpr int TS_IMG = 2 * 3 * 2 * 128 * 8;             // bf16 elements of one 32-step G image (24,576 B)
constexpr int TS_XT_FLOATS = 32 * 32;                   // a wave's X tile [32 time steps][32 columns] (4 KB), transposed through LDS
constexpr int TS_SMEM_BYTES = 2 * 2 * TS_IMG * 2 + 8 * TS_XT_FLOATS * 4;   // two buffers x two images + eight X tiles: 131,072 B

template <bool TWO_G, bool DROP, class AfterLoop>
__device__ __forceinline__ void ts_body(const TnParams &p, const int kc2, const int mc, const bool dual, const bool x0_act,
                                        uint16_t *smem, const int item, AfterLoop &&after_loop) {
    constexpr int NU = TWO_G ? 2 : 1;   // staging units (8 time steps of one channel) per thread and tile
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave >> 2, cg = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int b = mc / p.chunks_per_video;
    const int tbeg = (mc - b * p.chunks_per_video) * p.MC;
    const int tend = min(tbeg + p.MC, p.Trows);
    const int ntiles = (tend - tbeg + 31) >> 5;
    const int last = ntiles - 1;
    const int nch = p.nk0 + (dual ? 1 : 0);
    const int kc_raw = 2 * kc2 + hf;
    const bool active = kc_raw < nch;           // an odd chunk count leaves the last workgroup's second half without columns:
    const int kc = active ? kc_raw : 2 * kc2;   // it repeats the first half's work and writes nothing
    const bool second = dual && kc >= p.nk0;
    const int xoff = (!second && p.taps == 3) ? (kc - 1) * p.tap_step : 0;
    const int xcol = (second || p.taps == 3) ? 0 : kc * 128;
    const int ldx = second ? 128 : p.ldx;
    const int Tx = second ? p.Trows : p.Tx;
    const float *Xu = second ? p.X1 + (long)b * p.Trows * 128 : p.X0 + (long)b * p.x_bstride + xcol;   // wave-uniform
    // X is fetched in 16-byte pieces: lane -> (row lane >> 3 of 8, columns 4 (lane & 7) .. + 3), four instructions per 32-step tile
    // (a 4-byte load per element costs the memory pipeline as much per instruction: 16 of them per tile were its bottleneck)
    const int xrow = lane >> 3, xc4 = (lane & 7) * 4;
    const uint32_t x_lane = (uint32_t)(xrow * ldx + cg * 32 + xc4) * 4u;                              // per-lane byte offset
    float *xT = reinterpret_cast<float *>(smem + 2 * 2 * TS_IMG) + wave * TS_XT_FLOATS;               // this wave's transposition tile

    // staging role: SAME image -> unit (s, h) = (wave >> 2, (wave >> 1) & 1); TWO_G -> image wave >> 2, s = (wave >> 1) & 1, units h = 0, 1
    const int sn = tid & 127;
    const int s_hi = wave >> 2, s_lo = (wave >> 1) & 1;
    const int s_img = TWO_G ? s_hi : 0;
    const int s_s = TWO_G ? s_lo : s_hi;
    const float *Yu = ((TWO_G && s_img) ? p.Y1 : p.Y0) + (long)b * p.Trows * 128;   // wave-uniform
    const uint32_t y_lane = (uint32_t)sn * 4u;
    auto unit_h = [&](int u) { return TWO_G ? u : s_lo; };
    // dropout replay for image 1 only, branch-free (a branch would cut the woven schedule): image 0 keeps every element at scale 1
    DropCfg dcfg = p.drop;
    dcfg.thresh = s_img ? dcfg.thresh : 0u;
    dcfg.scale = s_img ? dcfg.scale : 1.f;
    // uniform base + 32-bit per-lane byte offset: the scalar-base form of global_load (no 64-bit vector address arithmetic per load)
    auto ld_su = [](const float *ubase, uint32_t lane_bytes) {
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + lane_bytes);
    };

    f32x4 rx[2][4];          // X: [set][8-row group], two tiles in flight
    float rawT[8];           // the eight time steps of this lane's column for the next MFMA step (read back from the X tile)
    float rgA[NU][4], rgB[NU][4];   // G: time slots 0-3 / 4-7 of the next image
    float bsum[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) bsum[u] = 0.f;

    // A tile is interior when none of its 32 rows needs a mask: inside the chunk (G) and, for this wave's tap, inside the video (X).
    // Interior tiles are loaded with wave-uniform row addresses and enter the MFMA body as they are; the others are loaded
    // from clamped rows and masked in their registers by the (rare) fix-up branches in front of the woven phases.
    auto g_int = [&](int tile) { return tbeg + tile * 32 + 32 <= tend; };
    auto x_int = [&](int tile) {
        const int t0 = tbeg + tile * 32;
        return !x0_act && t0 + 32 <= tend && t0 + xoff >= 0 && t0 + 31 + xoff < Tx;
    };
#ifndef TS_BRANCHFREE
#define TS_BRANCHFREE 1   // 0: round 4's loads (interior / edge variants of the X loads and the clamp of the G rows as branches inside the tile loop)
#endif
#if TS_BRANCHFREE
    // (r5) no branch between a load and its use inside the tile loop: see ts_body_st -- with the branches the compiler's wait counts made every X tile wait for
    // the loads issued one phase ago
    const int ldx4 = ldx * 4, xcb = (cg * 32 + xc4) * 4;
    auto gloadX = [&](int tile, auto SET) {
        constexpr int Q = decltype(SET)::value;
        const int row0 = tbeg + tile * 32 + xoff;   // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ts = min(max(row0 + 8 * i + xrow, 0), Tx - 1);
            rx[Q][i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(Xu) + (uint32_t)(ts * ldx4 + xcb));
        }
    };
#else
    auto gloadX = [&](int tile, auto SET) {
        constexpr int Q = decltype(SET)::value;
        if (x_int(tile)) {
            const float *ub = Xu + (long)(tbeg + tile * 32 + xoff) * ldx;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rx[Q][i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(ub + (long)(8 * i) * ldx) + x_lane);
        } else {
            const int row0 = tbeg + tile * 32 + xrow + xoff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ts = min(max(row0 + 8 * i, 0), Tx - 1);
                rx[Q][i] = *reinterpret_cast<const f32x4 *>(Xu + (long)ts * ldx + cg * 32 + xc4);
            }
        }
    };
#endif
    // a tile's 32 x 32 values go through the wave's own LDS tile: written as they were loaded (rows), read back by column into
    // the MFMA operand order (lane (r, h): column r, time steps 8h .. 8h + 7 of step s).  Wave-private: program order is all it needs.
    auto stageX = [&](auto SET) {
        constexpr int Q = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(xT + (8 * i + xrow) * 32 + xc4) = rx[Q][i];
    };
    auto readX = [&](int s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rawT[j] = xT[(16 * s + 8 * h + j) * 32 + r];
    };
    auto fixX = [&](float (&raw)[8], int tile, int s) {   // non-linearity of the last_conv job, zero padding, chunk end
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = tbeg + tile * 32 + 16 * s + 8 * h + j;
            const int ts = t + xoff;
            float x = raw[j];
            if (x0_act) x = act_f(x, p.slope);
            raw[j] = (t < tend && ts >= 0 && ts < Tx) ? x : 0.f;
        }
    };
    auto gloadG = [&](int tile, auto HALF) {
        constexpr int HB = decltype(HALF)::value;
        const bool inner = TS_BRANCHFREE ? false : g_int(tile);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;   // wave-uniform
                if (!inner) t = min(t, p.Trows - 1);
                const float v = ld_su(Yu + (long)t * 128, y_lane);
                if constexpr (HB) rgB[u][jj] = v;
                else rgA[u][jj] = v;
            }
    };
    auto fixG = [&](int tile, auto HALF) {   // rows past the chunk end are zero
        constexpr int HB = decltype(HALF)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;
                if constexpr (HB) rgB[u][jj] = t < tend ? rgB[u][jj] : 0.f;
                else rgA[u][jj] = t < tend ? rgA[u][jj] : 0.f;
            }
    };
    // four time slots of every unit: dropout replay, bias sums, exact split, three 8-byte LDS stores
    // (bw = 0 for the image past the chunk's last tile, which the tail of the pipeline builds from a re-load and nobody reads)
    auto splitstoreG = [&](int tile, int buf, auto HALF, float bw) {
        constexpr int HB = decltype(HALF)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                v[jj] = HB ? rgB[u][jj] : rgA[u][jj];
                if (DROP) {
                    const int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;
                    v[jj] *= drop_mul(dcfg, (uint32_t)(b * p.Trows + t) * 128u + (uint32_t)sn);
                }
                bsum[u] = fmaf(v[jj], bw, bsum[u]);
            }
            uint32_t a0, m0, l0, a1, m1, l1;
            sp_split2(v[0], v[1], a0, m0, l0);
            sp_split2(v[2], v[3], a1, m1, l1);
            uint16_t *dst = smem + (buf * 2 + s_img) * TS_IMG + (((s_s * 3) * 2 + unit_h(u)) * 128 + sn) * 8 + 4 * HB;
            *reinterpret_cast<u32x2 *>(dst) = u32x2{a0, a1};
            *reinterpret_cast<u32x2 *>(dst + 2 * 128 * 8) = u32x2{m0, m1};
            *reinterpret_cast<u32x2 *>(dst + 4 * 128 * 8) = u32x2{l0, l1};
        }
    };
    struct Planes { bf16x8 pl[3]; };
    auto convertX = [&](const float (&x)[8]) {
        u32x4 hh, mm, ll;
        uint32_t a, bb, c;
        if (TS_ABL & 2) {
            Planes P;
            hh = u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
            mm = u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
            P.pl[0] = __builtin_bit_cast(bf16x8, hh);
            P.pl[1] = __builtin_bit_cast(bf16x8, mm);
            P.pl[2] = __builtin_bit_cast(bf16x8, hh);
            return P;
        }
        sp_split2(x[0], x[1], a, bb, c); hh[0] = a; mm[0] = bb; ll[0] = c;
        sp_split2(x[2], x[3], a, bb, c); hh[1] = a; mm[1] = bb; ll[1] = c;
        sp_split2(x[4], x[5], a, bb, c); hh[2] = a; mm[2] = bb; ll[2] = c;
        sp_split2(x[6], x[7], a, bb, c); hh[3] = a; mm[3] = bb; ll[3] = c;
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

    // G fragment (step s, plane pl, channel block nb): image[s][pl][h][nb*32 + r][8]
    const int g_off = (TWO_G ? hf : 0) * TS_IMG + (h * 128 + r) * 8;
    auto mfma_step = [&](int buf, int s, const Planes &X) {
        const uint16_t *base = smem + buf * 2 * TS_IMG + s * (3 * 2 * 128 * 8) + g_off;
        bf16x8 w[4][3];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[nb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (2 * 128 * 8) + nb * 32 * 8);
        if (TS_ABL & 1) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, w[nb][pl])));
            return;
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {   // small terms first; all six land in the same fp32 accumulator
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][1], X.pl[1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][2], X.pl[0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][0], X.pl[2], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][1], X.pl[0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][0], X.pl[1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][0], X.pl[0], acc[nb], 0, 0, 0);
        }
    };
    // 24 MFMAs with the fragment reads of the first column blocks in front, VPM vector instructions behind every MFMA and the
    // LDS stores (which need the split results) in the second half
    constexpr int VPM = DROP ? 7 : (TWO_G ? 4 : 3);
    auto weave = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            if (i >= 12 && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };

    auto pin = [](auto &arr) {
#pragma unroll
        for (auto &v : arr) asm volatile("" : "+v"(v));
    };
    auto use = [](const Planes &P) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, P.pl[pl])));
    };
    Planes cur;
    {   // prologue: image of tile 0, X of tiles 0 .. 2 requested, tile 0 staged, first half of image 1, operand of step (0, 0)
        gloadG(0, I0{});
        gloadG(0, I1{});
        gloadX(0, I0{});
        gloadX(min(1, last), I1{});
        if (!g_int(0)) {
            fixG(0, I0{});
            fixG(0, I1{});
        }
        splitstoreG(0, 0, I0{}, 1.f);
        splitstoreG(0, 0, I1{}, 1.f);
        gloadG(min(1, last), I0{});   // (r5: requested with the loads above instead -- neutral, 0.6982 against 0.6987 ms per step; not kept)
        stageX(I0{});
        gloadX(min(2, last), I0{});
        readX(0);
        if (!x_int(0)) fixX(rawT, 0, 0);
        cur = convertX(rawT);
        __syncthreads();
    }

    // tile mt (image in buffer Q; the wave's X tile holds tile mt; set O holds tile mt+1, set Q tile mt+2 on its way):
    //   { second half of image mt+1 requested; X step 1 read back }
    //   { MFMAs of step 0 | split of X step 1, first half of image mt+1 -> buffer O }
    //   { X tile <- tile mt+1 (set O), set O <- tile mt+3 requested; first half of image mt+2 requested; X (mt+1, step 0) read back }
    //   { MFMAs of step 1 | split of X (mt+1, step 0), second half of image mt+1 }
#if TS_STAMP
    long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    long long st_prev = __builtin_amdgcn_s_memtime();
#endif
    auto tile = [&](int mt, auto SET, auto OTHER) {
        constexpr int Q = decltype(SET)::value, O = decltype(OTHER)::value;
        (void)Q;
        TS_T(5);
        const int n1 = min(mt + 1, last), n2 = min(mt + 2, last), n3 = min(mt + 3, last);
        const float bw = mt < last ? 1.f : 0.f;
        if (!(TS_ABL & 8)) gloadG(n1, I1{});
        readX(1);
        if (!x_int(mt)) fixX(rawT, mt, 1);
        if (!g_int(n1)) fixG(n1, I0{});
        __builtin_amdgcn_sched_barrier(0);
        TS_T(0);
        pin(rawT);   // (keeps the splits below in this block: without it they are duplicated into the fix-up branches, outside the weave)
#pragma unroll
        for (int u = 0; u < NU; ++u) pin(rgA[u]);
        mfma_step(Q, 0, cur);
        Planes nxt = convertX(rawT);
        if (!(TS_ABL & 4)) splitstoreG(n1, O, I0{}, bw);
        weave();
        use(nxt);   // (a use inside the phase: otherwise the split is sunk behind the branches below, out of the weave)
        __builtin_amdgcn_sched_barrier(0);
        TS_T(1);
        stageX(OTHER);
        if (!(TS_ABL & 8)) {
            gloadG(n2, I0{});
            gloadX(n3, OTHER);
        }
        readX(0);
        if (!x_int(n1)) fixX(rawT, n1, 0);
        if (!g_int(n1)) fixG(n1, I1{});
        __builtin_amdgcn_sched_barrier(0);
        TS_T(2);
        pin(rawT);
#pragma unroll
        for (int u = 0; u < NU; ++u) pin(rgB[u]);
        mfma_step(Q, 1, nxt);
        cur = convertX(rawT);
        if (!(TS_ABL & 4)) splitstoreG(n1, O, I1{}, bw);
        weave();
        use(cur);
        __builtin_amdgcn_sched_barrier(0);
        TS_T(3);
        __syncthreads();
        TS_T(4);
    };
    CLK_BEGIN();
    for (int mt = 0; mt < ntiles; mt += 2) {
        tile(mt, I0{}, I1{});
        if (mt + 1 < ntiles) tile(mt + 1, I1{}, I0{});
    }
    CLK_END(1, item);

#if TS_STAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 6; ++k) g_ts_stamps[wave * 8 + k] = st_acc[k];
#endif
    if (active) {
        float *slab = p.slabs + (long)mc * 128 * p.Ktot + kc_raw * 128 + cg * 32 + r;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = nb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                slab[(long)row * p.Ktot] = acc[nb][reg];
            }
    }
    after_loop();   // (the persistent launch draws its next item here: the ticket's round trip passes while the slab stores drain)
    // bias gradients = column sums of the staged gradient rows: image 0 (Y0) from the workgroup that owns chunk 0, image 1
    // (Y1, dropout replayed) from the TWO_G workgroup.  Fixed order: a thread's own time slots, then the units.
    const bool bias0 = p.bias_slabs != nullptr && kc2 == 0;
    const bool bias1 = TWO_G && p.bias_slabs != nullptr;
    if (bias0 || bias1) {   // workgroup-uniform
        float *red = reinterpret_cast<float *>(smem);   // the images are dead after the loop's last barrier
        float own = bsum[0];
        if (TWO_G) own += bsum[NU - 1];
        red[(s_hi * 2 + s_lo) * 128 + sn] = own;
        __syncthreads();
        if (TWO_G) {
            if (tid < 128 && bias0) p.bias_slabs[(long)mc * 256 + tid] = red[tid] + red[128 + tid];
            if (tid >= 128 && tid < 256 && bias1) p.bias_slabs[(long)mc * 256 + tid] = red[256 + sn] + red[384 + sn];
        } else if (tid < 128 && bias0) {
            p.bias_slabs[(long)mc * 256 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The staggered schedule (round 5; MUCON_TS_STAGGER, ts_batched_kernel<2>): ts_body's workgroup, operands, images, staging roles and
// arithmetic with the tile re-ordered into blocks so that the two waves of a SIMD never want the same unit -- see the comment at the
// tile loop below.
// ---------------------------------------------------------------------------------------------------------------------
template <bool TWO_G, bool DROP, class AfterLoop>
__device__ __forceinline__ void ts_body_st(const TnParams &p, const int kc2, const int mc, const bool dual, const bool x0_act,
                                        uint16_t *smem, const int item, AfterLoop &&after_loop) {
    constexpr int NU = TWO_G ? 2 : 1;   // staging units (8 time steps of one channel) per thread and tile
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave >> 2, cg = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int b = mc / p.chunks_per_video;
    const int tbeg = (mc - b * p.chunks_per_video) * p.MC;
    const int tend = min(tbeg + p.MC, p.Trows);
    const int ntiles = (tend - tbeg + 31) >> 5;
    const int last = ntiles - 1;
    const int nch = p.nk0 + (dual ? 1 : 0);
    const int kc_raw = 2 * kc2 + hf;
    const bool active = kc_raw < nch;           // an odd chunk count leaves the last workgroup's second half without columns:
    const int kc = active ? kc_raw : 2 * kc2;   // it repeats the first half's work and writes nothing
    const bool second = dual && kc >= p.nk0;
    const int xoff = (!second && p.taps == 3) ? (kc - 1) * p.tap_step : 0;
    const int xcol = (second || p.taps == 3) ? 0 : kc * 128;
    const int ldx = second ? 128 : p.ldx;
    const int Tx = second ? p.Trows : p.Tx;
    const float *Xu = second ? p.X1 + (long)b * p.Trows * 128 : p.X0 + (long)b * p.x_bstride + xcol;   // wave-uniform
    // X is fetched in 16-byte pieces: lane -> (row lane >> 3 of 8, columns 4 (lane & 7) .. + 3), four instructions per 32-step tile
    // (a 4-byte load per element costs the memory pipeline as much per instruction: 16 of them per tile were its bottleneck)
    const int xrow = lane >> 3, xc4 = (lane & 7) * 4;
    const uint32_t x_lane = (uint32_t)(xrow * ldx + cg * 32 + xc4) * 4u;                              // per-lane byte offset
    float *xT = reinterpret_cast<float *>(smem + 2 * 2 * TS_IMG) + wave * TS_XT_FLOATS;               // this wave's transposition tile

    // staging role: SAME image -> unit (s, h) = (wave >> 2, (wave >> 1) & 1); TWO_G -> image wave >> 2, s = (wave >> 1) & 1, units h = 0, 1
    const int sn = tid & 127;
    const int s_hi = wave >> 2, s_lo = (wave >> 1) & 1;
    const int s_img = TWO_G ? s_hi : 0;
    const int s_s = TWO_G ? s_lo : s_hi;
    const float *Yu = ((TWO_G && s_img) ? p.Y1 : p.Y0) + (long)b * p.Trows * 128;   // wave-uniform
    const uint32_t y_lane = (uint32_t)sn * 4u;
    auto unit_h = [&](int u) { return TWO_G ? u : s_lo; };
    // dropout replay for image 1 only, branch-free (a branch would cut the woven schedule): image 0 keeps every element at scale 1
    DropCfg dcfg = p.drop;
    dcfg.thresh = s_img ? dcfg.thresh : 0u;
    dcfg.scale = s_img ? dcfg.scale : 1.f;
    // uniform base + 32-bit per-lane byte offset: the scalar-base form of global_load (no 64-bit vector address arithmetic per load)
    auto ld_su = [](const float *ubase, uint32_t lane_bytes) {
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + lane_bytes);
    };

    constexpr int NXS = TWO_G ? 1 : 2;   // X register sets: two tiles in flight; one for the workgroups that stage two gradient images (register budget)
    f32x4 rx[NXS][4];        // X: [set][8-row group]
    float rawT[8], rawU[8];  // the eight time steps of this lane's column for the two MFMA steps of a tile (read back from the X tile)
    float rgA[NU][4], rgB[NU][4];   // G: time slots 0-3 / 4-7 of the next image
    float bsum[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) bsum[u] = 0.f;

    // A tile is interior when none of its 32 rows needs a mask: inside the chunk (G) and, for this wave's tap, inside the video (X).
    // Interior tiles are loaded with wave-uniform row addresses and enter the MFMA body as they are; the others are loaded
    // from clamped rows and masked in their registers by the (rare) fix-up branches in front of the woven phases.
    auto g_int = [&](int tile) { return tbeg + tile * 32 + 32 <= tend; };
    auto x_int = [&](int tile) {
        const int t0 = tbeg + tile * 32;
        return !x0_act && t0 + 32 <= tend && t0 + xoff >= 0 && t0 + 31 + xoff < Tx;
    };
    // X rows are ALWAYS fetched from per-lane clamped rows with 32-bit offsets from the wave-uniform video base, and G rows from clamped
    // wave-uniform rows: no branch between a load and its use anywhere in the tile loop.  (With the interior / edge choice as a branch
    // inside the loop the compiler's wait-count pass merged the two paths conservatively and made every X tile wait for the loads issued
    // one block ago instead of two -- one tile of tape in flight per wave, 2.2 TB/s, the staging blocks waiting ~1,500 cycles per tile.)
    const int ldx4 = ldx * 4, xcb = (cg * 32 + xc4) * 4;
    auto gloadX = [&](int tile, auto SET) {
        constexpr int Q = decltype(SET)::value;
        const int row0 = tbeg + tile * 32 + xoff;   // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ts = min(max(row0 + 8 * i + xrow, 0), Tx - 1);
            rx[Q][i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(Xu) + (uint32_t)(ts * ldx4 + xcb));
        }
    };
    // a tile's 32 x 32 values go through the wave's own LDS tile: written as they were loaded (rows), read back by column into
    // the MFMA operand order (lane (r, h): column r, time steps 8h .. 8h + 7 of step s).  Wave-private: program order is all it needs.
    auto stageX = [&](auto SET) {
        constexpr int Q = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(xT + (8 * i + xrow) * 32 + xc4) = rx[Q][i];
    };
    auto readX = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) rawT[j] = xT[(8 * h + j) * 32 + r];
#pragma unroll
        for (int j = 0; j < 8; ++j) rawU[j] = xT[(16 + 8 * h + j) * 32 + r];
    };
    auto fixX = [&](float (&raw)[8], int tile, int s) {   // non-linearity of the last_conv job, zero padding, chunk end
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = tbeg + tile * 32 + 16 * s + 8 * h + j;
            const int ts = t + xoff;
            float x = raw[j];
            if (x0_act) x = act_f(x, p.slope);
            raw[j] = (t < tend && ts >= 0 && ts < Tx) ? x : 0.f;
        }
    };
    auto gloadG = [&](int tile, auto HALF) {
        constexpr int HB = decltype(HALF)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int t = min(tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj, p.Trows - 1);   // wave-uniform
                const float v = ld_su(Yu + (long)t * 128, y_lane);
                if constexpr (HB) rgB[u][jj] = v;
                else rgA[u][jj] = v;
            }
    };
    auto fixG = [&](int tile, auto HALF) {   // rows past the chunk end are zero
        constexpr int HB = decltype(HALF)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;
                if constexpr (HB) rgB[u][jj] = t < tend ? rgB[u][jj] : 0.f;
                else rgA[u][jj] = t < tend ? rgA[u][jj] : 0.f;
            }
    };
    // four time slots of every unit: dropout replay, bias sums, exact split, three 8-byte LDS stores
    // (bw = 0 for the image past the chunk's last tile, which the tail of the pipeline builds from a re-load and nobody reads)
    auto splitstoreG = [&](int tile, int buf, auto HALF, float bw) {
        constexpr int HB = decltype(HALF)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                v[jj] = HB ? rgB[u][jj] : rgA[u][jj];
                if (DROP) {
                    const int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;
                    v[jj] *= drop_mul(dcfg, (uint32_t)(b * p.Trows + t) * 128u + (uint32_t)sn);
                }
                bsum[u] = fmaf(v[jj], bw, bsum[u]);
            }
            uint32_t a0, m0, l0, a1, m1, l1;
            sp_split2(v[0], v[1], a0, m0, l0);
            sp_split2(v[2], v[3], a1, m1, l1);
            uint16_t *dst = smem + (buf * 2 + s_img) * TS_IMG + (((s_s * 3) * 2 + unit_h(u)) * 128 + sn) * 8 + 4 * HB;
            *reinterpret_cast<u32x2 *>(dst) = u32x2{a0, a1};
            *reinterpret_cast<u32x2 *>(dst + 2 * 128 * 8) = u32x2{m0, m1};
            *reinterpret_cast<u32x2 *>(dst + 4 * 128 * 8) = u32x2{l0, l1};
        }
    };
    struct Planes { bf16x8 pl[3]; };
    auto convertX = [&](const float (&x)[8]) {
        u32x4 hh, mm, ll;
        uint32_t a, bb, c;
        if (TS_ABL & 2) {
            Planes P;
            hh = u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
            mm = u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
            P.pl[0] = __builtin_bit_cast(bf16x8, hh);
            P.pl[1] = __builtin_bit_cast(bf16x8, mm);
            P.pl[2] = __builtin_bit_cast(bf16x8, hh);
            return P;
        }
        sp_split2(x[0], x[1], a, bb, c); hh[0] = a; mm[0] = bb; ll[0] = c;
        sp_split2(x[2], x[3], a, bb, c); hh[1] = a; mm[1] = bb; ll[1] = c;
        sp_split2(x[4], x[5], a, bb, c); hh[2] = a; mm[2] = bb; ll[2] = c;
        sp_split2(x[6], x[7], a, bb, c); hh[3] = a; mm[3] = bb; ll[3] = c;
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

    // G fragment (step s, plane pl, channel block nb): image[s][pl][h][nb*32 + r][8]
    const int g_off = (TWO_G ? hf : 0) * TS_IMG + (h * 128 + r) * 8;
    auto mfma_step = [&](int buf, int s, const Planes &X) {
        const uint16_t *base = smem + buf * 2 * TS_IMG + s * (3 * 2 * 128 * 8) + g_off;
        bf16x8 w[4][3];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[nb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (2 * 128 * 8) + nb * 32 * 8);
        if (TS_ABL & 1) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, w[nb][pl])));
            return;
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {   // small terms first; all six land in the same fp32 accumulator
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][1], X.pl[1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][2], X.pl[0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][0], X.pl[2], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][1], X.pl[0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][0], X.pl[1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb][0], X.pl[0], acc[nb], 0, 0, 0);
        }
    };
    auto pin = [](auto &arr) {
#pragma unroll
        for (auto &v : arr) asm volatile("" : "+v"(v));
    };
    auto use = [](const Planes &P) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, P.pl[pl])));
    };
    Planes P0, P1;   // the operands of the two MFMA steps of the tile this wave multiplies next
    // One 32-step tile is two BLOCKS per wave: an MFMA block (both steps: 48 MFMAs, 24 fragment reads, nothing else) and a staging block
    // (everything else: this wave's units of the next G image -- fix-up, dropout replay, bias sums, split, LDS stores --, its X tile
    // through LDS into operand position and split, all global loads).  Waves 0-3 run MFMA block then staging block, waves 4-7 (each
    // shares a SIMD with one of them) staging block then MFMA block, one barrier per tile at the end of both: on every SIMD one wave's
    // MFMAs run beside the other wave's vector / LDS / memory work for the WHOLE tile.  (Round 4's schedule wove the split between
    // the MFMAs of each wave and ran all eight waves in lock step: behind the barrier both waves of a SIMD wanted the matrix pipe
    // together, later both were in their LDS read-backs together -- 5,140 cycles per tile for 3,072 of MFMA, 1,200 of them waves 0-3
    // waiting at the barrier: profiles/r05_weight_gradient_schedule.txt.)  Same double-buffered image, same staging roles, same
    // arithmetic and summation order as ts_body: bit-identical results.
    //   stage(g, u): image g (its raw values are in rgA / rgB) -> buffer g & 1, then image g + 1 requested;
    //                X tile u (register set u & 1) -> LDS tile -> read back by column -> split into the operands of both steps, then
    //                X tile u + 2 requested into the freed set.
    //   waves 0-3, tile mt:  MFMAs(mt) | stage(mt + 1, mt + 1)        waves 4-7, tile mt:  stage(mt + 1, mt) | MFMAs(mt)
#if TS_STAMP
    long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    long long st_prev = __builtin_amdgcn_s_memtime();
#endif
#ifndef TS_ST_PRIO
#define TS_ST_PRIO 0   // 1: s_setprio around the blocks (experiment)
#endif
    auto mfma_block = [&](int buf) {
        __builtin_amdgcn_sched_barrier(0);
        // (TS_ST_PRIO: the multiplying wave at priority 0, the staging wave at 2 -- measured without effect, 230.4 against 230.0 us; off)
        if (TS_ST_PRIO) __builtin_amdgcn_s_setprio(0);
        mfma_step(buf, 0, P0);
        mfma_step(buf, 1, P1);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 36 && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // a fragment arrives 12 MFMAs ahead of its first use; 6 - 9 are live
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stageG = [&](int g, int buf, float bw, int gnext, auto EDGE) {   // (buf given, not g & 1: past the chunk's end the last image is staged again, into the idle buffer)
        if (TS_ST_PRIO) __builtin_amdgcn_s_setprio(2);
        if (decltype(EDGE)::value && !g_int(g)) {
            fixG(g, I0{});
            fixG(g, I1{});
        }
        if (!(TS_ABL & 4)) {
            splitstoreG(g, buf, I0{}, bw);
            splitstoreG(g, buf, I1{}, bw);
        }
        if (!(TS_ABL & 8)) {
            gloadG(gnext, I0{});
            gloadG(gnext, I1{});
        }
    };
    auto stageXops = [&](int u, int unext, auto XSET, auto EDGE) {
        if (TS_ST_PRIO) __builtin_amdgcn_s_setprio(2);
        stageX(XSET);
        if (!(TS_ABL & 8)) gloadX(unext, XSET);
        readX();
        if (decltype(EDGE)::value && !x_int(u)) {
            fixX(rawT, u, 0);
            fixX(rawU, u, 1);
        }
        P0 = convertX(rawT);
        P1 = convertX(rawU);
    };
    const bool lead = hf == 0;   // (wave-uniform)
    gloadG(0, I0{});
    gloadG(0, I1{});
    gloadX(0, I0{});
    if constexpr (NXS == 2) gloadX(min(1, last), I1{});
    stageG(0, 0, 1.f, min(1, last), std::true_type{});
    if (lead) stageXops(0, min(NXS, last), I0{}, std::true_type{});
    __syncthreads();
    // (the two roles are two separate loops, not a branch inside one: a tile body holding both orders keeps the registers of both alive;
    //  the tiles that need a fix-up -- a video edge under a tap, the chunk's partial last tile, the non-linearity of the last_conv job --
    //  take a second copy of the tile body with the masks applied unconditionally: the interior copy is one straight line)
    auto tile = [&](int mt, auto SET, auto OTHER, auto LEAD, auto EDGE) {
        TS_T(5);
        const int n1 = min(mt + 1, last), n2 = min(mt + 2, last), n3 = min(mt + 3, last);
        const float bw = mt < last ? 1.f : 0.f;
        // (G before X inside a staging block: the gradient rows requested at the end of stageG have the rest of the block, the barrier and the
        // MFMA block to arrive -- requested at the block's end they were waited for, ~700 cycles per tile -- and the in-order memory counter
        // then never makes the X tile requested two blocks ago wait for anything younger than itself)
        if constexpr (decltype(LEAD)::value) {
            mfma_block(mt & 1);
            TS_T(0);
            stageG(n1, (mt + 1) & 1, bw, n2, EDGE);
            TS_T(2);
            if constexpr (NXS == 2) stageXops(n1, n3, OTHER, EDGE);
            else stageXops(n1, n2, I0{}, EDGE);
            TS_T(1);
        } else {
            stageG(n1, (mt + 1) & 1, bw, n2, EDGE);
            TS_T(2);
            if constexpr (NXS == 2) stageXops(mt, n2, SET, EDGE);
            else stageXops(mt, n1, I0{}, EDGE);
            TS_T(1);
            mfma_block(mt & 1);
            TS_T(0);
        }
        __syncthreads();
        TS_T(4);
    };
    auto tile2 = [&](int mt, auto SET, auto OTHER, auto LEAD) { tile(mt, SET, OTHER, LEAD, std::true_type{}); };
    CLK_BEGIN();
    if (lead) {
        for (int mt = 0; mt < ntiles; mt += 2) {
            tile2(mt, I0{}, I1{}, std::true_type{});
            if (mt + 1 < ntiles) tile2(mt + 1, I1{}, I0{}, std::true_type{});
        }
    } else {
        for (int mt = 0; mt < ntiles; mt += 2) {
            tile2(mt, I0{}, I1{}, std::false_type{});
            if (mt + 1 < ntiles) tile2(mt + 1, I1{}, I0{}, std::false_type{});
        }
    }
    CLK_END(1, item);

#if TS_STAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 6; ++k) g_ts_stamps[wave * 8 + k] = st_acc[k];
#endif
    if (active) {
        float *slab = p.slabs + (long)mc * 128 * p.Ktot + kc_raw * 128 + cg * 32 + r;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = nb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                slab[(long)row * p.Ktot] = acc[nb][reg];
            }
    }
    after_loop();   // (the persistent launch draws its next item here: the ticket's round trip passes while the slab stores drain)
    // bias gradients = column sums of the staged gradient rows: image 0 (Y0) from the workgroup that owns chunk 0, image 1
    // (Y1, dropout replayed) from the TWO_G workgroup.  Fixed order: a thread's own time slots, then the units.
    const bool bias0 = p.bias_slabs != nullptr && kc2 == 0;
    const bool bias1 = TWO_G && p.bias_slabs != nullptr;
    if (bias0 || bias1) {   // workgroup-uniform
        float *red = reinterpret_cast<float *>(smem);   // the images are dead after the loop's last barrier
        float own = bsum[0];
        if (TWO_G) own += bsum[NU - 1];
        red[(s_hi * 2 + s_lo) * 128 + sn] = own;
        __syncthreads();
        if (TWO_G) {
            if (tid < 128 && bias0) p.bias_slabs[(long)mc * 256 + tid] = red[tid] + red[128 + tid];
            if (tid >= 128 && tid < 256 && bias1) p.bias_slabs[(long)mc * 256 + tid] = red[256 + sn] + red[384 + sn];
        } else if (tid < 128 && bias0) {
            p.bias_slabs[(long)mc * 256 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same workgroup on v_mfma_f32_16x16x32_bf16 (round 5: MUCON_MFMA16 bit 1; why a second shape: gemm_split.hpp, and
// profiles/r05_mfma_shape.txt for the A/B).  Lane (r = lane & 15, h = lane >> 4) holds 8 consecutive TIME steps 8h .. 8h + 7 of a
// 32-step tile, for channel r of a 16-channel block (G) or column r of a 16-column half (X): a tile is ONE 32-deep step of
// 8 channel blocks x 2 column halves x 6 products = 96 MFMAs (32x32x16: 2 steps x 4 x 6 = 48 of twice the cycles).
//   * G image per 32 steps: [plane 3][time group 4][channel 128][8 steps] -- the same 24 KB, staged by the same thread roles
//     (unit (s, h) of the wide shape is time group 2s + h), conflict-free ds_read_b128 / ds_write_b64.
//   * a tile runs as two phases of four channel blocks each, both column halves in every phase: every G fragment is read from
//     LDS once (12 per phase, as before).  The operand planes of BOTH column halves of a tile are complete when it starts; the
//     split of the next tile's column half 0 / 1 is woven into phase 0 / 1, so the wave's X tile in LDS holds tile mt + 1
//     while tile mt multiplies (one tile further ahead than the wide shape; the register sets hold tiles mt + 2, mt + 3).
//   * the X tile is [32 steps][32] floats with columns 0-15 and 16-31 exchanged in the odd groups of 8 steps: ds_read_b32 serves
//     lanes 0-31 and 32-63 as two groups over 32 banks, a group's two time groups (h = 0, 1 / 2, 3) then sit in different bank halves
//     (plain row-major would collide 2-way).
//   * accumulators: [column half][channel block] float4 = channels 16 cb + 4 h + e of column 16 c + r.
// Summation order differs from the wide shape's (K = 32 per MFMA instead of 16): the gradients of the two shapes agree to
// fp32 rounding, not bitwise; each shape is bitwise repeatable and batch independent by itself.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TS16_XT_FLOATS = 32 * 32;
constexpr int TS16_SMEM_BYTES = 2 * 2 * TS_IMG * 2 + 8 * TS16_XT_FLOATS * 4;   // 131,072 B

template <bool TWO_G, bool DROP>
__device__ __forceinline__ void ts_body16(const TnParams &p, const int kc2, const int mc, const bool dual, const bool x0_act,
                                          uint16_t *smem) {
    constexpr int NU = TWO_G ? 2 : 1;   // staging units (8 time steps of one channel) per thread and tile
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave >> 2, cg = wave & 3;
    const int r = lane & 15, h = lane >> 4;
    const int b = mc / p.chunks_per_video;
    const int tbeg = (mc - b * p.chunks_per_video) * p.MC;
    const int tend = min(tbeg + p.MC, p.Trows);
    const int ntiles = (tend - tbeg + 31) >> 5;
    const int last = ntiles - 1;
    const int nch = p.nk0 + (dual ? 1 : 0);
    const int kc_raw = 2 * kc2 + hf;
    const bool active = kc_raw < nch;
    const int kc = active ? kc_raw : 2 * kc2;
    const bool second = dual && kc >= p.nk0;
    const int xoff = (!second && p.taps == 3) ? (kc - 1) * p.tap_step : 0;
    const int xcol = (second || p.taps == 3) ? 0 : kc * 128;
    const int ldx = second ? 128 : p.ldx;
    const int Tx = second ? p.Trows : p.Tx;
    const float *Xu = second ? p.X1 + (long)b * p.Trows * 128 : p.X0 + (long)b * p.x_bstride + xcol;   // wave-uniform
    const int xrow = lane >> 3, xc4 = (lane & 7) * 4;
    const uint32_t x_lane = (uint32_t)(xrow * ldx + cg * 32 + xc4) * 4u;
    float *xT = reinterpret_cast<float *>(smem + 2 * 2 * TS_IMG) + wave * TS16_XT_FLOATS;

    const int sn = tid & 127;
    const int s_hi = wave >> 2, s_lo = (wave >> 1) & 1;
    const int s_img = TWO_G ? s_hi : 0;
    const int s_s = TWO_G ? s_lo : s_hi;
    const float *Yu = ((TWO_G && s_img) ? p.Y1 : p.Y0) + (long)b * p.Trows * 128;   // wave-uniform
    const uint32_t y_lane = (uint32_t)sn * 4u;
    auto unit_h = [&](int u) { return TWO_G ? u : s_lo; };
    DropCfg dcfg = p.drop;
    dcfg.thresh = s_img ? dcfg.thresh : 0u;
    dcfg.scale = s_img ? dcfg.scale : 1.f;
    auto ld_su = [](const float *ubase, uint32_t lane_bytes) {
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + lane_bytes);
    };

    f32x4 rx[2][4];
    float rawT[8];
    float rgA[NU][4], rgB[NU][4];
    float bsum[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) bsum[u] = 0.f;

    auto g_int = [&](int tile) { return tbeg + tile * 32 + 32 <= tend; };
    auto x_int = [&](int tile) {
        const int t0 = tbeg + tile * 32;
        return !x0_act && t0 + 32 <= tend && t0 + xoff >= 0 && t0 + 31 + xoff < Tx;
    };
    auto gloadX = [&](int tile, auto SET) {
        constexpr int Q = decltype(SET)::value;
        if (x_int(tile)) {
            const float *ub = Xu + (long)(tbeg + tile * 32 + xoff) * ldx;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rx[Q][i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(ub + (long)(8 * i) * ldx) + x_lane);
        } else {
            const int row0 = tbeg + tile * 32 + xrow + xoff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ts = min(max(row0 + 8 * i, 0), Tx - 1);
                rx[Q][i] = *reinterpret_cast<const f32x4 *>(Xu + (long)ts * ldx + cg * 32 + xc4);
            }
        }
    };
    // wave-private X tile: rows as loaded (the odd groups of 8 steps with the two 16-column halves exchanged), read back by column
    auto stageX = [&](auto SET) {
        constexpr int Q = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(xT + (8 * i + xrow) * 32 + (xc4 ^ (16 * (i & 1)))) = rx[Q][i];
    };
    auto readX = [&](int c) {   // column 16 c + r, steps 8h .. 8h + 7
#pragma unroll
        for (int j = 0; j < 8; ++j) rawT[j] = xT[(8 * h + j) * 32 + ((16 * c + r) ^ (16 * (h & 1)))];
    };
    auto fixX = [&](float (&raw)[8], int tile) {   // non-linearity of the last_conv job, zero padding, chunk end
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = tbeg + tile * 32 + 8 * h + j;
            const int ts = t + xoff;
            float x = raw[j];
            if (x0_act) x = act_f(x, p.slope);
            raw[j] = (t < tend && ts >= 0 && ts < Tx) ? x : 0.f;
        }
    };
    auto gloadG = [&](int tile, auto HALF) {
        constexpr int HB = decltype(HALF)::value;
        const bool inner = g_int(tile);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;   // wave-uniform
                if (!inner) t = min(t, p.Trows - 1);
                const float v = ld_su(Yu + (long)t * 128, y_lane);
                if constexpr (HB) rgB[u][jj] = v;
                else rgA[u][jj] = v;
            }
    };
    auto fixG = [&](int tile, auto HALF) {
        constexpr int HB = decltype(HALF)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;
                if constexpr (HB) rgB[u][jj] = t < tend ? rgB[u][jj] : 0.f;
                else rgA[u][jj] = t < tend ? rgA[u][jj] : 0.f;
            }
    };
    auto splitstoreG = [&](int tile, int buf, auto HALF, float bw) {
        constexpr int HB = decltype(HALF)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                v[jj] = HB ? rgB[u][jj] : rgA[u][jj];
                if (DROP) {
                    const int t = tbeg + tile * 32 + 16 * s_s + 8 * unit_h(u) + 4 * HB + jj;
                    v[jj] *= drop_mul(dcfg, (uint32_t)(b * p.Trows + t) * 128u + (uint32_t)sn);
                }
                bsum[u] = fmaf(v[jj], bw, bsum[u]);
            }
            uint32_t a0, m0, l0, a1, m1, l1;
            sp_split2(v[0], v[1], a0, m0, l0);
            sp_split2(v[2], v[3], a1, m1, l1);
            // image [plane 3][time group 4][channel 128][8]: this unit is time group 2 s + h
            uint16_t *dst = smem + (buf * 2 + s_img) * TS_IMG + ((2 * s_s + unit_h(u)) * 128 + sn) * 8 + 4 * HB;
            *reinterpret_cast<u32x2 *>(dst) = u32x2{a0, a1};
            *reinterpret_cast<u32x2 *>(dst + 4 * 128 * 8) = u32x2{m0, m1};
            *reinterpret_cast<u32x2 *>(dst + 8 * 128 * 8) = u32x2{l0, l1};
        }
    };
    struct Planes { bf16x8 pl[3]; };
    auto convertX = [&](const float (&x)[8]) {
        u32x4 hh, mm, ll;
        uint32_t a, bb, c;
        sp_split2(x[0], x[1], a, bb, c); hh[0] = a; mm[0] = bb; ll[0] = c;
        sp_split2(x[2], x[3], a, bb, c); hh[1] = a; mm[1] = bb; ll[1] = c;
        sp_split2(x[4], x[5], a, bb, c); hh[2] = a; mm[2] = bb; ll[2] = c;
        sp_split2(x[6], x[7], a, bb, c); hh[3] = a; mm[3] = bb; ll[3] = c;
        Planes P;
        P.pl[0] = __builtin_bit_cast(bf16x8, hh);
        P.pl[1] = __builtin_bit_cast(bf16x8, mm);
        P.pl[2] = __builtin_bit_cast(bf16x8, ll);
        return P;
    };

    f32x4 acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // G fragment (plane pl, channel block cb): image[pl][h][cb*16 + r][8]
    const int g_off = (TWO_G ? hf : 0) * TS_IMG + (h * 128 + r) * 8;
    auto mfma_phase = [&](int buf, auto HALF, const Planes &X0, const Planes &X1) {
        constexpr int CB0 = 4 * decltype(HALF)::value;
        const uint16_t *base = smem + buf * 2 * TS_IMG + g_off + CB0 * 16 * 8;
        bf16x8 w[4][3];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[cb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (4 * 128 * 8) + cb * 16 * 8);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int c = 0; c < 2; ++c) {   // small terms first; all six land in the same fp32 accumulator
                const Planes &X = c ? X1 : X0;
                f32x4 a = acc[c][CB0 + cb];
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][1], X.pl[1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][2], X.pl[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], X.pl[2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][1], X.pl[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], X.pl[1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], X.pl[0], a, 0, 0, 0);
                acc[c][CB0 + cb] = a;
            }
    };
    // 48 MFMAs: the first two channel blocks' fragments in front, the other six under the first 12 MFMAs (a fragment read issued four
    // 16-cycle MFMAs ahead of its use still exposed its latency); 2 * VPM vector instructions behind every pair of MFMAs (an MFMA
    // holds the SIMD's issue for 8 of its 16 cycles), the LDS stores in the second half
    constexpr int VPM = DROP ? 7 : (TWO_G ? 4 : 3);
    auto weave = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 12 && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i & 1) __builtin_amdgcn_sched_group_barrier(0x002, VPM - VPM / 2, 0);
            else __builtin_amdgcn_sched_group_barrier(0x002, VPM / 2, 0);
            if (i >= 24 && (i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };

    auto pin = [](auto &arr) {
#pragma unroll
        for (auto &v : arr) asm volatile("" : "+v"(v));
    };
    auto use = [](const Planes &P) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, P.pl[pl])));
    };
    Planes cur0, cur1;
    {   // prologue: image of tile 0; X tiles 0 .. 3 requested, tile 0 split into both operands, tile 1 staged; first half of image 1 requested
        gloadG(0, I0{});
        gloadG(0, I1{});
        gloadX(0, I0{});
        gloadX(min(1, last), I1{});
        if (!g_int(0)) {
            fixG(0, I0{});
            fixG(0, I1{});
        }
        splitstoreG(0, 0, I0{}, 1.f);
        splitstoreG(0, 0, I1{}, 1.f);
        gloadG(min(1, last), I0{});
        stageX(I0{});
        gloadX(min(2, last), I0{});
        readX(0);
        if (!x_int(0)) fixX(rawT, 0);
        cur0 = convertX(rawT);
        readX(1);
        if (!x_int(0)) fixX(rawT, 0);
        cur1 = convertX(rawT);
        stageX(I1{});
        gloadX(min(3, last), I1{});
        __syncthreads();
    }

    // tile mt (image in buffer Q; operands cur0 / cur1; the wave's X tile holds tile mt+1; set Q holds tile mt+2, set O tile mt+3 on its way):
    //   { second half of image mt+1 requested; X (mt+1, columns 0-15) read back }
    //   { MFMAs of channel blocks 0-3 | split of X (mt+1, columns 0-15), first half of image mt+1 -> buffer O }
    //   { X (mt+1, columns 16-31) read back; X tile <- tile mt+2 (set Q), set Q <- tile mt+4 requested; first half of image mt+2 requested }
    //   { MFMAs of channel blocks 4-7 | split of X (mt+1, columns 16-31), second half of image mt+1 }
#if TS_STAMP
    long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    long long st_prev = __builtin_amdgcn_s_memtime();
#endif
    auto tile = [&](int mt, auto SET, auto OTHER) {
        constexpr int Q = decltype(SET)::value, O = decltype(OTHER)::value;
        TS_T(5);
        const int n1 = min(mt + 1, last), n2 = min(mt + 2, last), n4 = min(mt + 4, last);
        const float bw = mt < last ? 1.f : 0.f;
        gloadG(n1, I1{});
        readX(0);
        if (!x_int(n1)) {
            asm volatile("");   // (keeps the fix-up a branch: if-converted it is ~80 selects on every tile)
            fixX(rawT, n1);
        }
        if (!g_int(n1)) {
            asm volatile("");
            fixG(n1, I0{});
        }
        __builtin_amdgcn_sched_barrier(0);
        TS_T(0);
        pin(rawT);
#pragma unroll
        for (int u = 0; u < NU; ++u) pin(rgA[u]);
        mfma_phase(Q, I0{}, cur0, cur1);
        Planes nxt0 = convertX(rawT);
        splitstoreG(n1, O, I0{}, bw);
        weave();
        use(nxt0);
        __builtin_amdgcn_sched_barrier(0);
        TS_T(1);
        readX(1);
        stageX(SET);
        gloadG(n2, I0{});
        gloadX(n4, SET);
        if (!x_int(n1)) {
            asm volatile("");
            fixX(rawT, n1);
        }
        if (!g_int(n1)) {
            asm volatile("");
            fixG(n1, I1{});
        }
        __builtin_amdgcn_sched_barrier(0);
        TS_T(2);
        pin(rawT);
#pragma unroll
        for (int u = 0; u < NU; ++u) pin(rgB[u]);
        mfma_phase(Q, I1{}, cur0, cur1);
        Planes nxt1 = convertX(rawT);
        splitstoreG(n1, O, I1{}, bw);
        weave();
        use(nxt1);
        __builtin_amdgcn_sched_barrier(0);
        TS_T(3);
        cur0 = nxt0;
        cur1 = nxt1;
        __syncthreads();
        TS_T(4);
    };
    CLK_BEGIN();
    for (int mt = 0; mt < ntiles; mt += 2) {
        tile(mt, I0{}, I1{});
        if (mt + 1 < ntiles) tile(mt + 1, I1{}, I0{});
    }
    CLK_END(1, blockIdx.x);

#if TS_STAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 6; ++k) g_ts_stamps[wave * 8 + k] = st_acc[k];
#endif
    if (active) {
        float *slab = p.slabs + (long)mc * 128 * p.Ktot + kc_raw * 128 + cg * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) slab[(long)(cb * 16 + 4 * h + e) * p.Ktot + 16 * c] = acc[c][cb][e];
    }
    const bool bias0 = p.bias_slabs != nullptr && kc2 == 0;
    const bool bias1 = TWO_G && p.bias_slabs != nullptr;
    if (bias0 || bias1) {   // workgroup-uniform
        float *red = reinterpret_cast<float *>(smem);   // the images are dead after the loop's last barrier
        float own = bsum[0];
        if (TWO_G) own += bsum[NU - 1];
        red[(s_hi * 2 + s_lo) * 128 + sn] = own;
        __syncthreads();
        if (TWO_G) {
            if (tid < 128 && bias0) p.bias_slabs[(long)mc * 256 + tid] = red[tid] + red[128 + tid];
            if (tid >= 128 && tid < 256 && bias1) p.bias_slabs[(long)mc * 256 + tid] = red[256 + sn] + red[384 + sn];
        } else if (tid < 128 && bias0) {
            p.bias_slabs[(long)mc * 256 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
        }
    }
}

// All weight gradients of a backward pass in one launch (the job table of gemm_tn.hpp): a job with n 128-column chunks has
// ceil(n / 2) workgroups per time chunk.  ts_run_item is one such workgroup's work ("item" = its block index in the table's layout).
template <int MODE, class AfterLoop>   // 0: ts_body (32x32x16, round 4's lock-step schedule), 1: ts_body16 (16x16x32), 2: ts_body_st (32x32x16, staggered blocks)
__device__ __forceinline__ bool ts_run_item(const TnBatch &tb, const int item, uint16_t *ts_smem, AfterLoop &&after_loop) {
    // which job: the number of jobs that start at or before the item -- all first blocks compared at once (walking the table job by job
    // was a chain of dependent scalar loads, 1.0 - 1.7 us in front of a workgroup's first load: profiles/r05_weight_gradient_schedule.txt §9)
    int ji = -1;
#pragma unroll
    for (int k = 0; k < TN_MAX_BATCH; ++k) ji += item >= tb.first_block[k] ? 1 : 0;
    const TnJob &job = tb.j[ji];
    if (item - job.block0 >= ((job.nkc + 1) >> 1) * job.nmc) return false;   // padding block between two jobs
#if CLK_STAMP
    {
        long long t_;   // (the job's words are an input of the stamp: it cannot be read before they have arrived)
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "s"(job.nkc), "s"(job.nmc));
        if (threadIdx.x == 0 && item < 4096) g_clk_ph[item][0] = t_;
    }
#endif
    const int nkc2 = (job.nkc + 1) >> 1;
    const int local = item - job.block0;
    int mc = local / nkc2, kc2 = local - mc * nkc2;
    if (tb.xcd_order) {
        // The nkc2 workgroups of a time chunk read the same gradient rows.  Workgroups are dealt round-robin over the 8 XCDs
        // (block b and b + 8 share one -- observed, used for speed only), so inside every run of 8 * nkc2 blocks the chunk is
        // the block index mod 8: the workgroups that share rows share an L2 (4 MB per XCD; first_conv's 1 MB of rows per chunk
        // is fetched from the Infinity Cache once instead of eight times).
        const int nmc = job.nmc, grp = 8 * nkc2;
        const int G = local / grp;
        if ((G + 1) * 8 <= nmc) {
            const int in = local - G * grp;
            mc = G * 8 + (in & 7);
            kc2 = in >> 3;
        }
    }
    const bool two_g = job.dual && 2 * kc2 + 1 == job.p.nk0;
    if constexpr (MODE == 2) {
        // The staggered schedule is taken by the jobs with ONE gradient image and time chunks of at least tb.st_min_steps steps (first_conv's:
        // 128 workgroups of 64 tiles at the bench shape, 155-160 -> 136-141 us each).  The residual layers' jobs keep round 4's body: their
        // workgroups that stage two images and replay the dropout mask have twice the staging work per tile and the registers for one X
        // tile in flight only -- staggered they ran 218 us instead of 200 (profiles/r05_weight_gradient_schedule.txt).
        if (job.dual || tb.st_min_steps <= 0 || job.p.MC < tb.st_min_steps) {
            if (!two_g) ts_body<false, false>(job.p, kc2, mc, job.dual != 0, job.x0_act != 0, ts_smem, item, after_loop);
            else if (job.p.drop.thresh) ts_body<true, true>(job.p, kc2, mc, true, false, ts_smem, item, after_loop);
            else ts_body<true, false>(job.p, kc2, mc, true, false, ts_smem, item, after_loop);
        } else if (!two_g) ts_body_st<false, false>(job.p, kc2, mc, job.dual != 0, job.x0_act != 0, ts_smem, item, after_loop);
        else if (job.p.drop.thresh) ts_body_st<true, true>(job.p, kc2, mc, true, false, ts_smem, item, after_loop);
        else ts_body_st<true, false>(job.p, kc2, mc, true, false, ts_smem, item, after_loop);
    } else if constexpr (MODE == 1) {
        if (!two_g) ts_body16<false, false>(job.p, kc2, mc, job.dual != 0, job.x0_act != 0, ts_smem);
        else if (job.p.drop.thresh) ts_body16<true, true>(job.p, kc2, mc, true, false, ts_smem);
        else ts_body16<true, false>(job.p, kc2, mc, true, false, ts_smem);
        after_loop();
    } else {
        if (!two_g) ts_body<false, false>(job.p, kc2, mc, job.dual != 0, job.x0_act != 0, ts_smem, item, after_loop);
        else if (job.p.drop.thresh) ts_body<true, true>(job.p, kc2, mc, true, false, ts_smem, item, after_loop);
        else ts_body<true, false>(job.p, kc2, mc, true, false, ts_smem, item, after_loop);
    }
    return true;
}

template <int MODE>
__global__ __launch_bounds__(512) void ts_batched_kernel(const TnBatch tb) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ts_smem[];
#if CLK_STAMP
    const long long wg_t0_ = __builtin_amdgcn_s_memrealtime();   // (diagnostic build: the workgroup's whole life beside its tile loop, g_clk_wg)
#endif
    if (!ts_run_item<MODE>(tb, (int)blockIdx.x, ts_smem, [] {})) return;
#if CLK_STAMP
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_clk_wg[blockIdx.x][0] = wg_t0_;
        g_clk_wg[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// The same items on PERSISTENT workgroups (round 5; MUCON_TS_PERSIST, default): one workgroup per CU walks the item list -- its own block
// index first, then whatever item the launch's ticket counter hands it.  What that buys (profiles/r05_weight_gradient_schedule.txt §9): the
// hardware dispatcher deals workgroups to the XCDs strictly round-robin, so a freed CU waited a median 1.4 us and up to 15 us (another XCD's
// turn) for its next workgroup; the ticket is drawn behind the item's last tile and arrives under its slab write-out.  Every item computes
// exactly what its workgroup computed (which workgroup runs it changes nothing in its sums): bitwise the same gradients.
// `tickets` is one zeroed word of the caller's workspace (encoder_bwd's first kernel zeroes it: gn_bwd_kernel).
template <int MODE>
__global__ __launch_bounds__(512) void ts_persist_kernel(const TnBatch tb, unsigned *tickets) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ts_smem[];
    int *s_next = reinterpret_cast<int *>(ts_smem + TS_SMEM_BYTES / 2);   // one word behind the tiles
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int item = blockIdx.x;
    while (item < tb.nblocks) {
#if CLK_STAMP
        const long long wg_t0_ = __builtin_amdgcn_s_memrealtime();
#endif
        // The ticket is a SCALAR-memory atomic (s_atomic_add, wave 0): it returns through lgkmcnt, so it neither queues behind the slab stores
        // nor makes the compiler wait for them (a vector atomic behind a `threadIdx.x == 0` branch did: its round trip, 1.2 - 2.3 us,
        // stood in front of the item's slab write-out: profiles/r05_weight_gradient_schedule.txt §9).  Issue and wait are one asm
        // statement -- the compiler never sees a register with a load in flight -- placed behind wave 0's slab stores
        // so that it waits while they drain.
        bool have = false;
        auto draw = [&]() {
            if (wave == 0) {
                unsigned t = 1u;
                asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(tickets) : "memory");
                if (threadIdx.x == 0) *s_next = (int)(gridDim.x + t);
            }
            have = true;
        };
        const bool ran = ts_run_item<MODE>(tb, item, ts_smem, draw);
        if (!have) {         // a padding item: no barrier has been passed since the item was read -- every wave must have read it before wave 0 overwrites the word
            __syncthreads();
            draw();
        }
#if CLK_STAMP
        if (ran) {
            __syncthreads();
            if (threadIdx.x == 0 && item < 4096) {
                g_clk_wg[item][0] = wg_t0_;
                g_clk_wg[item][1] = __builtin_amdgcn_s_memrealtime();
            }
        }
#endif
        (void)ran;
        __syncthreads();   // ... which also puts the item's last LDS reads (the bias sums) in front of the next item's first stores
        item = __builtin_amdgcn_readfirstlane(*s_next);
    }
}

extern int g_ts_persist;   // mucon_hip.hip (MUCON_TS_PERSIST)
static hipError_t launch_ts_batch(TnBatch &tb, hipStream_t s, unsigned *tickets = nullptr) {
    if (tb.njobs == 0) return hipSuccess;
    const int mode = (g_mfma16 & 2) ? 1 : (g_ts_stagger ? 2 : 0);
    const bool persist = tickets != nullptr && g_ts_persist != 0;
    static bool attr_set[2][3] = {{false, false, false}, {false, false, false}};
    static int ncu = 0;
    const void *k = persist ? (mode == 1 ? reinterpret_cast<const void *>(ts_persist_kernel<1>)
                                         : mode == 2 ? reinterpret_cast<const void *>(ts_persist_kernel<2>) : reinterpret_cast<const void *>(ts_persist_kernel<0>))
                            : (mode == 1 ? reinterpret_cast<const void *>(ts_batched_kernel<1>)
                                         : mode == 2 ? reinterpret_cast<const void *>(ts_batched_kernel<2>) : reinterpret_cast<const void *>(ts_batched_kernel<0>));
    const int smem = (mode == 1 ? TS16_SMEM_BYTES : TS_SMEM_BYTES) + (persist ? 16 : 0);
    if (!attr_set[persist][mode]) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set[persist][mode] = true;
    }
    if (persist && ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        ncu = prop.multiProcessorCount;
    }
    // jobs were queued coarse levels first; the fine levels have the longest workgroups: lay them out first
    TnBatch lb;
    lb.njobs = tb.njobs;
    int blocks = 0;
    for (int i = 0; i < TN_MAX_BATCH; ++i) lb.first_block[i] = INT_MAX;
    for (int i = 0; i < tb.njobs; ++i) {
        const TnJob &src = tb.j[tb.njobs - 1 - i];
        lb.j[i] = src;
        lb.j[i].block0 = blocks;
        lb.first_block[i] = blocks;
        lb.j[i].nmc = src.block0;                     // block0 carried the time-chunk count while queued
        blocks += ((src.nkc + 1) / 2) * src.block0;
        if (kTsXcdOrder) blocks = (blocks + 7) & ~7;     // every job starts on a multiple of 8 (the padding blocks exit at once)
    }
    lb.nblocks = blocks;
    lb.xcd_order = kTsXcdOrder;
    lb.st_min_steps = g_ts_stagger;
    if (persist) {
        const dim3 grid(std::min(blocks, ncu));
        if (mode == 1) hipLaunchKernelGGL(ts_persist_kernel<1>, grid, dim3(512), smem, s, lb, tickets);
        else if (mode == 2) hipLaunchKernelGGL(ts_persist_kernel<2>, grid, dim3(512), smem, s, lb, tickets);
        else hipLaunchKernelGGL(ts_persist_kernel<0>, grid, dim3(512), smem, s, lb, tickets);
    } else if (mode == 1) hipLaunchKernelGGL(ts_batched_kernel<1>, dim3(blocks), dim3(512), smem, s, lb);
    else if (mode == 2) hipLaunchKernelGGL(ts_batched_kernel<2>, dim3(blocks), dim3(512), smem, s, lb);
    else hipLaunchKernelGGL(ts_batched_kernel<0>, dim3(blocks), dim3(512), smem, s, lb);
    tb.njobs = 0;
    return hipGetLastError();
}
