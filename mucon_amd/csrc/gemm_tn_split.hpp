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

constexpr int TS_IMG = 2 * 3 * 2 * 128 * 8;             // bf16 elements of one 32-step G image (24,576 B)
constexpr int TS_XT_FLOATS = 32 * 32;                   // a wave's X tile [32 time steps][32 columns] (4 KB), transposed through LDS
constexpr int TS_SMEM_BYTES = 2 * 2 * TS_IMG * 2 + 8 * TS_XT_FLOATS * 4;   // two buffers x two images + eight X tiles: 131,072 B

template <bool TWO_G, bool DROP>
__device__ __forceinline__ void ts_body(const TnParams &p, const int kc2, const int b, const int tbeg, const int tend, const bool dual, const bool x0_act,
                                        uint16_t *smem, const int item, f32x16 (&acc)[4], float (&bsum)[2]) {
    constexpr int NU = TWO_G ? 2 : 1;   // staging units (8 time steps of one channel) per thread and tile
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // (the thread index through an opaque asm: a run's per-lane offsets and roles are then computed inside the run -- hoisted out of the caller's run and
    // column loops they stayed alive across every body of the kernel: spills inside the staggered body's staging blocks, 277 us per share instead of 168)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave >> 2, cg = wave & 3;
    const int r = lane & 31, h = lane >> 5;
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

    // A tile is interior when none of its 32 rows needs a mask: inside the chunk (G) and, for this wave's tap, inside the video (X).
    // Interior tiles are loaded with wave-uniform row addresses and enter the MFMA body as they are; the others are loaded
    // from clamped rows and masked in their registers by the (rare) fix-up branches in front of the woven phases.
    auto g_int = [&](int tile) { return tbeg + tile * 32 + 32 <= tend; };
    auto x_int = [&](int tile) {
        const int t0 = tbeg + tile * 32;
        return !x0_act && t0 + 32 <= tend && t0 + xoff >= 0 && t0 + 31 + xoff < Tx;
    };
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
        const bool inner = false;
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


    // G fragment (step s, plane pl, channel block nb): image[s][pl][h][nb*32 + r][8]
    const int g_off = (TWO_G ? hf : 0) * TS_IMG + (h * 128 + r) * 8;
    auto mfma_step = [&](int buf, int s, const Planes &X) {
        const uint16_t *base = smem + buf * 2 * TS_IMG + s * (3 * 2 * 128 * 8) + g_off;
        bf16x8 w[4][3];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[nb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (2 * 128 * 8) + nb * 32 * 8);
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
    auto tile = [&](int mt, auto SET, auto OTHER) {
        constexpr int Q = decltype(SET)::value, O = decltype(OTHER)::value;
        (void)Q;
        const int n1 = min(mt + 1, last), n2 = min(mt + 2, last), n3 = min(mt + 3, last);
        const float bw = mt < last ? 1.f : 0.f;
        gloadG(n1, I1{});
        readX(1);
        if (!x_int(mt)) fixX(rawT, mt, 1);
        if (!g_int(n1)) fixG(n1, I0{});
        __builtin_amdgcn_sched_barrier(0);
        pin(rawT);   // (keeps the splits below in this block: without it they are duplicated into the fix-up branches, outside the weave)
#pragma unroll
        for (int u = 0; u < NU; ++u) pin(rgA[u]);
        mfma_step(Q, 0, cur);
        Planes nxt = convertX(rawT);
        splitstoreG(n1, O, I0{}, bw);
        weave();
        use(nxt);   // (a use inside the phase: otherwise the split is sunk behind the branches below, out of the weave)
        __builtin_amdgcn_sched_barrier(0);
        stageX(OTHER);
        gloadG(n2, I0{});
        gloadX(n3, OTHER);
        readX(0);
        if (!x_int(n1)) fixX(rawT, n1, 0);
        if (!g_int(n1)) fixG(n1, I1{});
        __builtin_amdgcn_sched_barrier(0);
        pin(rawT);
#pragma unroll
        for (int u = 0; u < NU; ++u) pin(rgB[u]);
        mfma_step(Q, 1, nxt);
        cur = convertX(rawT);
        splitstoreG(n1, O, I1{}, bw);
        weave();
        use(cur);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    CLK_BEGIN();
    for (int mt = 0; mt < ntiles; mt += 2) {
        tile(mt, I0{}, I1{});
        if (mt + 1 < ntiles) tile(mt + 1, I1{}, I0{});
    }
    CLK_END(1, item);

}

// ---------------------------------------------------------------------------------------------------------------------
// The staggered schedule (round 5; MUCON_TS_STAGGER, ts_batched_kernel<2>): ts_body's workgroup, operands, images, staging roles and
// arithmetic with the tile re-ordered into blocks so that the two waves of a SIMD never want the same unit -- see the comment at the
// tile loop below.
// ---------------------------------------------------------------------------------------------------------------------
template <bool TWO_G, bool DROP>
__device__ __forceinline__ void ts_body_st(const TnParams &p, const int kc2, const int b, const int tbeg, const int tend, const bool dual, const bool x0_act,
                                        uint16_t *smem, const int item, f32x16 (&acc)[4], float (&bsum)[2]) {
    constexpr int NU = TWO_G ? 2 : 1;   // staging units (8 time steps of one channel) per thread and tile
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // (the thread index through an opaque asm: a run's per-lane offsets and roles are then computed inside the run -- hoisted out of the caller's run and
    // column loops they stayed alive across every body of the kernel: spills inside the staggered body's staging blocks, 277 us per share instead of 168)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave >> 2, cg = wave & 3;
    const int r = lane & 31, h = lane >> 5;
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


    // G fragment (step s, plane pl, channel block nb): image[s][pl][h][nb*32 + r][8]
    const int g_off = (TWO_G ? hf : 0) * TS_IMG + (h * 128 + r) * 8;
    auto mfma_step = [&](int buf, int s, const Planes &X) {
        const uint16_t *base = smem + buf * 2 * TS_IMG + s * (3 * 2 * 128 * 8) + g_off;
        bf16x8 w[4][3];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[nb][pl] = *reinterpret_cast<const bf16x8 *>(base + pl * (2 * 128 * 8) + nb * 32 * 8);
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
    auto mfma_block = [&](int buf) {
        __builtin_amdgcn_sched_barrier(0);
        // (s_setprio around the blocks -- the multiplying wave at priority 0, the staging wave at 2 -- measured without effect: 230.4 against 230.0 us)
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
        if (decltype(EDGE)::value && !g_int(g)) {
            fixG(g, I0{});
            fixG(g, I1{});
        }
        splitstoreG(g, buf, I0{}, bw);
        splitstoreG(g, buf, I1{}, bw);
        gloadG(gnext, I0{});
        gloadG(gnext, I1{});
    };
    auto stageXops = [&](int u, int unext, auto XSET, auto EDGE) {
        stageX(XSET);
        gloadX(unext, XSET);
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
        const int n1 = min(mt + 1, last), n2 = min(mt + 2, last), n3 = min(mt + 3, last);
        const float bw = mt < last ? 1.f : 0.f;
        // (G before X inside a staging block: the gradient rows requested at the end of stageG have the rest of the block, the barrier and the
        // MFMA block to arrive -- requested at the block's end they were waited for, ~700 cycles per tile -- and the in-order memory counter
        // then never makes the X tile requested two blocks ago wait for anything younger than itself)
        if constexpr (decltype(LEAD)::value) {
            mfma_block(mt & 1);
            stageG(n1, (mt + 1) & 1, bw, n2, EDGE);
            if constexpr (NXS == 2) stageXops(n1, n3, OTHER, EDGE);
            else stageXops(n1, n2, I0{}, EDGE);
        } else {
            stageG(n1, (mt + 1) & 1, bw, n2, EDGE);
            if constexpr (NXS == 2) stageXops(mt, n2, SET, EDGE);
            else stageXops(mt, n1, I0{}, EDGE);
            mfma_block(mt & 1);
        }
        __syncthreads();
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

}

// ---------------------------------------------------------------------------------------------------------------------
// What a workgroup leaves behind for one COLUMN (a job's pair of 128-column chunks kc2): its 128 x 256 partial tile and the bias
// partials of the gradient images it staged.  `slab` = the partial's [128][ld] block at this column's first column, `bias` = 256
// floats (image 0's column sums in the first 128 words -- written by the workgroups of column 0 only --, image 1's in the second).
// ---------------------------------------------------------------------------------------------------------------------
template <bool TWO_G>
__device__ __forceinline__ void ts_flush(const TnParams &p, const int kc2, const bool dual, uint16_t *smem, const f32x16 (&acc)[4], const float (&bsum)[2],
                                         float *slab, const int ld, float *bias) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave >> 2, cg = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int nch = p.nk0 + (dual ? 1 : 0);
    if (2 * kc2 + hf < nch) {   // (an odd chunk count leaves the last column's second half without columns)
        float *o = slab + hf * 128 + cg * 32 + r;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = nb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                o[(long)row * ld] = acc[nb][reg];
            }
    }
    // bias gradients = column sums of the staged gradient rows: image 0 (Y0) from the workgroups of column 0, image 1 (Y1, dropout
    // replayed) from the TWO_G workgroups.  Fixed order: a thread's own time slots, then the units.
    const bool bias0 = bias != nullptr && kc2 == 0;
    const bool bias1 = TWO_G && bias != nullptr;
    if (bias0 || bias1) {   // workgroup-uniform
        const int sn = tid & 127;
        const int s_hi = wave >> 2, s_lo = (wave >> 1) & 1;
        float *red = reinterpret_cast<float *>(smem);   // the images are dead after the tile loop's last barrier
        float own = bsum[0];
        if (TWO_G) own += bsum[1];
        red[(s_hi * 2 + s_lo) * 128 + sn] = own;
        __syncthreads();
        if (TWO_G) {
            if (tid < 128 && bias0) bias[tid] = red[tid] + red[128 + tid];
            if (tid >= 128 && tid < 256 && bias1) bias[tid] = red[256 + sn] + red[384 + sn];
        } else if (tid < 128 && bias0) {
            bias[tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
        }
        __syncthreads();   // ... and the next column's first image must not land under these reads
    }
}

// One RUN = time steps [tbeg, tend) of video b for column (job, kc2), added to the accumulators the caller keeps.  The staggered
// schedule (ts_body_st) is taken by the single-image columns of jobs without a second gradient set (first_conv's: 128 workgroups x 64 tiles
// at the bench shape, 155-160 -> 136-141 us each); the residual layers' columns keep the lock-step body: the ones that stage two images and
// replay the dropout mask have twice the staging work per tile and the registers for one X tile in flight only -- staggered they ran 218 us
// instead of 200 (profiles/r05_weight_gradient_schedule.txt).  The two bodies are bit-identical.
__device__ __forceinline__ void ts_run(const TnJob &job, const int kc2, const int b, const int tbeg, const int tend, const bool stagger,
                                       uint16_t *ts_smem, const int item, f32x16 (&acc)[4], float (&bsum)[2]) {
    const bool two_g = job.dual && 2 * kc2 + 1 == job.p.nk0;
    if (!two_g) {
        if (stagger && !job.dual) ts_body_st<false, false>(job.p, kc2, b, tbeg, tend, false, job.x0_act != 0, ts_smem, item, acc, bsum);
        else ts_body<false, false>(job.p, kc2, b, tbeg, tend, job.dual != 0, job.x0_act != 0, ts_smem, item, acc, bsum);
    } else if (job.p.drop.thresh) ts_body<true, true>(job.p, kc2, b, tbeg, tend, true, false, ts_smem, item, acc, bsum);
    else ts_body<true, false>(job.p, kc2, b, tbeg, tend, true, false, ts_smem, item, acc, bsum);
}
__device__ __forceinline__ void ts_zero(f32x16 (&acc)[4], float (&bsum)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bsum[0] = bsum[1] = 0.f;
}

// ---- one workgroup per ITEM (= column x time chunk of p.MC steps; slabs [time chunk][128][Ktot]): the form of the stand-alone entry points
// (mucon_linear_bwd, mucon_conv128_wgrad) and, under MUCON_TS_RUNS=0, of the batched launch -- the tests' second schedule of the same sums
__global__ __launch_bounds__(512) void ts_batched_kernel(const TnBatch tb) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ts_smem[];
    const int item = blockIdx.x;
    int ji = -1;
#pragma unroll
    for (int k = 0; k < TN_MAX_BATCH; ++k) ji += item >= tb.first_block[k] ? 1 : 0;
    const TnJob &job = tb.j[ji];
    const int nkc2 = (job.nkc + 1) >> 1;
    const int local = item - job.block0;
    if (local >= nkc2 * job.nmc) return;   // padding block between two jobs
    const int mc = local / nkc2, kc2 = local - mc * nkc2;
    const int b = mc / job.p.chunks_per_video;
    const int tbeg = (mc - b * job.p.chunks_per_video) * job.p.MC;
    const int tend = min(tbeg + job.p.MC, job.p.Trows);
    f32x16 acc[4];
    float bsum[2];
    ts_zero(acc, bsum);
    ts_run(job, kc2, b, tbeg, tend, tb.st_min_steps > 0 && job.p.MC >= tb.st_min_steps, ts_smem, item, acc, bsum);
    float *slab = job.p.slabs + (long)mc * 128 * job.p.Ktot + kc2 * 256;
    float *bias = job.p.bias_slabs ? job.p.bias_slabs + (long)mc * 256 : nullptr;
    if (job.dual && 2 * kc2 + 1 == job.p.nk0) ts_flush<true>(job.p, kc2, true, ts_smem, acc, bsum, slab, job.p.Ktot, bias);
    else ts_flush<false>(job.p, kc2, job.dual != 0, ts_smem, acc, bsum, slab, job.p.Ktot, bias);
}

// ---------------------------------------------------------------------------------------------------------------------
// STATIC RUNS (round 6; the batched launch of encoder_bwd): every weight gradient of the pass on G persistent workgroups, each of which
// takes ONE contiguous share of the pass's work and keeps its accumulators in registers for as long as it stays inside a unit.
//
// Round 5's persistent launch handed out (column, <= 512-step chunk) items by ticket: 720 items, each with its own 128 KB partial
// tile -- 96 MB of slabs written and read back per step (HBM traffic 1.35 x the algorithmic bytes, a 20-us reduction pass behind the launch),
// and every item paid ~6.5 us of prologue / slab epilogue (2.8 items per CU).  Longer chunks lose under a ticket (it balances whole
// items: profiles/r06_same_box_abs.txt section 1).  Here the pass is ONE line of work (gemm_tn.hpp: TsLine -- jobs in launch order, a job's
// groups of videos / panels, a group's columns), priced in cost units (a 32-step tile of the column's kind + a fixed cost per video a run
// enters); share s is [s S, (s + 1) S) of it, S = ceil(W / G) ("stream-K" over the reduction dimension), run by workgroup ts_share_of^-1(s).
// A share is cut into runs at video ends (the taps' zero padding and the chunk masks are per video) and into units where the column or the
// group changes; one partial tile is written per (share, unit).  The schedule is a pure function of the job shapes and G -- no ticket, no
// atomics: the same sums in the same order run after run.
// ---------------------------------------------------------------------------------------------------------------------
// position inside a unit -> tile index on the unit's own line (video-major): the SAME function gives a share's end and the next share's start
__device__ __host__ __forceinline__ uint32_t ts_qmap(const uint32_t vcost, const uint32_t tcost, const uint32_t ovh, const uint32_t tv, const uint32_t off) {
    const uint32_t b = off / vcost, rem = off - b * vcost;
    const uint32_t t = rem <= ovh ? 0u : (rem - ovh) / tcost;
    return b * tv + (t < tv ? t : tv);
}
__global__ __launch_bounds__(512) void ts_runs_kernel(const TnBatch tb, const TsLine ln) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ts_smem[];
    const uint32_t w = blockIdx.x;
    const uint32_t sh = ts_share_of(w, (uint32_t)ln.G);
    const uint32_t lo = min(sh * ln.S, ln.W), hi = min(lo + ln.S, ln.W);
    if (lo >= hi) return;
#if CLK_STAMP
    const long long wg_t0_ = __builtin_amdgcn_s_memrealtime();   // (diagnostic build: a workgroup's whole life, g_clk_wg; tools/ts_runs_times.py)
    const long long wg_c0_ = __builtin_amdgcn_s_memtime();       // ... and in shader cycles (g_clk_ph[share][0]: tools/kernel_cycles.py)
#endif
    int ji = -1;   // the job `lo` lies in: all starts compared at once (independent scalar loads)
#pragma unroll
    for (int k = 0; k < TS_MAX_JOBS; ++k) ji += lo >= ln.j[k].pos0 ? 1 : 0;
    uint32_t pos = lo;
    while (pos < hi) {
        if (pos >= ln.j[ji + 1].pos0) {
            ++ji;
            continue;
        }
        const TsJobLine &L = ln.j[ji];
        const TnJob &job = tb.j[ji];
        const uint32_t off = pos - L.pos0, g = off / L.gcost, r = off - g * L.gcost;
        const uint32_t col = min(r / L.ucost, (uint32_t)L.ncols - 1u);
        uint32_t u0, u1;
        ts_unit_span(L, g, col, u0, u1);
        const bool lastc = col + 1 == L.ncols;
        const uint32_t tc = lastc ? L.tc_last : L.tc, vcost = ln.ovh + L.gt * tc;
        const uint32_t nvid = L.flat ? 1u : (uint32_t)L.vg;
        // tiles of the unit on its own line: [q0, q1) of nvid x gt (a flat job's last panel may be short: its missing tiles are priced, not run).
        // An ALIGNED unit (pad) is exactly one share: whoever holds any of it holds all of it.
        const uint32_t q0 = (pos > u0 && !L.aligned) ? ts_qmap(vcost, tc, ln.ovh, L.gt, pos - u0) : 0u;
        const uint32_t q1 = (hi >= u1 || L.aligned) ? nvid * L.gt : ts_qmap(vcost, tc, ln.ovh, L.gt, hi - u0);
        // (a share that touches the unit without holding one of its tiles still writes its -- zero -- slab: the reduction reads every visitor's)
        const long si = (long)ts_slab_index(ln, ji, g, col, sh);
        float *slab = ln.ct.slabs + si * (128 * 256);
        float *bias = ln.ct.bias ? ln.ct.bias + si * 256 : nullptr;
        const int kc2 = (int)col;
        using T = std::true_type;
        using F = std::false_type;
        if (L.flat) {
            // ONE run: the batch as one long video (no taps: no padding at the video ends), rows [panel start + q0, + q1)
            const int flat_rows = (int)L.vg * job.p.Trows;                   // (vg = B for a flat job)
            const uint32_t t0 = g * L.gt + q0, t1 = min(g * L.gt + q1, L.tiles);
            TnParams pf = job.p;
            pf.Trows = pf.Tx = flat_rows;
            f32x16 acc[4];
            float bsum[2];
            ts_zero(acc, bsum);
            if (t0 < t1) {
                // the staggered schedule with the accumulators zeroed in front of it: with accumulators that are alive through the body's prologue (a
                // second run of a unit) the compiler spilled the X tiles in flight -- loads waited for and parked in scratch inside the tile
                // loop, 277 us per share instead of 168 (tools/ts_runs_times.py; the schedule prices these columns at the staggered tile cost)
                if (t1 - t0 >= 8 && tb.st_min_steps > 0 && !job.x0_act)
                    ts_body_st<false, false>(pf, kc2, 0, (int)t0 * 32, min((int)t1 * 32, flat_rows), false, false, ts_smem, (int)w, acc, bsum);
                else
                    ts_body<false, false>(pf, kc2, 0, (int)t0 * 32, min((int)t1 * 32, flat_rows), false, job.x0_act != 0, ts_smem, (int)w, acc, bsum);
            }
            ts_flush<false>(pf, kc2, false, ts_smem, acc, bsum, slab, 256, bias);
        } else {
            // the lock-step body, chosen per unit, each choice its own zero / runs / write-out sequence (one loop around all bodies kept the
            // accumulators alive across the ones that do not run: 162 spilled registers)
            const uint32_t tv = L.gt, b0 = g * L.vg;
            auto unit = [&](auto TWO, auto DRP) {
                constexpr bool TWO_G = decltype(TWO)::value, DROP = decltype(DRP)::value;
                f32x16 acc[4];
                float bsum[2];
                ts_zero(acc, bsum);
                for (uint32_t q = q0; q < q1;) {
                    const uint32_t b = q / tv, t0 = q - b * tv, t1 = min(tv, t0 + (q1 - q));
                    ts_body<TWO_G, DROP>(job.p, kc2, (int)(b0 + b), (int)t0 * 32, min((int)t1 * 32, job.p.Trows), job.dual != 0, !TWO_G && job.x0_act != 0,
                                         ts_smem, (int)w, acc, bsum);
                    q += t1 - t0;
                }
                ts_flush<TWO_G>(job.p, kc2, job.dual != 0, ts_smem, acc, bsum, slab, 256, bias);
            };
            if (job.dual && 2 * kc2 + 1 == job.p.nk0) {
                if (job.p.drop.thresh) unit(T{}, T{});
                else unit(T{}, F{});
            } else unit(F{}, F{});
        }
        pos = u1;
    }
#if CLK_STAMP
    __syncthreads();
    if (threadIdx.x == 0 && sh < 4096) {
        g_clk_wg[sh][0] = wg_t0_;
        g_clk_wg[sh][1] = __builtin_amdgcn_s_memrealtime();
        g_clk_ph[sh][0] = __builtin_amdgcn_s_memtime() - wg_c0_;
        g_clk_ph[sh][1] = 1;
    }
#endif
}

// Cost units (1/32 us; MUCON_TS_COSTS = "staggered,plain,two-image,video"): a 32-step tile of a staggered single-image column (first_conv's), of a
// lock-step single-image column, of a column that stages two gradient images and replays the dropout mask, and what a run pays for entering a video
// (prologue: one memory round trip + the first image + the first X tile through LDS) -- the least-squares fit of the 256 shares' lives at the bench
// shape (tools/ts_runs_times.py, profiles/r06_weight_gradient_runs.txt: 2.14 / 2.32 / 2.97 us per tile, 3.4 us per run)
extern int g_ts_cost[4];
extern int g_ts_group_rows;   // a residual layer's groups are single videos when a video has at least this many rows, else the whole batch (MUCON_TS_GROUP_ROWS)
// The line of work of a queued batch for at most maxG workgroups.  Host only.
struct TsSchedule {
    TsLine ln;
    int nslabs;   // slabs the launch writes
};
static bool ts_make_schedule(const TnBatch &lb, int B, int maxG, TsSchedule &sc) {
    TsLine &ln = sc.ln;
    memset(&ln, 0, sizeof(ln));
    ln.ovh = (uint32_t)g_ts_cost[3];
    ln.njobs = lb.njobs;
    if (lb.njobs > TS_MAX_JOBS || B > 255) return false;
    // pass 1: tiles and tile costs (the panel length of the flat jobs wants the share, the share wants the line: estimated without the run overheads)
    uint64_t tiles = 0, work = 0;
    for (int i = 0; i < lb.njobs; ++i) {
        const TnJob &job = lb.j[i];
        const int nkc2 = (job.nkc + 1) >> 1;
        if (nkc2 > 255) return false;
        TsJobLine &L = ln.j[i];
        // no taps, one gradient set, videos back to back in both operands: the batch is one video of B * Trows rows
        L.flat = (!job.dual && job.p.taps != 3 && job.p.Tx == job.p.Trows && job.p.x_bstride == (long)job.p.Tx * job.p.ldx &&
                  (long)B * job.p.Trows < (1L << 30)) ? 1 : 0;
        L.ncols = (uint8_t)nkc2;
        const uint32_t tv = (uint32_t)(((L.flat ? B : 1) * (long)job.p.Trows + 31) >> 5);
        L.tiles = tv;
        // (a column with the non-linearity on its X operand -- last_conv's -- masks every tile: 2.7 us per tile on either body, priced like a two-image column)
        const bool stag = L.flat && tv >= 8 && lb.st_min_steps > 0 && !job.x0_act;
        L.tc = (uint16_t)(job.x0_act ? g_ts_cost[2] : stag ? g_ts_cost[0] : g_ts_cost[1]);
        L.tc_last = (uint16_t)(job.dual ? g_ts_cost[2] : L.tc);
        const uint64_t nv = L.flat ? 1 : B;
        tiles += nv * tv * nkc2;
        work += nv * tv * ((uint64_t)(nkc2 - 1) * L.tc + L.tc_last);
    }
    // at least four tiles per share (a share pays ~6 us of prologue and write-out whatever its length)
    int G = (int)std::min<uint64_t>((uint64_t)maxG, std::max<uint64_t>(1, tiles / 4));
    // pass 2: groups, unit costs, positions.  A multi-column flat job at the head of the line (first_conv: 8 columns that all read the same gradient
    // rows, 128 B of them per 2 KB of tape) is ALIGNED: its units are exactly one share each -- panel p, column c = share p * ncols + c -- so the
    // columns of a panel start together on neighbouring shares of one XCD (ts_share_of) and walk the panel's gradient rows in step: one fetch
    // serves all of them.  (Units that drift against the shares desynchronise within a few panels: a share's rows are ~1.2 MB, an XCD's L2 turns over
    // every ~10 us.)  The share S is then what the REST of the line needs: S = rest / (G - aligned shares), and an aligned unit is PRICED at S.
    uint64_t aligned_shares = 0, aligned_work = 0;
    for (int i = 0; i < lb.njobs; ++i) {
        TsJobLine &L = ln.j[i];
        if (i == 0 && L.flat && L.ncols > 1 && G >= 64) {
            const uint64_t w_i = (uint64_t)L.tiles * L.ncols * L.tc;
            uint64_t panels = (w_i * G / std::max<uint64_t>(work, 1) + L.ncols / 2) / L.ncols;     // shares this job deserves / columns, rounded
            panels = std::min<uint64_t>(std::max<uint64_t>(panels, 1), (uint64_t)(G - 1) / L.ncols);
            if (panels >= 1 && L.tiles / panels >= 8) {
                L.ngroups = (uint32_t)panels;
                L.gt = (uint32_t)((L.tiles + panels - 1) / panels);
                L.aligned = 1;   // aligned
                aligned_shares = panels * L.ncols;
                aligned_work = w_i;
            }
        }
    }
    const uint64_t s_est = std::max<uint64_t>(1, (work - aligned_work) / std::max<uint64_t>(1, (uint64_t)G - aligned_shares));
    uint64_t pos = 0, rest = 0;
    for (int pass = 0; pass < 2; ++pass) {   // (first the rest of the line -> S, then the positions)
        pos = 0;
        for (int i = 0; i < lb.njobs; ++i) {
            const TnJob &job = lb.j[i];
            TsJobLine &L = ln.j[i];
            if (L.aligned) {                    // aligned: a unit = a share
                L.vg = (uint8_t)B;
                L.ucost = L.ucost_last = pass ? ln.S : 0;
            } else if (L.flat) {
                // a panel = about what one share holds of ONE column
                const uint64_t gt = std::max<uint64_t>(8, std::min<uint64_t>(L.tiles, s_est > ln.ovh ? (s_est - ln.ovh) / L.tc : 8));
                L.gt = (uint32_t)gt;
                L.ngroups = (uint32_t)((L.tiles + gt - 1) / gt);
                L.vg = (uint8_t)B;   // (the batch: the kernel's row count of the flat video)
                L.ucost = ln.ovh + L.gt * L.tc;
                L.ucost_last = ln.ovh + L.gt * L.tc_last;
            } else {
                L.gt = L.tiles;
                L.vg = (uint8_t)((job.p.Trows >= g_ts_group_rows || B == 1) ? 1 : B);
                L.ngroups = (uint32_t)(B / L.vg);
                L.ucost = L.vg * (ln.ovh + L.gt * L.tc);
                L.ucost_last = L.vg * (ln.ovh + L.gt * L.tc_last);
            }
            L.gcost = (uint32_t)(L.ncols - 1) * L.ucost + L.ucost_last;
            L.pos0 = (uint32_t)pos;
            pos += (uint64_t)L.ngroups * L.gcost;
            if (pos >= 0xf0000000ull) return false;   // (33 M frames x 16 columns fit; anything longer takes the per-item launch)
        }
        if (pass == 0) {
            rest = pos;
            ln.S = (uint32_t)std::max<uint64_t>(1, (rest + (G - aligned_shares) - 1) / ((uint64_t)G - aligned_shares));
        }
    }
    ln.W = (uint32_t)pos;
    for (int k = lb.njobs; k <= TS_MAX_JOBS; ++k) ln.j[k].pos0 = k == lb.njobs ? ln.W : 0xffffffffu;
    ln.G = G;
    if (aligned_shares == 0) ln.S = (uint32_t)((pos + G - 1) / G);
    uint32_t ns = 0;   // a column's visits take consecutive slabs
    for (int i = 0; i < lb.njobs; ++i) {
        const TsJobLine &L = ln.j[i];
        if (L.ncols > TS_MAX_NCOLS) return false;
        for (uint32_t c = 0; c < L.ncols; ++c) {
            uint32_t n = 0, wf, wl;
            for (uint32_t g = 0; g < L.ngroups; ++g) {
                ts_unit_visitors(L, ln.S, g, c, wf, wl);
                n += wl - wf + 1;
            }
            if (ns + n > 65000) return false;
            ln.ct.slab0[i][c] = (uint16_t)ns;
            ln.ct.n[i][c] = (uint16_t)n;
            ns += n;
        }
    }
    sc.nslabs = (int)ns;
    return true;
}

// jobs were queued coarse levels first; the fine levels have the longest workgroups: lay them out first
static void ts_layout(const TnBatch &tb, TnBatch &lb) {
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
    }
    lb.nblocks = blocks;
    lb.xcd_order = 0;
    lb.st_min_steps = g_ts_stagger;
}
static hipError_t ts_smem_attr(const void *k, bool &done) {
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, TS_SMEM_BYTES);
    done = e == hipSuccess;
    return e;
}
static hipError_t launch_ts_batch(TnBatch &tb, hipStream_t s) {   // one workgroup per item
    if (tb.njobs == 0) return hipSuccess;
    static bool attr = false;
    hipError_t e = ts_smem_attr(reinterpret_cast<const void *>(ts_batched_kernel), attr);
    if (e != hipSuccess) return e;
    TnBatch lb;
    ts_layout(tb, lb);
    hipLaunchKernelGGL(ts_batched_kernel, dim3(lb.nblocks), dim3(512), TS_SMEM_BYTES, s, lb);
    tb.njobs = 0;
    return hipGetLastError();
}
static hipError_t launch_ts_runs(const TnBatch &lb, const TsSchedule &sc, hipStream_t s) {
    static bool attr = false;
    hipError_t e = ts_smem_attr(reinterpret_cast<const void *>(ts_runs_kernel), attr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ts_runs_kernel, dim3(sc.ln.G), dim3(512), TS_SMEM_BYTES, s, lb, sc.ln);
    return hipGetLastError();
}
