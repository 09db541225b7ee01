// TN GEMM core (weight gradients) on the f32-input MFMA:
//
//     dW[n][k] = sum_{b,t} Y[b][t][n] * X'[b][t][k]        n in [0,128), k in one 128-wide chunk
//
// i.e. the reduction runs over time (rows of both operands), which is the autograd of every
// nn.Conv1d of the reference's encoder (src/core/modules/temporal.py:23-32,108-114) w.r.t. its
// weight.  X' is the layer input gathered with the conv's tap offset (zero padded), the tape for
// first_conv, or the activated input of last_conv.  grid = (k-chunks, time-chunks); every
// workgroup writes its partial 128x128 tile to a slab [time-chunk][128][Ktot]; reduce_slabs_kernel
// then sums the slabs in a fixed order (bitwise reproducible -- no float atomics) and writes the
// gradient in the reference's [out][in][k] layout.
//
// LDS tiles are [32 time steps][128 channels] exactly as in HBM (512-byte rows, no padding
// needed: the MFMA operand read is ds_read_b32 with 32 consecutive lanes on consecutive floats).
#pragma once
#include <climits>

#include "common.hpp"
#include "dispatch.hpp"

constexpr int TN_SMEM_BYTES = 2 * 2 * 32 * 128 * 4;  // Y and X tiles, double buffered = 64 KiB

// One launch covers every weight gradient of a residual layer: k-chunks [0, nk0) use operand set 0
// (Y0 = gradient at the dilated conv's pre-activation, X0 = the layer input gathered per tap ->
// dilated_conv.weight), and with DUAL the extra k-chunk nk0 uses set 1 (Y1 = gradient at the layer
// output with the dropout mask replayed, X1 = the activated dilated-conv output -> conv_1x1.weight).
struct TnParams {
    int Trows;
    const float *Y0;  // [B][Trows][128]
    const float *X0;  // [B][Tx][ldx]
    long x_bstride;
    int ldx, Tx;
    int taps;         // 3: k-chunk kc is tap kc (row offset (kc-1)*tap_step), columns 0..127
    int tap_step;     // 1: k-chunk kc is columns kc*128.. of X0
    int nk0;          // k-chunks of set 0
    const float *Y1;  // DUAL: [B][Trows][128], multiplied by the dropout mask
    const float *X1;  // DUAL: [B][Trows][128]
    int Ktot;         // slab row length = 128 * (nk0 + DUAL)
    float *slabs;       // [n_time_chunks][128][Ktot]
    float *bias_slabs;  // [n_time_chunks][2][128]: column sums of Y0 (and of Y1 with DUAL)
    int MC;           // time steps per chunk (multiple of 32)
    int chunks_per_video;
    float slope;
    DropCfg drop;     // set 1: Y1 element index (b*Trows + t)*128 + n
};

// KS = 2: 512 threads = 8 waves; waves 4-7 multiply the second half (time steps 8-15 of each lane's 16) of every
// 32-step m-tile into their own accumulators, which are added to those of waves 0-3 through LDS before the slab is
// written.  A weight-gradient launch of a layer has one workgroup per CU (more workgroups = more slab traffic), i.e.
// ONE wave per SIMD with KS = 1: every LDS-read latency and barrier is exposed (MFMA pipe 38 % busy at B=8 x T=4096);
// the second wave per SIMD fills those gaps without adding slabs.
template <bool X0_ACT, bool DUAL, int KS>
__device__ __forceinline__ void tn_body(const TnParams &p, const int kc, const int mc, float *smem) {
    constexpr int NTHR = 256 * KS;
    constexpr int NQ = 4 / KS;           // float4 loads per thread and operand per m-tile
    float *Ys = smem;
    float *Xs = smem + 2 * 32 * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgrp = wave_all >> 2;
    const int wave = wave_all & 3;
    const int wr = wave >> 1, wc = wave & 1;
    const int b = mc / p.chunks_per_video;
    const int tbeg = (mc - b * p.chunks_per_video) * p.MC;
    const int tend = min(tbeg + p.MC, p.Trows);
    const int ntiles = (tend - tbeg + 31) >> 5;
    const bool second = DUAL && kc >= p.nk0;   // workgroup-uniform
    const int xoff = (!second && p.taps == 3) ? (kc - 1) * p.tap_step : 0;
    const int xcol = (second || p.taps == 3) ? 0 : kc * 128;
    const float *Yb = (second ? p.Y1 : p.Y0) + (long)b * p.Trows * 128;
    const float *Xb = second ? p.X1 + (long)b * p.Trows * 128 : p.X0 + (long)b * p.x_bstride + xcol;
    const int ldx = second ? 128 : p.ldx;
    const int Tx = second ? p.Trows : p.Tx;

    f32x4 ry[NQ], rx[NQ];
    int rty[NQ];      // time step of the staged Y row, -1 = padding
    bool rokx[NQ];
    // loads are only ISSUED here (always, from clamped rows); zeroing, dropout replay and the ReLU prologue are
    // applied at the LDS store one m-tile later, so the loads stay in flight under the MFMAs
    auto gload = [&](int mtile) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = tid + NTHR * q;
            const int row = f >> 5, c4 = (f & 31) * 4;
            const int t = tbeg + mtile * 32 + row;
            const int ts = t + xoff;
            const bool oky = t < tend;
            rty[q] = oky ? t : -1;
            rokx[q] = oky && ts >= 0 && ts < Tx;
            const int tyc = t < p.Trows ? t : p.Trows - 1;
            const int txc = ts < 0 ? 0 : (ts >= Tx ? Tx - 1 : ts);
            ry[q] = *reinterpret_cast<const f32x4 *>(Yb + (long)tyc * 128 + c4);
            rx[q] = *reinterpret_cast<const f32x4 *>(Xb + (long)txc * ldx + c4);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = tid + NTHR * q;
            const int c4 = (f & 31) * 4;
            f32x4 y = ry[q], x = rx[q];
            if (DUAL) {
                if (second && p.drop.thresh) {
                    const uint32_t idx = (uint32_t)(b * p.Trows + rty[q]) * 128u + (uint32_t)c4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] *= drop_mul(p.drop, idx + e);
                }
            }
            if (X0_ACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = act_f(x[e], p.slope);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = rty[q] >= 0 ? y[e] : 0.f;
                x[e] = rokx[q] ? x[e] : 0.f;
            }
            // rows 16..31 are rotated by 32 columns: the two lane halves of an MFMA operand read (rows s and 16 + s)
            // then hit disjoint bank halves instead of colliding 2-way
            const int sw = (f >> 5) * 128 + ((c4 + ((f >> 9) & 1) * 32) & 127);
            *reinterpret_cast<f32x4 *>(Ys + buf * 4096 + sw) = y;
            *reinterpret_cast<f32x4 *>(Xs + buf * 4096 + sw) = x;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (p.bias_slabs != nullptr) && (kc == 0 || (DUAL && kc == p.nk0)) && (tid < 128);

    // lane (i = lane&31, h = lane>>5): MFMA step s consumes time step 16h + s of the tile.
    const int hrow = (lane >> 5) * 16 * 128, hrot = (lane >> 5) * 32;
    const int y_off0 = hrow + ((wr * 64 + (lane & 31) + hrot) & 127), y_off1 = hrow + ((wr * 64 + 32 + (lane & 31) + hrot) & 127);
    const int x_off0 = hrow + ((wc * 64 + (lane & 31) + hrot) & 127), x_off1 = hrow + ((wc * 64 + 32 + (lane & 31) + hrot) & 127);

    // ntiles >= 1.  Guard-free loop body (the tail re-loads the last tile into the buffer nobody reads) with the
    // phases pinned: loads issued first, MFMAs, then the first use of the loaded data at the LDS store -- so the
    // loads of m-tile mt+1 are in flight under the 64 MFMAs of tile mt and hipcc can count its vmcnt waits.
    gload(0);
    sstore(0);
    __syncthreads();
    for (int mt = 0; mt < ntiles; ++mt) {
        const int cur = mt & 1;
        gload(min(mt + 1, ntiles - 1));
        __builtin_amdgcn_sched_barrier(0);
        const float *Yw0 = Ys + cur * 4096 + y_off0, *Yw1 = Ys + cur * 4096 + y_off1;
        const float *Xw0 = Xs + cur * 4096 + x_off0, *Xw1 = Xs + cur * 4096 + x_off1;
#pragma unroll
        for (int s0 = 0; s0 < 16 / KS; ++s0) {
            const int s = KS == 2 ? kgrp * 8 + s0 : s0;
            const float a0 = Yw0[s * 128], a1 = Yw1[s * 128];
            const float b0 = Xw0[s * 128], b1 = Xw1[s * 128];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (do_bias) {
            const float *Yc = Ys + cur * 4096;
#pragma unroll 8
            for (int m = 0; m < 32; ++m) bsum += Yc[m * 128 + ((tid + (m >> 4) * 32) & 127)];
        }
        __builtin_amdgcn_sched_barrier(0);
        sstore(cur ^ 1);
        __syncthreads();
    }

    if (KS == 2) {   // the tiles are dead after the loop's last barrier: 64 KiB of LDS carry the second half's sums
        float *xch = smem + ((wave * 4) * 16) * 64 + lane;
        if (kgrp == 1) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) xch[((mt * 2 + nt) * 16 + reg) * 64] = acc[mt][nt][reg];
        }
        __syncthreads();
        if (kgrp == 1) return;   // (KS = 2 exists only in the one-job kernel: nothing follows the body there)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) acc[mt][nt][reg] += xch[((mt * 2 + nt) * 16 + reg) * 64];
    }
    float *slab = p.slabs + (long)mc * 128 * p.Ktot + kc * 128;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = wc * 64 + nt * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = wr * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                slab[(long)row * p.Ktot + col] = acc[mt][nt][reg];
            }
        }
    if (do_bias) p.bias_slabs[(long)mc * 256 + (second ? 128 : 0) + tid] = bsum;
}

template <bool X0_ACT, bool DUAL, int KS>
__global__ __launch_bounds__(256 * KS) void tn_gemm_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    tn_body<X0_ACT, DUAL, KS>(p, blockIdx.x, blockIdx.y, smem);
}

// Every weight gradient of the residual layers (and last_conv's) in ONE launch at the end of the data-gradient chain:
// a launch per layer costs ~13 us of ramp, prologue and slab write-out each, and on the coarse levels has too few
// workgroups to fill the chip.  Jobs are laid out longest-first; a workgroup finds its job by its block index.
constexpr int TN_MAX_BATCH = 16;
struct TnJob {
    TnParams p;
    int block0, nkc;      // first block of the job, k-chunks per time chunk
    int x0_act, dual;
    int nmc;              // time chunks (gemm_tn_split.hpp)
};
struct TnBatch {
    TnJob j[TN_MAX_BATCH];
    int first_block[TN_MAX_BATCH];   // j[i].block0 once more, contiguous (unused entries: INT_MAX): gemm_tn_split.hpp's job lookup
    int njobs, nblocks;
    int xcd_order;        // gemm_tn_split.hpp
    int st_min_steps;     // gemm_tn_split.hpp: jobs with time chunks of at least this many steps run the staggered schedule
};
template <int KS>
__global__ __launch_bounds__(256 * KS) void tn_batched_kernel(const TnBatch tb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int ji = 0;
    while (ji + 1 < tb.njobs && (int)blockIdx.x >= tb.j[ji + 1].block0) ++ji;
    const TnJob &job = tb.j[ji];
    const int local = blockIdx.x - job.block0;
    const int mc = local / job.nkc, kc = local - mc * job.nkc;
    if (job.dual) tn_body<false, true, KS>(job.p, kc, mc, smem);
    else if (job.x0_act) tn_body<true, false, KS>(job.p, kc, mc, smem);
    else tn_body<false, false, KS>(job.p, kc, mc, smem);
}
static hipError_t launch_tn_batch(TnBatch &tb, hipStream_t s) {
    if (tb.njobs == 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tn_batched_kernel<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(tn_batched_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    // jobs were queued coarse levels first; the fine levels have the longest workgroups: lay them out first
    TnBatch lb;
    lb.njobs = tb.njobs;
    int blocks = 0;
    for (int i = 0; i < tb.njobs; ++i) {
        lb.j[i] = tb.j[tb.njobs - 1 - i];
        lb.j[i].block0 = blocks;
        blocks += lb.j[i].nkc * (tb.j[tb.njobs - 1 - i].block0);   // block0 carried the time-chunk count while queued
    }
    lb.nblocks = blocks;
    if (kTnBatchKs == 2) hipLaunchKernelGGL(tn_batched_kernel<2>, dim3(blocks), dim3(512), TN_SMEM_BYTES, s, lb);
    else hipLaunchKernelGGL(tn_batched_kernel<1>, dim3(blocks), dim3(256), TN_SMEM_BYTES, s, lb);
    tb.njobs = 0;
    return hipGetLastError();
}

template <bool X0_ACT, bool DUAL, int KS>
static hipError_t launch_tn(const TnParams &p, int B, hipStream_t s) {
    auto k = tn_gemm_kernel<X0_ACT, DUAL, KS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(p.Ktot / 128, B * p.chunks_per_video);
    hipLaunchKernelGGL(k, grid, dim3(256 * KS), TN_SMEM_BYTES, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Batched slab reduction: ONE launch sums every gradient's partial slabs of a backward pass, in a
// fixed order (bitwise reproducible -- no float atomics).  64*G threads = G slab lanes x 64
// float4 elements: lane group g sums slabs g, g+G, ... (four independent loads in flight per thread),
// then the G partials are combined through LDS in lane-group order.
//   element (row, col) of a job reads slabs[s*slab_stride + row*ld + coff + col]
//   mode 0: out[row*ncols + col]      mode 1 (conv k=3 weight, ncols = 384): col = tap*128 + i ->
//   out[(row*128 + i)*3 + tap], the reference's [out][in][k] layout
// ------------------------------------------------------------------------------------------
//   sched >= 0 (round 6: the partial tiles of the static-runs weight-gradient launch, gemm_tn_split.hpp): the job's columns lie in the 256-column
//   blocks of job `sched` of the launch's line of work (TsLine); block cb of group g was visited by the shares floor(unit start / S) ..
//   floor((unit end - 1) / S), each of which left one partial tile [128][256] (a bias job: 256 floats); a column's visits take consecutive
//   slabs (TsColTab: first slab and count per column)
constexpr int REDUCE_MAX_JOBS = 56;

// ---- the line of work of the static-runs launch (gemm_tn_split.hpp), shared with the reduction --------------------------------------------
// A job's work is cut into UNITS = (group g, column c): a group is a stretch of time -- `vg` whole videos, or for a job without taps over
// back-to-back videos (flat: first_conv, last_conv) a panel of `gt` 32-step tiles of the batch taken as one long video -- and a column is a
// pair of 128-column chunks kc2.  The line runs job by job, inside a job group by group, inside a group column by column: the columns of a
// group read the SAME gradient rows (and, for the taps of a residual layer, the same input rows), so the shares that work on them at the same time
// sit next to each other on the line -- and on one XCD (ts_share_of: eight consecutive shares per XCD), where the second reader hits the L2.
// (Column-major lines -- a column through all videos, then the next -- re-read every shared operand from memory: 800 MB of HBM traffic per
// launch for 531 MB of operands, profiles/r06_traffic.json first cut.)
struct TsJobLine {
    uint32_t pos0;          // start of the job on the line (cost units)
    uint32_t ngroups;
    uint32_t gcost;         // cost of a group = (ncols - 1) * ucost + ucost_last
    uint32_t ucost;         // cost of a unit of a plain column = nvid * (ovh + gt * tc)
    uint32_t ucost_last;    // ... of the job's last column (a residual layer's: two gradient images, dropout replay)
    uint32_t gt;            // tiles per video of a unit (flat: per panel)
    uint32_t tiles;         // flat: tiles of the whole batch (the last panel may be short)
    uint16_t tc, tc_last;   // cost of a 32-step tile
    uint8_t ncols, flat, vg, aligned;   // aligned: every unit of the job is exactly one share (gemm_tn_split.hpp: ts_make_schedule)
};
constexpr int TS_MAX_JOBS = 16;   // = TN_MAX_BATCH
constexpr int TS_MAX_NCOLS = 16;  // columns of a job (first_conv at D = 2048: 8)
// where the partial tiles of a column lie: its visits (group by group, inside a group share by share) take consecutive slabs from slab0 on
struct TsColTab {
    uint16_t slab0[TS_MAX_JOBS][TS_MAX_NCOLS], n[TS_MAX_JOBS][TS_MAX_NCOLS];
    float *slabs, *bias;            // arenas: [slab][128][256], [slab][256]
};
struct TsLine {
    TsJobLine j[TS_MAX_JOBS + 1];   // [njobs].pos0 = W: the end of the line; unused: 0xffffffff
    uint32_t W, S, ovh;             // length of the line, share per workgroup, fixed cost of entering a video
    int njobs, G;
    TsColTab ct;
};
// share -> workgroup: eight consecutive shares on one XCD (workgroups are dealt to the 8 XCDs round-robin: block b and b + 8 share one --
// observed, used for speed only), while every run of 64 workgroups still covers 64 consecutive shares
__device__ __host__ __forceinline__ uint32_t ts_share_of(const uint32_t w, const uint32_t G) {
    if (w >= (G & ~63u)) return w;
    return (w & ~63u) | ((w & 7u) << 3) | ((w >> 3) & 7u);
}
// unit (g, c) of a job: [start, end) on the line
__device__ __host__ __forceinline__ void ts_unit_span(const TsJobLine &J, const uint32_t g, const uint32_t c, uint32_t &u0, uint32_t &u1) {
    u0 = J.pos0 + g * J.gcost + c * J.ucost;
    u1 = u0 + (c + 1 == J.ncols ? J.ucost_last : J.ucost);
}
// visits of unit (g, c): the shares floor(start / S) .. floor((end - 1) / S)
__device__ __host__ __forceinline__ void ts_unit_visitors(const TsJobLine &J, const uint32_t S, const uint32_t g, const uint32_t c, uint32_t &wf, uint32_t &wl) {
    uint32_t u0, u1;
    ts_unit_span(J, g, c, u0, u1);
    wf = u0 / S;
    wl = (u1 - 1) / S;
}
// the slab share `sh` leaves for unit (g, c) of job ji: the column's visits in line order (an aligned job: one visit per group)
__device__ __host__ __forceinline__ uint32_t ts_slab_index(const TsLine &ln, const int ji, const uint32_t g, const uint32_t c, const uint32_t sh) {
    const TsJobLine &J = ln.j[ji];
    uint32_t v = g;
    if (!J.aligned) {
        v = 0;
        uint32_t wf, wl;
        for (uint32_t gg = 0; gg < g; ++gg) {
            ts_unit_visitors(J, ln.S, gg, c, wf, wl);
            v += wl - wf + 1;
        }
        ts_unit_visitors(J, ln.S, g, c, wf, wl);
        v += sh - wf;
    }
    return (uint32_t)ln.ct.slab0[ji][c] + v;
}

struct ReduceJob {
    const float *slabs;
    float *out;
    long slab_stride;
    int nslabs, ld, coff, ncols, n_elems;
    int block0;      // first workgroup of this job
    int8_t mode;
    int8_t vec;      // 1: ncols, coff, ld, slab_stride and n_elems are multiples of 4 (float4 path)
    int8_t sched;    // -1: slabs / nslabs / slab_stride / ld above; else the job of `line`
    int8_t isbias;   // ... the 256-float bias partials of column `bcol`
    int8_t bcol;
};
struct ReduceBatch {
    ReduceJob j[REDUCE_MAX_JOBS];
    int first_block[REDUCE_MAX_JOBS];   // j[i].block0 once more, contiguous (unused entries: INT_MAX): see reduce_batch_kernel
    int njobs, nblocks;
    TsColTab ct;                        // the static-runs launch's columns (jobs with sched >= 0)
};

// (r6) U consecutive 256-element chunks of a job per workgroup (U = 1: one).  The big pass of a training step is ~3,900 chunks whose workgroups live ~3 us, a
// fifth of it the two scalar-memory round trips in front of the first load (argument block, then the job's record): with U = 4 a workgroup pays them once for four
// chunks and keeps four times the loads in flight.  Chunk and slab-lane assignment per ELEMENT are unchanged: the same sums in the same order.
template <int G, int U = 1>
__global__ __launch_bounds__(64 * G) void reduce_batch_kernel(const ReduceBatch rb) {
    // a workgroup owns U x 256 consecutive elements of one job: 64 lanes x float4, G slab lanes
    __shared__ f32x4 part[G][U][64];
    // Which job this workgroup belongs to: the number of jobs that start at or before it.  All first blocks are read at once
    // (64 contiguous kernel-argument words, independent scalar loads) and compared in registers -- walking the table job by
    // job was a chain of up to ~50 dependent scalar-cache round trips in front of every workgroup's first load (2 - 4 us each:
    // the 3,900-workgroup pass took 21 us for 8 MB).
    int ji = -1;
#pragma unroll
    for (int k = 0; k < REDUCE_MAX_JOBS; ++k) ji += (int)blockIdx.x >= rb.first_block[k] ? 1 : 0;
    const ReduceJob &J = rb.j[ji];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    f32x4 s[U];
    int row[U], col[U], e[U];
    if (!J.vec) {  // odd shapes (e.g. a class count that is not a multiple of 4): element-wise
#pragma unroll
        for (int u = 0; u < U; ++u) {
            e[u] = ((((int)blockIdx.x - J.block0) * U + u) * 64 + lane) * 4;   // ncols, coff, ld are multiples of 4
            s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            row[u] = col[u] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ek = e[u] + k;
                if (ek < J.n_elems) {
                    const int r = ek / J.ncols, c = ek - r * J.ncols;
                    const float *p = J.slabs + (long)r * J.ld + J.coff + c;
                    float a = 0.f;
                    for (int i = g; i < J.nslabs; i += G) a += p[(long)i * J.slab_stride];
                    s[u][k] = a;
                }
            }
        }
    } else {
        const float *p[U];
        int nsl[U];
        long stride[U];
        int nmax = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            e[u] = ((((int)blockIdx.x - J.block0) * U + u) * 64 + lane) * 4;
            s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int ee = e[u] < J.n_elems ? e[u] : 0;     // (a chunk past the job's end: loads of element 0, nothing stored)
            row[u] = ee / J.ncols;
            col[u] = ee - row[u] * J.ncols;
            p[u] = J.slabs + (long)row[u] * J.ld + J.coff + col[u];
            nsl[u] = J.nslabs;
            stride[u] = J.slab_stride;
            if (J.sched >= 0) {   // the static-runs launch's partial tiles: this lane's 256-column block of the job says where they are and how many
                const int gcol = J.coff + col[u], cb = J.isbias ? (int)J.bcol : gcol >> 8;
                nsl[u] = rb.ct.n[J.sched][cb];
                stride[u] = J.isbias ? 256 : 128 * 256;
                p[u] = (J.isbias ? rb.ct.bias + gcol : rb.ct.slabs + (long)row[u] * 256 + (gcol & 255)) + (long)rb.ct.slab0[J.sched][cb] * stride[u];
            }
            if (e[u] >= J.n_elems) nsl[u] = 0;
            nmax = max(nmax, nsl[u]);
        }
        if (U == 1) {
            int i = g;
            if (G >= 16) {   // few deep jobs (the y-head's 256 slabs): sixteen loads per lane in flight -- the pass is its chain of round trips
                for (; i + 15 * G < nsl[0]; i += 16 * G) {
                    f32x4 v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const f32x4 *>(p[0] + (long)(i + G * q) * stride[0]);
#pragma unroll
                    for (int q = 0; q < 16; ++q) s[0] += v[q];
                }
            }
            for (; i + 3 * G < nsl[0]; i += 4 * G) {
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4 *>(p[0] + (long)(i + G * q) * stride[0]);
#pragma unroll
                for (int q = 0; q < 4; ++q) s[0] += v[q];
            }
            for (; i < nsl[0]; i += G) s[0] += *reinterpret_cast<const f32x4 *>(p[0] + (long)i * stride[0]);
        } else {
            // one loop for the U chunks: every round requests four slabs per chunk (4 U loads in flight); a chunk with fewer slabs re-reads its last one and
            // adds nothing (the loads are unconditional: a load behind a branch would make the compiler wait for everything in flight at the join)
            for (int i = g; i < nmax; i += 4 * G) {
                f32x4 v[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = min(i + G * q, max(nsl[u] - 1, 0));
                        v[u][q] = *reinterpret_cast<const f32x4 *>(p[u] + (long)k * stride[u]);
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (i + G * q < nsl[u]) s[u] += v[u][q];
            }
        }
    }
    if (G > 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) part[g][u][lane] = s[u];
        __syncthreads();
    }
    if (g == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e[u] >= J.n_elems) continue;
            f32x4 t = s[u];
#pragma unroll
            for (int k = 1; k < G; ++k) t += part[k][u][lane];
            if (!J.vec) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e[u] + k < J.n_elems) J.out[e[u] + k] = t[k];
            } else if (J.mode == 0) {
                *reinterpret_cast<f32x4 *>(J.out + e[u]) = t;
            } else {
                const int tap = col[u] >> 7, i = col[u] & 127;
#pragma unroll
                for (int k = 0; k < 4; ++k) J.out[(row[u] * 128 + i + k) * 3 + tap] = t[k];
            }
        }
    }
}
