// Transcript-constrained Viterbi decode on gfx950 -- bit-exact against the reference's
// Viterbi.decode (src/core/viterbi/viterbi.py:49-158) with SingleTranscriptGrammar
// (src/core/viterbi/grammar.py:196-217) and an f64 length table (PoissonModel,
// src/core/viterbi/length_model.py:76-80).  Compiled with -ffp-contract=off: every score is a
// chain of single IEEE adds in the reference's order and dtype.
//
// Formulation.  A hypothesis is (n, k0): transcript position n entered at column k0 (column k
// <-> frame (k+1)*fs-1); its segment length at column k is (k-k0+1)*fs.  Instead of shifting a
// [N x J] table every column (J = max_len/fs length slots, 66 by default), slot r = k0 mod J of
// state n holds the hypothesis for its whole life: a slot is re-used exactly when its old tenant
// reaches the maximum length.  Which slots are alive is a closed form of (n, k0, k):
// entries into state n >= 1 exist for n <= k0 <= J*n, state 0 is entered at k0 = 0 only.
//
// Kernels (dispatch: vit_launch below):
//   phase 1  viterbi_framescore_cols_kernel -- the sequential float32 cumsum of the emissions (np.cumsum, viterbi.py:51) and its
//            fs-strided differences (frame_score, viterbi.py:68-72) -> F[K][C]: one wave adds (lane = class), seven stage the
//            emissions through LDS three chunks ahead.  (viterbi_framescore_kernel: the plain version for unaligned inputs / fs > 256.)
//   phase 2  viterbi_dp_lanes_kernel -- J <= 66 slots, N <= 128 states: hypothesis scores in registers, lanes = (state, slot group),
//            length-indexed slots, DPP reductions and hand-over (VitLanes).  viterbi_dp_kernel: the general LDS version (one wave per
//            transcript state, lanes = length slots, the column double buffered in LDS), also the tests' second implementation.
//   both     viterbi_fused_kernel -- one short video in one launch, phase 2 running under phase 1 in the same workgroup.
//   Back-pointers: one byte per (column, state), LDS for latency calls, HBM scratch otherwise; thread 0 walks them
//   (viterbi.py:140-158) and all threads expand the labels.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include <chrono>

#include "../../include/mucon_hip.h"
#include "../../include/mucon_hip_test.h"

namespace {

#ifndef VIT_STAMP
#define VIT_STAMP 0
#endif
#if VIT_STAMP   // timing builds (MUCON_HIPCC_FLAGS=-DVIT_STAMP=1, tools/vit_stamps.py): s_memtime of thread 0 of block 0 at the phase edges
__device__ long long g_vit_stamps[16];
#define VSTAMP(i)                                                                    \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_vit_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define VSTAMP(i) \
    do {          \
    } while (0)
#endif

constexpr int VIT_THREADS = 1024;
constexpr int VIT_FCHUNK = 16;  // columns of frame scores staged in LDS at a time

// Phase 1.  The cumulative sum is a strictly sequential float32 chain per class (np.cumsum; a parallel scan
// would round differently), so one wave adds -- lane c owns class c -- while the whole workgroup keeps it fed:
// 256 threads stream the emissions in chunks of FS_ROWS frames into a double-buffered LDS tile, wave 0
// walks the tile row by row.  Column boundaries are wave-uniform (scalar branch, no divergence).
constexpr int FS_ROWS = 256;
constexpr int FS_THREADS = 256;
typedef float vit_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(FS_THREADS) void viterbi_framescore_kernel(const mucon_viterbi_job *jobs, char *ws, int C, int fs) {
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];   // [2][FS_ROWS * C]
    const mucon_viterbi_job job = jobs[blockIdx.x];
    const int K = job.T / fs;
    if (K < 1) return;
    float *F = reinterpret_cast<float *>(ws + job.ws_off);
    const int n = K * fs;                       // frames that enter the decode
    const int tid = threadIdx.x;
    // (global address space spelled out: through the job record's generic pointer these are FLAT loads, which count against the
    // LDS wait counter as well -- see framescore_cols_body)
    typedef const __attribute__((address_space(1))) float *gptr1;
    typedef const __attribute__((address_space(1))) vit_f32x4 *gptr4v;
    const gptr1 src = (gptr1)job.lp;
    const int chunk = FS_ROWS * C;              // floats per chunk: a contiguous range of the [T][C] array
    const int nchunks = (n + FS_ROWS - 1) / FS_ROWS;
    const long total = (long)n * C;
    const bool vec = (C & 3) == 0;              // float4 path: every offset is a multiple of 4 floats
    const int nper = C >> 2;                    // float4 per thread per chunk = chunk / 4 / FS_THREADS  (<= 16)

    vit_f32x4 r[16];
    auto gload = [&](int ci) {                  // issue only; clamped so that no load needs a branch
        const long base4 = ((long)ci * chunk) >> 2;
        const long last4 = (total >> 2) - 1;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < nper) {
                long e = base4 + u * FS_THREADS + tid;
                e = e < last4 ? e : last4;
                r[u] = ((gptr4v)src)[e];
            }
    };
    auto sstore = [&](int buf) {
        vit_f32x4 *dst = reinterpret_cast<vit_f32x4 *>(fs_smem + buf * chunk);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < nper) dst[u * FS_THREADS + tid] = r[u];
    };
    auto copy_scalar = [&](int ci, int buf) {   // odd class counts: plain element-wise staging
        float *dst = fs_smem + buf * chunk;
        const long base = (long)ci * chunk;
        for (int i = tid; i < chunk; i += FS_THREADS) {
            const long e = base + i;
            dst[i] = src[e < total ? e : total - 1];
        }
    };

    // -0.0f is the additive identity of IEEE float addition for EVERY x (incl. both zeros), so starting the chain
    // there reproduces cs[0] = lp[0], cs[t] = cs[t-1] + lp[t] bit for bit without a first-row special case
    float run = -0.0f, prev = 0.f;
    int until = fs, k = 0;
    if (vec) {
        gload(0);
        sstore(0);
    } else {
        copy_scalar(0, 0);
    }
    __syncthreads();
    for (int ci = 0; ci < nchunks; ++ci) {
        const int cur = ci & 1;
        const int nxt = ci + 1 < nchunks ? ci + 1 : ci;
        if (vec) gload(nxt);
        if (tid < 64) {   // wave 0: the sequential chain, lane = class; column bookkeeping is wave-uniform
            const float *col = fs_smem + cur * chunk + (tid < C ? tid : C - 1);
            const int rows = min(FS_ROWS, n - ci * FS_ROWS);
            int rr = 0;
            while (rr < rows) {
                const int take = min(until, rows - rr);   // rows of the current column inside this chunk
                int q = 0;
                for (; q + 6 <= take; q += 6) {
                    float v[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) v[u] = col[(rr + q + u) * C];
#pragma unroll
                    for (int u = 0; u < 6; ++u) run = run + v[u];      // sequential float32 chain
                }
                for (; q < take; ++q) run = run + col[(rr + q) * C];
                rr += take;
                until -= take;
                if (until == 0) {                                      // end of column k
                    if (tid < C) F[(long)k * C + tid] = (k == 0) ? run : run - prev;
                    prev = run;
                    until = fs;
                    ++k;
                }
            }
        }
        if (vec) sstore(cur ^ 1);
        else copy_scalar(nxt, cur ^ 1);
        __syncthreads();
    }
}

// The same chain at its own speed (fs <= 256, C a multiple of 4).  One wave issues an instruction every ~4-5 cycles and a
// dependent float32 add returns after ~7 (tools/chain_probe.hip: 2.8 ns per add from registers), so what a row costs is the
// number of instructions wave 0 spends on it; the kernel above spends a read, an address and a wait on every add.  Here
//   * a chunk holds WHOLE columns and lies TRANSPOSED in LDS ([class][row], each column padded to a multiple of four rows
//     with -0.0f, the additive identity of IEEE addition for every x including both zeros: the padded adds change no bit), so
//     lane c fetches four consecutive rows of class c with one ds_read_b128 at an immediate offset, eight reads in flight;
//   * wave 0 does nothing but add: after every read's four adds it drops the running sum into LDS (one ds_write_b32, no
//     column bookkeeping, no branch), and its loop body is 16 adds, 4 reads, 4 writes;
//   * waves 1..7 do the rest: global float4 -> four transposed LDS words one chunk ahead of the chain (the global loads two
//     chunks ahead), and the fs-strided differences of the previous chunk's running sums -> F.
constexpr int FSC_THREADS = 512;
constexpr int FSC_MAX_LDS = 160 * 1024;
constexpr int FSC_DEPTH = 8;                 // b128 reads in flight
constexpr int FSC_SETS = 3;                  // register sets of staged global loads (chunks in flight)
constexpr size_t VF_DYN_MAX = 160 * 1024 - 8 * 1024;    // dynamic LDS of the one-launch kernel: 160 KiB minus its static arrays
constexpr size_t VP_DYN_MAX = 160 * 1024 - 40 * 1024;   // dynamic LDS of the pair kernel: 160 KiB minus the decoding role's static arrays
constexpr int VF_MAX_K = 640;                           // columns of a video the one-launch kernel takes
constexpr size_t VL_BP_LDS_MAX = 96 * 1024;             // dynamic LDS of the register DP kernels: back-pointers [K][N] of a latency call
constexpr int FSC_SLACK = 8 * FSC_DEPTH;     // rows the read-ahead and the last round may run past a chunk's end
__host__ __device__ inline int fsc_pitch(int rows) { return ((rows + FSC_SLACK + 15) & ~15) + 4; }   // = 4 mod 16, in floats
__host__ __device__ inline int fsc_floats(int C, int fs, int cols) {                                 // dynamic LDS, in floats
    const int nq = (fs + 3) >> 2;
    return 2 * C * fsc_pitch(cols * nq * 4) + 2 * (cols * nq + FSC_DEPTH) * 64;
}
// Workgroup barrier for LDS traffic only (release / acquire on the LDS address space: s_waitcnt lgkmcnt(0) + s_barrier).  It never
// waits for global loads or stores in flight -- the staging waves' read-ahead, the decoding wave's length scores on their way from
// pinned host memory.  (What paced the chain until r3 were the staging waves' FLAT loads -- see gptr4 below -- which count against
// lgkmcnt as well: every barrier waited for the read-ahead just issued, one HBM latency per chunk.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
struct FscNoDp {
    __device__ __forceinline__ void operator()(int, int) const {}
};
// W4: the column length in reads is a multiple of four (fs = 29..32, the default 30): only every fourth sum can end a column.
// NSTG staging waves: 7 -- waves 1..7 stage; 6 -- wave 1 has a job of its own, once per round between the same barriers:
//   PUB = false  the one-launch kernel's DECODING wave: dp(k_lo, k_hi) on the columns whose frame scores the round before made
//                visible (two rounds behind the chain: phase 2 runs UNDER phase 1 instead of after it);
//   PUB = true   the pair kernel's PUBLISHING wave: it takes the column differences itself (it has no loads in flight, so its
//                release fence waits for nothing but its own stores) and then calls dp(0, columns done) -- which publishes the count.
template <bool W4, int NSTG, bool PUB, typename Dp>
__device__ __forceinline__ void framescore_cols_body(const float *lp_video, float *F, const int K, const int C, const int fs, const int cols,
                                                     float *fs_smem, Dp dp) {   // [2][C][pitch] rows, then [2][cols * nq + 8][64] sums
    constexpr int STG0 = FSC_THREADS - 64 * NSTG;     // first staging thread
    constexpr int NSTAGE = 64 * NSTG;                 // staging threads
    constexpr int NPER = (64 + NSTG - 1) / NSTG;      // tiles per staging wave per chunk: 16 row blocks x 4 float4 blocks over the waves
    const int tid = threadIdx.x;
    const int nq = (fs + 3) >> 2;                 // b128 reads per column
    const int P = nq * 4;                         // LDS rows per column
    const int pitch = fsc_pitch(cols * P);
    const int bufsz = C * pitch;
    const int runsz = (cols * nq + FSC_DEPTH) * 64;   // (the unrolled loop's last round may write up to seven sums too many)
    float *runs = fs_smem + 2 * bufsz;
    const int C4 = C >> 2;
    const int chunk4 = cols * fs * C4;            // float4 per chunk in memory (contiguous: whole columns)
    const int nchunks = (K + cols - 1) / cols;
    // rounds: chunk c is added in round c, differenced in round c + 1, decoded in round c + 2; a multiple of three (see below)
    const int nch3 = (nchunks + (NSTG == 6 && !PUB ? 2 : 1) + FSC_SETS - 1) / FSC_SETS * FSC_SETS;
    // F[k][c] = run(end of column k) - run(end of column k - 1) for the columns of chunk ci (frame_score, viterbi.py:68-72), by
    // the `nthr` threads j.  Runs while wave 0 fills the OTHER sums buffer, so the end of the chunk before travels in a register
    // (threads j < 64 own column 0 of every chunk).
    float carry = 0.f;
    auto diffs = [&](const int ci, const int j, const int nthr) {
        const float *rn = runs + (ci & 1) * runsz;
        const int ncols = min(cols, K - ci * cols);
        for (int i = j; i < ncols * 64; i += nthr) {
            const int kc = i >> 6, c = i & 63;
            if (c < C) {
                const float hi = rn[((kc + 1) * nq - 1) * 64 + c];
                const float lo = kc > 0 ? rn[(kc * nq - 1) * 64 + c] : carry;
                const int k = ci * cols + kc;
                F[(long)k * C + c] = k == 0 ? hi : hi - lo;
            }
        }
        if (j < 64) carry = rn[(cols * nq - 1) * 64 + j];   // (a short last chunk has no successor)
    };
    // the padding rows of both buffers: -0.0f (every thread its share)
    VSTAMP(12);
    auto zero_fill = [&]() {
        vit_f32x4 *z = reinterpret_cast<vit_f32x4 *>(fs_smem);
        for (int i = tid; i < bufsz / 2; i += FSC_THREADS) z[i] = vit_f32x4{-0.0f, -0.0f, -0.0f, -0.0f};
    };
    if (tid >= STG0) {
        // ---- the staging waves: global -> LDS, and the column differences ----
        const int j = tid - STG0;
        // (global address space spelled out: through the generic pointer of the job record these were FLAT loads, which count
        // against the LDS wait counter as well -- every LDS wait of the staging waves then waited for the read-ahead)
        typedef const __attribute__((address_space(1))) vit_f32x4 *gptr4;
        const gptr4 src4 = (gptr4)(reinterpret_cast<const vit_f32x4 *>(lp_video));
        const int total4 = K * fs * C4;               // < 2^31: T * C / 4
        const unsigned inv_fs = (1u << 20) / fs + 1;  // x / fs == (x * inv_fs) >> 20 for x < 4096
        // One wave instruction moves a tile of 16 rows x 4 float4 (lane = 4 * row + float4): the global load reads 64
        // contiguous bytes per row, and each of the four transposed ds_write_b32 hits 64 different banks (consecutive rows are
        // consecutive words; 4 * pitch = 16 mod 64 spreads the four float4) -- with one float4 per lane in memory order the
        // twelve float4 of a row land on two banks, and the conflicts stall the chain's own reads.
        const int wv = j >> 6, l = j & 63;
        const int ncb = (C4 + 3) >> 2;                // tiles across the classes
        const int inv_ncb = 65536 / ncb + 1;
        int dsto[NPER], srco[NPER];                   // LDS word / chunk-relative float4 of this thread's u-th element; -1 = none
#pragma unroll
        for (int u = 0; u < NPER; ++u) {
            const int t = u * NSTG + wv;                                   // < 72
            const int rb = (t * inv_ncb) >> 16, cb = t - rb * ncb;         // t / ncb (ncb <= 4: exact); a division is ~30 instructions,
                                                                          // and this set-up is in front of the first load
            const int row = rb * 16 + (l >> 2), c4 = cb * 4 + (l & 3);
            const int col = (int)(((unsigned)row * inv_fs) >> 20);
            const bool on = row < cols * fs && c4 < C4;
            dsto[u] = on ? 4 * c4 * pitch + row + col * (P - fs) : -1;
            srco[u] = row * C4 + c4;
        }
        // Three register sets: the loads of chunk ci + 3 are issued when chunk ci's set has been stored, so a load has three
        // chunk times (3 x ~0.7 us of chain) to return -- with one set the chain ran at one HBM latency (~1.4 us) per chunk.
        vit_f32x4 rs[FSC_SETS][NPER];
        auto gload = [&](int ci, vit_f32x4 (&r)[NPER]) {
            const int base4 = ci * chunk4;
#pragma unroll
            for (int u = 0; u < NPER; ++u) r[u] = src4[min(base4 + srco[u], total4 - 1)];   // (unconditional: the clamp keeps idle threads inside the video, and a branch per load serialises them)
        };
        auto sstore = [&](int ci, const vit_f32x4 (&r)[NPER]) {
            float *dst = fs_smem + (ci & 1) * bufsz;
            const int left4 = total4 - ci * chunk4;   // the last chunk may be short: keep the clamped re-loads out
#pragma unroll
            for (int u = 0; u < NPER; ++u)
                if (dsto[u] >= 0 && srco[u] < left4) {
                    float *d = dst + dsto[u];
                    d[0] = r[u].x;
                    d[pitch] = r[u].y;
                    d[2 * pitch] = r[u].z;
                    d[3 * pitch] = r[u].w;
                }
        };
        // Every load below is issued on every path (past the video's end the clamp re-reads its last float4): the wait in front of a
        // set's stores counts the loads issued after it, and LLVM takes the smallest count over all paths that reach it.  For the
        // same reason both sides run the chunk loop to a multiple of three (the extra rounds find nothing to do).
        gload(0, rs[0]);
        zero_fill();                                  // (behind the first chunk's loads: they are in flight meanwhile.  All three sets in
        lds_barrier();                                // front of it kept the waves at the load queue for ~3,000 cycles)
        sstore(0, rs[0]);                             // (the chain starts on chunk 0 while the next three are fetched)
        lds_barrier();
        gload(1, rs[1]);
        gload(2, rs[2]);
        gload(3, rs[0]);
        for (int c0 = 0; c0 < nch3; c0 += FSC_SETS) {
#pragma unroll
            for (int u3 = 0; u3 < FSC_SETS; ++u3) {      // chunk ci + 1 lives in set (ci + 1) % 3 = (u3 + 1) % 3
                const int ci = c0 + u3;
                vit_f32x4(&r)[NPER] = rs[(u3 + 1) % FSC_SETS];
                sstore(ci + 1, r);                        // (behind the last chunk: left4 <= 0, nothing is stored)
                gload(ci + 1 + FSC_SETS, r);
                if (!PUB && ci > 0) diffs(ci - 1, j, NSTAGE);
                lds_barrier();
            }
        }
        return;
    }
    if (NSTG == 6 && tid >= 64) {
        // ---- wave 1: the decoding wave (one-launch kernel) / the publishing wave (pair kernel) ----
        __builtin_amdgcn_s_setprio(2);               // in front of the staging wave it shares a SIMD with
        zero_fill();
        lds_barrier();
        lds_barrier();
        for (int r = 0; r < nch3; ++r) {
            if constexpr (PUB) {
                if (r > 0 && r <= nchunks) {
                    diffs(r - 1, tid - 64, 64);
                    dp(0, min(K, r * cols));
                }
            } else {
                const int c = r - 2;
                if (c >= 0 && c < nchunks) dp(c * cols, min(K, (c + 1) * cols));
            }
            lds_barrier();
        }
        return;
    }
    // ---- wave 0: the chain ----
    const int lane_c = tid < C ? tid : C - 1;
    float run = -0.0f;
    zero_fill();
    lds_barrier();
    VSTAMP(13);
    __builtin_amdgcn_s_setprio(3);                 // the chain goes first on the SIMD it shares with a staging wave
    lds_barrier();
    VSTAMP(14);
    for (int ci = 0; ci < nch3; ++ci) {
        const vit_f32x4 *q = reinterpret_cast<const vit_f32x4 *>(fs_smem + (ci & 1) * bufsz + lane_c * pitch);
        float *w = runs + (ci & 1) * runsz + tid;
        const int nquads = min(cols, K - ci * cols) * nq;
#if VIT_STAMP
        const long long tl0 = (long long)__builtin_amdgcn_s_memtime();
#endif
        // eight reads in flight (an LDS read that waves 1..7 are writing into takes ~150 cycles to return, a read's four adds
        // ~27); the sched_barriers keep each re-load right behind the adds that free its registers (placing the re-load BETWEEN the
        // dependent adds was measured: 16.4 instead of 11.5 cycles per row -- back-to-back dependent adds are the fast path).  The last
        // round may add up to seven reads too many: their sums land in the slack of `runs`, `run` is restored from the last real one.
        vit_f32x4 qr[FSC_DEPTH];
        // (issued in the loop's own order: the wait counter is in-order, and the count LLVM puts in front of read u's adds is the
        // smaller of what the prologue and the loop edge allow -- with the prologue's reads reordered it waited for a read issued
        // three groups earlier instead of eight: 11.8 instead of 7.5 cycles per row)
#pragma unroll
        for (int u = 0; u < FSC_DEPTH; ++u) {
            qr[u] = q[u];
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int s = 0; s < nquads; s += FSC_DEPTH) {
            q += FSC_DEPTH;
#pragma unroll
            for (int u = 0; u < FSC_DEPTH; ++u) {
                __builtin_amdgcn_sched_barrier(0);
                const vit_f32x4 v = qr[u];
                run = run + v.x;                   // sequential float32 chain
                run = run + v.y;
                run = run + v.z;
                run = run + v.w;
                if (!W4 || (u & 3) == 3) w[u * 64] = run;
                qr[u] = q[u];                      // behind the chunk's end: slack rows
            }
            __builtin_amdgcn_sched_barrier(0);
            w += FSC_DEPTH * 64;
        }
        if (nquads > 0 && (nquads & (FSC_DEPTH - 1))) run = runs[(ci & 1) * runsz + (nquads - 1) * 64 + tid];
#if VIT_STAMP
        const long long tb0 = (long long)__builtin_amdgcn_s_memtime();
        lds_barrier();
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            const long long tb1 = (long long)__builtin_amdgcn_s_memtime();
            if (ci == 0) g_vit_stamps[10] = 0, g_vit_stamps[11] = 0;
            g_vit_stamps[10] += tb0 - tl0;     // adds of the chunk
            g_vit_stamps[11] += tb1 - tb0;     // waiting for the stagers
        }
#else
        lds_barrier();
#endif
    }
}

template <bool W4>
__global__ __launch_bounds__(FSC_THREADS) void viterbi_framescore_cols_kernel(const mucon_viterbi_job *jobs, char *ws, int C, int fs, int cols) {
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    const mucon_viterbi_job job = jobs[blockIdx.x];
    const int K = job.T / fs;
    if (K < 1) return;
    VSTAMP(8);
    framescore_cols_body<W4, 7, false>(job.lp, reinterpret_cast<float *>(ws + job.ws_off), K, C, fs, cols, fs_smem, FscNoDp());
    VSTAMP(9);
}

struct Cand {
    double v;
    int j;
};
__device__ __forceinline__ Cand better(Cand a, Cand b) {
    // the later (longer) of equal candidates wins: HypDict.update uses `<=` (viterbi.py:27)
    const bool take = (b.v > a.v) || (b.v == a.v && b.j > a.j);
    return take ? b : a;
}
// Wave-wide arg-max with the reference's tie rule, on the DPP crossbar (no LDS round trips): first the
// maximum score (six DPP steps of v_max_f64), then the largest length index among the lanes that hold it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // lanes whose DPP source is disabled / out of range keep their own value (old = self): max(v, v) = v
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return fmax(v, __hiloint2double(hi2, lo2));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max_i32(int v) {
    return max(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ Cand wave_best(Cand c) {
    // candidates are never NaN here: NaN length tables are resolved on the host (mucon_amd/core/viterbi/viterbi.py)
    double m = c.v;
    m = dpp_max_f64<0xB1, 0xF>(m);    // quad_perm [1,0,3,2]
    m = dpp_max_f64<0x4E, 0xF>(m);    // quad_perm [2,3,0,1]
    m = dpp_max_f64<0x141, 0xF>(m);   // row_half_mirror
    m = dpp_max_f64<0x140, 0xF>(m);   // row_mirror: every lane of a 16-lane row holds the row maximum
    m = dpp_max_f64<0x142, 0xA>(m);   // row_bcast15 into rows 1 and 3
    m = dpp_max_f64<0x143, 0xC>(m);   // row_bcast31 into rows 2 and 3: lane 63 holds the wave maximum
    const int mlo = __builtin_amdgcn_readlane(__double2loint(m), 63);
    const int mhi = __builtin_amdgcn_readlane(__double2hiint(m), 63);
    const double vmax = __hiloint2double(mhi, mlo);
    int j = (c.v == vmax) ? c.j : -1;  // `<=` in HypDict.update keeps the LAST of equal candidates: the largest j
    j = dpp_max_i32<0xB1, 0xF>(j);
    j = dpp_max_i32<0x4E, 0xF>(j);
    j = dpp_max_i32<0x141, 0xF>(j);
    j = dpp_max_i32<0x140, 0xF>(j);
    j = dpp_max_i32<0x142, 0xA>(j);
    j = dpp_max_i32<0x143, 0xC>(j);
    Cand r;
    r.v = vmax;
    r.j = __builtin_amdgcn_readlane(j, 63);
    return r;
}
__device__ __forceinline__ bool alive_at(int n, int k0, int J) {
    return k0 >= 0 && (n == 0 ? (k0 == 0) : (k0 >= n && k0 <= J * n));
}
// hyp.score + frame_score (viterbi.py:99,113): float32 + float32 while in the first transcript
// state, float64 + float32 afterwards (NumPy 2 promotion; the length table makes later states f64)
__device__ __forceinline__ double add_frame(double s, float f, int n) {
    const float t32 = (float)s + f;          // state 0: s is float32-valued, the add rounds to float32
    const double t64 = s + (double)f;
    return n == 0 ? (double)t32 : t64;
}

// Where the per-frame labels go and in which form (include/mucon_hip.h: MUCON_VIT_LABELS_*).  The decode's own result is the
// segmentation (seg_len, n_seg); the per-frame labels are its expansion -- 4 T bytes as the reference's ints, T bytes as uint8
// (C <= 64), or nothing at all when the caller expands the segments itself: at 256 videos of T = 16,384 in flight the int32 labels
// were 16.8 MB crossing PCIe, longer than the decode (r3 profiles).
struct VitLabels {
    void *p;
    int fmt;
};
__device__ __forceinline__ void vit_fill_i32(int32_t *lab, const int t0, const int t1, const int l, const int tid, const int nthreads) {
    for (int t = t0 + tid; t < t1; t += nthreads) lab[t] = l;
}
// uint8 labels: 16-byte stores over the aligned interior of a run (a wave writes 1 KB per instruction), byte stores at its two ends
__device__ __forceinline__ void vit_fill_u8(uint8_t *lab, const int t0, const int t1, const int l, const int tid, const int nthreads) {
    if (t1 <= t0) return;
    const int mis = (int)(reinterpret_cast<uintptr_t>(lab + t0) & 15);
    const int head = min(t1 - t0, (16 - mis) & 15);
    if (tid < head) lab[t0 + tid] = (uint8_t)l;
    const int a0 = t0 + head, nvec = (t1 - a0) >> 4;
    const unsigned w = (unsigned)(l & 0xFF) * 0x01010101u;
    const uint4 v = make_uint4(w, w, w, w);
    uint4 *dst = reinterpret_cast<uint4 *>(lab + a0);
    for (int i = tid; i < nvec; i += nthreads) dst[i] = v;
    const int b0 = a0 + (nvec << 4);
    if (b0 + tid < t1) lab[b0 + tid] = (uint8_t)l;
}

// The tail both DP kernels share: thread 0 walks the back-pointers (viterbi.py:140-158), then the workgroup expands the
// segments into per-frame labels.  `pre` is an LDS scratch of N + 1 ints, `a` the transcript in LDS.
__device__ __forceinline__ void vit_traceback_and_labels(const mucon_viterbi_job &job, int vid, int fin_n, int fin_j, double fin_score,
                                                         bool forced, const uint8_t *bp, const int *a, int *pre, const VitLabels labels,
                                                         int32_t *seg_len, int32_t *n_seg, double *score, int32_t *status, int fs,
                                                         const uint8_t *bp_l = nullptr, const bool bp_lds = false,
                                                         const int live_threads = 0) {   // bp_lds: the back-pointers are in bp_l (LDS), bp is unused; live_threads: the threads that got here, if not all
    const int T = job.T, N = job.N, K = T / fs;
    const int tid = threadIdx.x, nthreads = live_threads ? live_threads : (int)blockDim.x;
    // traceback
    const int nseg = fin_n + 1;
    const int missing = T - K * fs;
    if (tid == 0) {
        int n = fin_n, j = fin_j, k = K - 1;
        int32_t *sl = seg_len + job.seg_off;
        // (the outputs may live in pinned host memory: nothing written there is read back -- the lengths also go to `pre`)
        for (int s = nseg - 1; s >= 0; --s) {
            const int len = (j + 1) * fs;
            pre[s + 1] = len;
            sl[s] = s == nseg - 1 ? len + missing : len;   // leftover frames are added to the last segment's length
            const int k0 = k - j;
            if (n > 0) {
                // (a dependent chain of N reads: ~100 cycles each from LDS, ~1,500 from memory.  The volatile access keeps LLVM from merging
                // the two loads into one of a selected pointer: clang 22's inliner pass crashes on that form)
                j = bp_lds ? (int)*(volatile const __attribute__((address_space(3))) uint8_t *)(bp_l + k0 * N + n) : (int)bp[(size_t)k0 * N + n];
                k = k0 - 1;
                --n;
            }
        }
        int acc = 0;
        for (int s = 0; s < nseg; ++s) {                   // pre[s] = frames in front of segment s
            const int len = pre[s + 1];
            pre[s] = acc;
            acc += len;
        }
        pre[nseg] = acc;
        for (int s = nseg; s < N; ++s) sl[s] = 0;
        n_seg[vid] = nseg;
        score[vid] = fin_score;
        status[vid] = forced ? MUCON_VIT_TRUNCATED : MUCON_VIT_OK;
    }
    __syncthreads();
    // ... and labelled, at the START of the video, with the last segment's label.  Segment by segment (a wave-uniform loop, the
    // threads stride over the segment's frames): a binary search per frame cost ~6 dependent LDS reads for every label.
    if (labels.fmt == MUCON_VIT_LABELS_NONE) return;           // the caller expands (transcript, seg_len) itself
    if (labels.fmt == MUCON_VIT_LABELS_U8) {
        uint8_t *lab = static_cast<uint8_t *>(labels.p) + job.label_off;
        vit_fill_u8(lab, 0, missing, a[nseg - 1], tid, nthreads);
        for (int sg = 0; sg < nseg; ++sg) vit_fill_u8(lab, missing + pre[sg], missing + pre[sg + 1], a[sg], tid, nthreads);
        vit_fill_u8(lab, missing + pre[nseg], T, a[nseg - 1], tid, nthreads);
        return;
    }
    int32_t *lab = static_cast<int32_t *>(labels.p) + job.label_off;
    vit_fill_i32(lab, 0, missing, a[nseg - 1], tid, nthreads);
    for (int sg = 0; sg < nseg; ++sg) vit_fill_i32(lab, missing + pre[sg], missing + pre[sg + 1], a[sg], tid, nthreads);
    vit_fill_i32(lab, missing + pre[nseg], T, a[nseg - 1], tid, nthreads);   // (nothing, when the segments cover the columns)
}

// Phase 2, J <= 66 length slots (every shipped configuration: max_len 2000, fs 30) and up to 128 transcript states: the whole DP
// runs with the hypothesis scores in REGISTERS -- one wave (<= 16 states: no barrier, no LDS round trip per column) or four / eight
// waves (one barrier per column for the hand-over across the wave boundaries).
//
// Lane n * G + g owns state n's hypotheses with length index j in [g * JG, (g + 1) * JG) -- G = 8 or 4 lanes per state.
// Indexing by LENGTH instead of by entry column turns "every hypothesis grows by one column" into S[j] = S[j-1] + f with a
// different destination register (the shift costs nothing; across a lane boundary it is one DPP row_shr), and makes every table
// index a compile-time constant: the length scores Pl[n][j] sit in registers too.  Dead hypotheses hold -inf (-inf + f stays
// -inf: emissions are log-probabilities), so the column has no liveness test:
//     S[j]  = S[j-1] + f                         stay in state n (viterbi.py:96-104); slot J only feeds the candidates
//     cand  = S[j] + Pl[n][j-1]                   leave state n after (old) length index j-1 (viterbi.py:105-121)
//     (v,j) = arg max, the larger j among equals   HypDict.update's `<=` (viterbi.py:27) -> enters state n + 1 at index 0
// What one column costs is its DEPENDENT chain, entry -> S[1] -> candidate -> maximum -> next state's entry, so (r3) the arg max
// is taken apart: the VALUE is a max tree over the lane's candidates with the one that hangs on the fresh entry (slot 1) joined
// last, three DPP butterflies inside the state's lane group, and a DPP hand-over to the next state's lanes (row_shr:G inside a
// 16-lane row, row_bcast:15 across rows: after the butterflies every lane of a group holds the maximum) -- no LDS instruction
// on the chain; the INDEX (largest slot whose candidate equals the maximum -- the reference's tie rule) is computed behind it
// and only feeds the back-pointer store.  (r2 walked a compare-and-select chain over the slots and handed over by ds_bpermute:
// 1,150 cycles per column at G = 8; now ~ 165 -> ~ 100 instructions and ~ 700 cycles: one wave issues an instruction every 5-7
// cycles whatever the dependences -- what a column costs is its instruction count.)
// If every candidate is -inf the reference still picks the longest LIVE one: that index is the closed form min(J, k-n) - 1.
// The reference's `+ 0.0` on every candidate only turns -0.0 into +0.0; comparisons do not see the sign of zero, so it is
// applied once, to the winner.  State 0 holds a single hypothesis (entered at column 0) whose score is float32 + float32
// (NumPy promotion, see add_frame): a scalar side chain, its one candidate enters state 1 directly.
constexpr int VL_THREADS = 256;   // NW = 1: wave 0 decodes; waves 1..3 wait at the barrier and help to expand the labels
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false));
}
// butterflies (every lane has a valid source): no `old` operand to keep alive, and an integer max takes the DPP operand itself
template <int CTRL>
__device__ __forceinline__ double dpp_f64_all(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32_all(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
// max of two non-NaN doubles, one of them fresh from a DPP move: fmax() would first canonicalise the moved value (LLVM cannot know
// it is no signalling NaN) -- one more v_max_f64 on the column's dependent chain.  (The hazard recogniser treats the statement's
// result as a vector write: it puts the two wait states a DPP read needs behind it itself -- checked in the listing.)
__device__ __forceinline__ double max_f64_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the value lane - G holds (G = 4, 8; every lane of a G-group holds the same): row_shr:G inside a 16-lane row, lane 15 of the
// row below for a row's first group (bank masks steer the two moves into the same register: no select).  Row 0's first group keeps
// its own value (its source is another wave, or nothing).
template <int G>
__device__ __forceinline__ int hand_up_i32(int v) {
    constexpr int LOW = G == 8 ? 0x3 : 0x1;                                           // the banks (4 lanes each) of a row's first G lanes
    const int a = __builtin_amdgcn_update_dpp(v, v, 0x110 + G, 0xF, 0xF & ~LOW, false);   // row_shr:G into the lanes >= G
    return __builtin_amdgcn_update_dpp(a, v, 0x142, 0xE, LOW, false);                    // row_bcast:15 into the first G of rows 1..3
}
template <int G>
__device__ __forceinline__ double hand_up_f64(double v) {
    return __hiloint2double(hand_up_i32<G>(__double2hiint(v)), hand_up_i32<G>(__double2loint(v)));
}

// The decoding lanes' registers and one column of the DP (see the comment above).  Lane dl = n * G + g of the NW decoding waves.
// The length scores P[j][n] = length_model.score((j + 1) fs, a_n) of a video, two forms:
//   * lf == nullptr: a host-built table [J][N] at double offset p_off (mucon_viterbi_decode_host / _batch);
//   * lf != nullptr (ABI 7, the *_poisson entries): p_off points at the transcript's PoissonModel parameters [3][N] = (ln mu, mu, norm) of the
//     classes a_n, lf[j] is the reference's running log-factorial at length (j + 1) fs, and the score is built here exactly as reference
//     src/core/viterbi/length_model.py:65-71 builds it -- `l * np.log(mu) - mu - logFak - norms`, left to right: four single IEEE double
//     operations (this translation unit is compiled with -ffp-contract=off: no fused multiply-add), NumPy's `log` stays on the host.
//     Lengths >= max_len score -inf (length_model.py:76-80).  3 doubles per transcript state cross PCIe instead of J = 66.
struct VitTab {
    const double *t;
    const double *lf;
    int fs, max_len;
    __device__ __forceinline__ double at(const int64_t p_off, const int j, const int n, const int N) const {
        if (lf == nullptr) return t[p_off + (size_t)j * N + n];
        const double *q = t + p_off;
        const long l = (long)(j + 1) * fs;
        const double v = (((double)l * q[n] - q[N + n]) - lf[j]) - q[2 * N + n];
        return l < max_len ? v : -INFINITY;
    }
};

template <int G, int JG>
struct VitLanes {
    static_assert(G * JG >= 67 && (G == 4 || G == 8), "slots 0..66 over the G lanes of a state");
    double S[JG];     // hypothesis scores by slot; S[0] of a state's first lane (g == 0): the entry the last column produced
    double PlS[JG];   // length scores against the slot index s = j + 1 (the candidate of slot s uses the OLD length index s - 1)
    double zmask;     // state 0's candidate + zmask: the reference's `+ 0.0` in state 1's first lane, -inf everywhere else
    float s0;         // state 0: one hypothesis, a float32 chain
    int n, g, lane;
    int lo_s, R;      // this lane stores a back-pointer at column k <=> (unsigned)(k - lo_s) <= R   (g == 0 and n <= k <= J n)
    bool is_n1;       // a lane of state 1 (its entry comes from state 0's side chain, not from the hand-over)

    // the loads that do not depend on the frame scores (issued early; `dl` = decoding lane index)
    __device__ __forceinline__ void load(const mucon_viterbi_job &job, const VitTab &tables, const int dl, const int J) {
        const int N = job.N;
        n = dl / G;
        g = dl - n * G;
        lane = dl & 63;
        const bool state_on = n >= 1 && n < N;
        const double NEG = -INFINITY;
#pragma unroll
        for (int i = 0; i < JG; ++i) {
            const int sidx = g * JG + i;
            PlS[i] = (state_on && sidx >= 1 && sidx <= J) ? tables.at(job.p_off, sidx - 1, n, N) : NEG;
            S[i] = NEG;
        }
        zmask = (state_on && n == 1 && g == 0) ? 0.0 : NEG;
        lo_s = (state_on && g == 0) ? n : 0x40000000;
        R = (J - 1) * n;
        is_n1 = n == 1;
        s0 = 0.f;

    }
    // init_decoding (viterbi.py:81-90): score = 0.0 + frame_score(fs-1, a_0), float32.  (State 0 is lane 0 of wave 0; the other
    // waves carry a side chain of their own first lane's scores that nothing reads.)
    __device__ __forceinline__ void init(const float f_col0) { s0 = 0.0f + __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(f_col0))); }

    // Column k >= 1, first half: everything up to the state's best candidate.  f = frame score of this lane's label, pl0 = state 0's
    // length score at the old length index k - 1 (-inf from J on).  -> vout: the winner (+ 0.0), handed to the next state's lanes
    // in vin; jm / jin: its length index likewise.
    __device__ __forceinline__ void col_a(const int k, const float f, const double pl0, const int J, double &vout, double &vin, int &jm,
                                          int &jin, double &alt) {
        const double fd = (double)f;
        // every hypothesis grows by one column (viterbi.py:96-104); across a lane boundary: row_shr:1 from the slot group below
        const double in = dpp_f64<0x111>(S[JG - 1]);
#pragma unroll
        for (int i = JG - 1; i >= 1; --i) S[i] = S[i - 1] + fd;
        S[0] = in + fd;                                   // (g == 0: overwritten by col_b; its candidate below is -inf whatever this is)
        // state 0's float32 chain and its candidate for state 1: ((s + f) + P) + 0.0
        const float f0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(f)));
        const float t0 = s0 + f0;
        s0 = t0;
        const double a1 = ((double)t0 + pl0) + zmask;     // state 1's first lane: its entry; -inf elsewhere
        alt = g == 0 ? a1 : S[0];                         // what S[0] becomes where no entry arrives through the hand-over
        // candidates; the maximum of the lane's, slot 1 (behind the fresh entry) last
        double c[JG];
#pragma unroll
        for (int i = 0; i < JG; ++i) c[i] = S[i] + PlS[i];
        double t[JG];
        t[0] = c[0];
#pragma unroll
        for (int i = 1; i < JG - 1; ++i) t[i] = c[i + 1];
#pragma unroll
        for (int st = 1; st < JG - 1; st *= 2)
#pragma unroll
            for (int i = 0; i + st < JG - 1; i += 2 * st) t[i] = fmax(t[i], t[i + st]);
        double vm = fmax(t[0], c[1]);
        // ... of the state: butterflies inside the lane group, every lane ends up with the maximum
        vm = max_f64_raw(vm, dpp_f64_all<0xB1>(vm));                // quad_perm [1,0,3,2]
        if (G >= 4) vm = max_f64_raw(vm, dpp_f64_all<0x4E>(vm));    // quad_perm [2,3,0,1]
        if (G >= 8) vm = max_f64_raw(vm, dpp_f64_all<0x141>(vm));   // row_half_mirror
        vout = vm + 0.0;
        vin = hand_up_f64<G>(vout);
        // the index, behind the chain: the largest slot whose candidate is the maximum (HypDict.update's `<=`, viterbi.py:27).
        // (Taking it one column late, next to the next column's value chain, was measured: the copies of the kept candidates cost
        // more than the stalls they fill.)
        const double NEG = -INFINITY;
        int m3[3] = {-128, -128, -128};                             // three short select chains instead of one long one
#pragma unroll
        for (int i = 0; i < JG; ++i) m3[i % 3] = max(m3[i % 3], c[i] == vm ? i : -128);
        jm = g * JG - 1 + max(max(m3[0], m3[1]), m3[2]);            // (< 0 in lanes that do not hold the maximum)
        jm = max(jm, dpp_i32_all<0xB1>(jm));
        if (G >= 4) jm = max(jm, dpp_i32_all<0x4E>(jm));
        if (G >= 8) jm = max(jm, dpp_i32_all<0x141>(jm));
        // all -inf: the reference still picks the longest live hypothesis.  (State 0's lanes always take this path, and for them it
        // is k - 1 while state 1 can be entered: exactly the length index of state 0's one hypothesis -- state 1 needs no special case.)
        if (vm == NEG) jm = min(J - 1, k - n - 1);
        jin = hand_up_i32<G>(jm);
    }
    // second half: the entry arrives (vin / jin: from the hand-over, or across the wave boundary).  -> whether this lane stores jin
    __device__ __forceinline__ bool col_b(const int k, const double vin, const double alt) {
        const bool store = (unsigned)(k - lo_s) <= (unsigned)R;     // g == 0 and state n can be entered at column k
        const bool take = store && !is_n1;
        S[0] = take ? vin : alt;
        return store;
    }
};

// finalize_decoding (viterbi.py:125-138) on the decoding lanes: the best final hypothesis -> *fin_n / *fin_j / *fin_score (LDS)
template <int G, int JG>
__device__ __forceinline__ void vit_lanes_finalize(const VitLanes<G, JG> &L, const mucon_viterbi_job &job, const int K, const int J,
                                                   const bool forced, const double *Pl0, const int wave, int *fin_n, int *fin_j,
                                                   double *fin_score) {
    const int N = job.N, n = L.n, g = L.g;
    const double NEG = -INFINITY;
    const double pl_up = dpp_f64<0x101>(L.PlS[0]);                   // row_shl:1 -- the next lane's first length score
    if (forced) {
        // Degenerate outcomes of the reference, see viterbi_dp_kernel
        if (L.lane == 0 && wave == 0) {
            *fin_n = job.force_n >= 0 ? job.force_n : K - 1;
            *fin_j = job.force_n >= 0 ? job.force_j : 0;
            *fin_score = -INFINITY;
        }
        return;
    }
    const int nf = N - 1, c = K - 1;
    Cand best;
    best.v = NEG;
    best.j = -1;
    if (nf == 0) {
        if (c < J) {
            best.v = ((double)L.s0 + Pl0[c]) + 0.0;
            best.j = c;
        }
    } else if (n == nf) {
        // live length indices at the last column: entered at k0 = c - j with nf <= k0 <= J nf, and j < J
        const int j_lo = max(0, c - J * nf), j_hi = min(J - 1, c - nf);
#pragma unroll
        for (int i = 0; i < JG; ++i) {
            const int j = g * JG + i;
            // Pl[nf][j] is the candidates' table one slot up: the next register, the next lane's first for the lane's last slot
            // (j beyond J - 1 is not live: whatever arrives there is not used)
            const double pl = i + 1 < JG ? L.PlS[i + 1 < JG ? i + 1 : 0] : pl_up;
            Cand d;
            d.v = (L.S[i] + pl) + 0.0;
            d.j = j;
            if (j >= j_lo && j <= j_hi) best = better(best, d);
        }
    }
    best = wave_best(best);
    if (L.lane == 0 && wave == (nf * G) / 64) {                       // the wave that holds the last state
        *fin_n = nf;
        *fin_j = best.j;
        *fin_score = best.v;
    }
}

// The two-launch DP: frame scores from memory (phase 1 wrote them), NW decoding waves (1: wave 0 decodes, the workgroup's other
// waves only help with the labels).  bp_lds: the back-pointers go to bp_l (LDS) instead of bp.
template <int G, int JG, int NW>
__device__ __forceinline__ void viterbi_dp_lanes_body(
    const mucon_viterbi_job &job, const int vid, const float *F, uint8_t *bp, uint8_t *bp_l, const bool bp_lds, const int32_t *transcripts,
    const VitTab tables, const VitLabels labels, int32_t *seg_len, int32_t *n_seg, double *score, int32_t *status, int C, int fs, int J,
    const unsigned long long *progress = nullptr, const uint32_t seq = 0, const int live_threads = 0) {
    // progress != nullptr (pair kernel): the frame scores are being written by another workgroup, which publishes
    // (seq << 32 | columns done) there; live_threads: the workgroup's threads that run this body (the others have left)
    constexpr int VL_CH = NW >= 8 ? 8 : (NW >= 4 ? 16 : 32);  // columns of frame scores staged at a time (one register each while in flight)
    constexpr int NL = 64 * NW;               // decoding lanes
    __shared__ float Fb[2 * VL_CH + 1][NL];   // frame scores of the lanes' own labels, two chunks (+ a row the read-ahead may touch)
    __shared__ double xch_v[2][NW];           // NW > 1: a wave's last state hands its best candidate to the next wave's first
    __shared__ int xch_j[2][NW];
    __shared__ double Pl0[128];               // state 0's length scores (runtime index: its hypothesis has j = column), -inf from J on
    __shared__ int a[NL], pre[NL + 1];
    __shared__ double fin_score;
    __shared__ int fin_n, fin_j;
    const int T = job.T, N = job.N;
    const int K = T / fs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (K < 1) {  // frame_scores[fs-1] does not exist: IndexError in the reference (viterbi.py:87)
        if (tid == 0) {
            status[vid] = MUCON_VIT_INDEX_ERROR;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }
    if (job.force_n < 0 && K > J * N) {  // every hypothesis has outlived max_length: empty set
        if (tid == 0) {
            status[vid] = MUCON_VIT_NO_HYPOTHESIS;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }
    const bool forced = job.force_n >= 0 || K < N;
    const double NEG = -INFINITY;
    VitLanes<G, JG> L;
    if (tid < NL) {
        L.load(job, tables, tid, J);
        a[tid] = tid < N ? transcripts[job.tr_off + tid] : 0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + u * NL < 128) Pl0[tid + u * NL] = tid + u * NL < J ? tables.at(job.p_off, tid + u * NL, 0, N) : NEG;
    }
    __syncthreads();

    if (tid < NL) {
        const int n = L.n;
        const int an = a[n < N ? n : 0];
        // frame scores: chunk q in Fb[(q & 1) * VL_CH ..]; the loads of chunk q + 1 are issued at the start of chunk q
        float fq[VL_CH];
        auto fetch = [&](int q) {
            if (progress) {                                              // (the adding workgroup runs ~3x ahead of the decode: a wait at the start only)
                const int need = min(K, (q + 1) * VL_CH);
                for (;;) {
                    const unsigned long long v = __hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(v >> 32) == seq && (int)(uint32_t)v >= need) break;
                    __builtin_amdgcn_s_sleep(8);
                }
            }
#pragma unroll
            for (int u = 0; u < VL_CH; ++u) {
                const int col = q * VL_CH + u;
                fq[u] = F[(size_t)(col < K ? col : K - 1) * C + an];
            }
        };
        auto stash = [&](int q) {
#pragma unroll
            for (int u = 0; u < VL_CH; ++u) Fb[(q & 1) * VL_CH + u][tid] = fq[u];
        };
        VSTAMP(2);
        fetch(0);
        stash(0);
        VSTAMP(3);
        L.init(Fb[0][tid]);
        const int nchunks = (K + VL_CH - 1) / VL_CH;
        for (int q = 0; q < nchunks; ++q) {
            if (q + 1 < nchunks) fetch(q + 1);
            const int k_lo = q == 0 ? 1 : q * VL_CH, k_hi = min(K, (q + 1) * VL_CH);
            const float *fp = &Fb[(q & 1) * VL_CH + (k_lo & (VL_CH - 1))][tid];
            float f_nx = fp[0];
            double pl0_nx = Pl0[min(k_lo - 1, 127)];
            for (int k = k_lo; k < k_hi; ++k) {
                const float f = f_nx;
                const double pl0 = pl0_nx;
                fp += NL;
                f_nx = fp[0];                                            // the next column's reads overlap this column (behind a chunk's
                pl0_nx = Pl0[min(k, 127)];                               // end: a value nothing uses)
                double vout, vin, alt;
                int jm, jin;
                L.col_a(k, f, pl0, J, vout, vin, jm, jin, alt);
                if constexpr (NW > 1) {
                    // Across a wave boundary: through LDS, one barrier per column.  (A mailbox per column that only the next wave polls,
                    // so that the waves run skewed and none waits for all, was measured: the polling reads cost more than the barrier.)
                    if (lane == 63) {
                        xch_v[k & 1][wave] = vout;
                        xch_j[k & 1][wave] = jm;
                    }
                    // LDS-only barrier (r4): __syncthreads() also waits for vmcnt(0) -- the read-ahead of the next chunk's frame scores
                    // once per chunk and, with the back-pointers in HBM scratch, a global byte store's round trip EVERY column: the
                    // 256-in-flight DP launch of config 5 lasted 730 us for a 280 us decode (r3 profile) because of that, not
                    // because of its labels
                    lds_barrier();
                    if (wave >= 1 && lane < G) {
                        vin = xch_v[k & 1][wave - 1];
                        jin = xch_j[k & 1][wave - 1];
                    }
                }
                if (L.col_b(k, vin, alt)) {
                    if (bp_lds) bp_l[k * N + n] = (uint8_t)jin;
                    else bp[(size_t)k * N + n] = (uint8_t)jin;
                }
            }
            if (q + 1 < nchunks) stash(q + 1);
        }
        VSTAMP(4);
        vit_lanes_finalize<G, JG>(L, job, K, J, forced, Pl0, wave, &fin_n, &fin_j, &fin_score);
    }
    __syncthreads();
    VSTAMP(5);
    if (fin_j < 0) {  // no comparable final hypothesis (NaN scores): traceback is None in the reference
        if (tid == 0) {
            status[vid] = MUCON_VIT_NO_HYPOTHESIS;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }
    vit_traceback_and_labels(job, vid, fin_n, fin_j, fin_score, forced, bp, a, pre, labels, seg_len, n_seg, score, status, fs, bp_l, bp_lds,
                             live_threads);
    VSTAMP(6);
}

template <int G, int JG, int NW>
__global__ __launch_bounds__(NW == 1 ? VL_THREADS : 64 * NW) void viterbi_dp_lanes_kernel(
    const mucon_viterbi_job *jobs, const int32_t *transcripts, const VitTab tables, const VitLabels labels,
    int32_t *seg_len, int32_t *n_seg, double *score, int32_t *status, char *ws, int C, int fs, int J, int bp_lds_bytes,
    volatile int32_t *done_flag, int32_t done_value) {
    extern __shared__ __attribute__((aligned(16))) uint8_t vl_bp[];   // [K][N] back-pointers when they fit (bp_lds_bytes of them)
    const mucon_viterbi_job job = jobs[blockIdx.x];
    const int K = job.T / fs;
    const size_t f_bytes = ((size_t)(K > 0 ? K : 0) * C * sizeof(float) + 15) & ~(size_t)15;
    viterbi_dp_lanes_body<G, JG, NW>(job, blockIdx.x, reinterpret_cast<const float *>(ws + job.ws_off),
                                     reinterpret_cast<uint8_t *>(ws + job.ws_off + f_bytes), vl_bp, K > 0 && K * job.N <= bp_lds_bytes, transcripts, tables, labels, seg_len,
                                     n_seg, score, status, C, fs, J);
    if (done_flag) {                                 // (one video per call: see viterbi_fused_kernel)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(const_cast<int32_t *>(done_flag), done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// A few longer videos (or more transcript states than the one-launch kernel takes): phase 1 and phase 2 as TWO WORKGROUPS of one
// launch, blockIdx.y = 0 adds (framescore_cols_body, wave 1 publishing "columns done" behind a release fence), blockIdx.y = 1 decodes
// (viterbi_dp_lanes_body, waiting on that count before each chunk of frame scores it fetches).  The decode is the longer of the two
// (~1,200 against ~350 cycles per column), so after the first chunk it never waits: the call costs the decode, not the sum
// (T = 16,384 / N = 64: 0.44 -> 0.34 ms).  For latency calls (vit_launch: <= 8 videos).
template <int G, int JG, int NW, bool W4>
__global__ __launch_bounds__(FSC_THREADS) void viterbi_pair_kernel(
    const mucon_viterbi_job *jobs, const int32_t *transcripts, const VitTab tables, const VitLabels labels, int32_t *seg_len, int32_t *n_seg,
    double *score, int32_t *status, char *ws, int C, int fs, int J, int cols, int bp_lds_bytes, volatile int32_t *done_flag,
    int32_t done_value, unsigned long long *progress_words, uint32_t seq) {
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    const mucon_viterbi_job job = jobs[blockIdx.x];
    const int K = job.T / fs;
    const size_t f_bytes = ((size_t)(K > 0 ? K : 0) * C * sizeof(float) + 15) & ~(size_t)15;
    float *F = reinterpret_cast<float *>(ws + job.ws_off);
    // one 64-bit word per video, 64 bytes apart, in a buffer of the library's own that holds nothing else and starts as zeros: the
    // tag (a call counter that never repeats within 2^32 calls) cannot be met by stale scratch of an earlier call of another shape
    unsigned long long *progress = progress_words + 8 * blockIdx.x;
    if (blockIdx.y == 0) {
        if (K < 1) return;
        framescore_cols_body<W4, 6, true>(job.lp, F, K, C, fs, cols, fs_smem, [&](int, const int done) {
            __threadfence();                             // this wave's frame-score stores are visible on the device ...
            if (threadIdx.x == 64) __hip_atomic_store(progress, ((unsigned long long)seq << 32) | (unsigned)done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... before the count
        });
        return;
    }
    constexpr int LIVE = NW == 1 ? VL_THREADS : 64 * NW;
    if ((int)threadIdx.x >= LIVE) return;                // (a finished wave no longer counts at the workgroup's barriers)
    viterbi_dp_lanes_body<G, JG, NW>(job, blockIdx.x, F, reinterpret_cast<uint8_t *>(ws + job.ws_off + f_bytes),
                                     reinterpret_cast<uint8_t *>(fs_smem), K > 0 && K * job.N <= bp_lds_bytes, transcripts, tables, labels,
                                     seg_len, n_seg, score, status, C, fs, J, progress, seq, LIVE);
    if (done_flag) {                                     // (one video per call: see viterbi_fused_kernel)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(const_cast<int32_t *>(done_flag), done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ONE launch for a short video (the evaluation's case: T ~ 2,000 frames, up to 16 transcript states): the frame scores never leave
// the workgroup, and phase 2 runs UNDER phase 1 -- wave 0 adds (framescore_cols_body's float32 chain), waves 2..7 stage and take the
// column differences into LDS, wave 1 decodes the columns two rounds behind the chain (VitLanes), back-pointers in LDS.  The job
// record is a kernel argument; transcript and length scores are read straight from the caller's PINNED host staging buffer by the
// decoding wave before it does anything else; the results go straight into pinned host memory (no copy calls on either side);
// thread 0 publishes `*done_flag = done_value` behind a system-scope fence, which is what the host waits for.
// (r2, two launches + copies: 0.100 ms for T = 2,000 / N = 6; r3: 0.050 ms.)
template <int G, int JG, bool W4>
__global__ __launch_bounds__(FSC_THREADS) void viterbi_fused_kernel(const mucon_viterbi_job job, const int32_t *transcripts,
                                                                    const VitTab tables, const VitLabels labels, int32_t *seg_len,
                                                                    int32_t *n_seg, double *score, int32_t *status, int C,
                                                                    int fs, int J, int cols, int dyn_floats, volatile int32_t *done_flag,
                                                                    int32_t done_value) {
    // (the job travels as a kernel argument: read from the pinned table it was one more PCIe round trip in front of everything)
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    __shared__ double Pl0[VF_MAX_K];                 // state 0's length scores by column, -inf from J on
    __shared__ int a[64], pre[65];
    __shared__ double fin_score;
    __shared__ int fin_n, fin_j;
    const int T = job.T, N = job.N, K = T / fs;
    const int tid = threadIdx.x, vid = 0;
    float *F = fs_smem + dyn_floats;                 // [K][C] behind phase 1's buffers, then the back-pointers [K][N]
    uint8_t *bp_l = reinterpret_cast<uint8_t *>(F + (K > 0 ? K : 0) * C);
    VSTAMP(0);
    bool early = false;
    if (K < 1) {  // frame_scores[fs-1] does not exist: IndexError in the reference (viterbi.py:87)
        if (tid == 0) status[vid] = MUCON_VIT_INDEX_ERROR;
        early = true;
    } else if (job.force_n < 0 && K > J * N) {  // every hypothesis has outlived max_length: empty set
        if (tid == 0) status[vid] = MUCON_VIT_NO_HYPOTHESIS;
        early = true;
    }
    if (early) {
        if (tid == 0) {
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
    } else {
        const bool forced = job.force_n >= 0 || K < N;
        // wave 1 decodes: its loads from pinned host memory (~2 us away) are issued in front of everything else it does
        // (they stay in registers until its first round: an LDS store would have it wait for them in front of the first barrier)
        VitLanes<G, JG> L;
        const int dl = tid - 64;
        int an = 0, a_own = 0;
        double pl0r[2] = {0.0, 0.0};
        if (tid >= 64 && tid < 128) {
            L.load(job, tables, dl, J);
            an = transcripts[job.tr_off + (L.n < N ? L.n : 0)];
            a_own = dl < N ? transcripts[job.tr_off + dl] : 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) pl0r[u] = dl + 64 * u < J ? tables.at(job.p_off, dl + 64 * u, 0, N) : -INFINITY;
        }
        framescore_cols_body<W4, 6, false>(job.lp, F, K, C, fs, cols, fs_smem, [&](const int k_lo, const int k_hi) {
            int k = k_lo;
            if (k_lo == 0) {                             // the first round: the prefetched values, then init_decoding on column 0
                a[dl] = a_own;
                Pl0[dl] = pl0r[0];
                Pl0[dl + 64] = pl0r[1];
                for (int j = dl + 128; j < K; j += 64) Pl0[j] = -INFINITY;
                L.init(F[an]);
                k = 1;
            }
            const float *fp = F + k * C + an;
            const double *pp = Pl0 + (k - 1);
            for (; k < k_hi; ++k, fp += C, ++pp) {
                double vout, vin, alt;
                int jm, jin;
                L.col_a(k, *fp, *pp, J, vout, vin, jm, jin, alt);
                if (L.col_b(k, vin, alt)) bp_l[k * N + L.n] = (uint8_t)jin;
            }
        });
        VSTAMP(1);
        if (tid >= 64 && tid < 128) vit_lanes_finalize<G, JG>(L, job, K, J, forced, Pl0, 0, &fin_n, &fin_j, &fin_score);
        __syncthreads();
        VSTAMP(5);
        if (fin_j < 0) {  // no comparable final hypothesis (NaN scores): traceback is None in the reference
            if (tid == 0) {
                status[vid] = MUCON_VIT_NO_HYPOTHESIS;
                n_seg[vid] = 0;
                score[vid] = -INFINITY;
            }
        } else {
            vit_traceback_and_labels(job, vid, fin_n, fin_j, fin_score, forced, nullptr, a, pre, labels, seg_len, n_seg, score, status, fs,
                                     bp_l, true);
        }
        VSTAMP(6);
    }
    if (done_flag) {
        __threadfence_system();                      // every thread's label stores are visible to the host ...
        __syncthreads();
        VSTAMP(7);
        if (threadIdx.x == 0) {
            __hip_atomic_store(const_cast<int32_t *>(done_flag), done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the flag
        }
    }
}


// SPW = transcript states per wave (a wave owns states wave, wave+16, ...): the per-state work of a column is
// written out SPW times in straight-line code -- all LDS reads first, then the adds, then the SPW independent
// DPP reductions, then the writes -- so that the LDS and DPP latencies of the states overlap instead of adding up.
template <int SPW>
__global__ __launch_bounds__(VIT_THREADS) void viterbi_dp_kernel(
    const mucon_viterbi_job *jobs, const int32_t *transcripts, const VitTab tables, const VitLabels labels,
    int32_t *seg_len, int32_t *n_seg, double *score, int32_t *status, char *ws, int C, int fs, int J) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const mucon_viterbi_job job = jobs[blockIdx.x];
    const int T = job.T, N = job.N;
    const int K = T / fs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x;             // 64 x (waves that own a transcript state), at least 8 C: see the launch
    const int nwaves = nthreads >> 6;
    const int vid = blockIdx.x;

    if (K < 1) {  // frame_scores[fs-1] does not exist: IndexError in the reference (viterbi.py:87)
        if (tid == 0) {
            status[vid] = MUCON_VIT_INDEX_ERROR;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }
    if (job.force_n < 0 && K > J * N) {  // every hypothesis has outlived max_length: empty set
        if (tid == 0) {
            status[vid] = MUCON_VIT_NO_HYPOTHESIS;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }

    // LDS carve (all dynamic, 16-byte aligned base):
    //   fin[16 B] | S[2][N][J] f64 | P[N][J] f64 | Fs[2][16][C] f32 | a[N] i32 | pre[N+1] i32
    double &fin_score = *reinterpret_cast<double *>(smem_raw);
    int &fin_n = *reinterpret_cast<int *>(smem_raw + 8);
    int &fin_j = *reinterpret_cast<int *>(smem_raw + 12);
    double *S = reinterpret_cast<double *>(smem_raw + 16);
    double *Pl = S + 2 * (size_t)N * J;
    float *Fs = reinterpret_cast<float *>(Pl + (size_t)N * J);
    int *a = reinterpret_cast<int *>(Fs + 2 * VIT_FCHUNK * C);
    int *pre = a + N;

    const float *F = reinterpret_cast<const float *>(ws + job.ws_off);
    const size_t f_bytes = ((size_t)K * C * sizeof(float) + 15) & ~(size_t)15;
    uint8_t *bp = reinterpret_cast<uint8_t *>(ws + job.ws_off + f_bytes);  // [K][N]

    for (int e = tid; e < N; e += nthreads) a[e] = transcripts[job.tr_off + e];
    for (int e = tid; e < N * J; e += nthreads) {
        const int n = e / J, j = e - n * J;
        Pl[e] = tables.at(job.p_off, j, n, N);
    }
    // frame-score chunks 0 and 1
    for (int e = tid; e < 2 * VIT_FCHUNK * C; e += nthreads) {
        const int col = e / C;
        Fs[e] = (col < K) ? F[e] : 0.f;
    }
    __syncthreads();

    // init_decoding (viterbi.py:81-90): score = 0.0 + frame_score(fs-1, a_0), float32
    if (tid == 0) {
        const float s0 = 0.0f + Fs[a[0]];
        S[0] = (double)s0;  // buffer 0, state 0, slot 0
    }
    __syncthreads();

    // slot -> current length index: at old column c = k-1 slot r holds j = (c - r) mod J
    int jr0 = (J - (lane % J)) % J;                    // c = 0, r = lane       (lane < J assumed for r)
    int jr1 = (J - ((lane + 64) % J)) % J;             // c = 0, r = lane + 64
    const bool has0 = lane < J, has1 = lane + 64 < J;  // J <= 128 slots per state
    float fpre[2] = {0.f, 0.f};

    // per-wave state set, hoisted out of the column loop
    int sm[SPW], sam[SPW], sapm[SPW];
    bool sv[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
        const int m = wave + nwaves * i;
        sv[i] = m < N;
        sm[i] = sv[i] ? m : 0;
        sam[i] = a[sm[i]];
        sapm[i] = a[sm[i] >= 1 ? sm[i] - 1 : 0];
    }
    const int l0 = has0 ? lane : J - 1, l1 = has1 ? lane + 64 : J - 1;   // clamped slot indices (reads stay in bounds)

    for (int k = 1; k < K; ++k) {
        double *So = S + (size_t)((k - 1) & 1) * N * J;
        double *Sn = S + (size_t)(k & 1) * N * J;
        // stage frame scores: chunk q+1 is fetched at the start of chunk q and stored half way
        const int kin = k & (VIT_FCHUNK - 1);
        const int q = k / VIT_FCHUNK;
        if (kin == 0 && k >= VIT_FCHUNK) {   // two elements per thread: nthreads >= 8 C covers the 16 C of a chunk
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + u * nthreads;
                const int col = (q + 1) * VIT_FCHUNK + e / C;
                fpre[u] = (e < VIT_FCHUNK * C && col < K) ? F[(size_t)(q + 1) * VIT_FCHUNK * C + e] : 0.f;
            }
        }
        if (kin == VIT_FCHUNK / 2 && k >= VIT_FCHUNK) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + u * nthreads;
                if (e < VIT_FCHUNK * C) Fs[((q + 1) & 1) * VIT_FCHUNK * C + e] = fpre[u];
            }
        }
        const float *Fk = Fs + (q & 1) * VIT_FCHUNK * C + kin * C;
        const int c_old = k - 1;
        const int kslot = k % J;
        const int k00 = c_old - jr0, k01 = c_old - jr1;  // entry columns of this lane's two slots

        // ---- all LDS reads of the column (unconditional; out-of-range work reads a clamped, valid address)
        float fm[SPW], fp[SPW];
        double own0[SPW], own1[SPW], prv0[SPW], prv1[SPW], pl0[SPW], pl1[SPW];
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int m = sm[i], pm = m >= 1 ? m - 1 : 0;
            fm[i] = Fk[sam[i]];
            fp[i] = Fk[sapm[i]];
            own0[i] = So[m * J + l0];
            own1[i] = So[m * J + l1];
            prv0[i] = So[pm * J + l0];
            prv1[i] = So[pm * J + l1];
            pl0[i] = Pl[pm * J + jr0];
            pl1[i] = Pl[pm * J + jr1];
        }
        // ---- (1) stay in state m (viterbi.py:96-104) and (2) the candidates for entering m from m-1
        //      (viterbi.py:105-121; the frame score is the OLD label's)
        Cand best[SPW];
        bool enter[SPW];
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int m = sm[i], pm = m >= 1 ? m - 1 : 0;
            if (sv[i] && has0 && alive_at(m, k00, J) && jr0 + 1 < J) Sn[m * J + lane] = add_frame(own0[i], fm[i], m);
            if (sv[i] && has1 && alive_at(m, k01, J) && jr1 + 1 < J) Sn[m * J + lane + 64] = add_frame(own1[i], fm[i], m);
            enter[i] = sv[i] && m >= 1 && k >= m && k <= J * m;
            Cand c0, c1;
            c0.v = (add_frame(prv0[i], fp[i], pm) + pl0[i]) + 0.0;
            c0.j = jr0;
            c1.v = (add_frame(prv1[i], fp[i], pm) + pl1[i]) + 0.0;
            c1.j = jr1;
            const bool a0 = has0 && alive_at(pm, k00, J), a1 = has1 && alive_at(pm, k01, J);
            if (!a0) {
                c0.v = -INFINITY;
                c0.j = -1;
            }
            if (!a1) {
                c1.v = -INFINITY;
                c1.j = -1;
            }
            best[i] = better(c0, c1);
        }
#pragma unroll
        for (int i = 0; i < SPW; ++i) best[i] = wave_best(best[i]);
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            if (enter[i] && lane == 0) {
                Sn[sm[i] * J + kslot] = best[i].v;
                bp[(size_t)k * N + sm[i]] = (uint8_t)best[i].j;
            }
        }
        jr0 = (jr0 + 1 == J) ? 0 : jr0 + 1;
        jr1 = (jr1 + 1 == J) ? 0 : jr1 + 1;
        __syncthreads();
    }

    // finalize_decoding (viterbi.py:125-138)
    const double *Sf = S + (size_t)((K - 1) & 1) * N * J;
    const bool forced = job.force_n >= 0 || K < N;
    if (forced) {
        // Degenerate outcomes of the reference: no hypothesis of the last transcript state has a
        // comparable score, every final score is -inf and the LAST hypothesis in dictionary order
        // wins (`>=`, viterbi.py:135).  For K < N that is (K-1, just entered); the NaN-length-model
        // cases are resolved on the host (mucon_amd/core/viterbi/viterbi.py) and passed in force_*.
        if (tid == 0) {
            fin_n = job.force_n >= 0 ? job.force_n : K - 1;
            fin_j = job.force_n >= 0 ? job.force_j : 0;
            fin_score = -INFINITY;
        }
    } else if (wave == 0) {
        const int nf = N - 1;
        const int c = K - 1;
        Cand best;
        best.v = -INFINITY;
        best.j = -1;
        // jr0/jr1 now describe column c = K-1
        if (has0 && alive_at(nf, c - jr0, J)) {
            Cand d;
            d.v = (Sf[nf * J + lane] + Pl[nf * J + jr0]) + 0.0;
            d.j = jr0;
            best = better(best, d);
        }
        if (has1 && alive_at(nf, c - jr1, J)) {
            Cand d;
            d.v = (Sf[nf * J + lane + 64] + Pl[nf * J + jr1]) + 0.0;
            d.j = jr1;
            best = better(best, d);
        }
        best = wave_best(best);
        if (lane == 0) {
            fin_n = nf;
            fin_j = best.j;
            fin_score = best.v;
        }
    }
    __syncthreads();
    if (fin_j < 0) {  // no comparable final hypothesis (NaN scores): traceback is None in the reference
        if (tid == 0) {
            status[vid] = MUCON_VIT_NO_HYPOTHESIS;
            n_seg[vid] = 0;
            score[vid] = -INFINITY;
        }
        return;
    }

    vit_traceback_and_labels(job, vid, fin_n, fin_j, fin_score, forced, bp, a, pre, labels, seg_len, n_seg, score, status, fs);
}

thread_local char g_err[256];

}  // namespace

int g_vit_lanes = 1;   // MUCON_VIT_LANES=0: always the one-wave-per-state LDS kernel (tests)
// Calls of up to this many videos are LATENCY calls (pair kernel, back-pointers in LDS, results through the library's own pinned
// staging buffer); larger ones are throughput calls (two launches, many videos per CU, a pinned caller array written in place).
// include/mucon_hip.h states the same number as MUCON_VIT_LATENCY_VIDEOS; mucon_amd/ops.py reads it from _lib.
constexpr int kVitLatencyVideos = MUCON_VIT_LATENCY_VIDEOS;

void mucon_internal_set_error(const char *msg);  // mucon_hip.hip: feeds mucon_last_error()
#define VIT_FAIL(code)                    \
    do {                                  \
        mucon_internal_set_error(g_err);  \
        return (code);                    \
    } while (0)

#if VIT_STAMP
extern "C" int mucon_test_vit_stamps(long long *out) {
    if (hipDeviceSynchronize() != hipSuccess) return MUCON_E_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vit_stamps), sizeof(long long) * 16) == hipSuccess ? MUCON_OK : MUCON_E_HIP;
}
#endif

extern "C" size_t mucon_viterbi_job_workspace_bytes(int32_t T, int32_t C, int32_t N, int32_t fs) {
    const size_t K = fs > 0 ? (size_t)(T / fs) : 0;
    const size_t f_bytes = (K * (size_t)C * sizeof(float) + 15) & ~(size_t)15;
    const size_t bp_bytes = (K * (size_t)N + 15) & ~(size_t)15;
    return f_bytes + bp_bytes + 16;
}

// Launches of one decode: `fused` = the one-launch kernel (one short video, <= 16 states), else frame scores + DP.
static int vit_launch(int32_t n_videos, const mucon_viterbi_job *jobs, int32_t C, int32_t fs, int32_t max_len, int32_t max_N,
                      const int32_t *transcripts, const VitTab length_tables, const VitLabels labels, int32_t *seg_len, int32_t *n_seg,
                      double *score, int32_t *status, void *workspace, hipStream_t s, bool fused, int max_K, bool cols_ok,
                      volatile int32_t *done_flag, int32_t done_value, const mucon_viterbi_job *host_jobs,
                      unsigned long long *progress_words = nullptr, hipEvent_t tables_ready = nullptr) {
    // tables_ready: an event the DP launch (not the frame-score launch, which reads neither transcripts nor length tables) has to wait for
    // host_jobs: the job table where the host can read it (the one-launch kernel takes its job as a kernel argument), or nullptr;
    // max_K: the longest video's column count when the caller knows it (LDS sizing of the latency paths), else 0
    if (fs <= 0 || max_len < fs || C <= 0 || C > 64 || max_N <= 0) {
        snprintf(g_err, sizeof(g_err), "viterbi: unsupported arguments (C=%d must be <= 64, fs=%d, max_len=%d, max_N=%d)",
                 C, fs, max_len, max_N);
        VIT_FAIL(MUCON_E_ARG);
    }
    const int J = max_len / fs;
    if (J > 128) {
        snprintf(g_err, sizeof(g_err), "viterbi: max_len/fs = %d length slots > 128 not supported", J);
        VIT_FAIL(MUCON_E_ARG);
    }
    const size_t smem = 16 + (size_t)3 * max_N * J * sizeof(double) + (size_t)2 * VIT_FCHUNK * C * sizeof(float) +
                        (size_t)(2 * max_N + 1) * sizeof(int) + 16;
    const bool lanes = g_vit_lanes && max_N <= 128 && J <= 66;   // the register kernel (below) needs no such table
    if (!lanes && smem > 160 * 1024 - 64) {
        snprintf(g_err, sizeof(g_err), "viterbi: transcript of %d states x %d slots needs %zu B of LDS (> 160 KiB)",
                 max_N, J, smem);
        VIT_FAIL(MUCON_E_ARG);
    }
    // the > 64 KB LDS opt-ins are per device: keyed by the current device
    static int attr_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "viterbi: hipGetDevice failed");
        VIT_FAIL(MUCON_E_HIP);
    }
    if (attr_dev != dev) {
        const void *ks[4] = {reinterpret_cast<const void *>(viterbi_dp_kernel<1>), reinterpret_cast<const void *>(viterbi_dp_kernel<2>),
                             reinterpret_cast<const void *>(viterbi_dp_kernel<4>), reinterpret_cast<const void *>(viterbi_dp_kernel<8>)};
        bool ok = true;
        for (const void *kp : ks) ok = ok && hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_framescore_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       2 * FS_ROWS * 64 * 4) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_framescore_cols_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, FSC_MAX_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_framescore_cols_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, FSC_MAX_LDS) == hipSuccess;
#define VF_ATTR(G, JG)                                                                                                          \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_fused_kernel<G, JG, true>),                            \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, VF_DYN_MAX) == hipSuccess &&                      \
         hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_fused_kernel<G, JG, false>),                                 \
                             hipFuncAttributeMaxDynamicSharedMemorySize, VF_DYN_MAX) == hipSuccess
        VF_ATTR(8, 9);
        VF_ATTR(4, 17);
#undef VF_ATTR
#define VP_ATTR(G, JG, NW)                                                                                                      \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_pair_kernel<G, JG, NW, true>),                         \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, VP_DYN_MAX) == hipSuccess &&                      \
         hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_pair_kernel<G, JG, NW, false>),                              \
                             hipFuncAttributeMaxDynamicSharedMemorySize, VP_DYN_MAX) == hipSuccess
        VP_ATTR(8, 9, 1);
        VP_ATTR(4, 17, 1);
        VP_ATTR(8, 9, 4);
        VP_ATTR(4, 17, 4);
        VP_ATTR(4, 17, 8);
#undef VP_ATTR
#define VL_ATTR(G, JG, NW)                                                                                                      \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_dp_lanes_kernel<G, JG, NW>),                           \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, VL_BP_LDS_MAX) == hipSuccess
        VL_ATTR(8, 9, 1);
        VL_ATTR(4, 17, 1);
        VL_ATTR(8, 9, 4);
        VL_ATTR(4, 17, 4);
        VL_ATTR(4, 17, 8);
#undef VL_ATTR
        if (!ok) {
            snprintf(g_err, sizeof(g_err), "viterbi: hipFuncSetAttribute failed");
            VIT_FAIL(MUCON_E_HIP);
        }
        attr_dev = dev;
    }
    const bool w4 = (((fs + 3) >> 2) & 3) == 0;
    if (fused && lanes && max_N <= 16 && max_K <= VF_MAX_K && cols_ok && host_jobs && n_videos == 1) {
        // phase 1's buffers + the frame scores [K][C] + the back-pointers [K][N] share the dynamic LDS: as many columns per chunk as fit beside them
        const size_t f_bytes = (((size_t)max_K * C * sizeof(float) + (size_t)max_K * max_N) + 15) & ~(size_t)15;
        int cols = FS_ROWS / ((fs + 3) & ~3);
        while (cols > 1 && (size_t)fsc_floats(C, fs, cols) * 4 + f_bytes > VF_DYN_MAX) --cols;
        if ((size_t)fsc_floats(C, fs, cols) * 4 + f_bytes <= VF_DYN_MAX) {
            const int dyn = fsc_floats(C, fs, cols);
            const size_t bytes = (size_t)dyn * 4 + f_bytes;
#define VF_LAUNCH(G, JG)                                                                                                          \
    do {                                                                                                                          \
        if (w4) hipLaunchKernelGGL((viterbi_fused_kernel<G, JG, true>), dim3(1), dim3(FSC_THREADS), bytes, s, host_jobs[0], transcripts, \
                                   length_tables, labels, seg_len, n_seg, score, status, C, fs, J, cols, dyn, done_flag, done_value); \
        else hipLaunchKernelGGL((viterbi_fused_kernel<G, JG, false>), dim3(1), dim3(FSC_THREADS), bytes, s, host_jobs[0], transcripts,  \
                                length_tables, labels, seg_len, n_seg, score, status, C, fs, J, cols, dyn, done_flag, done_value);   \
    } while (0)
            if (max_N <= 8) VF_LAUNCH(8, 9);
            else VF_LAUNCH(4, 17);
#undef VF_LAUNCH
            if (hipGetLastError() != hipSuccess) {
                snprintf(g_err, sizeof(g_err), "viterbi: kernel launch failed");
                VIT_FAIL(MUCON_E_HIP);
            }
            return 1;   // (fused: the completion flag, if any, will be published)
        }
    }
    // a few videos of known size: both phases as two workgroups of ONE launch, the decode running beside the chain (viterbi_pair_kernel)
    // (latency calls only: measured at 64 videos per call the pair is slower than two launches -- 17.0 against 15.3 us per video at
    // T = 16,384: every workgroup then holds the chain's 112 KB of LDS.  Any number would be SAFE: workgroups are dispatched in
    // order, every adding workgroup before any decoding one, and the adding ones wait for nothing.)
    if (lanes && cols_ok && fs <= FS_ROWS && max_K > 0 && n_videos <= kVitLatencyVideos && done_flag && progress_words) {
        int cols = FS_ROWS / ((fs + 3) & ~3);
        while (cols > 1 && (size_t)fsc_floats(C, fs, cols) * 4 > VP_DYN_MAX) --cols;
        const size_t fsc_bytes = (size_t)fsc_floats(C, fs, cols) * 4;
        if (fsc_bytes <= VP_DYN_MAX) {
            const size_t bp_need = ((size_t)max_K * max_N + 15) & ~(size_t)15;
            const size_t bp_lds = bp_need <= fsc_bytes ? bp_need : 0;      // (the decoding workgroup uses the same dynamic LDS for them)
            volatile int32_t *flag1 = n_videos == 1 ? done_flag : nullptr;
            const uint32_t seq = (uint32_t)done_value;
#define VP_LAUNCH(G, JG, NW)                                                                                                        \
    do {                                                                                                                            \
        if (w4) hipLaunchKernelGGL((viterbi_pair_kernel<G, JG, NW, true>), dim3(n_videos, 2), dim3(FSC_THREADS), fsc_bytes, s, jobs,  \
                                   transcripts, length_tables, labels, seg_len, n_seg, score, status, static_cast<char *>(workspace), \
                                   C, fs, J, cols, (int)bp_lds, flag1, done_value, progress_words, seq);                             \
        else hipLaunchKernelGGL((viterbi_pair_kernel<G, JG, NW, false>), dim3(n_videos, 2), dim3(FSC_THREADS), fsc_bytes, s, jobs,    \
                                transcripts, length_tables, labels, seg_len, n_seg, score, status, static_cast<char *>(workspace),   \
                                C, fs, J, cols, (int)bp_lds, flag1, done_value, progress_words, seq);                                \
    } while (0)
            if (max_N <= 8) VP_LAUNCH(8, 9, 1);
            else if (max_N <= 16) VP_LAUNCH(4, 17, 1);
            else if (max_N <= 32) VP_LAUNCH(8, 9, 4);
            else if (max_N <= 64) VP_LAUNCH(4, 17, 4);
            else VP_LAUNCH(4, 17, 8);
#undef VP_LAUNCH
            if (hipGetLastError() != hipSuccess) {
                snprintf(g_err, sizeof(g_err), "viterbi: kernel launch failed");
                VIT_FAIL(MUCON_E_HIP);
            }
            return flag1 ? 1 : MUCON_OK;
        }
    }
    const size_t fs_smem = (size_t)2 * FS_ROWS * C * sizeof(float);
    if (fs <= FS_ROWS && cols_ok) {   // whole columns per chunk: the pipelined chain
        // columns padded to whole b128 reads in LDS; as many columns per chunk as 256 rows and the LDS hold
        int cols = FS_ROWS / ((fs + 3) & ~3);
        while (cols > 1 && fsc_floats(C, fs, cols) * 4 > FSC_MAX_LDS) --cols;
        if (w4)
            hipLaunchKernelGGL(viterbi_framescore_cols_kernel<true>, dim3(n_videos), dim3(FSC_THREADS), (size_t)fsc_floats(C, fs, cols) * sizeof(float), s,
                               jobs, static_cast<char *>(workspace), C, fs, cols);
        else
            hipLaunchKernelGGL(viterbi_framescore_cols_kernel<false>, dim3(n_videos), dim3(FSC_THREADS), (size_t)fsc_floats(C, fs, cols) * sizeof(float), s,
                               jobs, static_cast<char *>(workspace), C, fs, cols);
    } else {
        hipLaunchKernelGGL(viterbi_framescore_kernel, dim3(n_videos), dim3(FS_THREADS), fs_smem, s, jobs,
                           static_cast<char *>(workspace), C, fs);
    }
    if (tables_ready && hipStreamWaitEvent(s, tables_ready, 0) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "viterbi: hipStreamWaitEvent failed");
        VIT_FAIL(MUCON_E_HIP);
    }
    // the DP: up to 66 length slots run in the registers of one wave (<= 16 states), four (<= 64) or eight (<= 128) ...
#define VL_LAUNCH(G, JG, NW)                                                                                          \
    hipLaunchKernelGGL((viterbi_dp_lanes_kernel<G, JG, NW>), dim3(n_videos), dim3(NW == 1 ? VL_THREADS : 64 * NW), bp_lds, s, jobs, \
                       transcripts, length_tables, labels, seg_len, n_seg, score, status, static_cast<char *>(workspace), C, \
                       fs, J, (int)bp_lds, flag1, done_value)
    if (lanes) {
        // back-pointers in LDS when the caller knows the sizes and they fit -- the traceback is a chain of N dependent reads (~100
        // cycles each from LDS, ~1,500 from memory: 45 us of a config-5 video).  Latency calls: up to 96 KB; larger calls: up to
        // 40 KB (beside the kernel's ~35 KB of static LDS that still leaves two workgroups per CU), and only while the call is
        // at most two workgroups per CU anyway -- beyond that the LDS goes to more resident videos.  One video: the kernel
        // publishes the completion flag.
        const size_t bp_need = ((size_t)max_K * max_N + 15) & ~(size_t)15;
        const size_t bp_cap = n_videos <= kVitLatencyVideos ? VL_BP_LDS_MAX : (n_videos <= 512 ? (size_t)40 * 1024 : 0);
        const size_t bp_lds = (max_K > 0 && bp_need <= bp_cap) ? bp_need : 0;
        volatile int32_t *flag1 = n_videos == 1 ? done_flag : nullptr;
        // lanes per state x waves, by measurement (single T = 4,000 .. 16,384 decodes): more lanes per state shorten the per-lane
        // slot loop, more waves pay a barrier per column, and past four waves two share a SIMD
        if (max_N <= 8) VL_LAUNCH(8, 9, 1);
        else if (max_N <= 16) VL_LAUNCH(4, 17, 1);
        else if (max_N <= 32) VL_LAUNCH(8, 9, 4);
        else if (max_N <= 64) VL_LAUNCH(4, 17, 4);
        else VL_LAUNCH(4, 17, 8);
#undef VL_LAUNCH
        if (hipGetLastError() != hipSuccess) {
            snprintf(g_err, sizeof(g_err), "viterbi: kernel launch failed");
            VIT_FAIL(MUCON_E_HIP);
        }
        return flag1 ? 1 : MUCON_OK;
    }
    // ... longer transcripts or more slots: one wave per state, the column in LDS
    const int spw = (max_N + 15) / 16;   // transcript states per wave
    // one wave per SPW transcript states; fewer waves make the per-column barrier cheaper (a Breakfast-typical transcript has 6
    // states: 6 waves instead of 16); the frame-score staging needs 8 C threads (two elements each)
    const int spwt = spw <= 1 ? 1 : (spw <= 2 ? 2 : (spw <= 4 ? 4 : 8));   // the instantiated states-per-wave
    int dp_threads = 64 * ((max_N + spwt - 1) / spwt);
    const int min_threads = (8 * C + 63) / 64 * 64;
    dp_threads = dp_threads < min_threads ? min_threads : dp_threads;
    dp_threads = dp_threads > VIT_THREADS ? VIT_THREADS : dp_threads;
#define VIT_LAUNCH(SPW)                                                                                              \
    hipLaunchKernelGGL(viterbi_dp_kernel<SPW>, dim3(n_videos), dim3(dp_threads), smem, s, jobs, transcripts,           \
                       length_tables, labels, seg_len, n_seg, score, status, static_cast<char *>(workspace), C, fs, J)
    if (spw <= 1) VIT_LAUNCH(1);
    else if (spw <= 2) VIT_LAUNCH(2);
    else if (spw <= 4) VIT_LAUNCH(4);
    else if (spw <= 8) VIT_LAUNCH(8);
    else {
        snprintf(g_err, sizeof(g_err), "viterbi: transcripts longer than 128 states are not supported (max_N=%d)", max_N);
        VIT_FAIL(MUCON_E_ARG);
    }
#undef VIT_LAUNCH
    if (hipGetLastError() != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "viterbi: kernel launch failed");
        VIT_FAIL(MUCON_E_HIP);
    }
    return MUCON_OK;
}

static int vit_decode_batch_impl(int32_t n_videos, const mucon_viterbi_job *jobs, int32_t C, int32_t fs,
                                 int32_t max_len, int32_t max_N, const int32_t *transcripts, const VitTab length_tables,
                                 void *labels, int32_t label_format, int32_t *seg_len, int32_t *n_seg, double *score,
                                 int32_t *status, void *workspace, void *stream) {
    if (n_videos <= 0) return MUCON_OK;
    if (label_format < MUCON_VIT_LABELS_I32 || label_format > MUCON_VIT_LABELS_NONE || (!labels && label_format != MUCON_VIT_LABELS_NONE)) {
        snprintf(g_err, sizeof(g_err), "viterbi: label_format %d / labels %s", label_format, labels ? "given" : "NULL");
        VIT_FAIL(MUCON_E_ARG);
    }
    // the job table lives on the device: the emission pointers cannot be inspected here, so the 16-byte loads of the pipelined
    // frame-score kernel are only taken for class counts where every row of an aligned base is aligned (the caller keeps `lp`
    // 16-byte aligned: include/mucon_hip.h)
    const int rc = vit_launch(n_videos, jobs, C, fs, max_len, max_N, transcripts, length_tables, VitLabels{labels, label_format}, seg_len, n_seg, score, status,
                              workspace, static_cast<hipStream_t>(stream), false, 0, (C & 3) == 0, nullptr, 0, nullptr);
    return rc > 0 ? MUCON_OK : rc;
}
extern "C" int mucon_viterbi_decode_batch(int32_t n_videos, const mucon_viterbi_job *jobs, int32_t C, int32_t fs,
                                          int32_t max_len, int32_t max_N, const int32_t *transcripts, const double *length_tables,
                                          void *labels, int32_t label_format, int32_t *seg_len, int32_t *n_seg, double *score,
                                          int32_t *status, void *workspace, void *stream) {
    return vit_decode_batch_impl(n_videos, jobs, C, fs, max_len, max_N, transcripts, VitTab{length_tables, nullptr, fs, max_len}, labels, label_format,
                                 seg_len, n_seg, score, status, workspace, stream);
}
extern "C" int mucon_viterbi_decode_batch_poisson(int32_t n_videos, const mucon_viterbi_job *jobs, int32_t C, int32_t fs,
                                                  int32_t max_len, int32_t max_N, const int32_t *transcripts, const double *poisson_params,
                                                  const double *log_fact, void *labels, int32_t label_format, int32_t *seg_len, int32_t *n_seg,
                                                  double *score, int32_t *status, void *workspace, void *stream) {
    if (!poisson_params || !log_fact) {
        snprintf(g_err, sizeof(g_err), "viterbi: null Poisson parameters / log-factorial row");
        VIT_FAIL(MUCON_E_ARG);
    }
    return vit_decode_batch_impl(n_videos, jobs, C, fs, max_len, max_N, transcripts, VitTab{poisson_params, log_fact, fs, max_len}, labels, label_format,
                                 seg_len, n_seg, score, status, workspace, stream);
}

// ---- decode with host-side inputs and outputs (what Viterbi.decode is: numpy in, Python objects out) -------------------------
// Library-owned, per device, grown on demand: a pinned host buffer the kernels READ the job table / transcripts / length tables from
// (no upload call), a pinned host buffer they WRITE the results to (no download call), device scratch for frame scores and
// back-pointers.  One host thread per process (include/mucon_hip.h), so no locking.
namespace {
struct VitHostState {
    char *pin_in = nullptr, *pin_out = nullptr, *ws = nullptr;
    char *dev_in = nullptr;                   // device copy of the pinned input buffer's tables / transcripts (throughput calls)
    size_t dev_in_cap = 0;
    hipStream_t side = nullptr;               // ... made on this stream, under the frame-score launch
    hipEvent_t tables_ready = nullptr;
    unsigned long long *progress = nullptr;   // the pair kernel's "columns done" words: kVitLatencyVideos x 64 bytes, zeroed once, nothing else
    size_t in_cap = 0, out_cap = 0, ws_cap = 0;
    int32_t seq = 0;
};
VitHostState g_vh[16];

int vh_grow(char **buf, size_t *cap, size_t need, bool pinned) {
    if (need <= *cap) return MUCON_OK;
    if (hipDeviceSynchronize() != hipSuccess) return MUCON_E_HIP;     // nothing may still be using the old buffer
    if (*buf) (void)(pinned ? hipHostFree(*buf) : hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    const size_t want = need + need / 2 + 4096;
    const hipError_t e = pinned ? hipHostMalloc(reinterpret_cast<void **>(buf), want, hipHostMallocMapped | hipHostMallocCoherent)   // fine-grained: the kernels' stores (results, completion flag) are visible to the host while the launch runs
                                : hipMalloc(reinterpret_cast<void **>(buf), want);
    if (e != hipSuccess) return MUCON_E_HIP;
    *cap = want;
    return MUCON_OK;
}
inline size_t up16(size_t n) { return (n + 15) & ~(size_t)15; }
double g_vh_phase_us[4];
inline double vh_now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" int mucon_test_vit_host_phases(double *us4) {
    if (!us4) return MUCON_E_ARG;
    for (int i = 0; i < 4; ++i) us4[i] = g_vh_phase_us[i];
    return MUCON_OK;
}

// log_fact != nullptr: videos[v].table is the video's [3][N] PoissonModel parameter rows, the length scores are built on the device (VitTab)
static int vit_decode_host_impl(int32_t n_videos, const mucon_viterbi_video *videos, const double *log_fact, int32_t C, int32_t fs, int32_t max_len,
                                double *score, int32_t *n_seg, int32_t *status, void *labels, int32_t label_format,
                                int32_t *seg_len, void *stream) {
    if (n_videos <= 0) return MUCON_OK;
    const double t_begin = vh_now_us();
    if (label_format < MUCON_VIT_LABELS_I32 || label_format > MUCON_VIT_LABELS_NONE) {
        snprintf(g_err, sizeof(g_err), "viterbi: label_format %d", label_format);
        VIT_FAIL(MUCON_E_ARG);
    }
    const size_t lab_elem = label_format == MUCON_VIT_LABELS_I32 ? 4 : (label_format == MUCON_VIT_LABELS_U8 ? 1 : 0);
    if (!videos || !score || !n_seg || !status || (!labels && lab_elem) || !seg_len || fs <= 0 || max_len < fs) {
        snprintf(g_err, sizeof(g_err), "viterbi: null argument / bad frame sampling");
        VIT_FAIL(MUCON_E_ARG);
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
        snprintf(g_err, sizeof(g_err), "viterbi: unsupported device index");
        VIT_FAIL(MUCON_E_HIP);
    }
    VitHostState &st = g_vh[dev];
    const int J = max_len / fs;
    size_t sum_T = 0, sum_N = 0, ws_bytes = 0;
    int max_N = 0, max_K = 0;
    bool aligned = (C & 3) == 0;
    for (int v = 0; v < n_videos; ++v) {
        const mucon_viterbi_video &q = videos[v];
        if (q.T < 0 || q.N <= 0 || !q.lp || !q.transcript || !q.table) {
            snprintf(g_err, sizeof(g_err), "viterbi: video %d: T=%d N=%d or a null pointer", v, q.T, q.N);
            VIT_FAIL(MUCON_E_ARG);
        }
        sum_T += (size_t)(q.T > 0 ? q.T : 1);
        sum_N += (size_t)q.N;
        ws_bytes += (mucon_viterbi_job_workspace_bytes(q.T, C, q.N, fs) + 255) & ~(size_t)255;
        max_N = q.N > max_N ? q.N : max_N;
        max_K = q.T / fs > max_K ? q.T / fs : max_K;
        aligned = aligned && (reinterpret_cast<uintptr_t>(q.lp) & 15) == 0;   // the pipelined frame-score kernel loads 16 bytes per lane
    }
    // input staging: [jobs][length tables][transcripts]; output staging: [flag, 64 B][score][n_seg][status][seg_len][labels]
    const size_t tab_rows = log_fact ? 3 : (size_t)J;   // doubles per transcript state in the staging buffer; parameter form: + the shared lf[J] behind them
    const size_t o_tab = up16(sizeof(mucon_viterbi_job) * n_videos), o_tr = o_tab + up16(sizeof(double) * (tab_rows * sum_N + (log_fact ? (size_t)J : 0)));
    const size_t in_bytes = o_tr + up16(sizeof(int32_t) * sum_N);
    const size_t o_score = 64, o_nseg = o_score + up16(8 * (size_t)n_videos), o_stat = o_nseg + up16(4 * (size_t)n_videos);
    // `labels` is by far the largest output (4 T bytes per video).  When the caller's array is itself pinned host memory the kernels
    // write it directly (no staging copy: at 256 videos of T = 16,384 the copy out of the staging buffer was ~1 ms of a 2.6 ms call).
    void *labels_dev = nullptr;
    if (n_videos >= kVitLatencyVideos && lab_elem) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, labels) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
            labels_dev = at.devicePointer;
        else
            (void)hipGetLastError();   // (an ordinary host pointer: the query fails, and leaves that error behind)
    }
    const size_t o_seg = o_stat + up16(4 * (size_t)n_videos), o_lab = o_seg + up16(4 * sum_N),
                 out_bytes = o_lab + (labels_dev ? 0 : up16(lab_elem * sum_T));
    if (vh_grow(&st.pin_in, &st.in_cap, in_bytes, true) != MUCON_OK || vh_grow(&st.pin_out, &st.out_cap, out_bytes, true) != MUCON_OK ||
        vh_grow(&st.ws, &st.ws_cap, ws_bytes + 256, false) != MUCON_OK) {
        snprintf(g_err, sizeof(g_err), "viterbi: staging allocation failed (%zu / %zu / %zu bytes)", in_bytes, out_bytes, ws_bytes);
        VIT_FAIL(MUCON_E_HIP);
    }
    if (!st.progress) {
        if (hipMalloc(reinterpret_cast<void **>(&st.progress), 64 * (size_t)kVitLatencyVideos) != hipSuccess ||
            hipMemset(st.progress, 0, 64 * (size_t)kVitLatencyVideos) != hipSuccess) {   // (synchronous on the null stream: done before any launch below)
            st.progress = nullptr;
            snprintf(g_err, sizeof(g_err), "viterbi: allocation of the progress words failed");
            VIT_FAIL(MUCON_E_HIP);
        }
    }
    mucon_viterbi_job *jobs = reinterpret_cast<mucon_viterbi_job *>(st.pin_in);
    double *tabs = reinterpret_cast<double *>(st.pin_in + o_tab);
    int32_t *trs = reinterpret_cast<int32_t *>(st.pin_in + o_tr);
    size_t tr_off = 0, lab_off = 0, ws_off = 0;
    for (int v = 0; v < n_videos; ++v) {
        const mucon_viterbi_video &q = videos[v];
        mucon_viterbi_job &j = jobs[v];
        j.lp = q.lp;
        j.tr_off = (int64_t)tr_off;
        j.p_off = (int64_t)(tr_off * tab_rows);
        j.label_off = (int64_t)lab_off;
        j.seg_off = (int64_t)tr_off;
        j.ws_off = (int64_t)ws_off;
        j.T = q.T;
        j.N = q.N;
        j.force_n = q.force_n;
        j.force_j = q.force_j;
        memcpy(tabs + tr_off * tab_rows, q.table, sizeof(double) * tab_rows * q.N);
        memcpy(trs + tr_off, q.transcript, sizeof(int32_t) * (size_t)q.N);
        tr_off += (size_t)q.N;
        lab_off += (size_t)(q.T > 0 ? q.T : 1);
        ws_off += (mucon_viterbi_job_workspace_bytes(q.T, C, q.N, fs) + 255) & ~(size_t)255;
    }
    if (log_fact) memcpy(tabs + tab_rows * sum_N, log_fact, sizeof(double) * (size_t)J);
    char *din = nullptr, *dout = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&din), st.pin_in, 0) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&dout), st.pin_out, 0) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "viterbi: hipHostGetDevicePointer failed");
        VIT_FAIL(MUCON_E_HIP);
    }
    volatile int32_t *flag_h = reinterpret_cast<volatile int32_t *>(st.pin_out);
    const int32_t seq = ++st.seq == 0 ? ++st.seq : st.seq;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Throughput calls with large tables: every decoding workgroup reads its transcript and its J x N length scores (33 KB at N = 64) when
    // it starts -- from PINNED HOST memory, i.e. over PCIe, all workgroups at once: 9 MB at 256 videos of config 5, ~0.2 ms in front of
    // the first column (r4: why the DP launch took 530 us at 256 in flight and 315 us at 9; profiles/r04_viterbi_inflight_sweep.txt).
    // They are copied to the device on a side stream instead, UNDER the frame-score launch, which needs neither; the DP launch waits for
    // the copy's event.  (The job table stays pinned: 64 bytes per workgroup.)
    const char *tab_base = din;
    hipEvent_t wait_ev = nullptr;
    if (n_videos > kVitLatencyVideos && in_bytes - o_tab >= ((size_t)1 << 20)) {
        if (!st.side && (hipStreamCreateWithFlags(&st.side, hipStreamNonBlocking) != hipSuccess ||
                         hipEventCreateWithFlags(&st.tables_ready, hipEventDisableTiming) != hipSuccess)) {
            snprintf(g_err, sizeof(g_err), "viterbi: side stream / event creation failed");
            VIT_FAIL(MUCON_E_HIP);
        }
        if (vh_grow(&st.dev_in, &st.dev_in_cap, in_bytes, false) != MUCON_OK ||
            hipMemcpyAsync(st.dev_in + o_tab, st.pin_in + o_tab, in_bytes - o_tab, hipMemcpyHostToDevice, st.side) != hipSuccess ||
            hipEventRecord(st.tables_ready, st.side) != hipSuccess) {
            snprintf(g_err, sizeof(g_err), "viterbi: table upload failed");
            VIT_FAIL(MUCON_E_HIP);
        }
        tab_base = st.dev_in;
        wait_ev = st.tables_ready;
    }
    const double t_staged = vh_now_us();
    // latency path: a handful of short videos in ONE launch each; throughput path: two launches whose second packs many per CU
    const bool want_fused = n_videos == 1;
    const int rc = vit_launch(n_videos, reinterpret_cast<const mucon_viterbi_job *>(din), C, fs, max_len, max_N,
                              reinterpret_cast<const int32_t *>(tab_base + o_tr),
                              VitTab{reinterpret_cast<const double *>(tab_base + o_tab),
                                     log_fact ? reinterpret_cast<const double *>(tab_base + o_tab) + tab_rows * sum_N : nullptr, fs, max_len},
                              VitLabels{labels_dev ? labels_dev : static_cast<void *>(dout + o_lab), label_format}, reinterpret_cast<int32_t *>(dout + o_seg),
                              reinterpret_cast<int32_t *>(dout + o_nseg), reinterpret_cast<double *>(dout + o_score),
                              reinterpret_cast<int32_t *>(dout + o_stat), st.ws, s, want_fused, max_K, aligned,
                              reinterpret_cast<volatile int32_t *>(dout), seq, jobs, st.progress, wait_ev);
    if (rc < 0) return rc;
    const double t_launched = vh_now_us();
    bool done = false;
    if (rc > 0) {   // a one-video call's last kernel publishes the flag: spin on it (a stream synchronisation costs several microseconds more)
        for (long spin = 0; spin < 40000000L; ++spin) {
            if (__atomic_load_n(const_cast<const int32_t *>(flag_h), __ATOMIC_ACQUIRE) == seq) {
                done = true;
                break;
            }
            __builtin_ia32_pause();
        }
    }
    if (!done && hipStreamSynchronize(s) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "viterbi: the decode failed on the device: %s", hipGetErrorString(hipGetLastError()));
        VIT_FAIL(MUCON_E_HIP);
    }
    const double t_waited = vh_now_us();
    memcpy(score, st.pin_out + o_score, 8 * (size_t)n_videos);
    memcpy(n_seg, st.pin_out + o_nseg, 4 * (size_t)n_videos);
    memcpy(status, st.pin_out + o_stat, 4 * (size_t)n_videos);
    memcpy(seg_len, st.pin_out + o_seg, 4 * sum_N);
    if (!labels_dev && lab_elem) memcpy(labels, st.pin_out + o_lab, lab_elem * sum_T);
    g_vh_phase_us[0] = t_staged - t_begin;
    g_vh_phase_us[1] = t_launched - t_staged;
    g_vh_phase_us[2] = t_waited - t_launched;
    g_vh_phase_us[3] = vh_now_us() - t_waited;
    return MUCON_OK;
}

extern "C" int mucon_viterbi_decode_host(int32_t n_videos, const mucon_viterbi_video *videos, int32_t C, int32_t fs, int32_t max_len,
                                         double *score, int32_t *n_seg, int32_t *status, void *labels, int32_t label_format,
                                         int32_t *seg_len, void *stream) {
    return vit_decode_host_impl(n_videos, videos, nullptr, C, fs, max_len, score, n_seg, status, labels, label_format, seg_len, stream);
}
extern "C" int mucon_viterbi_decode_host_poisson(int32_t n_videos, const mucon_viterbi_video *videos, const double *log_fact, int32_t C, int32_t fs,
                                                 int32_t max_len, double *score, int32_t *n_seg, int32_t *status, void *labels,
                                                 int32_t label_format, int32_t *seg_len, void *stream) {
    if (!log_fact) {
        snprintf(g_err, sizeof(g_err), "viterbi: null log-factorial row");
        VIT_FAIL(MUCON_E_ARG);
    }
    return vit_decode_host_impl(n_videos, videos, log_fact, C, fs, max_len, score, n_seg, status, labels, label_format, seg_len, stream);
}

// Test hook (include/mucon_hip_test.h): the length rows the kernels build from a [3][N] parameter block -- out[j][n] = VitTab::at -- so that a test can hold
// them against PoissonModel.rows_for bit for bit.  All pointers DEVICE.
__global__ void vit_rows_probe_kernel(const VitTab tab, int N, int J, double *out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N * J; e += gridDim.x * blockDim.x) out[e] = tab.at(0, e / N, e % N, N);
}
extern "C" int mucon_test_vit_rows(const double *poisson_params, const double *log_fact, int32_t N, int32_t J, int32_t fs, int32_t max_len,
                                   double *out, void *stream) {
    if (!poisson_params || !log_fact || !out || N < 1 || J < 1 || fs < 1) {
        snprintf(g_err, sizeof(g_err), "test_vit_rows: bad arguments");
        VIT_FAIL(MUCON_E_ARG);
    }
    hipLaunchKernelGGL(vit_rows_probe_kernel, dim3(64), dim3(256), 0, static_cast<hipStream_t>(stream), VitTab{poisson_params, log_fact, fs, max_len}, N, J, out);
    if (hipGetLastError() != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "test_vit_rows: launch failed");
        VIT_FAIL(MUCON_E_HIP);
    }
    return MUCON_OK;
}
