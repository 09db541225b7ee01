// NT GEMM core on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact f32 = an fmaf chain):
//
//     out[t][n] = epilogue( sum_{tap, c} A[t + (tap - taps/2) * tap_step][c] * W[n][tap*Kc + c] )
//
// One workgroup = 256 threads = 4 waves computes BM time steps x 128 output channels of one video, BM in
// {128, 64, 32} picked per level (nt_pick_bm: the coarse levels get many short workgroups, the fine ones two
// workgroups per CU).  A and W tiles (BM / 128 rows x 32 k) are staged global -> registers -> LDS, double
// buffered, one barrier per k-tile, with TWO register sets so that the loads of k-tile kt+2 are in flight
// while kt is multiplied and kt+1 waits in the other set.  LDS rows are padded to 36 floats so that the
// 16-lane groups of ds_read_b128 hit 64 distinct banks (row r -> bank offset 36r mod 64: a permutation of the
// multiples of 4).
//
// Three things keep hipcc from serialising the memory pipeline (each was worth 5-10 % of the step):
//   * loads are never under a divergent `if`: padding rows load from a clamped valid row and are zeroed later;
//   * every USE of loaded data (zeroing, ReLU / dropout prologue) sits in the LDS-store phase, and the
//     phases load / MFMA / store are pinned with sched_barrier, so waits are counted vmcnt(N), never vmcnt(0);
//   * full tiles take a straight-line epilogue (16 residual / mask loads, math, 16 stores per 32x32 tile).
//
// It serves every convolution of the reference's encoder (src/core/modules/temporal.py):
//   first_conv (:133)  taps=1 Kc=2048        dilated_conv (:48) taps=3 Kc=128 (zero pad = dilation)
//   conv_1x1 (:50)     taps=1 Kc=128         last_conv (:145)   taps=1 Kc=128, ReLU on the input
// and, with transposed / re-packed weights, their data gradients.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "dispatch.hpp"

constexpr int NT_BK = 32;
constexpr int NT_LDS = 36;  // padded row length (floats)
// workgroup tile = BM time steps x 128 channels; BM in {128, 64, 32}: 4 waves arranged WAVES_M x (4/WAVES_M),
// each owning WM x WN MFMA tiles of 32x32.  Small BM = more, shorter workgroups for the coarse levels.
constexpr int nt_smem_bytes(int BM) { return 2 * (BM + 128) * NT_LDS * 4; }  // A and W tiles, double buffered

struct NtParams {
    const float *A;     // [B][Ta][lda] source rows
    long a_bstride;     // floats between videos in A
    int lda, Ta;
    int Trows;          // output rows per video (before pooling)
    int taps, tap_step; // 1 or 3 taps; signed row offset between taps
    int Kc;             // channels per tap (multiple of 32)
    const float *W;     // [128][ldw]: columns tap*Kc + c
    int ldw;            // row stride of W (floats); taps*Kc unless a tap subset is used
    const float *bias;  // [128] or null
    float *out;         // [B][Tout][128]; Tout = Trows (POOL 0) or Trows/2 (POOL 1,2)
    float *out_pre;     // POOL 1: un-pooled rows [B][Trows][128], kept for the max-pool backward
    const float *ypre;  // POOL 3: the forward's un-pooled rows [B][Tfine][128] (arg-max routing of the max-pool backward)
    int Tfine;          // POOL 3 / 4: rows per video of the un-pooled level; out is [B][Tfine][128]
    long tap_shift;     // floats added to A per tap index: tap k reads A + k * tap_shift (0 = one source; a two-input 1x1 convolution over
                        // the concatenation [a; b] is taps = 2, tap_step = 0, tap_shift = b - a: mucon_mstcn_fuse_fwd)
    float *out_act;     // EPI_RES, when non-null: the branch value after bias / act / dropout, BEFORE the residual is added
                        // ([B][Trows][128]; what its backward needs: x > 0 <=> the element passed both the ReLU and the dropout)
    const float *res;   // EPI_RES : [B][Trows][128] added after bias/act/dropout
    const float *mask;  // EPI_MASK: [B][Trows][128]; result *= act'(mask) (skipped when null)
    float slope;        // 0 = ReLU, 0.01 = leaky ReLU
    DropCfg drop;       // element index (b*Trows + t)*128 + c
    // TAG 1 (first_conv) on few rows: grid.z k-chunks, chunk z writes its raw partial sums to part[z] ([B][Trows][128]);
    // first_conv_combine_kernel then adds them in order, with bias and activation (ksplit <= 1: off)
    int ksplit;
    float *part[4];
};

// TAG only names the instantiation (TAG 1 = first_conv forward, so that profilers list it separately)
template <int MT, typename ACC>
__device__ __forceinline__ void nt_mfma(ACC &c, float a, float b) {
    if constexpr (MT == 32) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// MT = MFMA tile edge: 32 (v_mfma_f32_32x32x2_f32, 16 accumulator registers per tile) or 16 (v_mfma_f32_16x16x4_f32, 4
// registers; same 64 FLOP/clk/SIMD).  MT = 16 exists for BM = 16: the latency-bound coarse levels get twice the workgroups
// with half the MFMA chain each.
template <int MT>
struct NtTile {
    using Acc = typename std::conditional<MT == 32, f32x16, f32x4>::type;
    static constexpr int NREG = MT == 32 ? 16 : 4;
    // C/D layout: column = lane & (MT-1); row of register `reg`:
    //   32x32: (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)        16x16: reg + 4 * (lane >> 4)
    __device__ static __forceinline__ int row0(int lane) { return MT == 32 ? 4 * (lane >> 5) : 4 * (lane >> 4); }
    __device__ static __forceinline__ int rowr(int reg) { return MT == 32 ? (reg & 3) + 8 * (reg >> 2) : reg; }
    // operand lane (i = lane & (MT-1), kg = lane / MT) reads k = kg * (32 / groups) + s of the 32-deep k-tile at step s
    __device__ static __forceinline__ int koff(int lane) { return MT == 32 ? (lane >> 5) * 16 : (lane >> 4) * 8; }
};

// NW = waves per workgroup: 4, or 8 for the 128-row first_conv tile (two such workgroups per CU = 4 waves per SIMD, and
// the W tile is fetched once per 128 rows instead of once per 64).
template <int WM, int WAVES_M, bool PRO_ACT, bool PRO_DROP, bool EPI_ACT, bool EPI_DROP, bool EPI_RES, bool EPI_MASK, int POOL, int TAG,
          int MT = 32, int NW = 4>
__global__ __launch_bounds__(64 * NW) void nt_gemm_kernel(const NtParams p) {
    using TL = NtTile<MT>;
    constexpr int NREG = TL::NREG;
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int WN = (128 / WAVES_N) / MT;   // MT-wide column tiles per wave (128 columns in all)
    constexpr int BM = WAVES_M * WM * MT;
    constexpr int LROWS = NW * 8;                 // rows one pass of the cooperative loader covers
    constexpr int NQA = BM >= LROWS ? BM / LROWS : 1;   // A-tile float4 loads per thread (BM < LROWS: rows wrap, duplicate loads)
    constexpr int NQW = 128 / LROWS;              // W-tile float4 loads per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 2 * BM * NT_LDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * BM;
    const int ktiles_per_tap = p.Kc >> 5;
    const bool ksplit = TAG == 1 && p.ksplit > 1;
    const int nkt = ksplit ? (p.taps * ktiles_per_tap) / p.ksplit : p.taps * ktiles_per_tap;   // k-tiles of this workgroup
    const int kt_base = ksplit ? (int)blockIdx.z * nkt : 0;
    const int Ktot = p.ldw;
    const int lrow = tid >> 3;
    const int lc4 = (tid & 7) * 4;
    const float *Ab = p.A + (long)b * p.a_bstride;

    // two register sets: tile kt+2 is requested while tile kt is being multiplied and tile kt+1 (requested one
    // iteration earlier) waits in the other set -- with 32-row tiles one k-tile of MFMAs (1024 cycles) is
    // shorter than an L2 round trip, so a single tile of look-ahead leaves the loop latency-bound
    f32x4 ra[2][NQA], rb[2][NQW];
    int rt[2][NQA];   // time step of each staged A row, or -1 when the row is padding (zeroed at the LDS store)
    int rkk[2];       // channel offset of the staged k-tile (dropout replay)

    // gload only ISSUES loads (always, from a clamped valid row: a load under a divergent `if` makes hipcc wait
    // vmcnt(0) around it); every use of the loaded values -- zeroing of padding rows, ReLU / dropout prologues --
    // happens in sstore, one or two k-tiles later, so the loads stay in flight under the MFMAs.
    auto gload = [&](int kt_local, auto SET) {
        constexpr int S = decltype(SET)::value;
        const int kt = kt_local + kt_base;
        const int tap = kt / ktiles_per_tap;
        const int kk = (kt - tap * ktiles_per_tap) * 32;
        const int off = (tap - (p.taps >> 1)) * p.tap_step;
        rkk[S] = kk;
#pragma unroll
        for (int q = 0; q < NQW; ++q)
            rb[S][q] = *reinterpret_cast<const f32x4 *>(p.W + (long)(lrow + LROWS * q) * Ktot + kt * 32 + lc4);
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            const int t = t0 + ((lrow + LROWS * q) & (BM - 1));
            const int ts = t + off;
            const bool ok = (t < p.Trows) && (ts >= 0) && (ts < p.Ta);
            const int tc = ts < 0 ? 0 : (ts >= p.Ta ? p.Ta - 1 : ts);
            ra[S][q] = *reinterpret_cast<const f32x4 *>(Ab + (long)tap * p.tap_shift + (long)tc * p.lda + kk + lc4);
            rt[S][q] = ok ? t : -1;
        }
    };
    auto sstore = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        float *a = As + buf * BM * NT_LDS;
        float *w = Bs + buf * 128 * NT_LDS;
#pragma unroll
        for (int q = 0; q < NQW; ++q) *reinterpret_cast<f32x4 *>(w + (lrow + LROWS * q) * NT_LDS + lc4) = rb[S][q];
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            f32x4 v = ra[S][q];
            const int t = rt[S][q];
            if (PRO_ACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.slope);
            }
            if (PRO_DROP) {
                if (p.drop.thresh) {
                    const uint32_t idx = (uint32_t)(b * p.Trows + t) * 128u + (uint32_t)(rkk[S] + lc4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= drop_mul(p.drop, idx + e);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t >= 0 ? v[e] : 0.f;
            *reinterpret_cast<f32x4 *>(a + ((lrow + LROWS * q) & (BM - 1)) * NT_LDS + lc4) = v;   // duplicates store equal values
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    typename TL::Acc acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < NREG; ++e) acc[i][j][e] = 0.f;

    // lane (i = lane & (MT-1), group = lane / MT) feeds row i of its tile and a fixed k sub-range of the 32-deep tile
    // (k = 16h + s for 32x32x2, k = 8g + s for 16x16x4): both operands use the same k permutation, so the sum is unchanged.
    const int a_off = (wr * WM * MT + (lane & (MT - 1))) * NT_LDS + TL::koff(lane);
    const int b_off = (wc * WN * MT + (lane & (MT - 1))) * NT_LDS + TL::koff(lane);

    auto compute = [&](int cur) {
        const float *Aw = As + cur * BM * NT_LDS + a_off;
        const float *Bw = Bs + cur * 128 * NT_LDS + b_off;
        if (MT == 32) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 av[WM][2], bv[WN][2];
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    av[m][0] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * NT_LDS + ks * 8);
                    av[m][1] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * NT_LDS + ks * 8 + 4);
                }
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    bv[n][0] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS + ks * 8);
                    bv[n][1] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS + ks * 8 + 4);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            nt_mfma<MT>(acc[m][n], av[m][s >> 2][s & 3], bv[n][s >> 2][s & 3]);
                }
            }
        } else {   // 16x16x4: four k per MFMA, the lane's eight k of the tile in two 16-byte reads
            f32x4 av[WM][2], bv[WN][2];
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                av[m][0] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * NT_LDS);
                av[m][1] = *reinterpret_cast<const f32x4 *>(Aw + m * MT * NT_LDS + 4);
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                bv[n][0] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS);
                bv[n][1] = *reinterpret_cast<const f32x4 *>(Bw + n * MT * NT_LDS + 4);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        nt_mfma<MT>(acc[m][n], av[m][s >> 2][s & 3], bv[n][s >> 2][s & 3]);
            }
        }
    };

    // nkt is even (Kc is a multiple of 64).  The loop body is ONE basic block -- no guards around the loads or
    // the LDS stores (the tail re-loads the last tile and stores it where nobody reads it) -- so that hipcc
    // can count its vmcnt waits exactly: at sstore(set) only that set's loads must have landed while the
    // other set's five loads, issued one half-iteration later, stay in flight under the MFMAs.
    const int last = nkt - 1;
    gload(0, S0{});
    gload(1, S1{});
    sstore(0, S0{});
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        // sched_barrier pins the three phases: hipcc otherwise sinks the loads below the MFMAs and hoists the
        // first use of their data (the padding select) to the loop top, i.e. a vmcnt(0) stall per half-iteration
        gload(min(kt + 2, last), S0{});
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        sstore(1, S1{});
        __syncthreads();
        gload(min(kt + 3, last), S1{});
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        sstore(0, S0{});
        __syncthreads();
    }

    // Epilogue.  C/D layout (NtTile): col = lane & (MT-1); registers (2i, 2i+1) hold adjacent time steps: max_pool1d(2) pairs.
    // Two instantiations, chosen by a workgroup-uniform branch: FULL tiles (every row inside the video, the
    // common case) run straight-line code -- all residual / mask loads of a tile first, then the math,
    // then the stores -- because per-element bounds checks put every load and store under a divergent branch
    // and hipcc then separates them with vmcnt(0) waits, which serialises the whole epilogue.
    const long vbase = (long)b * p.Trows;
    if (TAG == 1) {
        if (ksplit) {   // raw partial sums of this k-chunk; bias, activation and the sum over chunks: first_conv_combine_kernel
            float *dst = p.part[blockIdx.z];
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt) {
                    const int col = (wc * WN + nt) * MT + (lane & (MT - 1));
#pragma unroll
                    for (int reg = 0; reg < NREG; ++reg) {
                        const int t = t0 + (wr * WM + mt) * MT + TL::row0(lane) + TL::rowr(reg);
                        if (t < p.Trows) dst[(vbase + t) * 128 + col] = acc[mt][nt][reg];
                    }
                }
            return;
        }
    }
    auto epilogue = [&](auto FULLT) {
        constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int col = (wc * WN + nt) * MT + (lane & (MT - 1));
                const float bias = p.bias ? p.bias[col] : 0.f;
                const int rbase = (wr * WM + mt) * MT + TL::row0(lane);
                float rres[NREG], rmask[NREG];
                const bool use_mask = EPI_MASK && (p.mask != nullptr);
#pragma unroll
                for (int reg = 0; reg < NREG; ++reg) {
                    const int t = t0 + rbase + TL::rowr(reg);
                    const long g = (vbase + (FULL ? t : min(t, p.Trows - 1))) * 128 + col;
                    if (EPI_RES) rres[reg] = p.res[g];
                    if (EPI_MASK) rmask[reg] = use_mask ? p.mask[g] : 1.f;
                }
                // POOL 3: the un-pooled pair of every output row, for the max-pool un-routing below
                float y0[POOL == 3 ? NREG : 1], y1[POOL == 3 ? NREG : 1];
                if (POOL == 3) {
#pragma unroll
                    for (int reg = 0; reg < NREG; ++reg) {
                        const int t = t0 + rbase + TL::rowr(reg);
                        const long gf = ((long)b * p.Tfine + 2 * (FULL ? t : min(t, p.Trows - 1))) * 128 + col;
                        y0[reg] = p.ypre[gf];
                        y1[reg] = p.ypre[gf + 128];
                    }
                }
                float v[NREG];
#pragma unroll
                for (int reg = 0; reg < NREG; ++reg) {
                    const int t = t0 + rbase + TL::rowr(reg);
                    const long g = (vbase + t) * 128 + col;
                    float x = acc[mt][nt][reg] + bias;
                    if (EPI_ACT) x = act_f(x, p.slope);
                    if (EPI_DROP) {
                        if (p.drop.thresh) x *= drop_mul(p.drop, (uint32_t)g);
                    }
                    if (EPI_RES) {
                        if (p.out_act && (FULL || t < p.Trows)) p.out_act[g] = x;
                        x += rres[reg];
                    }
                    if (EPI_MASK) {
                        if (use_mask) x *= act_grad(rmask[reg], p.slope);
                    }
                    v[reg] = x;
                }
#pragma unroll
                for (int reg = 0; reg < NREG; ++reg) {
                    const int t = t0 + rbase + TL::rowr(reg);
                    const long g = (vbase + t) * 128 + col;
                    if (FULL || t < p.Trows) {
                        if (POOL == 0) p.out[g] = v[reg];
                        if (POOL == 1) p.out_pre[g] = v[reg];
                    }
                }
                if (POOL == 3 || POOL == 4) {
                    // backward of max_pool1d(2) / the x2 sum pooling, fused into the producer of the pooled level's
                    // gradient: row t of this (coarse) level goes to rows 2t, 2t+1 of the fine level -- to the arg-max
                    // of the forward pair (first wins ties, as torch) or to both; an odd trailing fine row gets 0
#pragma unroll
                    for (int reg = 0; reg < NREG; ++reg) {
                        const int t = t0 + rbase + TL::rowr(reg);
                        if (FULL || t < p.Trows) {
                            const long gf = ((long)b * p.Tfine + 2 * t) * 128 + col;
                            const bool second = POOL == 3 ? (y1[reg] > y0[reg]) : false;
                            p.out[gf] = (POOL == 4 || !second) ? v[reg] : 0.f;
                            p.out[gf + 128] = (POOL == 4 || second) ? v[reg] : 0.f;
                            if (t == p.Trows - 1 && 2 * p.Trows < p.Tfine) p.out[gf + 256] = 0.f;
                        }
                    }
                } else if (POOL != 0) {
#pragma unroll
                    for (int rp = 0; rp < NREG / 2; ++rp) {
                        const int te = t0 + rbase + TL::rowr(2 * rp);   // even time step of the pair
                        if (FULL || te + 1 < p.Trows) {  // floor pooling drops an odd last step
                            const long g = ((long)b * (p.Trows >> 1) + (te >> 1)) * 128 + col;
                            p.out[g] = (POOL == 1) ? fmaxf(v[2 * rp], v[2 * rp + 1]) : (v[2 * rp] + v[2 * rp + 1]);
                        }
                    }
                }
            }
        }
    };
    if (t0 + BM <= p.Trows) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

template <int WM, int WAVES_M, bool PRO_ACT, bool PRO_DROP, bool EPI_ACT, bool EPI_DROP, bool EPI_RES, bool EPI_MASK, int POOL, int TAG,
          int MT = 32, int NW = 4>
static hipError_t launch_nt_cfg(const NtParams &p, int B, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * MT;
    auto k = nt_gemm_kernel<WM, WAVES_M, PRO_ACT, PRO_DROP, EPI_ACT, EPI_DROP, EPI_RES, EPI_MASK, POOL, TAG, MT, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, nt_smem_bytes(BM));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((p.Trows + BM - 1) / BM, B, (TAG == 1 && p.ksplit > 1) ? p.ksplit : 1);
    hipLaunchKernelGGL(k, grid, dim3(64 * NW), nt_smem_bytes(BM), s, p);
    return hipGetLastError();
}

// out = act(part[0] + part[1] + ... + bias): the k-chunks of a split first_conv, summed in chunk order (out may be part[0])
__global__ __launch_bounds__(256) void first_conv_combine_kernel(const NtParams p, long n4) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    f32x4 v = *reinterpret_cast<const f32x4 *>(p.part[0] + e * 4);
    for (int z = 1; z < p.ksplit; ++z) v += *reinterpret_cast<const f32x4 *>(p.part[z] + e * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + (e & 31) * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = act_f(v[k] + bv[k], p.slope);
    *reinterpret_cast<f32x4 *>(p.out + e * 4) = v;
}

// Tile height by problem size: keep >= ~2 workgroups per CU in flight where the level allows it; below ~one workgroup
// per CU at BM = 32 the launch is latency-bound and 16-row tiles (twice the workgroups, half the MFMA chain) are faster.
extern int g_nt_force_bm;  // 0 = automatic (tuning hook: MUCON_NT_BM)
extern long g_nt_bm16_rows;  // levels with fewer rows in the batch than this use BM = 16 (MUCON_NT_BM16_ROWS; 0 = never)
static inline int nt_pick_bm(int B, int Trows) {
    if (g_nt_force_bm) return g_nt_force_bm;
    const long rows = (long)B * Trows;
    if (rows >= 512L * 128) return 128;
    if (rows >= 512L * 64) return 64;
    if (rows < g_nt_bm16_rows) return 16;
    return 32;
}

template <bool PRO_ACT, bool PRO_DROP, bool EPI_ACT, bool EPI_DROP, bool EPI_RES, bool EPI_MASK, int POOL, int TAG = 0>
static hipError_t launch_nt(const NtParams &p, int B, hipStream_t s) {
    if ((TAG == 1 || kFirstConv8w == 2) && kFirstConv8w && POOL < 3 && (long)B * p.Trows >= 512L * 64)   // 128 rows x 8 waves
        return launch_nt_cfg<1, 4, PRO_ACT, PRO_DROP, EPI_ACT, EPI_DROP, EPI_RES, EPI_MASK, POOL, TAG, 32, 8>(p, B, s);
    switch (nt_pick_bm(B, p.Trows)) {
        case 128: return launch_nt_cfg<2, 2, PRO_ACT, PRO_DROP, EPI_ACT, EPI_DROP, EPI_RES, EPI_MASK, POOL, TAG>(p, B, s);
        case 64: return launch_nt_cfg<1, 2, PRO_ACT, PRO_DROP, EPI_ACT, EPI_DROP, EPI_RES, EPI_MASK, POOL, TAG>(p, B, s);
        case 16: return launch_nt_cfg<1, 1, PRO_ACT, PRO_DROP, EPI_ACT, EPI_DROP, EPI_RES, EPI_MASK, POOL, TAG, 16>(p, B, s);
        default: return launch_nt_cfg<1, 1, PRO_ACT, PRO_DROP, EPI_ACT, EPI_DROP, EPI_RES, EPI_MASK, POOL, TAG>(p, B, s);
    }
}
