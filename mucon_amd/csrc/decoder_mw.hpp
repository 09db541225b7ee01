// The attention decoder's forward step loop on EIGHT workgroups (r4).
//
// decoder_fwd_kernel (decoder.hpp) walks the steps on ONE workgroup: a step touches ~1 MB of weights (LSTM cell 512 KB,
// attn_combine 196 KB, the encoder memory 128 KB, ...) and costs what one CU's L1 path needs to stream them, 13.5 us.
// Here MW_G = 8 workgroups each keep an eighth of every per-step operand RESIDENT IN LDS for all steps -- nothing is
// streamed inside the loop -- and exchange three short vectors per step:
//   workgroup j owns   hidden units 16 j .. 16 j + 15:  their 64 gate rows of W_ih | W_hh            (64 KB, transposed [256][64])
//                      rows 16 j .. of attention_l2 (its slice of q), the same columns of V and of the memory projection mp
//                      memory columns 32 j .. 32 j + 31 (its slice of the context)
//                      columns {16 j .. (embedding), 128 + 32 j .. (context)} of attn_combine, all 128 rows   (24 KB)
//   per step           q slice -> PARTIAL scores over its 16 columns            -> exchange 1: all 8 partial score vectors [Tz]
//                      softmax (every workgroup, same bits) -> context slice -> PARTIAL attn_combine over its 48 inputs
//                                                                               -> exchange 2: all 8 partial `mixed` vectors [128]
//                      its 64 gate rows -> cell update of its 16 units          -> exchange 3: the 8 slices of h [16]
// An exchange is an all-gather through global memory of 8-byte {value, tag} granules (tag = step + 1; the buffers are zeroed by the
// launch in front, dec_memproj_kernel): a producer's relaxed agent-scope 8-byte atomic store is the write-through `sc1` store,
// consumers poll the granules themselves with relaxed agent-scope atomic loads -- no flags, no fences (MI355X_MICROARCH.md:
// "8-B agent atomics both sides"); two buffers alternate by step parity (a workgroup can be at most one exchange ahead of the
// slowest one).  Partial sums are added in workgroup order by every consumer, so all eight hold the same bits and results do not
// depend on timing.  Correctness needs the eight workgroups to be co-resident (8 << 256 CUs) and nothing about their placement.
// A poll that does not complete within MW_SPIN_LIMIT re-reads gives up, marks the launch failed (*nsteps_out = -1; the heads kernel then
// fills the outputs with NaN, the backward poisons dV) and lets the kernel run to its end: a broken hand-over can not hang the GPU.
// The transcript / length heads do not feed the recurrence (teacher forcing) or only through the arg-max token (greedy decoding):
// they run after the loop for all steps at once in decoder_heads_kernel; with greedy decoding the loop computes t1 / logits /
// arg-max itself, every workgroup the same bits.
#pragma once
#include "decoder.hpp"
#include "head_body.hpp"

constexpr int MW_G = 8;                       // workgroups
constexpr int MW_T = 256;                     // threads each
constexpr int MW_U = DEC_D / MW_G;            // 16: hidden units / q rows / embedding columns per workgroup
constexpr int MW_MC = DEC_MAXME / MW_G;       // 32: memory (context) columns per workgroup; the kernel takes ME == 256 only
constexpr int MW_XI = MW_U + MW_MC;           // 48: attn_combine inputs per workgroup
constexpr int MW_TZ = 192;                    // longest encoder sequence whose slices fit the LDS beside the weights
constexpr int MW_SPIN_LIMIT = 1 << 21;
// granules of the exchange buffers: [2 parities][MW_G producers][vector length]
constexpr size_t MW_X_SCORE = 2 * MW_G * MW_TZ, MW_X_MIXED = 2 * MW_G * DEC_D, MW_X_H = 2 * MW_G * MW_U;
constexpr size_t MW_X_WORDS = MW_X_SCORE + MW_X_MIXED + MW_X_H;
constexpr int MW_WTP = 65, MW_CMBP = 129;    // LDS pitches of the transposed gate / attn_combine slices: odd, so that the one-time fill (lanes along the
                                              // input index) and the per-step reads (lanes along the row) both fall on 32 different banks
static inline size_t mw_fwd_lds_bytes(int Tz) { return sizeof(float) * ((size_t)256 * MW_WTP + MW_XI * MW_CMBP + MW_U * DEC_D + (size_t)Tz * (MW_U + MW_MC)); }

__device__ __forceinline__ void mw_pub(unsigned long long *p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the MW_G partial values of one element (producer stride `ps` granules), all requested together, re-read until every tag matches;
// summed in producer order
__device__ __forceinline__ float mw_gather_sum(const unsigned long long *p, const int ps, const unsigned tag, int *err) {
    unsigned long long v[MW_G];
    int spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int g = 0; g < MW_G; ++g) v[g] = __hip_atomic_load(p + (size_t)g * ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int g = 0; g < MW_G; ++g) ok = ok && (unsigned)(v[g] >> 32) == tag;
        if (ok) break;
        if (++spins > MW_SPIN_LIMIT) {
            *err = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    float s = __uint_as_float((unsigned)v[0]);
#pragma unroll
    for (int g = 1; g < MW_G; ++g) s += __uint_as_float((unsigned)v[g]);
    return s;
}
__device__ __forceinline__ float mw_get(const unsigned long long *p, const unsigned tag, int *err) {
    unsigned long long v;
    int spins = 0;
    while ((unsigned)((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag) {
        if (++spins > MW_SPIN_LIMIT) {
            *err = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return __uint_as_float((unsigned)v);
}
// out[r] = act(b[r] + W[r][0..127] . x) for r < rows, W from global memory, 4 waves: wave w takes rows w, w + 4, ... in groups of 8,
// every group's loads (16 per lane) requested before the first is used; all groups of a wave requested up front (rows <= 128: 64 loads)
template <int ACT>
__device__ __forceinline__ void mw_matvec_rows128(const int tid, const float *__restrict__ W, const float *__restrict__ b, const int rows,
                                                  const float *x, float *out) {
    const int lane = tid & 63, wave = tid >> 6;
    const float x0 = x[lane], x1 = x[lane + 64];
    const int ngroups = (rows + 31) / 32;                 // groups of 8 rows per wave (4 waves x 8 rows = 32 rows a round)
    float w[4][8][2];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
        if (gq < ngroups) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = min(gq * 32 + wave * 8 + i, rows - 1);
                w[gq][i][0] = W[(long)r * DEC_D + lane];
                w[gq][i][1] = W[(long)r * DEC_D + 64 + lane];
            }
        }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
        if (gq < ngroups) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = w[gq][i][0] * x0 + w[gq][i][1] * x1;
            const float sum = wave_sum_rows<8>(lane, acc);
            const int r = gq * 32 + wave * 8 + ((lane >> 3) & 7);
            if ((lane & 7) == 0 && r < rows) {
                const float v = sum + b[r];
                out[r] = ACT ? fmaxf(v, 0.f) : v;
            }
        }
}

// grid MW_G, MW_T threads, dynamic LDS mw_fwd_lds_bytes(Tz).  Requires ME == 256, Tz <= MW_TZ, NC <= 128 (host: mw_fwd_ok).
__global__ __launch_bounds__(MW_T) void decoder_fwd_mw_kernel(DecDims dm, DecParams p, DecSaved sv, const float *memory, const float *hn,
                                                             const float *cn, const long *tf_input, const float *dropmask,
                                                             float *logits_out, unsigned long long *xbuf, int *nsteps_out) {
    extern __shared__ __attribute__((aligned(16))) float mw_dyn[];
    float *s_wt = mw_dyn;                               // [256 inputs: mixed | h][64 gate rows: gate * 16 + unit]
    float *s_cmb = s_wt + 256 * MW_WTP;                 // [48 inputs: 16 embedding | 32 context][128 rows]
    float *s_l2 = s_cmb + MW_XI * MW_CMBP;              // [16 rows][128]
    float *s_mp = s_l2 + MW_U * DEC_D;                  // [Tz][16]
    float *s_mem = s_mp + (size_t)dm.Tz * MW_U;         // [Tz][32]
    __shared__ float s_h[DEC_D], s_c[MW_U], s_q[MW_U], s_x[MW_XI], s_mixed[DEC_D], s_score[MW_TZ], s_attn[MW_TZ];
    __shared__ float s_part[256], s_gates[64], s_hc[2 * DEC_MAXME], s_t1[DEC_D], s_logits[DEC_MAXNC];
    __shared__ int s_tok, s_stop, s_err;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = blockIdx.x;
    const int Tz = dm.Tz, ME = dm.ME, NC = dm.NC, CW = DEC_D + ME;
    unsigned long long *x_score = xbuf, *x_mixed = xbuf + MW_X_SCORE, *x_h = x_mixed + MW_X_MIXED;

    // ---- residents
    for (int e = tid; e < 64 * 256; e += MW_T) {        // gate row r = gate * 16 + unit  <-  global row gate * 128 + 16 j + unit
        const int r = e >> 8, c = e & 255, grow = (r >> 4) * DEC_D + MW_U * j + (r & 15);
        s_wt[c * MW_WTP + r] = c < DEC_D ? p.w_ih[(long)grow * DEC_D + c] : p.w_hh[(long)grow * DEC_D + c - DEC_D];
    }
    for (int e = tid; e < DEC_D * MW_XI; e += MW_T) {   // attn_combine [128][128 + 256]: columns 16 j .. and 128 + 32 j ..
        const int row = e / MW_XI, c = e - row * MW_XI;
        s_cmb[c * MW_CMBP + row] = p.cmb_w[(long)row * CW + (c < MW_U ? MW_U * j + c : DEC_D + MW_MC * j + c - MW_U)];
    }
    for (int e = tid; e < MW_U * DEC_D; e += MW_T) s_l2[e] = p.l2_w[(long)(MW_U * j) * DEC_D + e];
    for (int e = tid; e < Tz * MW_U; e += MW_T) s_mp[e] = sv.mp[(long)(e >> 4) * DEC_D + MW_U * j + (e & 15)];
    for (int e = tid; e < Tz * MW_MC; e += MW_T) s_mem[e] = memory[(long)(e >> 5) * ME + MW_MC * j + (e & 31)];
    // initial state (models.py:612-617), every workgroup all 128 units of h, its own 16 of c
    for (int e = tid; e < ME; e += MW_T) {
        s_hc[e] = hn[e];
        s_hc[DEC_MAXME + e] = cn[e];
    }
    if (tid == 0) {
        s_tok = (int)tf_input[0];
        s_stop = 0;
        s_err = 0;
    }
    __syncthreads();
    {   // rows of hidden_out (all 128) and cn_out (this workgroup's 16): lanes along the 256 columns, a wave takes rows w, w + 4, ...;
        // two round trips (rows 0..127, then the 16 rows of cn_out), every load of a round requested before the first is used
        const float xh[4] = {s_hc[lane], s_hc[64 + lane], s_hc[128 + lane], s_hc[192 + lane]};
        float w[4][8][4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float *wrow = p.ho_w + (long)(gq * 32 + wave * 8 + i) * ME;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[gq][i][q] = wrow[q * 64 + lane];
            }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = ((w[gq][i][0] * xh[0] + w[gq][i][1] * xh[1]) + w[gq][i][2] * xh[2]) + w[gq][i][3] * xh[3];
            const float sum = wave_sum_rows<8>(lane, acc);
            const int r = gq * 32 + wave * 8 + ((lane >> 3) & 7);
            if ((lane & 7) == 0) s_h[r] = sum + p.ho_b[r];
        }
        if (wave < 2) {
            const float xc[4] = {s_hc[DEC_MAXME + lane], s_hc[DEC_MAXME + 64 + lane], s_hc[DEC_MAXME + 128 + lane], s_hc[DEC_MAXME + 192 + lane]};
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float *wrow = p.co_w + (long)(MW_U * j + wave * 8 + i) * ME;
                acc[i] = ((wrow[lane] * xc[0] + wrow[64 + lane] * xc[1]) + wrow[128 + lane] * xc[2]) + wrow[192 + lane] * xc[3];
            }
            const float sum = wave_sum_rows<8>(lane, acc);
            const int r = wave * 8 + ((lane >> 3) & 7);
            if ((lane & 7) == 0) s_c[r] = sum + p.co_b[MW_U * j + r];
        }
    }
    __syncthreads();
    if (j == 0 && tid < DEC_D) sv.h[tid] = s_h[tid];
    if (tid < MW_U) sv.c[MW_U * j + tid] = s_c[tid];
    const float l2b = p.l2_b[MW_U * j + (tid >> 4)];
    float vsl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vsl[i] = p.v[MW_U * j + (tid & 1) * 8 + i];
    const float cmb_b = tid < DEC_D ? p.cmb_b[tid] : 0.f;
    const int grow_g = tid < 64 ? (tid >> 4) * DEC_D + MW_U * j + (tid & 15) : 0;
    const float gate_b = tid < 64 ? p.b_ih[grow_g] + p.b_hh[grow_g] : 0.f;
    const bool heads_in_loop = !(dm.teacher_forcing && !dm.stop_on_eos);

    int s = 0, err = 0;
    for (; s < dm.S; ++s) {
        const unsigned tag = (unsigned)s + 1u;
        const int par = s & 1;
        int tok = dm.teacher_forcing ? (int)tf_input[s] : s_tok;
        tok = tok < 0 ? 0 : tok >= dm.n_emb ? dm.n_emb - 1 : tok;
        // -- q slice = attention_l2(h)[16 j ..]: 16 lanes (a DPP row) share a row, 8 columns each
        {
            const int row = tid >> 4, cg = tid & 15;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) a += s_l2[row * DEC_D + cg * 8 + i] * s_h[cg * 8 + i];
            a += dpp_f<DPP_XOR1>(a);
            a += dpp_f<DPP_XOR2>(a);
            a += dpp_f<DPP_HALF_MIRROR>(a);
            a += dpp_f<DPP_MIRROR>(a);
            if (cg == 0) {
                s_q[row] = a + l2b;
                sv.q[s * DEC_D + MW_U * j + row] = a + l2b;
            }
        }
        __syncthreads();
        // -- partial score[t] over this workgroup's 16 columns: a lane pair per encoder state
        for (int t0 = 0; t0 < Tz; t0 += 128) {
            const int t = t0 + (tid >> 1), hf = tid & 1;
            const int tc = min(t, Tz - 1);
            const f32x4 m0 = *reinterpret_cast<const f32x4 *>(s_mp + tc * MW_U + hf * 8), m1 = *reinterpret_cast<const f32x4 *>(s_mp + tc * MW_U + hf * 8 + 4);
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) a += vsl[i] * tanh_f((i < 4 ? m0[i] : m1[i - 4]) + s_q[hf * 8 + i]);
            a += dpp_f<DPP_XOR1>(a);
            if (hf == 0 && t < Tz) mw_pub(x_score + ((size_t)par * MW_G + j) * MW_TZ + t, a, tag);
        }
        // embedded = dropout(relu(embedding(input)))[16 j ..]
        if (tid < MW_U) {
            float e = fmaxf(p.emb[(long)tok * DEC_D + MW_U * j + tid], 0.f);
            if (dropmask) e *= dropmask[s * DEC_D + MW_U * j + tid];
            s_x[tid] = e;
            sv.cat[(long)s * CW + MW_U * j + tid] = e;
        }
        if (j == 0 && tid == 0) sv.toks[s] = tok;
        // -- exchange 1: the scores
        if (tid < Tz) s_score[tid] = mw_gather_sum(x_score + (size_t)par * MW_G * MW_TZ + tid, MW_TZ, tag, &err);
        __syncthreads();
        // softmax: every wave reduces all scores itself (same bits in every wave of every workgroup)
        {
            float mx = -INFINITY;
            for (int t = lane; t < Tz; t += 64) mx = fmaxf(mx, s_score[t]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int t = lane; t < Tz; t += 64) sum += expf(s_score[t] - mx);
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            for (int t = tid; t < Tz; t += MW_T) {
                const float a = expf(s_score[t] - mx) * inv;
                s_attn[t] = a;
                if (j == 0) sv.attn[(long)s * Tz + t] = a;
            }
        }
        __syncthreads();
        // -- context slice: 8 groups of states x 32 columns, the groups meet in LDS in order
        {
            const int col = tid & 31, tq = tid >> 5;
            float a = 0.f;
            for (int t = tq; t < Tz; t += 8) a += s_attn[t] * s_mem[t * MW_MC + col];
            s_part[tq * 32 + col] = a;
        }
        __syncthreads();
        if (tid < MW_MC) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += s_part[g * 32 + tid];
            s_x[MW_U + tid] = v;
            sv.cat[(long)s * CW + DEC_D + MW_MC * j + tid] = v;
        }
        __syncthreads();
        // -- partial attn_combine over this workgroup's 48 inputs: two halves of 24 per row
        {
            const int row = tid & 127, hf = tid >> 7;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 24; ++c) a += s_cmb[(hf * 24 + c) * MW_CMBP + row] * s_x[hf * 24 + c];
            s_part[hf * 128 + row] = a;
        }
        __syncthreads();
        if (tid < DEC_D) mw_pub(x_mixed + ((size_t)par * MW_G + j) * DEC_D + tid, s_part[tid] + s_part[128 + tid], tag);
        // -- exchange 2: mixed = relu(sum of the partials + bias)
        if (tid < DEC_D) {
            const float m = fmaxf(mw_gather_sum(x_mixed + (size_t)par * MW_G * DEC_D + tid, DEC_D, tag, &err) + cmb_b, 0.f);
            s_mixed[tid] = m;
            if (j == 0) sv.mixed[s * DEC_D + tid] = m;
        }
        __syncthreads();
        // -- this workgroup's 64 gate rows: wave w multiplies inputs 64 w .. 64 w + 63 of [mixed | h], lane = gate row
        {
            const float xv = wave < 2 ? s_mixed[64 * wave + lane] : s_h[64 * (wave - 2) + lane];
            const float *wt = s_wt + (64 * wave) * MW_WTP + lane;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 64; ++c) a += wt[c * MW_WTP] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c));
            s_part[wave * 64 + lane] = a;
        }
        __syncthreads();
        if (tid < 64) s_gates[tid] = (((s_part[tid] + s_part[64 + tid]) + s_part[128 + tid]) + s_part[192 + tid]) + gate_b;
        __syncthreads();
        if (tid < MW_U) {
            const float gi = sigmoid_f(s_gates[tid]), gf = sigmoid_f(s_gates[16 + tid]);
            const float gg = tanh_f(s_gates[32 + tid]), go = sigmoid_f(s_gates[48 + tid]);
            const float c = gf * s_c[tid] + gi * gg;
            const float h = go * tanh_f(c);
            s_c[tid] = c;
            const int u = MW_U * j + tid;
            float *gs = sv.gates + (long)s * 4 * DEC_D;
            gs[u] = gi;
            gs[DEC_D + u] = gf;
            gs[2 * DEC_D + u] = gg;
            gs[3 * DEC_D + u] = go;
            sv.c[(s + 1) * DEC_D + u] = c;
            sv.h[(s + 1) * DEC_D + u] = h;
            mw_pub(x_h + ((size_t)par * MW_G + j) * MW_U + tid, h, tag);
        }
        // -- exchange 3: h
        if (tid < DEC_D) s_h[tid] = mw_get(x_h + ((size_t)par * MW_G + (tid >> 4)) * MW_U + (tid & 15), tag, &err);
        __syncthreads();
        if (!heads_in_loop) continue;
        // greedy decoding: the arg-max word feeds the next step -- t1, logits and arg-max in every workgroup (the same bits)
        mw_matvec_rows128<1>(tid, p.t1_w, p.t1_b, DEC_D, s_h, s_t1);
        __syncthreads();
        mw_matvec_rows128<0>(tid, p.t2_w, p.t2_b, NC, s_t1, s_logits);
        __syncthreads();
        if (j == 0 && tid < DEC_D) sv.t1[s * DEC_D + tid] = s_t1[tid];
        if (j == 0 && tid < NC) logits_out[(long)s * NC + tid] = s_logits[tid];
        if (wave == 0) {   // arg-max, lowest index on ties
            const float x0 = lane < NC ? s_logits[lane] : -INFINITY, x1 = lane + 64 < NC ? s_logits[lane + 64] : -INFINITY;
            const float mx = wave_max(fmaxf(x0, x1));
            int cand = x0 == mx ? lane : x1 == mx ? lane + 64 : 1 << 20;
#pragma unroll
            for (int o = 32; o; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
            if (lane == 0) {
                s_tok = cand;
                if (dm.stop_on_eos && cand == dm.eos) s_stop = 1;
            }
        }
        __syncthreads();
        if (s_stop) {
            ++s;
            break;
        }
    }
    if (err) s_err = 1;
    __syncthreads();
    if (tid == 0) {
        if (s_err) atomicExch(nsteps_out, -1);               // (any workgroup that gave up marks the launch)
        else if (j == 0) atomicCAS(nsteps_out, 0, s);         // (0 = the value dec_memproj_kernel left: not marked failed)
    }
}

// The heads of every step that ran, ONE WORKGROUP PER STEP (grid = max steps, DEC_THREADS threads; decoder_fwd_kernel's per-step head code):
// transcript MLP -> log-softmax, length MLP.  The steps are independent of each other here, so they run side by side instead of as a walk
// over the steps inside one workgroup (13.5 -> ~6.5 us at 7 steps).  have_logits: t1 and the raw logits were computed inside the loop
// (greedy decoding); *nsteps = the steps that ran (-1: a hand-over gave up -- every output is poisoned with NaN).
__global__ __launch_bounds__(DEC_THREADS) void decoder_heads_kernel(DecDims dm, DecParams p, DecSaved sv, float *logp_out, float *len_out,
                                                                    const int *nsteps, const int have_logits) {
    __shared__ float s_h[DEC_D], s_t1[DEC_D], s_logits[DEC_MAXNC], s_lencat[DEC_D + DEC_MAXNC], s_l1[DEC_NL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, st = blockIdx.x;
    const int S = min(*nsteps, dm.S), NC = dm.NC, LW = DEC_D + NC;
    if (S < 0) {   // (a caller that never reads the step count -- teacher-forced training -- cannot mistake this for a result)
        for (int e = tid; e < NC; e += DEC_THREADS) logp_out[(long)st * NC + e] = NAN;
        if (tid == 0) len_out[st] = NAN;
        return;
    }
    if (st >= S) return;
    if (!have_logits) {
        if (tid < DEC_D) s_h[tid] = sv.h[(st + 1) * DEC_D + tid];
        __syncthreads();
        matvec_rows<1, 8, DEC_D, false>(tid, p.t1_w, p.t1_b, DEC_D, DEC_D, s_h, s_t1);
        __syncthreads();
        if (tid < DEC_D) sv.t1[st * DEC_D + tid] = s_t1[tid];
        matvec_rows<0, 4, DEC_D, false>(tid, p.t2_w, p.t2_b, NC, DEC_D, s_t1, s_logits);
    } else if (tid < NC) {
        s_logits[tid] = logp_out[(long)st * NC + tid];
    }
    __syncthreads();
    if (tid < LW) {
        const float v = tid < DEC_D ? sv.mixed[st * DEC_D + tid] : fmaxf(s_logits[tid - DEC_D], 0.f);
        s_lencat[tid] = v;
        sv.lencat[(long)st * LW + tid] = v;
    }
    __syncthreads();
    matvec_rows<1, 4, 0, false>(tid, p.n1_w, p.n1_b, DEC_NL, LW, s_lencat, s_l1);
    __syncthreads();
    if (wave == 0) {
        sv.l1[st * DEC_NL + lane] = s_l1[lane];
        const float a = wave_sum(p.n2_w[lane] * s_l1[lane]);
        if (lane == 0) len_out[st] = a + p.n2_b[0];
    } else if (wave == 1) {
        const float x0 = lane < NC ? s_logits[lane] : -INFINITY, x1 = lane + 64 < NC ? s_logits[lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(x0, x1));
        const float se = wave_sum((lane < NC ? expf(x0 - mx) : 0.f) + (lane + 64 < NC ? expf(x1 - mx) : 0.f));
        const float lse = mx + logf(se);
        if (lane < NC) logp_out[(long)st * NC + lane] = x0 - lse;
        if (lane + 64 < NC) logp_out[(long)st * NC + lane + 64] = x1 - lse;
    }
}

// ============================================================================================================ backward
// The backward step loop on the same eight workgroups, same slices, three exchanges per step as well:
//   cell backward of the workgroup's 16 units -> its 64 gate deltas
//   PARTIAL d mixed = W_ih[its 64 rows]^T dgates and PARTIAL dh_a = W_hh[its 64 rows]^T dgates      -> exchange A (256 values)
//   d mixed (+ the length head's share, through the ReLU) -> d cat over its 48 attn_combine columns: d embedding, d context slice
//   PARTIAL d attn[t] = memory[t][its 32 columns] . d context slice                                     -> exchange B (Tz values)
//   softmax backward (every workgroup, same bits) -> d score -> d q over its 16 rows (through the tanh; dV alongside)
//   PARTIAL dh_b = attention_l2[its 16 rows]^T d q                                                      -> exchange C (128 values)
//   dh(s - 1) = dh_a + dh_b (+ the transcript head's share, added by the next step's cell backward)
// The heads' backward runs in front as a launch of its own, a workgroup per step (decoder_heads_bwd_kernel: what decoder_bwd_kernel does in
// front of its loop); it also clears the exchange granules.  Every per-step delta the weight-gradient kernels read (dl.gates, dl.mixed,
// dl.ctx, dl.score, dl.q) is written exactly as decoder_bwd_kernel writes it.
constexpr size_t MWB_X_A = 2 * MW_G * 256, MWB_X_B = 2 * MW_G * MW_TZ, MWB_X_C = 2 * MW_G * DEC_D, MWB_X_D = MW_G * MW_U;
constexpr size_t MWB_X_WORDS = MWB_X_A + MWB_X_B + MWB_X_C + MWB_X_D;
static_assert(MWB_X_WORDS <= 2 * MW_X_WORDS, "the backward's granules fit the buffer dec_layout reserves (twice the forward's)");
static inline size_t mw_bwd_lds_bytes(int Tz) { return sizeof(float) * ((size_t)2 * 64 * DEC_D + MW_XI * MW_CMBP + MW_U * DEC_D + (size_t)Tz * MW_U + (size_t)MW_MC * (Tz | 1)); }

// decoder_bwd_kernel's prologue as a kernel, ONE WORKGROUP PER STEP (grid = steps, DEC_THREADS threads): back-propagates the heads of its step
// and parks what reaches the recurrence in dl.q[s] (d dec_out from the transcript MLP) and dl.mixed[s] (d mixed from the length MLP); the
// workgroups together zero d_emb and the exchange granules of the step kernel behind them.
__global__ __launch_bounds__(DEC_THREADS) void decoder_heads_bwd_kernel(DecDims dm, DecParams p, DecSaved sv, DecDeltas dl, const float *logp,
                                                                        const float *d_logp, const float *d_len, float *d_emb,
                                                                        unsigned long long *zero, int zero_words) {
    __shared__ __attribute__((aligned(16))) float s_scr[DEC_THREADS];
    __shared__ float s_dlog[DEC_MAXNC], s_dl1[DEC_NL], s_out[DEC_D + DEC_MAXNC], s_dt1[DEC_D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, st = blockIdx.x;
    const int NC = dm.NC, LW = DEC_D + NC;
    for (int e = st * DEC_THREADS + tid; e < zero_words; e += gridDim.x * DEC_THREADS) zero[e] = 0ull;
    for (long e = (long)st * DEC_THREADS + tid; e < (long)dm.n_emb * DEC_D; e += (long)gridDim.x * DEC_THREADS) d_emb[e] = 0.f;
    if (wave == 0) {            // log-softmax backward
        const float g0 = (d_logp && lane < NC) ? d_logp[(long)st * NC + lane] : 0.f;
        const float g1 = (d_logp && lane + 64 < NC) ? d_logp[(long)st * NC + lane + 64] : 0.f;
        const float tot = wave_sum(g0 + g1);
        if (lane < NC) s_dlog[lane] = g0 - expf(logp[(long)st * NC + lane]) * tot;
        if (lane + 64 < NC) s_dlog[lane + 64] = g1 - expf(logp[(long)st * NC + lane + 64]) * tot;
    } else if (wave == 1) {     // length MLP output layer backward
        const float dlen = d_len ? d_len[st] : 0.f;
        const float v = sv.l1[st * DEC_NL + lane] > 0.f ? dlen * p.n2_w[lane] : 0.f;
        s_dl1[lane] = v;
        dl.l1[st * DEC_NL + lane] = v;
        if (lane == 0) dl.len[st] = dlen;
    }
    __syncthreads();
    matvec_cols<false>(tid, p.n1_w, DEC_NL, LW, s_dl1, s_out, s_scr);
    __syncthreads();
    if (tid < LW) {
        const float v = sv.lencat[(long)st * LW + tid] > 0.f ? s_out[tid] : 0.f;
        if (tid < DEC_D) dl.mixed[st * DEC_D + tid] = v;
        else s_dlog[tid - DEC_D] += v;
    }
    __syncthreads();
    if (tid < NC) dl.logits[(long)st * NC + tid] = s_dlog[tid];
    matvec_cols<false>(tid, p.t2_w, NC, DEC_D, s_dlog, s_out, s_scr);       // transcript MLP backward -> d dec_out
    __syncthreads();
    if (tid < DEC_D) {
        const float v = sv.t1[st * DEC_D + tid] > 0.f ? s_out[tid] : 0.f;
        s_dt1[tid] = v;
        dl.t1[st * DEC_D + tid] = v;
    }
    __syncthreads();
    matvec_cols<false>(tid, p.t1_w, DEC_D, DEC_D, s_dt1, s_out, s_scr);
    __syncthreads();
    if (tid < DEC_D) dl.q[st * DEC_D + tid] = s_out[tid];
}

// grid MW_G, MW_T threads, dynamic LDS mw_bwd_lds_bytes(Tz); behind decoder_heads_bwd_kernel.
__global__ __launch_bounds__(MW_T) void decoder_bwd_mw_kernel(DecDims dm, DecParams p, DecSaved sv, DecDeltas dl, const float *memory,
                                                             const float *dropmask, float *d_emb, float *d_v, float *d_hn, float *d_cn,
                                                             unsigned long long *xbuf, const HeadBwdArgs ha, const int hgx, const int hblocks) {
    if ((int)blockIdx.x >= MW_G) {   // (r6) workgroups behind the eight: a deferred y-head backward (head_body.hpp), one of its 256-thread blocks each
        const int hb = (int)blockIdx.x - MW_G;
        if (hb < hblocks) head_bwd_z_body(ha, hb % hgx, hb / hgx, hgx, (int)threadIdx.x);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float mwb_dyn[];
    float *s_wih = mwb_dyn;                             // [64 gate rows: gate * 16 + unit][128]
    float *s_whh = s_wih + 64 * DEC_D;                  // [64][128]
    float *s_cmb = s_whh + 64 * DEC_D;                  // [48 columns: 16 embedding | 32 context][128 rows], pitch MW_CMBP
    float *s_l2 = s_cmb + MW_XI * MW_CMBP;              // [16 rows][128]
    float *s_mp = s_l2 + MW_U * DEC_D;                  // [Tz][16]
    float *s_mem = s_mp + (size_t)dm.Tz * MW_U;         // [32 columns][Tz | 1]: transposed, the d attn phase runs a lane per state
    __shared__ float s_dh[DEC_D], s_dha[DEC_D], s_dc[MW_U], s_dg[64], s_dmixed[DEC_D], s_dctx[MW_MC], s_ds[MW_TZ], s_dq[MW_U], s_part[256];
    __shared__ float s_attn[MW_TZ];
    __shared__ int s_errb;
    __shared__ float s_dcfull[DEC_D];
    const int tid = threadIdx.x, lane = tid & 63, j = blockIdx.x;
    const int Tz = dm.Tz, ME = dm.ME, CW = DEC_D + ME, S = dm.S;
    unsigned long long *x_a = xbuf, *x_b = xbuf + MWB_X_A, *x_c = x_b + MWB_X_B, *x_d = x_c + MWB_X_C;

    for (int e = tid; e < 64 * DEC_D; e += MW_T) {
        const int r = e >> 7, c = e & 127, grow = (r >> 4) * DEC_D + MW_U * j + (r & 15);
        s_wih[e] = p.w_ih[(long)grow * DEC_D + c];
        s_whh[e] = p.w_hh[(long)grow * DEC_D + c];
    }
    for (int e = tid; e < DEC_D * MW_XI; e += MW_T) {
        const int row = e / MW_XI, c = e - row * MW_XI;
        s_cmb[c * MW_CMBP + row] = p.cmb_w[(long)row * CW + (c < MW_U ? MW_U * j + c : DEC_D + MW_MC * j + c - MW_U)];
    }
    for (int e = tid; e < MW_U * DEC_D; e += MW_T) s_l2[e] = p.l2_w[(long)(MW_U * j) * DEC_D + e];
    for (int e = tid; e < Tz * MW_U; e += MW_T) s_mp[e] = sv.mp[(long)(e >> 4) * DEC_D + MW_U * j + (e & 15)];
    const int TzP = Tz | 1;
    for (int e = tid; e < Tz * MW_MC; e += MW_T) s_mem[(e & 31) * TzP + (e >> 5)] = memory[(long)(e >> 5) * ME + MW_MC * j + (e & 31)];
    if (tid < DEC_D) s_dh[tid] = 0.f;
    if (tid < MW_U) s_dc[tid] = 0.f;
    if (tid == 0) s_errb = 0;
    const float vq = p.v[MW_U * j + (tid & 15)];       // d q phase: thread (d = tid & 15, state group tid >> 4)
    float dv_acc = 0.f;
    int err = 0;
    __syncthreads();

    // the cell backward's operands travel one step ahead (threads < 16: their unit's gates, cell states, the heads' share of dh)
    float nx[7];
    auto fetch_step = [&](int st) {
        const int u = MW_U * j + tid;
        const float *gs = sv.gates + (long)st * 4 * DEC_D;
        nx[0] = gs[u];
        nx[1] = gs[DEC_D + u];
        nx[2] = gs[2 * DEC_D + u];
        nx[3] = gs[3 * DEC_D + u];
        nx[4] = sv.c[(st + 1) * DEC_D + u];
        nx[5] = sv.c[st * DEC_D + u];
        nx[6] = dl.q[st * DEC_D + u];
    };
    if (tid < MW_U) fetch_step(S - 1);
    for (int s = S - 1; s >= 0; --s) {
        const unsigned tag = (unsigned)(S - s);
        const int par = (S - 1 - s) & 1;
        // operands of the later phases, requested now: the length head's share of d mixed and the ReLU mask, the attention weights, q
        const float park_mixed = tid < DEC_D ? dl.mixed[s * DEC_D + tid] : 0.f, sv_mixed = tid < DEC_D ? sv.mixed[s * DEC_D + tid] : 0.f;
        const float attn_t = tid < Tz ? sv.attn[(long)s * Tz + tid] : 0.f;
        if (tid < Tz) s_attn[tid] = attn_t;       // (read behind exchange B: several barriers away)
        const float q_d = sv.q[s * DEC_D + MW_U * j + (tid & 15)];
        const int tok = sv.toks[s];
        // -- LSTM cell backward of this workgroup's units
        if (tid < MW_U) {
            const float gi = nx[0], gf = nx[1], gg = nx[2], go = nx[3], ct = nx[4], cp = nx[5];
            const float dh = s_dh[MW_U * j + tid] + nx[6], th = tanh_f(ct);
            const float dct = s_dc[tid] + dh * go * (1.f - th * th);
            const float dpi = dct * gg * gi * (1.f - gi), dpf = dct * cp * gf * (1.f - gf);
            const float dpg = dct * gi * (1.f - gg * gg), dpo = dh * th * go * (1.f - go);
            s_dc[tid] = dct * gf;
            s_dg[tid] = dpi;
            s_dg[16 + tid] = dpf;
            s_dg[32 + tid] = dpg;
            s_dg[48 + tid] = dpo;
            float *o = dl.gates + (long)s * 4 * DEC_D + MW_U * j + tid;
            o[0] = dpi;
            o[DEC_D] = dpf;
            o[2 * DEC_D] = dpg;
            o[3 * DEC_D] = dpo;
        }
        __syncthreads();
        if (tid < MW_U && s > 0) fetch_step(s - 1);
        // -- partial d mixed (threads 0..127) and partial dh_a (threads 128..255) over this workgroup's 64 gate rows
        {
            const float dgv = s_dg[lane];
            const float *w = (tid < DEC_D ? s_wih : s_whh) + (tid & 127);
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 64; ++r) a += w[r * DEC_D] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dgv), r));
            mw_pub(x_a + ((size_t)par * MW_G + j) * 256 + tid, a, tag);
        }
        // -- exchange A
        {
            const float v = mw_gather_sum(x_a + (size_t)par * MW_G * 256 + tid, 256, tag, &err);
            if (tid < DEC_D) {
                const float dm_ = sv_mixed > 0.f ? v + park_mixed : 0.f;
                s_dmixed[tid] = dm_;
                if (j == 0) dl.mixed[s * DEC_D + tid] = dm_;
            } else {
                s_dha[tid - DEC_D] = v;
            }
        }
        __syncthreads();
        // -- d cat over this workgroup's 48 columns of attn_combine: four lanes per column (32 rows each)
        if (tid < 4 * MW_XI) {
            const int col = tid >> 2, qd = tid & 3;
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const int rr = qd * 32 + ((r + 8 * qd) & 31);     // (the four lanes of a column start eight banks apart)
                a += s_cmb[col * MW_CMBP + rr] * s_dmixed[rr];
            }
            a += dpp_f<DPP_XOR1>(a);
            a += dpp_f<DPP_XOR2>(a);
            if (qd == 0) {
                if (col < MW_U) {   // embedding row gradient (this workgroup is the only writer of its 16 columns)
                    float g = p.emb[(long)tok * DEC_D + MW_U * j + col] > 0.f ? a : 0.f;
                    if (dropmask) g *= dropmask[s * DEC_D + MW_U * j + col];
                    d_emb[(long)tok * DEC_D + MW_U * j + col] += g;
                } else {
                    s_dctx[col - MW_U] = a;
                    dl.ctx[(long)s * ME + MW_MC * j + col - MW_U] = a;
                }
            }
        }
        __syncthreads();
        // -- partial d attn[t] over this workgroup's 32 memory columns: a lane per state
        {
            const float dcv = s_dctx[lane & 31];
            const int tc = min(tid, Tz - 1);
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < MW_MC; ++c) a += s_mem[c * TzP + tc] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dcv), c));
            if (tid < Tz) mw_pub(x_b + ((size_t)par * MW_G + j) * MW_TZ + tid, a, tag);
        }
        // -- exchange B, softmax backward (every wave reduces the dot product itself: same bits everywhere)
        if (tid < Tz) s_ds[tid] = mw_gather_sum(x_b + (size_t)par * MW_G * MW_TZ + tid, MW_TZ, tag, &err);
        __syncthreads();
        {
            float part = 0.f;
            for (int t = lane; t < Tz; t += 64) part += s_attn[t] * s_ds[t];
            const float dot = wave_sum(part);
            __syncthreads();
            if (tid < Tz) {
                const float v = attn_t * (s_ds[tid] - dot);
                s_ds[tid] = v;
                if (j == 0) dl.score[(long)s * Tz + tid] = v;
            }
        }
        __syncthreads();
        // -- d q over this workgroup's 16 rows through the tanh (dV alongside): thread (d, state group of 16)
        {
            const int d = tid & 15, tg = tid >> 4;
            float dq = 0.f;
            for (int t = tg; t < Tz; t += 16) {
                const float u = tanh_f(s_mp[t * MW_U + d] + q_d);
                const float dsv = s_ds[t];
                dv_acc += dsv * u;
                dq += dsv * vq * (1.f - u * u);
            }
            s_part[tg * 16 + d] = dq;
        }
        __syncthreads();
        if (tid < MW_U) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) v += s_part[g * 16 + tid];
            s_dq[tid] = v;
            dl.q[s * DEC_D + MW_U * j + tid] = v;
        }
        __syncthreads();
        // -- partial dh_b = attention_l2[its 16 rows]^T d q                                              -> exchange C
        if (tid < DEC_D) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < MW_U; ++r) a += s_l2[r * DEC_D + tid] * s_dq[r];
            mw_pub(x_c + ((size_t)par * MW_G + j) * DEC_D + tid, a, tag);
            s_dh[tid] = s_dha[tid] + mw_gather_sum(x_c + (size_t)par * MW_G * DEC_D + tid, DEC_D, tag, &err);
        }
        __syncthreads();
    }
    // ---- initial state: d h0 / d c0, d h_n / d c_n through hidden_out / cn_out (this workgroup: 32 of the 256 columns), dV
    if (tid < MW_U) mw_pub(x_d + (size_t)j * MW_U + tid, s_dc[tid], (unsigned)S + 1u);
    if (tid < DEC_D) s_dcfull[tid] = mw_get(x_d + (size_t)(tid >> 4) * MW_U + (tid & 15), (unsigned)S + 1u, &err);
    __syncthreads();
    if (j == 0 && tid < DEC_D) {
        dl.h0[tid] = s_dh[tid];
        dl.c0[tid] = s_dcfull[tid];
    }
    {
        const int col = MW_MC * j + (tid & 31), ig = tid >> 5;     // 8 groups of 16 rows
        float ah = 0.f, ac = 0.f;
#pragma unroll 8
        for (int i = ig * 16; i < ig * 16 + 16; ++i) {
            ah += p.ho_w[(long)i * ME + col] * s_dh[i];
            ac += p.co_w[(long)i * ME + col] * s_dcfull[i];
        }
        s_part[tid] = ah;
        __syncthreads();
        if (tid < MW_MC) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += s_part[g * 32 + tid];
            d_hn[MW_MC * j + tid] = v;
        }
        __syncthreads();
        s_part[tid] = ac;
        __syncthreads();
        if (tid < MW_MC) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += s_part[g * 32 + tid];
            d_cn[MW_MC * j + tid] = v;
        }
        __syncthreads();
    }
    s_part[tid] = dv_acc;
    if (err) s_errb = 1;
    __syncthreads();
    if (tid < MW_U) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += s_part[g * 16 + tid];
        d_v[MW_U * j + tid] = s_errb ? NAN : v;      // a hand-over that gave up poisons this workgroup's slice of dV: the step is loudly invalid
    }
}
