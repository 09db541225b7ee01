// libmucon_hip.so -- host side of the C ABI (include/mucon_hip.h): sequences the gfx950 kernels of
// the encoder / y-head forward and backward on the caller's stream.  No allocation, no
// synchronisation (except the explicit bench helper); all state lives in the caller's workspace.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "../../include/mucon_hip.h"
#include "../../include/mucon_hip_test.h"
#include "common.hpp"
#include "dispatch.hpp"
#include "gemm_nt.hpp"
#include "gemm_fused.hpp"
#include "gemm_tn.hpp"
#include "gemm_split.hpp"
#include "gemm_tn_split.hpp"
#include "gemm_fused_split.hpp"
#include "gemm_coarse_split.hpp"
#include "small_kernels.hpp"
#include "probe.hpp"

int g_mfma16 = 1;             // MUCON_MFMA16: bit 0 = first_conv forward / layer 0's data gradient (gemm_split.hpp), bit 1 = the weight gradients
                              // (gemm_tn_split.hpp) on v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (profiles/r05_mfma_shape.txt: the A/B)
int g_ts_runs = 1;            // the batched split weight-gradient launch of encoder_bwd as static runs on persistent workgroups (MUCON_TS_RUNS; gemm_tn_split.hpp); 0: one workgroup per item
int g_ts_cost[4] = {69, 74, 95, 109};   // ... its cost units (1/32 us): tile of a staggered / lock-step / two-image column, a run's fixed cost per video (fixed since r6: was the knob TS_COSTS=a,b,c,d)
int g_ts_max_wg = 0;          // ... on at most this many workgroups (fixed since r6: was the knob TS_MAX_WG; 0 = one per CU)
// One-shot options of the NEXT mucon_encoder_bwd call (mucon_encoder_bwd_overlap; data-parallel training): an event to record on the pass's stream
// once every gradient EXCEPT first_conv's is final, and a cap on the workgroups of the weight-gradient launches (CUs left free for RCCL's kernel)
hipEvent_t g_bwd_event = nullptr;
int g_bwd_max_wg = 0;
// One-shot option of the NEXT mucon_head_bwd call (mucon_head_bwd_defer), and what such a call leaves behind: the y-head's slab reduction, to be taken
// by extra workgroups of the next mucon_encoder_bwd's first launch on the same stream (small_kernels.hpp: head_reduce_tail) -- or by mucon_head_bwd_flush
struct HeadPending {
    bool armed = false, pending = false;
    const float *w_slabs = nullptr, *b_slabs = nullptr;
    float *d_w = nullptr, *d_b = nullptr;
    int nblk = 0, C = 0, H = 0;
    hipStream_t stream = nullptr;
};
HeadPending g_head_pending;
HeadFwdPending g_head_fwd_pending;        // (head_body.hpp) mucon_head_fwd_defer: the forward kernel, left to the next mucon_lstm_fwd (shead.hip)
HeadKernelPending g_head_kernel_pending;   // (head_body.hpp) mucon_head_bwd_defer bit 1: the z-level backward kernel itself, left to the next mucon_decoder_bwd (shead.hip)
int g_ts_group_rows = 1 << 30;    // ... a residual layer's groups: single videos when a video has at least this many rows, else the whole batch (fixed since r6: was the knob TS_GROUP_ROWS)
int g_ts_stagger = 1024;      // fixed since r6: was the knob TS_STAGGER=n: the single-image weight-gradient jobs (first_conv's) with time chunks >= n steps on the staggered block schedule
                              // (gemm_tn_split.hpp: ts_body_st); 0: every job on round 4's lock-step schedule
int g_cs_rb4_wgs = 512;       // forward launches of the coarse kernel take 64 rows per workgroup where 16-row workgroups would number more than this (fixed since r6: was the knob COARSE_RB4_WGS; 0: never)
int g_cs_rb = 0;              // row blocks (16 rows each) per workgroup of the coarse-level split kernel: 0 = by level size (MUCON_COARSE_RB)
int g_nt_force_bm = 0;
long g_nt_bm16_rows = 8193;   // see nt_pick_bm (MUCON_NT_BM16_ROWS; 0 = never use 16-row tiles)
int g_no_fuse = 0;  // the fused two-stage layer kernels (gemm_fused.hpp); MUCON_FUSE=0 runs two launches per layer


namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(MUCON_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }  // in floats: 256-byte alignment

// time-chunk length for a weight-gradient launch: ~256 workgroups, but never fewer than 4 m-tiles (128
// time steps) per workgroup -- every workgroup writes a 64 KB partial tile that has to be summed later
int g_first_conv_ksplit = 1;           // first_conv of small launches in four k-chunks (fixed since r6: was the knob FIRST_CONV_KSPLIT)
long g_first_conv_ksplit_rows = 6144;  // ... up to this many frames per launch (fixed since r6: was the knob FIRST_CONV_KSPLIT_ROWS; measured: 65 -> 47 us at 5,000, even at 8,000)
int g_first_conv_split = 1;            // first_conv forward on the bf16 MFMA, operands split exactly in three (fixed since r6: was the knob FIRST_CONV_SPLIT)
long g_first_conv_split_rows = 8192;   // ... for launches of at least this many frames (MUCON_FIRST_CONV_SPLIT_ROWS)
int g_cs = 1;                 // residual layers of the other (coarse, latency-bound) levels on the k-split split-bf16 kernel (gemm_coarse_split.hpp; MUCON_COARSE_SPLIT=0: f32 MFMA)
int g_fs = 1;                 // residual layers of chip-filling levels on the split-bf16 two-stage kernel (gemm_fused_split.hpp; fixed since r6: was the knob FUSED_SPLIT=0: f32 MFMA)
long g_fs_rows = 32768;       // ... from this many rows in the batch = one 128-row workgroup per CU (MUCON_FUSED_SPLIT_ROWS).  r3: 16,384 -> 32,768: at
                              // B = 8 x T = 4096 the T/2 level (16,384 rows: 256 workgroups of 64 rows on the 4-wave variant) is 3.7 us per step
                              // faster on the coarse kernel's 32-row workgroups (512 of them, two per CU)
int g_tn_split = 1;           // weight gradients on the bf16 MFMA with exactly split operands (gemm_tn_split.hpp; MUCON_TN_SPLIT=0: f32 MFMA)
int g_ts_mc_cap = 2048;       // ... whose workgroups (256 columns each) take time chunks of at most this many steps (fixed since r6: was the knob TS_MC_CAP; measured 2048: 209 us, 1024: 218, 512: 229 at B=8 x T=4096)
int g_ts_layer_mc_cap = 512;  // ... and the residual layers' jobs of the batched split launch chunks of at most this many (fixed since r6: was the knob TS_LAYER_MC_CAP): their workgroups
                              // stage two gradient images and replay the dropout mask, 3.1 us per 32-step tile against first_conv's 2.4 -- at 2,048 steps the
                              // 32 of them at the finest level ran 200 us while first_conv's 128 finished after 160 and the launch (215 us) waited for them (r5: 2048 -> 512,
                              // 0.2086 -> 0.2000 ms per launch; 1024: 0.2148, 768: 0.199, 384: 0.200, 256: 0.1975 with 8 us more slab reduction)
inline int pick_mc(int B, int Trows, int kchunks, bool batched = false, bool split = false, bool layer = false) {
    const int target = batched ? kTnBatchTarget : kTnTarget;
    const int cap = split ? ((layer && batched) ? std::min(g_ts_mc_cap, g_ts_layer_mc_cap) : g_ts_mc_cap) : kTnMcCap;
    long want = ((long)B * Trows * kchunks + target - 1) / target;
    long mc = ((want + 31) / 32) * 32;
    if (mc < 128) mc = 128;
    if (mc > cap) mc = cap;
    return (int)mc;
}

// slab / bias-partial floats ONE weight-gradient job needs under every schedule it can take: one workgroup per (column, time chunk) item
// (the shortest chunk any of those launches would pick), or static runs (one 128 x 256 partial per workgroup and column: gemm_tn_split.hpp)
inline void one_job_arena(int B, int Trows, int Ktot, bool layer, size_t &sf, size_t &bf) {
    int mc = pick_mc(B, Trows, Ktot / 128);
    for (int v = 1; v < 4; ++v) mc = std::min(mc, pick_mc(B, Trows, Ktot / 128, (v & 1) != 0, (v & 2) != 0, layer));
    const size_t nmc = (size_t)B * ((Trows + mc - 1) / mc);
    // static runs: one partial per (share, unit) = at most shares + units + 1 of them; units = groups x columns, groups <= B videos or, for a job
    // taken as one long video, panels of about a share each (<= shares + 1); + the launch's copy of its line behind the bias partials
    const size_t ncols = ((size_t)Ktot / 128 + 1) / 2, tiles = ncols * B * (((size_t)Trows + 31) / 32);
    const size_t shares = std::min<size_t>(kTsMaxWorkgroups, (tiles + 3) / 4);
    const size_t nsl = shares + ncols * std::max<size_t>((size_t)B, shares / ncols + 2) + 2;
    sf = std::max(align64(nmc * 128 * Ktot), align64(nsl * 128 * 256));
    bf = std::max(align64(nmc * 256), align64(nsl * 256));
}

struct Plan {
    int L, B, T, D, Tz;
    int Tl[MUCON_MAX_LAYERS + 1];
    size_t W1f, W1b, W2t, Wlt;
    size_t W0s;  // first_conv.weight as pre-split bf16 fragment images (gemm_split.hpp)
    size_t Wd0s; // layer 0's data-gradient operand (W1b) likewise: 3*128*384 bf16
    size_t Wfs;  // per layer: the four split images of gemm_fused_split.hpp (W1f, W1b, W2, W2t)
    size_t x[MUCON_MAX_LAYERS + 1], h[MUCON_MAX_LAYERS], ypre[MUCON_MAX_LAYERS];
    size_t z, gnstat, gnpart;
    size_t sync;   // 64 words the backward's first kernel zeroes: [0] the ticket counter of the persistent weight-gradient launch
    size_t gz, g[MUCON_MAX_LAYERS + 1], dpre[MUCON_MAX_LAYERS], dyd[MUCON_MAX_LAYERS];
    size_t slabs, slab_floats, bslabs, bslab_floats;
    size_t total;  // floats
};

int validate(const mucon_encoder_cfg *c) {
    if (!c) return fail(MUCON_E_ARG, "cfg is null");
    if (c->H != MUCON_H) return fail(MUCON_E_ARG, "hidden size %d unsupported (kernels are built for %d)", c->H, MUCON_H);
    if (c->D <= 0 || c->D % 128 != 0) return fail(MUCON_E_ARG, "input dim %d must be a positive multiple of 128", c->D);
    if (c->n_layers < 1 || c->n_layers > MUCON_MAX_LAYERS) return fail(MUCON_E_ARG, "n_layers %d not in [1,%d]", c->n_layers, MUCON_MAX_LAYERS);
    if (c->B < 1 || c->T < 1) return fail(MUCON_E_ARG, "B=%d T=%d", c->B, c->T);
    if ((long)c->B * c->T * 128 >= (1L << 32)) return fail(MUCON_E_ARG, "B*T*128 must be < 2^32 (dropout counter)");
    if (c->pool_type != 0 && c->pool_type != 1) return fail(MUCON_E_ARG, "pool_type %d", c->pool_type);
    if (c->last_gn) {
        const int G = c->gn_groups;
        if (G < 1 || 128 % G != 0 || (128 / G) % 4 != 0) return fail(MUCON_E_ARG, "gn_groups %d unsupported (need 128/G a multiple of 4)", G);
        const int cpg = 128 / G;
        if ((cpg & (cpg - 1)) != 0) return fail(MUCON_E_ARG, "gn_groups %d unsupported", G);
    }
    for (int l = 0; l < c->n_layers; ++l)
        if (c->dilation[l] < 1) return fail(MUCON_E_ARG, "dilation[%d]=%d", l, c->dilation[l]);
    if (c->p_drop_layer < 0.f || c->p_drop_layer >= 1.f || c->p_drop_last < 0.f || c->p_drop_last >= 1.f)
        return fail(MUCON_E_ARG, "dropout probabilities must be in [0,1)");
    int T = c->T;
    for (int l = 0; l < c->n_layers; ++l)
        if (c->pool_after[l]) T /= 2;
    if (T < 1) return fail(MUCON_E_ARG, "T=%d is too short for the pooling schedule", c->T);
    return MUCON_OK;
}

void make_plan(const mucon_encoder_cfg *c, Plan &p) {
    p.L = c->n_layers;
    p.B = c->B;
    p.T = c->T;
    p.D = c->D;
    p.Tl[0] = c->T;
    for (int l = 0; l < p.L; ++l) p.Tl[l + 1] = c->pool_after[l] ? p.Tl[l] / 2 : p.Tl[l];
    p.Tz = p.Tl[p.L];
    size_t o = 0;
    auto take = [&](size_t n) {
        size_t r = o;
        o += align64(n);
        return r;
    };
    p.W1f = take((size_t)p.L * 49152);
    p.W1b = take((size_t)p.L * 49152);
    p.W2t = take((size_t)p.L * 16384);
    p.Wlt = take(16384);
    p.W0s = take((size_t)3 * 128 * p.D / 2);
    p.Wd0s = take((size_t)3 * 128 * 384 / 2);
    p.Wfs = take((size_t)(p.L + 1) * FS_LAYER_ELEMS / 2);   // slot L: last_conv
    for (int l = 0; l <= p.L; ++l) p.x[l] = take((size_t)p.B * p.Tl[l] * 128);
    for (int l = 0; l < p.L; ++l) {
        p.h[l] = take((size_t)p.B * p.Tl[l] * 128);
        p.ypre[l] = (c->pool_after[l] && c->pool_type == 0) ? take((size_t)p.B * p.Tl[l] * 128) : 0;
    }
    p.z = take((size_t)p.B * p.Tz * 128);
    p.gnstat = take((size_t)p.B * 128 * 2);
    p.gnpart = take((size_t)p.B * 256);
    p.sync = take(64);
    // backward buffers, one per layer (no ping-pong): the weight-gradient launches run on a second
    // stream and may still be reading a layer's tensors while the data-gradient chain moves on
    p.gz = take((size_t)p.B * p.Tz * 128);
    for (int l = 0; l <= p.L; ++l) p.g[l] = take((size_t)p.B * p.Tl[l] * 128);     // gradient w.r.t. x[l]
    for (int l = 0; l < p.L; ++l) {
        p.dpre[l] = take((size_t)p.B * p.Tl[l] * 128);                             // at the dilated conv's pre-activation
        p.dyd[l] = c->pool_after[l] ? take((size_t)p.B * p.Tl[l] * 128) : 0;       // un-pooled gradient of a pooled layer
    }
    // slab arena: every weight-gradient launch of a backward pass keeps its own slabs until the
    // single batched reduction at the end
    size_t sf = 0, bf = 0;
    auto consider = [&](int Trows, int Ktot, bool layer = false) {   // the shortest time chunk any schedule (batched or not, split or f32) would take
        size_t s1, b1;
        one_job_arena(p.B, Trows, Ktot, layer, s1, b1);
        sf += s1;
        bf += b1;
    };
    consider(p.T, p.D);
    for (int l = 0; l < p.L; ++l) consider(p.Tl[l], 512, true);
    consider(p.Tz, 128);
    p.slab_floats = sf;
    p.bslab_floats = bf;
    p.slabs = take(sf);
    p.bslabs = take(bf);
    p.total = o;
}

void prof_mark(int slot, bool stop, hipStream_t s);

// Layer l's level fills the chip and its dilation reaches inside the sequence: its launches take the split-bf16 two-stage kernels
inline bool fs_level(const mucon_encoder_cfg *cfg, const Plan &pl, int l) {
    return g_fs && !g_no_fuse && l >= 0 && l < pl.L && (long)pl.B * pl.Tl[l] >= g_fs_rows && cfg->dilation[l] < pl.Tl[l];
}
inline bool cs_on() { return g_cs && !g_no_fuse; }
int g_pack_f32 = 0;     // fixed since r6: was the knob PACK_ALL=1: every weight re-layout is written whether or not a launch of the pass reads it (A/B of the r4 trimming; tests)
int g_tail_chain = 1;   // MUCON_TAIL_CHAIN=0: the row-local launches at the coarsest level one by one (cs_kernel) instead of chained (ct_kernel)

// Which weight re-layouts a forward pass writes.  The backward pass reads them, decides from the same predicates, and refuses to run when
// its answer differs from what the forward recorded for the workspace (a knob set between the two calls would otherwise make it multiply
// by workspace nobody wrote).
struct PackDecision {
    bool need_f32;       // the f32 operand layouts W1f / W1b / W2t / Wlt
    bool split_first;    // first_conv.weight's pre-split image (W0s)
    bool split_dgrad0;   // layer 0's data-gradient image (Wd0s)
    int img16;           // ... both in the order of the 16x16x32 kernel
    bool operator==(const PackDecision &o) const {
        return need_f32 == o.need_f32 && split_first == o.split_first && split_dgrad0 == o.split_dgrad0 && img16 == o.img16;
    }
};
PackDecision pack_decision(const mucon_encoder_cfg *cfg, const mucon_encoder_params *prm, const Plan &pl) {
    PackDecision d;
    const int L = pl.L;
    // The f32 operand layouts feed the f32-MFMA kernels only.  When every launch of a forward AND of its backward takes the split-bf16
    // images (the default configuration: k-split kernels on, pooled boundaries fused, biases present, no pooling behind the last layer)
    // they are not written at all: 5 MB of stores and 56 transposing workgroups less per pass (r4).
    d.need_f32 = !cs_on() || !kPoolFuse || cfg->pool_after[L - 1] || !prm->last_b || (long)pl.B * pl.T > kFuseMaxRows || g_pack_f32;
    for (int l = 0; l < L; ++l) d.need_f32 = d.need_f32 || !prm->dil_b[l] || !prm->pw_b[l];
    // first_conv on the bf16 MFMA with exactly split operands: worth it once the launch fills the chip
    d.split_first = g_first_conv_split && (long)pl.B * pl.T >= g_first_conv_split_rows;
    // layer 0's dilated-conv data gradient (the last launch of the backward chain) takes the same kernel
    // (mucon_encoder_bwd reaches that launch only when neither two-stage split kernel takes layer 0)
    d.split_dgrad0 = kNtSplitDgrad0 && (long)pl.B * pl.T >= g_first_conv_split_rows && cfg->dilation[0] < pl.T &&
                     ((!fs_level(cfg, pl, 0) && !cs_on()) || g_pack_f32);
    d.img16 = g_mfma16 & 1;
    return d;
}
// workspace -> what its last forward wrote, and for which problem (a workspace that was freed and re-allocated at the same address for another
// shape does not pass as "a forward was run on this workspace").  Bounded: the entries of workspaces that no longer exist are dropped wholesale
// when the table grows past kFwdRecordMax (the next forward of a live workspace simply records again).
struct FwdRecord {
    PackDecision d;
    uint64_t shape;   // B, T, D, L folded together
};
constexpr size_t kFwdRecordMax = 256;
std::mutex g_fwd_mu;
std::unordered_map<const void *, FwdRecord> g_fwd_record;
inline uint64_t fwd_shape_key(const Plan &pl) {
    return ((uint64_t)(uint32_t)pl.B << 48) ^ ((uint64_t)(uint32_t)pl.T << 20) ^ ((uint64_t)(uint32_t)pl.D << 6) ^ (uint64_t)(uint32_t)pl.L;
}
inline const uint16_t *fs_img(const float *ws, const Plan &pl, int l, int mat) {   // mat: 0 W1f, 1 W1b, 2 W2, 3 W2t, 4 / 5 centre taps of W1f / W1b in accumulator order
    const long off = mat == 0 ? 0 : (mat == 1 ? FS_IMG_K384 : 2L * FS_IMG_K384 + (long)(mat - 2) * FS_IMG_K128);
    return reinterpret_cast<const uint16_t *>(ws + pl.Wfs) + (long)l * FS_LAYER_ELEMS + off;
}

// The last two residual layers and last_conv are row-local (dilation past the sequence, no pooling in between) and run on the
// k-split kernels: they can be chained in one launch (ct_kernel).  The backward chain (last_conv's and layer L-1's data gradients)
// needs the same of layer L-1 only, but both directions switch together.
static bool tail_chain_ok(const mucon_encoder_cfg *cfg, const Plan &pl, const mucon_encoder_params *prm, int B) {
    const int L = pl.L;
    if (!g_tail_chain || !cs_on() || L < 2) return false;
    for (int l = L - 2; l < L; ++l)
        if (cfg->pool_after[l] || cfg->dilation[l] < pl.Tl[l] || fs_level(cfg, pl, l) || !prm->dil_b[l] || !prm->pw_b[l]) return false;
    return prm->last_b && (long)B * pl.Tl[L - 1] <= kFuseMaxRows;
}

// Collects the slab reductions of one backward pass; run() sums them all in one launch.
struct Reducer {
    ReduceBatch rb;
    hipStream_t stream;
    int max_slabs = 0;     // deepest job queued
    explicit Reducer(hipStream_t s) : stream(s) {
        rb.njobs = 0;
        rb.nblocks = 0;
    }
    bool add(const float *slabs, int nslabs, long slab_stride, int ld, int coff, int nrows, int ncols, float *out,
             int mode) {
        if (rb.njobs >= REDUCE_MAX_JOBS && run() != hipSuccess) return false;  // table full: flush what is queued
        ReduceJob &j = rb.j[rb.njobs++];
        j.slabs = slabs;
        j.out = out;
        j.slab_stride = slab_stride;
        j.nslabs = nslabs;
        j.ld = ld;
        j.coff = coff;
        j.ncols = ncols;
        j.n_elems = nrows * ncols;
        j.mode = (int8_t)mode;
        j.sched = -1;
        j.isbias = 0;
        j.bcol = 0;
        j.vec = ((ncols | coff | ld | j.n_elems) % 4 == 0 && slab_stride % 4 == 0) ? 1 : 0;
        j.block0 = rb.nblocks;
        max_slabs = rb.njobs == 1 ? nslabs : std::max(max_slabs, nslabs);
        if (rb.njobs == 1)
            for (int k = 0; k < REDUCE_MAX_JOBS; ++k) rb.first_block[k] = INT_MAX;
        rb.first_block[rb.njobs - 1] = j.block0;
        rb.nblocks += (j.n_elems + 255) / 256;
        return true;
    }
    bool line_used = false;   // rb.ct holds the column table of a static-runs launch whose jobs are queued
    // a job over the partial tiles of the static-runs launch: columns coff .. coff + ncols of job `sched` of the line (bias: the 256-float partials of its column bcol)
    bool add_sched(int sched, const TsJobLine &L, bool isbias, int bcol, int coff, int nrows, int ncols, float *out, int mode) {
        if (!add(nullptr, 1, 0, 256, coff, nrows, ncols, out, mode)) return false;
        ReduceJob &j = rb.j[rb.njobs - 1];
        j.sched = (int8_t)sched;
        j.isbias = isbias ? 1 : 0;
        j.bcol = (int8_t)bcol;
        line_used = true;
        for (int c = isbias ? bcol : coff / 256; c <= (isbias ? bcol : (coff + ncols - 1) / 256); ++c) max_slabs = std::max(max_slabs, (int)rb.ct.n[sched][c]);
        (void)L;
        return j.vec == 1;
    }
    hipError_t run() {
        line_used = false;
        if (rb.njobs == 0) return hipSuccess;
        // few, deep jobs (the y-head's 256 slabs of 6 K elements: 25 workgroups) want many slab lanes: the chain of dependent
        // loads per thread is what their time is; the big pass (3,900 workgroups, <= 32 slabs) is bandwidth-bound and wants 4
        // ... and a pass whose jobs have one or two slabs (a batch-1 step: every time chunk is the whole video) is a copy:
        // one wave per workgroup, no exchange
        const int lanes = rb.nblocks <= 128 ? 16 : (max_slabs <= 2 ? 1 : kReduceLanes);
        if (lanes == 4 && kReduceChunks > 1 && rb.nblocks >= 1024) {
            // the big pass of a training step: kReduceChunks chunks per workgroup (gemm_tn.hpp: reduce_batch_kernel<G, U>) -- the jobs' first workgroups re-counted
            int nb = 0;
            for (int k = 0; k < rb.njobs; ++k) {
                rb.j[k].block0 = nb;
                rb.first_block[k] = nb;
                nb += (rb.j[k].n_elems + 256 * kReduceChunks - 1) / (256 * kReduceChunks);
            }
            hipLaunchKernelGGL((reduce_batch_kernel<4, kReduceChunks>), dim3(nb), dim3(256), 0, stream, rb);
            rb.njobs = 0;
            rb.nblocks = 0;
            return hipGetLastError();
        }
        switch (lanes) {
            case 1: hipLaunchKernelGGL(reduce_batch_kernel<1>, dim3(rb.nblocks), dim3(64), 0, stream, rb); break;
            case 2: hipLaunchKernelGGL(reduce_batch_kernel<2>, dim3(rb.nblocks), dim3(128), 0, stream, rb); break;
            case 4: hipLaunchKernelGGL(reduce_batch_kernel<4>, dim3(rb.nblocks), dim3(256), 0, stream, rb); break;
            case 8: hipLaunchKernelGGL(reduce_batch_kernel<8>, dim3(rb.nblocks), dim3(512), 0, stream, rb); break;
            case 16: hipLaunchKernelGGL(reduce_batch_kernel<16>, dim3(rb.nblocks), dim3(1024), 0, stream, rb); break;
            default: hipLaunchKernelGGL(reduce_batch_kernel<4>, dim3(rb.nblocks), dim3(256), 0, stream, rb); break;
        }
        rb.njobs = 0;
        rb.nblocks = 0;
        return hipGetLastError();
    }
};

// One weight-gradient launch through the TN core.  Slabs are carved from the arena at *arena / *barena
// (floats, advanced); the reductions are queued on `red`.
//   set 0: Y0 x X0 (taps / column chunks) -> out_w0 (mode0: 0 = [128][nk0*128], 1 = conv3 layout), out_b0
//   set 1 (dual): (Y1 * dropout) x X1    -> out_w1 [128][128], out_b1
struct WgradArgs {
    const float *Y0, *X0;
    long x_bstride;
    int ldx, Tx, taps, tap_step, nk0;
    bool x0_act;
    const float *Y1, *X1;  // null: single set
    float *out_w0, *out_b0, *out_w1, *out_b1;
    int mode0;
    DropCfg drop;
};
// The weight-gradient jobs of a backward pass, queued for ONE launch behind the data-gradient chain (flush_wgrads).
struct WgradQueue {
    TnBatch tb;
    WgradArgs out[TN_MAX_BATCH];
    WgradQueue() { tb.njobs = 0; }
};
static TnParams wgrad_params(const Plan &pl, int Trows, const WgradArgs &a, float slope, bool batched) {
    const bool dual = a.Y1 != nullptr;
    TnParams t;
    memset(&t, 0, sizeof(t));
    t.Trows = Trows;
    t.Y0 = a.Y0;
    t.X0 = a.X0;
    t.x_bstride = a.x_bstride;
    t.ldx = a.ldx;
    t.Tx = a.Tx;
    t.taps = a.taps;
    t.tap_step = a.tap_step;
    t.nk0 = a.nk0;
    t.Y1 = a.Y1;
    t.X1 = a.X1;
    t.Ktot = 128 * (a.nk0 + (dual ? 1 : 0));
    t.MC = pick_mc(pl.B, Trows, t.Ktot / 128, batched, g_tn_split != 0, dual);
    t.chunks_per_video = (Trows + t.MC - 1) / t.MC;
    t.slope = slope;
    t.drop = a.drop;
    return t;
}
// slabs [time chunk][128][Ktot] for a job of the per-item launches, carved from the arena at arena / barena (floats, advanced); reductions queued on `red`
static int wgrad_slabs(const Plan &pl, float *ws, size_t &arena, size_t &barena, TnParams &t, const WgradArgs &a, Reducer &red) {
    const bool dual = a.Y1 != nullptr;
    const int nmc = pl.B * t.chunks_per_video;
    const size_t need = align64((size_t)nmc * 128 * t.Ktot), bneed = align64((size_t)nmc * 256);
    if (arena + need > pl.slab_floats || barena + bneed > pl.bslab_floats)
        return fail(MUCON_E_WORKSPACE, "internal: slab arena too small");
    t.slabs = ws + pl.slabs + arena;
    t.bias_slabs = ws + pl.bslabs + barena;
    arena += need;
    barena += bneed;
    const long ss = (long)128 * t.Ktot;
    bool ok = red.add(t.slabs, nmc, ss, t.Ktot, 0, 128, a.nk0 * 128, a.out_w0, a.mode0);
    if (a.out_b0) ok = ok && red.add(t.bias_slabs, nmc, 256, 256, 0, 1, 128, a.out_b0, 0);
    if (dual) {
        ok = ok && red.add(t.slabs, nmc, ss, t.Ktot, a.nk0 * 128, 128, 128, a.out_w1, 0);
        if (a.out_b1) ok = ok && red.add(t.bias_slabs, nmc, 256, 256, 128, 1, 128, a.out_b1, 0);
    }
    if (!ok) return fail(MUCON_E_ARG, "internal: too many reduction jobs");
    return MUCON_OK;
}
static int device_cus() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        ncu = prop.multiProcessorCount;
    }
    return ncu;
}
// Launches what is queued.  Default: the static-runs launch (gemm_tn_split.hpp) -- the schedule and the reduction's column table come from the
// same numbers; MUCON_TS_RUNS=0 (or a batch the schedule does not take): one workgroup per (column, time chunk) item; MUCON_TN_SPLIT=0: the f32 kernels.
static int flush_wgrads(const Plan &pl, float *ws, size_t &arena, size_t &barena, WgradQueue &q, Reducer &red, hipStream_t s) {
    TnBatch &tb = q.tb;
    if (tb.njobs == 0) return MUCON_OK;
    if (g_tn_split && g_ts_runs) {
        const int ncu = device_cus();
        if (ncu <= 0) return fail(MUCON_E_ARG, "internal: device properties unavailable");
        TnBatch lb;
        ts_layout(tb, lb);
        TsSchedule sc;
        int maxg = std::min(kTsMaxWorkgroups, g_ts_max_wg > 0 ? std::min(g_ts_max_wg, ncu) : ncu);
        if (g_bwd_max_wg > 0) maxg = std::max(1, std::min(maxg, g_bwd_max_wg));
        if (ts_make_schedule(lb, pl.B, maxg, sc)) {
            const size_t need = (size_t)sc.nslabs * 128 * 256, bneed = (size_t)sc.nslabs * 256;
            if (arena + need > pl.slab_floats || barena + bneed > pl.bslab_floats) return fail(MUCON_E_WORKSPACE, "internal: slab arena too small");
            if (red.rb.njobs > 0 && red.line_used && red.run() != hipSuccess) return fail(MUCON_E_ARG, "internal: slab reduction failed");
            sc.ln.ct.slabs = ws + pl.slabs + arena;
            sc.ln.ct.bias = ws + pl.bslabs + barena;
            arena += align64(need);
            barena += align64(bneed);
            red.rb.ct = sc.ln.ct;
            bool ok = true;
            for (int i = 0; i < lb.njobs; ++i) {
                const WgradArgs &a = q.out[tb.njobs - 1 - i];   // (ts_layout reverses the queue)
                const TsJobLine &L = sc.ln.j[i];
                const int nk0 = lb.j[i].p.nk0;
                ok = ok && red.add_sched(i, L, false, 0, 0, 128, nk0 * 128, a.out_w0, a.mode0);
                if (a.out_b0) ok = ok && red.add_sched(i, L, true, 0, 0, 1, 128, a.out_b0, 0);
                if (lb.j[i].dual) {
                    ok = ok && red.add_sched(i, L, false, 0, nk0 * 128, 128, 128, a.out_w1, 0);
                    if (a.out_b1) ok = ok && red.add_sched(i, L, true, nk0 / 2, 128, 1, 128, a.out_b1, 0);
                }
            }
            if (!ok) return fail(MUCON_E_ARG, "internal: too many reduction jobs");
            HIPCHK(launch_ts_runs(lb, sc, s));
            tb.njobs = 0;
            return MUCON_OK;
        }
    }
    for (int i = 0; i < tb.njobs; ++i) {
        int rc = wgrad_slabs(pl, ws, arena, barena, tb.j[i].p, q.out[i], red);
        if (rc != MUCON_OK) return rc;
    }
    HIPCHK(g_tn_split ? launch_ts_batch(tb, s) : launch_tn_batch(tb, s));
    return MUCON_OK;
}

// One weight-gradient job through the TN core: queued on `q` (encoder_bwd), or launched at once.
//   set 0: Y0 x X0 (taps / column chunks) -> out_w0 (mode0: 0 = [128][nk0*128], 1 = conv3 layout), out_b0
//   set 1 (dual): (Y1 * dropout) x X1    -> out_w1 [128][128], out_b1
int wgrad(const Plan &pl, float *ws, size_t &arena, size_t &barena, int Trows, const WgradArgs &a, float slope,
          Reducer &red, hipStream_t s, int prof_slot = -1, WgradQueue *q = nullptr) {
    const bool dual = a.Y1 != nullptr;
    const bool split = g_tn_split != 0;
    if (split && dual && a.nk0 % 2 == 0)   // the split kernel pairs the conv_1x1 chunk with the last tap (gemm_tn_split.hpp)
        return fail(MUCON_E_ARG, "internal: dual weight-gradient job with an even chunk count");
    TnParams t = wgrad_params(pl, Trows, a, slope, q != nullptr);
    auto enqueue = [&](WgradQueue &wq) {
        wq.out[wq.tb.njobs] = a;
        TnJob &jb = wq.tb.j[wq.tb.njobs++];
        jb.p = t;
        jb.nkc = t.Ktot / 128;
        jb.block0 = pl.B * t.chunks_per_video;   // time-chunk count while queued; the launch turns it into the block offset
        jb.x0_act = a.x0_act ? 1 : 0;
        jb.dual = dual ? 1 : 0;
    };
    if (q) {
        if (q->tb.njobs >= TN_MAX_BATCH) {
            int rc = flush_wgrads(pl, ws, arena, barena, *q, red, s);
            if (rc != MUCON_OK) return rc;
        }
        enqueue(*q);
        return MUCON_OK;
    }
    if (split) {   // a queue of one
        WgradQueue one;
        enqueue(one);
        if (prof_slot >= 0) prof_mark(prof_slot, false, s);
        int rc = flush_wgrads(pl, ws, arena, barena, one, red, s);
        if (prof_slot >= 0) prof_mark(prof_slot, true, s);
        return rc;
    }
    int rc = wgrad_slabs(pl, ws, arena, barena, t, a, red);
    if (rc != MUCON_OK) return rc;
    if (prof_slot >= 0) prof_mark(prof_slot, false, s);
    {
        // layer launches run one workgroup per CU: two waves per SIMD (KS = 2); first_conv's has two workgroups per CU
        const bool ks2 = kTnKs == 2 || (kTnKs == 0 && dual);
        if (dual) {
            if (ks2) HIPCHK((launch_tn<false, true, 2>(t, pl.B, s)));
            else HIPCHK((launch_tn<false, true, 1>(t, pl.B, s)));
        } else if (a.x0_act) {
            if (ks2) HIPCHK((launch_tn<true, false, 2>(t, pl.B, s)));
            else HIPCHK((launch_tn<true, false, 1>(t, pl.B, s)));
        } else {
            if (ks2) HIPCHK((launch_tn<false, false, 2>(t, pl.B, s)));
            else HIPCHK((launch_tn<false, false, 1>(t, pl.B, s)));
        }
    }
    if (prof_slot >= 0) prof_mark(prof_slot, true, s);
    return MUCON_OK;
}

NtParams nt_base(const float *A, long a_bstride, int lda, int Ta, int Trows, int taps, int tap_step, int Kc,
                 const float *W, const float *bias, float *out, float slope) {
    NtParams p;
    memset(&p, 0, sizeof(p));
    p.A = A;
    p.a_bstride = a_bstride;
    p.lda = lda;
    p.Ta = Ta;
    p.Trows = Trows;
    p.taps = taps;
    p.tap_step = tap_step;
    p.Kc = Kc;
    p.W = W;
    p.ldw = taps * Kc;
    p.bias = bias;
    p.out = out;
    p.slope = slope;
    p.drop.thresh = 0;
    p.drop.scale = 1.f;
    return p;
}

// --- optional in-library timing of the two tape-streaming kernels (bench.py's roofline leg) ---
// slot 0: first_conv forward (NT core on the tape); slot 1: first_conv weight gradient (TN core).
// An event record is not free: each one opens a ~6 us bubble on the stream (r4 timeline: the four records of a step were the step's
// whole 24 us of gaps), so a run times every `stride`-th launch of a slot only (mucon_profile_stride; 1 = every launch).
struct ProfState {
    bool on = false;
    int cap = 0, stride = 1;
    int n[2] = {0, 0}, seen[2] = {0, 0};
    bool armed[2] = {false, false};
    hipEvent_t *ev[2] = {nullptr, nullptr};  // pairs (start, stop)
} g_prof;

void prof_mark(int slot, bool stop, hipStream_t s) {
    if (!g_prof.on || g_prof.n[slot] >= g_prof.cap) return;
    if (!stop) g_prof.armed[slot] = (g_prof.seen[slot]++ % g_prof.stride) == 0;
    if (!g_prof.armed[slot]) return;
    (void)hipEventRecord(g_prof.ev[slot][2 * g_prof.n[slot] + (stop ? 1 : 0)], s);
    if (stop) ++g_prof.n[slot];
}

// every weight re-layout of a forward pass in one launch.  The grid is the list of blocks that HAVE work (r4: the (32, L + 3 + slots)
// grid of r3 launched 832 workgroups of 1,024 threads of which ~560 returned at once): [n_f32 = 4 (L + 1) blocks of the f32 layouts, if
// any launch reads them][32 blocks of first_conv's planes][6 blocks of layer 0's data-gradient planes][20 blocks per fragment-image slot]
struct PackGrid {
    int n_f32, n_first, n_d0, n_fs;   // block counts of the four sections, in this order
};
// (r4: staging the fragment images through LDS -- a block loading 32 output rows of a layer once and emitting every fragment that
// depends on them -- was built and measured: 13.5 us against 13.5 us for fs_pack_body's gather, bit-identical images.  Not kept.)
__global__ __launch_bounds__(PACK_THREADS) void pack_all_kernel(const PackArgs a, const FsPackArgs f, const PackGrid g) {
    __shared__ float lds[PACK_LDS_FLOATS];
    int b = blockIdx.x;
    if (b < g.n_f32) return pack_weights_body(a, lds, b >> 2, b & 3, 4);
    b -= g.n_f32;
    if (b < g.n_first) return pack_weights_body(a, lds, a.L + 1, b, g.n_first);
    b -= g.n_first;
    if (b < g.n_d0) return pack_weights_body(a, lds, a.L + 2, b, g.n_d0);
    b -= g.n_d0;
    fs_pack_body(f, b / 20, (b % 20) * PACK_THREADS + threadIdx.x);
}

__global__ void dropout_mask_kernel(uint8_t *mask, long n, DropCfg d) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        mask[e] = drop_mul(d, (uint32_t)e) != 0.f ? 1 : 0;
}

}  // namespace

// Tuning / regression knobs: read from the environment once when the library is first used; the test hook
// mucon_test_set_knob applies the same parsing at run time (tests compare code paths inside one process).
extern int g_vit_lanes;   // viterbi.hip
extern int g_dec_mw;      // shead.hip
static bool apply_knob(const char *name, const char *e) {
    if (!strcmp(name, "MUCON_TAIL_CHAIN")) {
        if (e) g_tail_chain = atoi(e);
        return true;
    }
    if (!strcmp(name, "MUCON_VIT_LANES")) {
        if (e) g_vit_lanes = atoi(e);
        return true;
    }
    if (!strcmp(name, "MUCON_DEC_MW")) {
        if (e) g_dec_mw = atoi(e) ? 1 : 0;
        return true;
    }
    if (!strcmp(name, "MUCON_COARSE_RB")) {
        g_cs_rb = (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4) ? atoi(e) : 0;
        return true;
    }
    if (!strcmp(name, "MUCON_COARSE_SPLIT")) {
        g_cs = atoi(e) ? 1 : 0;
        return true;
    }
    if (!strcmp(name, "MUCON_FUSED_SPLIT_ROWS")) {
        g_fs_rows = atol(e);
        return true;
    }
    if (!strcmp(name, "MUCON_MFMA16")) {
        if (e) g_mfma16 = atoi(e) & 1;
        return true;
    }
    if (!strcmp(name, "MUCON_TN_SPLIT")) {
        if (e) g_tn_split = atoi(e) ? 1 : 0;
        return true;
    }
    if (!strcmp(name, "MUCON_TS_RUNS")) {
        if (e) g_ts_runs = atoi(e) ? 1 : 0;
        return true;
    }
    if (!strcmp(name, "MUCON_FIRST_CONV_SPLIT_ROWS")) {
        if (e) g_first_conv_split_rows = atol(e);
        return true;
    }
    if (!strcmp(name, "MUCON_NT_BM16_ROWS")) {
        if (e) g_nt_bm16_rows = atol(e);
        return true;
    }
    if (!strcmp(name, "MUCON_FUSE")) {
        if (e) g_no_fuse = atoi(e) ? 0 : 1;
        return true;
    }
    return false;
}
static const char *const kKnobs[] = {"MUCON_TS_RUNS", "MUCON_MFMA16", "MUCON_TAIL_CHAIN", "MUCON_DEC_MW", "MUCON_VIT_LANES", "MUCON_COARSE_RB", "MUCON_COARSE_SPLIT", "MUCON_FUSED_SPLIT_ROWS", "MUCON_TN_SPLIT", "MUCON_FIRST_CONV_SPLIT_ROWS", "MUCON_NT_BM16_ROWS", "MUCON_FUSE"};

void mucon_internal_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

extern "C" {

int mucon_abi_version(void) {
    static bool once = false;
    if (!once) {
        for (const char *name : kKnobs) {
            const char *e = getenv(name);
            if (e) apply_knob(name, e);
        }
        once = true;
    }
    return MUCON_ABI_VERSION;
}
int mucon_test_set_knob(const char *name, const char *value) {
    mucon_abi_version();   // the environment first: a later first use must not overwrite this
    if (!name || !value || !apply_knob(name, value)) return fail(MUCON_E_ARG, "unknown tuning knob %s", name ? name : "(null)");
    return MUCON_OK;
}
int mucon_test_get_knob(const char *name) {
    mucon_abi_version();
    if (name && !strcmp(name, "MUCON_MFMA16")) return g_mfma16;
    if (name && !strcmp(name, "MUCON_TN_SPLIT")) return g_tn_split;
    if (name && !strcmp(name, "MUCON_TS_RUNS")) return g_ts_runs;
    if (name && !strcmp(name, "MUCON_FIRST_CONV_SPLIT")) return g_first_conv_split;
    return -1;
}
int mucon_test_mfma_probe(int32_t shape16, int32_t launches, int32_t iters, void *scratch, size_t scratch_bytes, float *tflops_host,
                          float *clock_ghz_host, float *ms_host, void *stream) {
    const int grid = 1024;
    if (launches < 1 || iters < 1 || !scratch || scratch_bytes < grid * sizeof(ProbeOut) + 64)
        return fail(MUCON_E_ARG, "mfma_probe: launches=%d iters=%d scratch=%zu B (need %zu)", launches, iters, scratch_bytes, grid * sizeof(ProbeOut) + 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProbeOut *po = static_cast<ProbeOut *>(scratch);
    float *sink = reinterpret_cast<float *>(po + grid);
    auto go = [&]() {
        if (shape16 == 2) hipLaunchKernelGGL(mfma_lds_probe_kernel, dim3(grid), dim3(256), 0, s, iters, po, sink);
        else if (shape16) hipLaunchKernelGGL(mfma_probe_kernel<true>, dim3(grid), dim3(256), 0, s, iters, po, sink);
        else hipLaunchKernelGGL(mfma_probe_kernel<false>, dim3(grid), dim3(256), 0, s, iters, po, sink);
        return hipGetLastError();
    };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(go());   // ramp
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < launches; ++i) HIPCHK(go());
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    static ProbeOut host[1024];
    HIPCHK(hipMemcpy(host, po, sizeof(host), hipMemcpyDeviceToHost));
    double ghz[1024];
    int n = 0;
    for (int i = 0; i < grid; ++i)
        if (host[i].ticks > 0) ghz[n++] = (double)host[i].cycles / (double)host[i].ticks * 0.1;
    std::sort(ghz, ghz + n);
    if (tflops_host) *tflops_host = (float)((double)launches * grid * 4 * iters * kProbeFlopsPerIter / (ms * 1e-3) / 1e12);
    if (clock_ghz_host) *clock_ghz_host = n ? (float)ghz[n / 2] : 0.f;
    if (ms_host) *ms_host = ms;
    return MUCON_OK;
}
int mucon_test_read_cs_stamps(long long *stamps, int32_t *info, int32_t n_slots) {
#if CS_STAMP
    if (!stamps || !info || n_slots < 64) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_cs_stamps), sizeof(long long) * 64 * 2 * 4 * 12) != hipSuccess) return -1;
    const int used = g_cs_slot < 64 ? g_cs_slot : 64;
    for (int i = 0; i < used; ++i) {
        const CsStampInfo &c = g_cs_info[i];
        const int v[8] = {c.bwd, c.pool, c.taps, c.one, c.rb, c.gx, c.gy, c.rows};
        for (int k = 0; k < 8; ++k) info[i * 8 + k] = v[k];
    }
    g_cs_slot = 0;
    return used;
#else
    (void)stamps;
    (void)info;
    (void)n_slots;
    return 0;
#endif
}
int mucon_test_read_clock(int32_t slot, long long *out, int32_t n) {
#if CLK_STAMP
    if (!out || slot < 0 || slot > 4 || n < 2 * 4096) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (slot == 3) {   // ... behind the job lookup / at the first tile
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk_ph), sizeof(long long) * 2 * 4096) != hipSuccess) return -1;
    } else if (slot == 4) {   // ... behind the last tile
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk_pe), sizeof(long long) * 2 * 4096) != hipSuccess) return -1;
    } else if (slot == 2) {   // the weight-gradient launch's workgroups: absolute entry / exit ticks
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk_wg), sizeof(long long) * 2 * 4096) != hipSuccess) return -1;
    } else if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), sizeof(long long) * 2 * 4096, sizeof(long long) * 2 * 4096 * slot) != hipSuccess) return -1;
    int used = 0;
    for (int i = 0; i < 4096; ++i)
        if (out[2 * i + 1] > 0) used = i + 1;
    return used;
#else
    (void)slot;
    (void)out;
    (void)n;
    return 0;
#endif
}
const char *mucon_last_error(void) { return g_err; }
int mucon_test_read_stamps(long long *out, int32_t n) {
#if FS_STAMP
    if (!out || n < 64 * 8 * 8) return fail(MUCON_E_ARG, "read_stamps: buffer of 64 * 8 * 8 values needed");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fs_stamps), sizeof(long long) * 64 * 8 * 8));
    return MUCON_OK;
#else
    (void)out;
    (void)n;
    return fail(MUCON_E_ARG, "read_stamps: not a timing build (MUCON_HIPCC_FLAGS=-DFS_STAMP=1)");
#endif
}

int32_t mucon_encoder_out_length(const mucon_encoder_cfg *cfg) {
    if (validate(cfg) != MUCON_OK) return -1;
    Plan p;
    make_plan(cfg, p);
    return p.Tz;
}

size_t mucon_encoder_workspace_bytes(const mucon_encoder_cfg *cfg) {
    if (validate(cfg) != MUCON_OK) return 0;
    Plan p;
    make_plan(cfg, p);
    return p.total * sizeof(float);
}

int mucon_encoder_saved_view(const mucon_encoder_cfg *cfg, int32_t kind, int32_t layer, size_t *byte_offset,
                             int32_t *rows_per_video) {
    int rc = validate(cfg);
    if (rc != MUCON_OK) return rc;
    Plan pl;
    make_plan(cfg, pl);
    if (!byte_offset || !rows_per_video) return fail(MUCON_E_ARG, "null pointer argument");
    size_t off;
    int rows;
    switch (kind) {
        case 0:  // x[layer]: input of layer `layer` (x[0] = activated first_conv output, x[L] = last_conv input)
            if (layer < 0 || layer > pl.L) return fail(MUCON_E_ARG, "saved_view: layer %d", layer);
            off = pl.x[layer];
            rows = pl.Tl[layer];
            break;
        case 1:  // h[layer]: activated dilated_conv output
            if (layer < 0 || layer >= pl.L) return fail(MUCON_E_ARG, "saved_view: layer %d", layer);
            off = pl.h[layer];
            rows = pl.Tl[layer];
            break;
        case 2:  // ypre[layer]: un-pooled output of a max-pooled layer
            if (layer < 0 || layer >= pl.L || !cfg->pool_after[layer] || cfg->pool_type != 0)
                return fail(MUCON_E_ARG, "saved_view: layer %d is not max-pooled", layer);
            off = pl.ypre[layer];
            rows = pl.Tl[layer];
            break;
        case 3:  // z: last_conv output
            off = pl.z;
            rows = pl.Tz;
            break;
        default:
            return fail(MUCON_E_ARG, "saved_view: kind %d", kind);
    }
    *byte_offset = off * sizeof(float);
    *rows_per_video = rows;
    return MUCON_OK;
}

int mucon_encoder_fwd(const mucon_encoder_cfg *cfg, const mucon_encoder_params *prm, const float *tape, float *enc,
                      void *workspace, size_t workspace_bytes, void *stream) {
    int rc = validate(cfg);
    if (rc != MUCON_OK) return rc;
    if (!prm || !tape || !enc || !workspace) return fail(MUCON_E_ARG, "null pointer argument");
    Plan pl;
    make_plan(cfg, pl);
    if (workspace_bytes < pl.total * sizeof(float))
        return fail(MUCON_E_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, pl.total * sizeof(float));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(workspace);
    const float slope = cfg->leaky ? 0.01f : 0.f;
    const int B = pl.B, L = pl.L;

    PackArgs pa;
    memset(&pa, 0, sizeof(pa));
    for (int l = 0; l < L; ++l) {
        pa.dil_w[l] = prm->dil_w[l];
        pa.pw_w[l] = prm->pw_w[l];
    }
    pa.last_w = prm->last_w;
    const PackDecision dec = pack_decision(cfg, prm, pl);
    {
        std::lock_guard<std::mutex> lk(g_fwd_mu);
        if (g_fwd_record.size() >= kFwdRecordMax && g_fwd_record.find(workspace) == g_fwd_record.end()) g_fwd_record.clear();
        g_fwd_record[workspace] = FwdRecord{dec, fwd_shape_key(pl)};
    }
    const bool need_f32 = dec.need_f32, split_first = dec.split_first, split_dgrad0 = dec.split_dgrad0;
    pa.W1f = need_f32 ? ws + pl.W1f : nullptr;
    pa.W1b = ws + pl.W1b;
    pa.W2t = ws + pl.W2t;
    pa.Wlt = ws + pl.Wlt;
    pa.L = L;
    pa.D = pl.D;
    pa.first_w = prm->first_w;
    pa.first_planes = split_first ? reinterpret_cast<uint16_t *>(ws + pl.W0s) : nullptr;
    pa.dgrad0_planes = split_dgrad0 ? reinterpret_cast<uint16_t *>(ws + pl.Wd0s) : nullptr;
    pa.img16 = dec.img16;
    {   // ... and, in the SAME launch, the split images of the layers whose launches take gemm_fused_split.hpp / gemm_coarse_split.hpp
        // (a layer's images sit at its own slot): two launches were 8 + 7 us at the head of every forward pass
        FsPackArgs fa;
        memset(&fa, 0, sizeof(fa));
        int lo = -1, hi = -1;
        for (int l = 0; l < L; ++l)
            if (cs_on() || fs_level(cfg, pl, l)) {
                if (lo < 0) lo = l;
                hi = l;
            }
        int yfs = 0;
        if (lo >= 0) {
            fa.nl = hi - lo + 1;
            for (int l = lo; l <= hi; ++l) {
                fa.dil_w[l - lo] = prm->dil_w[l];
                fa.pw_w[l - lo] = prm->pw_w[l];
            }
            fa.last_w = (cs_on() && hi == L - 1) ? prm->last_w : nullptr;
            fa.img = reinterpret_cast<uint16_t *>(ws + pl.Wfs) + (long)lo * FS_LAYER_ELEMS;
            yfs = fa.nl + (fa.last_w ? 1 : 0);
        }
        PackGrid pg;
        pg.n_f32 = need_f32 ? 4 * (L + 1) : 0;
        pg.n_first = split_first ? 32 : 0;
        pg.n_d0 = split_dgrad0 ? 6 : 0;
        pg.n_fs = 20 * yfs;
        const int nblocks = pg.n_f32 + pg.n_first + pg.n_d0 + pg.n_fs;
        if (nblocks > 0) hipLaunchKernelGGL(pack_all_kernel, dim3(nblocks), dim3(PACK_THREADS), 0, s, pa, fa, pg);
        HIPCHK(hipGetLastError());
    }

    // first_conv + non-linearity (temporal.py:133); the tape is consumed row-major, no permute
    {
        NtParams p = nt_base(tape, (long)pl.T * pl.D, pl.D, pl.T, pl.T, 1, 0, pl.D, prm->first_w, prm->first_b,
                             ws + pl.x[0], slope);
        prof_mark(0, false, s);
        if (split_first) {
            HIPCHK((launch_nt_split<true>(p, reinterpret_cast<const uint16_t *>(ws + pl.W0s), B, s)));
        } else if (g_first_conv_ksplit && (long)B * pl.T <= g_first_conv_ksplit_rows && pl.D % 256 == 0 && prm->first_b) {
            // few rows: every workgroup would walk all D/32 k-tiles alone (64 dependent steps, 38 us at T = 2000).  Four k-chunks
            // in grid.z, partial sums in level-0 buffers that are idle during the forward, one ordered combine pass.
            p.ksplit = 4;
            p.part[0] = ws + pl.x[0];
            p.part[1] = ws + pl.h[0];     // written only by layer 0, after this
            p.part[2] = ws + pl.g[0];     // backward buffers
            p.part[3] = ws + pl.dpre[0];
            HIPCHK((launch_nt<false, false, true, false, false, false, 0, 1>(p, B, s)));
            const long n4 = (long)B * pl.T * 32;
            hipLaunchKernelGGL(first_conv_combine_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p, n4);
            HIPCHK(hipGetLastError());
        } else {
            HIPCHK((launch_nt<false, false, true, false, false, false, 0, 1>(p, B, s)));
        }
        prof_mark(0, true, s);
    }
    bool tail_done = false;   // layers L-2, L-1 and last_conv ran as one chained launch
    for (int l = 0; l < L; ++l) {
        const int Tl = pl.Tl[l];
        // When the dilation reaches past the sequence (d >= T_l, e.g. d = 512, 1024 at T/16) the outer taps only
        // ever read zero padding: centre tap alone.
        const bool centre_only = cfg->dilation[l] >= Tl;
        const DropCfg dl = make_drop(cfg->seed, l, cfg->p_drop_layer, cfg->training != 0);
        const int pool = !cfg->pool_after[l] ? 0 : (cfg->pool_type == 0 ? 1 : 2);
        if (l == L - 2 && tail_chain_ok(cfg, pl, prm, B)) {
            // the last two residual layers (centre tap only: row-local) and last_conv as ONE launch (gemm_coarse_split.hpp: ct_kernel)
            CtParams c;
            memset(&c, 0, sizeof(c));
            c.A = ws + pl.x[l];
            c.keep0 = ws + pl.x[l];
            c.Trows = Tl;
            c.slope = slope;
            for (int i = 0; i < 2; ++i) {
                const int li = l + i;
                CtStage &a = c.st[2 * i], &b2 = c.st[2 * i + 1];
                a.img = i == 0 ? fs_img(ws, pl, li, 0) + 4 * FS_WSTEP : fs_img(ws, pl, li, 4);
                a.bias = prm->dil_b[li];
                a.out = ws + pl.h[li];
                b2.img = fs_img(ws, pl, li, 2);
                b2.bias = prm->pw_b[li];
                b2.drop = make_drop(cfg->seed, li, cfg->p_drop_layer, cfg->training != 0);
                b2.out = ws + pl.x[li + 1];
            }
            c.st[4].img = fs_img(ws, pl, L, 4);
            c.st[4].bias = prm->last_b;
            c.st[4].out = ws + pl.z;
            HIPCHK(launch_ct<false>(c, B, s));
            tail_done = true;
            break;
        }
        if (!g_no_fuse && (long)B * Tl <= kFuseMaxRows) {
            // one launch per residual layer: dilated_conv + non-linearity (temporal.py:48-49), then conv_1x1,
            // dropout, residual (:50-52) and the pooling of WaveNetBlock (:137-142); h crosses through LDS
            FusedParams f;
            memset(&f, 0, sizeof(f));
            f.Trows = Tl;
            f.A = ws + pl.x[l];
            f.taps = centre_only ? 1 : 3;
            f.tap_step = cfg->dilation[l];
            f.W1 = ws + pl.W1f + (size_t)l * 49152 + (centre_only ? 128 : 0);
            f.ldw1 = 384;
            f.bias1 = prm->dil_b[l];
            f.out1 = ws + pl.h[l];
            f.W2 = prm->pw_w[l];
            f.bias2 = prm->pw_b[l];
            f.res2 = ws + pl.x[l];
            f.out2 = ws + pl.x[l + 1];
            f.out_pre = pool == 1 ? ws + pl.ypre[l] : nullptr;
            f.slope = slope;
            f.drop = dl;
            if (fs_level(cfg, pl, l) && f.bias1 && f.bias2) {
                const uint16_t *w1 = fs_img(ws, pl, l, 0), *w2 = fs_img(ws, pl, l, 2);
                if (pool == 0) HIPCHK((launch_fs<false, 0>(f, w1, w2, B, s)));
                else if (pool == 1) HIPCHK((launch_fs<false, 1>(f, w1, w2, B, s)));
                else HIPCHK((launch_fs<false, 2>(f, w1, w2, B, s)));
                continue;
            }
            if (cs_on() && f.bias1 && f.bias2) {
                const uint16_t *w1 = fs_img(ws, pl, l, 0) + (centre_only ? 4 * FS_WSTEP : 0), *w2 = fs_img(ws, pl, l, 2);
                if (centre_only) {
                    if (pool == 0) HIPCHK((launch_cs<false, 0, 1>(f, w1, w2, B, s)));
                    else if (pool == 1) HIPCHK((launch_cs<false, 1, 1>(f, w1, w2, B, s)));
                    else HIPCHK((launch_cs<false, 2, 1>(f, w1, w2, B, s)));
                } else {
                    if (pool == 0) HIPCHK((launch_cs<false, 0, 3>(f, w1, w2, B, s)));
                    else if (pool == 1) HIPCHK((launch_cs<false, 1, 3>(f, w1, w2, B, s)));
                    else HIPCHK((launch_cs<false, 2, 3>(f, w1, w2, B, s)));
                }
                continue;
            }
            if (pool == 0) HIPCHK((launch_fused<false, 0>(f, B, s)));
            else if (pool == 1) HIPCHK((launch_fused<false, 1>(f, B, s)));
            else HIPCHK((launch_fused<false, 2>(f, B, s)));
            continue;
        }
        {   // dilated_conv + non-linearity (temporal.py:48-49)
            NtParams p = nt_base(ws + pl.x[l], (long)Tl * 128, 128, Tl, Tl, centre_only ? 1 : 3, cfg->dilation[l], 128,
                                 ws + pl.W1f + (size_t)l * 49152 + (centre_only ? 128 : 0), prm->dil_b[l],
                                 ws + pl.h[l], slope);
            p.ldw = 384;
            HIPCHK((launch_nt<false, false, true, false, false, false, 0>(p, B, s)));
        }
        {   // conv_1x1, dropout, residual (temporal.py:50-52) and the pooling of WaveNetBlock (:137-142)
            NtParams p = nt_base(ws + pl.h[l], (long)Tl * 128, 128, Tl, Tl, 1, 0, 128, prm->pw_w[l], prm->pw_b[l],
                                 ws + pl.x[l + 1], slope);
            p.res = ws + pl.x[l];
            p.drop = dl;
            if (pool == 0) {
                HIPCHK((launch_nt<false, false, false, true, true, false, 0>(p, B, s)));
            } else if (pool == 1) {
                p.out_pre = ws + pl.ypre[l];
                HIPCHK((launch_nt<false, false, false, true, true, false, 1>(p, B, s)));
            } else {
                HIPCHK((launch_nt<false, false, false, true, true, false, 2>(p, B, s)));
            }
        }
    }
    if (!tail_done) {   // non-linearity + last_conv (temporal.py:144-145)
        const int Tz = pl.Tz;
        if (cs_on() && prm->last_b && !g_no_fuse && (long)B * pl.Tl[L - 1] <= kFuseMaxRows) {
            FusedParams f;
            memset(&f, 0, sizeof(f));
            f.Trows = Tz;
            f.A = ws + pl.x[L];
            f.taps = 1;
            f.bias1 = prm->last_b;
            f.out1 = ws + pl.z;
            f.slope = slope;
            f.drop = make_drop(0, 0, 0.f, false);
            HIPCHK((launch_cs<false, 0, 1, true, true>(f, fs_img(ws, pl, L, 2), nullptr, B, s)));
        } else {
        NtParams p = nt_base(ws + pl.x[L], (long)Tz * 128, 128, Tz, Tz, 1, 0, 128, prm->last_w, prm->last_b,
                             ws + pl.z, slope);
        HIPCHK((launch_nt<true, false, false, false, false, false, 0>(p, B, s)));
        }
    }
    {   // GroupNorm, ReLU, Dropout (models.py:759-768)
        GnArgs g;
        g.z = ws + pl.z;
        g.enc = enc;
        g.gamma = prm->gn_w;
        g.beta = prm->gn_b;
        g.stats = ws + pl.gnstat;
        g.Tz = pl.Tz;
        g.G = cfg->last_gn ? cfg->gn_groups : 32;
        g.eps = cfg->gn_eps;
        g.use_gn = cfg->last_gn;
        g.use_relu = cfg->last_relu;
        g.drop = make_drop(cfg->seed, L, cfg->p_drop_last, cfg->training != 0);
        hipLaunchKernelGGL(gn_fwd_kernel, dim3(g.G, B), dim3(GN_THREADS), 0, s, g.z, g.enc, g.gamma, g.beta, g.stats, g.Tz, g.G, g);
        HIPCHK(hipGetLastError());
    }
    return MUCON_OK;
}

int mucon_encoder_bwd(const mucon_encoder_cfg *cfg, const mucon_encoder_params *prm, const float *tape,
                      const float *d_enc, void *workspace, size_t workspace_bytes, const mucon_encoder_params *gr,
                      void *stream) {
    int rc = validate(cfg);
    if (rc != MUCON_OK) return rc;
    if (!prm || !tape || !d_enc || !workspace || !gr) return fail(MUCON_E_ARG, "null pointer argument");
    Plan pl;
    make_plan(cfg, pl);
    if (workspace_bytes < pl.total * sizeof(float))
        return fail(MUCON_E_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, pl.total * sizeof(float));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(workspace);
    const float slope = cfg->leaky ? 0.01f : 0.f;
    const int B = pl.B, L = pl.L, Tz = pl.Tz;
    DropCfg nodrop = make_drop(0, 0, 0.f, false);
    {   // the weight layouts this pass will read are the ones the forward of this workspace wrote
        const PackDecision now = pack_decision(cfg, prm, pl);
        std::lock_guard<std::mutex> lk(g_fwd_mu);
        auto it = g_fwd_record.find(workspace);
        if (it == g_fwd_record.end() || it->second.shape != fwd_shape_key(pl))
            return fail(MUCON_E_ARG, "encoder_bwd: no forward pass of this problem (B=%d T=%d D=%d L=%d) was run on this workspace", pl.B, pl.T, pl.D, pl.L);
        if (!(it->second.d == now))
            return fail(MUCON_E_ARG, "encoder_bwd: the forward pass wrote other weight layouts than this pass would read (f32 layouts %d/%d, "
                        "first_conv image %d/%d, layer-0 data-gradient image %d/%d, 16x16x32 order %d/%d): a tuning knob or the parameter "
                        "set changed between the two calls", (int)it->second.d.need_f32, (int)now.need_f32, (int)it->second.d.split_first,
                        (int)now.split_first, (int)it->second.d.split_dgrad0, (int)now.split_dgrad0, it->second.d.img16, now.img16);
    }

    Reducer red(s);
    size_t arena = 0, barena = 0;
    float *gz = ws + pl.gz;
    WgradQueue wq;
    WgradQueue *batch = &wq;   // every weight gradient of the pass is queued for ONE launch after the data-gradient chain

    {   // GroupNorm / ReLU / Dropout backward -> dz
        GnBwdArgs g;
        g.z = ws + pl.z;
        g.denc = d_enc;
        g.dz = gz;
        g.gamma = prm->gn_w;
        g.beta = prm->gn_b;
        g.stats = ws + pl.gnstat;
        g.part = ws + pl.gnpart;
        g.Tz = Tz;
        g.G = cfg->last_gn ? cfg->gn_groups : 32;
        g.use_gn = cfg->last_gn;
        g.use_relu = cfg->last_relu;
        g.drop = make_drop(cfg->seed, L, cfg->p_drop_last, cfg->training != 0);
        g.zero = reinterpret_cast<unsigned *>(ws + pl.sync);   // (the pass's first kernel: the words the later launches count in start at zero)
        // (a deferred y-head backward KERNEL no mucon_decoder_bwd took: it must run before its sums are taken below)
        if (g_head_kernel_pending.pending) {
            rc = head_kernel_flush();
            if (rc != MUCON_OK) return rc;
        }
        // (r6) a y-head slab reduction left pending on this stream (mucon_head_bwd_defer) rides in extra rows of this launch's grid
        HeadReduceTail tail;
        memset(&tail, 0, sizeof(tail));
        int tail_rows = 0;
        if (g_head_pending.pending && g_head_pending.stream == s) {
            const HeadPending &hp = g_head_pending;
            tail.slabs[0] = hp.w_slabs;
            tail.slabs[1] = hp.b_slabs;
            tail.out[0] = hp.d_w;
            tail.out[1] = hp.d_b;
            tail.stride[0] = tail.n_elems[0] = hp.C * hp.H;
            tail.stride[1] = tail.n_elems[1] = hp.C;
            tail.nslabs = hp.nblk;
            tail.nblocks0 = (hp.C * hp.H + 255) / 256;
            tail.nblocks = tail.nblocks0 + (hp.C + 255) / 256;
            tail_rows = (tail.nblocks + g.G - 1) / g.G;
            g_head_pending.pending = false;
        } else if (g_head_pending.pending) {
            // left on ANOTHER stream: not this pass's to take -- but nobody else may come for it either: finished now, on the stream it was left on
            rc = mucon_head_bwd_flush();
            if (rc != MUCON_OK) return rc;
        }
        hipLaunchKernelGGL(gn_bwd_kernel, dim3(g.G, B + tail_rows), dim3(GN_THREADS), 0, s, g.z, g.denc, g.dz, g.stats, g.Tz, g.G, B, g, tail);
        HIPCHK(hipGetLastError());
        if (cfg->last_gn) {
            red.add(ws + pl.gnpart, B, 256, 256, 0, 1, 128, gr->gn_w, 0);
            red.add(ws + pl.gnpart, B, 256, 256, 128, 1, 128, gr->gn_b, 0);
        } else {
            HIPCHK(hipMemsetAsync(gr->gn_w, 0, 128 * sizeof(float), s));
            HIPCHK(hipMemsetAsync(gr->gn_b, 0, 128 * sizeof(float), s));
        }
    }
    // Data-gradient chain.  Stage pairs that meet at a layer boundary without a pooling step run as one fused
    // launch: (last_conv or layer l+1's dilated conv) data gradient -> gradient at layer l's output ->
    // through layer l's conv_1x1 and non-linearity -> dpre_l.
    auto fused_tail = [&](FusedParams &f, int l) {  // stage 2 = conv_1x1 backward of layer l
        f.W2 = ws + pl.W2t + (size_t)l * 16384;
        f.mask2 = ws + pl.h[l];
        f.out2 = ws + pl.dpre[l];
        f.slope = slope;
        f.drop = make_drop(cfg->seed, l, cfg->p_drop_layer, cfg->training != 0);
    };
    bool have_dpre = false;  // dpre[l] already produced by the previous (fused) launch
    bool tail_chained = false;  // ... and so are g[L-1] and dpre[L-2]: layer L-1's data gradient ran inside the chained launch
    bool unpooled_by_producer = false;  // dyd[l] (un-pooled gradient) already written by the launch that produced g[l+1]
    {   // last_conv backward: weight gradient queued, data gradient on the chain
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.Y0 = gz;
        a.X0 = ws + pl.x[L];
        a.x_bstride = (long)Tz * 128;
        a.ldx = 128;
        a.Tx = Tz;
        a.taps = 1;
        a.nk0 = 1;
        a.x0_act = true;
        a.out_w0 = gr->last_w;
        a.out_b0 = gr->last_b;
        a.drop = nodrop;
        rc = wgrad(pl, ws, arena, barena, Tz, a, slope, red, s, -1, batch);
        if (rc != MUCON_OK) return rc;
        if (tail_chain_ok(cfg, pl, prm, B)) {
            // last_conv's and layer L-1's data gradients (both row-local) as ONE launch: g[L], dpre[L-1], g[L-1], dpre[L-2]
            CtParams c;
            memset(&c, 0, sizeof(c));
            c.A = gz;
            c.Trows = Tz;
            c.slope = slope;
            c.st[0].img = fs_img(ws, pl, L, 3);
            c.st[0].mask = ws + pl.x[L];
            c.st[0].out = ws + pl.g[L];
            c.st[0].drop = make_drop(cfg->seed, L - 1, cfg->p_drop_layer, cfg->training != 0);
            c.st[1].img = fs_img(ws, pl, L - 1, 3);
            c.st[1].mask = ws + pl.h[L - 1];
            c.st[1].out = ws + pl.dpre[L - 1];
            c.st[2].img = fs_img(ws, pl, L - 1, 5);
            c.st[2].out = ws + pl.g[L - 1];
            c.st[2].drop = make_drop(cfg->seed, L - 2, cfg->p_drop_layer, cfg->training != 0);
            c.st[3].img = fs_img(ws, pl, L - 2, 3);
            c.st[3].mask = ws + pl.h[L - 2];
            c.st[3].out = ws + pl.dpre[L - 2];
            HIPCHK(launch_ct<true>(c, B, s));
            have_dpre = true;
            tail_chained = true;
        } else if (!g_no_fuse && !cfg->pool_after[L - 1] && (long)B * pl.Tl[L - 1] <= kFuseMaxRows) {
            FusedParams f;
            memset(&f, 0, sizeof(f));
            f.Trows = Tz;
            f.A = gz;
            f.taps = 1;
            f.W1 = ws + pl.Wlt;
            f.ldw1 = 128;
            f.mask1 = ws + pl.x[L];
            f.out1 = ws + pl.g[L];
            fused_tail(f, L - 1);
            if (cs_on()) HIPCHK((launch_cs<true, 0, 1>(f, fs_img(ws, pl, L, 3), fs_img(ws, pl, L - 1, 3), B, s)));
            else HIPCHK((launch_fused<true, 0>(f, B, s)));
            have_dpre = true;
        } else {
            NtParams p = nt_base(gz, (long)Tz * 128, 128, Tz, Tz, 1, 0, 128, ws + pl.Wlt, nullptr, ws + pl.g[L], slope);
            p.mask = ws + pl.x[L];
            HIPCHK((launch_nt<false, false, false, false, false, true, 0>(p, B, s)));
        }
    }
    for (int l = L - 1; l >= 0; --l) {
        const int Tl = pl.Tl[l];
        const DropCfg dl = make_drop(cfg->seed, l, cfg->p_drop_layer, cfg->training != 0);
        const float *dyd = ws + pl.g[l + 1];   // gradient at the layer output
        float *dpre = ws + pl.dpre[l];
        if (have_dpre && cfg->pool_after[l]) {   // un-pooled by the fused launch that also made dpre[l]
            dyd = ws + pl.dyd[l];
            unpooled_by_producer = false;
        }
        if (!have_dpre) {
            if (cfg->pool_after[l]) {
                float *u = ws + pl.dyd[l];
                if (!unpooled_by_producer) {
                    const long n4 = (long)B * Tl * 32;
                    const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
                    hipLaunchKernelGGL(unpool_kernel, dim3(blocks), dim3(256), 0, s, ws + pl.g[l + 1],
                                       cfg->pool_type == 0 ? ws + pl.ypre[l] : nullptr, u, B, Tl, cfg->pool_type);
                    HIPCHK(hipGetLastError());
                }
                dyd = u;
            }
            unpooled_by_producer = false;
            // gradient at the dilated conv's pre-activation: through conv_1x1 (dropout replayed) and the non-linearity
            NtParams p = nt_base(dyd, (long)Tl * 128, 128, Tl, Tl, 1, 0, 128, ws + pl.W2t + (size_t)l * 16384, nullptr,
                                 dpre, slope);
            p.mask = ws + pl.h[l];
            p.drop = dl;
            HIPCHK((launch_nt<false, true, false, false, false, true, 0>(p, B, s)));
        }
        {   // all four parameter gradients of the layer: one job of the batched launch
            WgradArgs a;
            memset(&a, 0, sizeof(a));
            a.Y0 = dpre;
            a.X0 = ws + pl.x[l];
            a.x_bstride = (long)Tl * 128;
            a.ldx = 128;
            a.Tx = Tl;
            a.taps = 3;
            a.tap_step = cfg->dilation[l];
            a.nk0 = 3;
            a.Y1 = dyd;
            a.X1 = ws + pl.h[l];
            a.out_w0 = gr->dil_w[l];
            a.out_b0 = gr->dil_b[l];
            a.mode0 = 1;
            a.out_w1 = gr->pw_w[l];
            a.out_b1 = gr->pw_b[l];
            a.drop = dl;
            rc = wgrad(pl, ws, arena, barena, Tl, a, slope, red, s, -1, batch);
            if (rc != MUCON_OK) return rc;
        }
        if (tail_chained && l == L - 1) {
            have_dpre = true;   // dpre[L-2] exists
        } else {   // data gradient of the dilated conv + the residual branch -> gradient w.r.t. the layer input
            const bool centre_only = cfg->dilation[l] >= Tl;
            const float *W1b = ws + pl.W1b + (size_t)l * 49152 + (centre_only ? 128 : 0);
            have_dpre = false;
            if (!g_no_fuse && l >= 1 && !cfg->pool_after[l - 1] && (long)B * pl.Tl[l] <= kFuseMaxRows) {
                FusedParams f;
                memset(&f, 0, sizeof(f));
                f.Trows = Tl;
                f.A = dpre;
                f.taps = centre_only ? 1 : 3;
                f.tap_step = -cfg->dilation[l];
                f.W1 = W1b;
                f.ldw1 = 384;
                f.res1 = dyd;
                f.out1 = ws + pl.g[l];
                fused_tail(f, l - 1);
                if (fs_level(cfg, pl, l) && fs_level(cfg, pl, l - 1)) HIPCHK((launch_fs<true, 0>(f, fs_img(ws, pl, l, 1), fs_img(ws, pl, l - 1, 3), B, s)));
                else if (cs_on()) {
                    const uint16_t *w1 = fs_img(ws, pl, l, 1) + (centre_only ? 4 * FS_WSTEP : 0), *w2 = fs_img(ws, pl, l - 1, 3);
                    if (centre_only) HIPCHK((launch_cs<true, 0, 1>(f, w1, w2, B, s)));
                    else HIPCHK((launch_cs<true, 0, 3>(f, w1, w2, B, s)));
                } else HIPCHK((launch_fused<true, 0>(f, B, s)));
                have_dpre = true;
            } else if (!g_no_fuse && kPoolFuse && l >= 1 && cfg->pool_after[l - 1] && (long)B * pl.Tl[l] <= kFuseMaxRows) {
                // pooled boundary: dilated-conv data gradient on this (coarse) level, max-pool backward in its epilogue,
                // layer l-1's conv_1x1 backward on the 2 x rows of the finer level -- one launch instead of two
                FusedParams f;
                memset(&f, 0, sizeof(f));
                f.Trows = Tl;
                f.A = dpre;
                f.taps = centre_only ? 1 : 3;
                f.tap_step = -cfg->dilation[l];
                f.W1 = W1b;
                f.ldw1 = 384;
                f.res1 = dyd;
                f.out1 = ws + pl.dyd[l - 1];
                f.ypre = cfg->pool_type == 0 ? ws + pl.ypre[l - 1] : nullptr;
                f.Tfine = pl.Tl[l - 1];
                fused_tail(f, l - 1);
                if (fs_level(cfg, pl, l) && fs_level(cfg, pl, l - 1)) {
                    if (cfg->pool_type == 0) HIPCHK((launch_fs<true, 3>(f, fs_img(ws, pl, l, 1), fs_img(ws, pl, l - 1, 3), B, s)));
                    else HIPCHK((launch_fs<true, 4>(f, fs_img(ws, pl, l, 1), fs_img(ws, pl, l - 1, 3), B, s)));
                } else if (cs_on()) {
                    const uint16_t *w1 = fs_img(ws, pl, l, 1) + (centre_only ? 4 * FS_WSTEP : 0), *w2 = fs_img(ws, pl, l - 1, 3);
                    if (cfg->pool_type == 0) {
                        if (centre_only) HIPCHK((launch_cs<true, 3, 1>(f, w1, w2, B, s)));
                        else HIPCHK((launch_cs<true, 3, 3>(f, w1, w2, B, s)));
                    } else {
                        if (centre_only) HIPCHK((launch_cs<true, 4, 1>(f, w1, w2, B, s)));
                        else HIPCHK((launch_cs<true, 4, 3>(f, w1, w2, B, s)));
                    }
                } else if (cfg->pool_type == 0) HIPCHK((launch_fused<true, 3>(f, B, s)));
                else HIPCHK((launch_fused<true, 4>(f, B, s)));
                have_dpre = true;
                unpooled_by_producer = true;
            } else {
                NtParams p = nt_base(dpre, (long)Tl * 128, 128, Tl, Tl, centre_only ? 1 : 3, -cfg->dilation[l], 128, W1b,
                                     nullptr, ws + pl.g[l], slope);
                p.ldw = 384;
                p.res = dyd;
                p.mask = (l == 0) ? ws + pl.x[0] : nullptr;  // through first_conv's non-linearity
                if (l >= 1 && cfg->pool_after[l - 1] && kUnpoolFuse) {
                    // layer l-1 was pooled: this launch's epilogue scatters the gradient straight onto the un-pooled
                    // rows of level l-1 (the max-pool backward), instead of a separate pass over that level
                    p.out = ws + pl.dyd[l - 1];
                    p.Tfine = pl.Tl[l - 1];
                    p.ypre = cfg->pool_type == 0 ? ws + pl.ypre[l - 1] : nullptr;
                    if (cfg->pool_type == 0) HIPCHK((launch_nt<false, false, false, false, true, true, 3>(p, B, s)));
                    else HIPCHK((launch_nt<false, false, false, false, true, true, 4>(p, B, s)));
                    unpooled_by_producer = true;
                } else if (l == 0 && fs_level(cfg, pl, 0)) {
                    // layer 0's dilated-conv data gradient (+ residual, through first_conv's non-linearity): stage 1 of the split-bf16
                    // two-stage kernel on its own (16x16x32 tiles, no k-split across waves)
                    FusedParams f;
                    memset(&f, 0, sizeof(f));
                    f.Trows = Tl;
                    f.A = dpre;
                    f.taps = 3;
                    f.tap_step = -cfg->dilation[0];
                    f.res1 = dyd;
                    f.mask1 = ws + pl.x[0];
                    f.out1 = ws + pl.g[0];
                    f.slope = slope;
                    f.drop = make_drop(0, 0, 0.f, false);
                    HIPCHK((launch_fs<true, 0, true>(f, fs_img(ws, pl, 0, 1), nullptr, B, s)));
                } else if (l == 0 && cs_on()) {   // ... and below the chip-filling sizes on the k-split kernel, stage 1 alone
                    FusedParams f;
                    memset(&f, 0, sizeof(f));
                    f.Trows = Tl;
                    f.A = dpre;
                    f.taps = centre_only ? 1 : 3;
                    f.tap_step = -cfg->dilation[0];
                    f.res1 = dyd;
                    f.mask1 = ws + pl.x[0];
                    f.out1 = ws + pl.g[0];
                    f.slope = slope;
                    f.drop = make_drop(0, 0, 0.f, false);
                    const uint16_t *w1 = fs_img(ws, pl, 0, 1) + (centre_only ? 4 * FS_WSTEP : 0);
                    if (centre_only) HIPCHK((launch_cs<true, 0, 1, true>(f, w1, nullptr, B, s)));
                    else HIPCHK((launch_cs<true, 0, 3, true>(f, w1, nullptr, B, s)));
                } else if (l == 0 && !centre_only && kNtSplitDgrad0 && (long)B * Tl >= g_first_conv_split_rows) {
                    // bf16 MFMA on exactly split operands (gemm_split.hpp); the W1b image was written by the forward's pack_weights
                    HIPCHK((launch_nt_split<false, true, true, true>(p, reinterpret_cast<const uint16_t *>(ws + pl.Wd0s), B, s)));
                } else {
                    HIPCHK((launch_nt<false, false, false, false, true, true, 0>(p, B, s)));
                }
            }
        }
    }
    const hipEvent_t overlap_ev = g_bwd_event;
    if (overlap_ev) {
        // data-parallel step (mucon_encoder_bwd_overlap): every gradient except first_conv's is made final HERE -- the residual layers', last_conv's
        // and GroupNorm's jobs launched and reduced -- and the caller's event recorded: its all-reduce of that part of the flat buffer (3 of the 4 MB)
        // travels under first_conv's weight-gradient launch, which follows on fewer workgroups (g_bwd_max_wg)
        rc = flush_wgrads(pl, ws, arena, barena, wq, red, s);
        if (rc != MUCON_OK) return rc;
        HIPCHK(red.run());
        HIPCHK(hipEventRecord(overlap_ev, s));
    }
    {   // first_conv: the tape needs no gradient; its weight gradient streams the tape once more
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.Y0 = ws + pl.g[0];
        a.X0 = tape;
        a.x_bstride = (long)pl.T * pl.D;
        a.ldx = pl.D;
        a.Tx = pl.T;
        a.taps = 1;
        a.nk0 = pl.D / 128;
        a.out_w0 = gr->first_w;
        a.out_b0 = gr->first_b;
        a.drop = nodrop;
        rc = wgrad(pl, ws, arena, barena, pl.T, a, slope, red, s, 1, batch);
        if (rc != MUCON_OK) return rc;
    }
    {   // ... first_conv's included (its workgroups first): profile slot 1 times this launch
        prof_mark(1, false, s);
        rc = flush_wgrads(pl, ws, arena, barena, wq, red, s);
        if (rc != MUCON_OK) return rc;
        prof_mark(1, true, s);
    }
    HIPCHK(red.run());
    g_bwd_event = nullptr;   // one-shot
    g_bwd_max_wg = 0;
    return MUCON_OK;
}

int mucon_encoder_bwd_overlap(void *event_after_layers, int32_t max_workgroups) {
    if (max_workgroups < 0) return fail(MUCON_E_ARG, "encoder_bwd_overlap: max_workgroups %d", max_workgroups);
    g_bwd_event = static_cast<hipEvent_t>(event_after_layers);
    g_bwd_max_wg = max_workgroups;
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ head
size_t mucon_head_workspace_bytes(int32_t B, int32_t Tz, int32_t H, int32_t C) {
    const size_t nblk = (size_t)B * ((Tz + HB_Z - 1) / HB_Z);   // HB_Z <= HEAD_ZC: the larger block count of the two backward kernels
    return sizeof(float) * (align64((size_t)B * Tz * C) + align64(nblk * C * H) + align64(nblk * C));
}

static int head_check(int B, int Tz, int Tf, int H, int C) {
    if (B < 1 || Tz < 1 || Tf < 1) return fail(MUCON_E_ARG, "head: B=%d Tz=%d Tf=%d", B, Tz, Tf);
    if (C < 1 || C > HEAD_MAXC) return fail(MUCON_E_ARG, "head: %d classes unsupported (max %d)", C, HEAD_MAXC);
    if (H < 1 || H > 512) return fail(MUCON_E_ARG, "head: hidden %d unsupported", H);
    return MUCON_OK;
}

int mucon_head_fwd(int32_t B, int32_t Tz, int32_t Tf, int32_t H, int32_t C, const float *enc, const float *w,
                   const float *b, float *logits, float *logp, void *workspace, size_t workspace_bytes,
                   void *stream) {
    int rc = head_check(B, Tz, Tf, H, C);
    if (rc != MUCON_OK) return rc;
    if (!enc || !w || !b || !workspace) return fail(MUCON_E_ARG, "null pointer argument");
    if (workspace_bytes < mucon_head_workspace_bytes(B, Tz, H, C)) return fail(MUCON_E_WORKSPACE, "head workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HeadFwdArgs a;
    a.enc = enc;
    a.w = w;
    a.b = b;
    a.logits = logits;
    a.logp = logp;
    a.logp_z = static_cast<float *>(workspace);
    a.Tz = Tz;
    a.Tf = Tf;
    a.H = H;
    a.C = C;
    a.scale = (float)Tz / (float)Tf;
    const size_t smem = head_smem_bytes(H, C);
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(head_fwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(head_bwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr = true;
    }
    if (g_head_fwd_pending.pending) {   // an earlier deferred forward nobody took: launched now, on its own
        rc = head_fwd_flush();
        if (rc != MUCON_OK) return rc;
    }
    const bool defer = g_head_fwd_pending.armed && H == 128;
    g_head_fwd_pending.armed = false;
    if (defer) {
        // (mucon_head_fwd_defer) the kernel waits for the next mucon_lstm_fwd on this stream: logits / logp / the saved z-level log-probs exist once THAT call has been enqueued
        HeadFwdPending &fp = g_head_fwd_pending;
        fp.a = a;
        fp.gx = (Tz + HF_Z - 1) / HF_Z;
        fp.gy = B;
        fp.stream = s;
        fp.pending = true;
        return MUCON_OK;
    }
    if (H % 16 == 0) hipLaunchKernelGGL(head_fwd_z_kernel, dim3((Tz + HF_Z - 1) / HF_Z, B), dim3(256), head_fwd_z_smem_bytes(H), s, a.enc, a.w, a.b, a.logp_z, a.Tz, a.H, a.C, a);
    else hipLaunchKernelGGL(head_fwd_kernel, dim3((Tf + HEAD_FB - 1) / HEAD_FB, B), dim3(256), smem, s, a);
    HIPCHK(hipGetLastError());
    return MUCON_OK;
}

int mucon_head_fwd_defer(int32_t enable) {
    g_head_fwd_pending.armed = enable != 0;
    return MUCON_OK;
}

int head_fwd_flush() {
    HeadFwdPending &fp = g_head_fwd_pending;
    if (!fp.pending) return MUCON_OK;
    fp.pending = false;
    const HeadFwdArgs &a = fp.a;
    hipLaunchKernelGGL(head_fwd_z_kernel, dim3(fp.gx, fp.gy), dim3(256), head_fwd_z_smem_bytes(a.H), fp.stream, a.enc, a.w, a.b, a.logp_z, a.Tz, a.H, a.C, a);
    HIPCHK(hipGetLastError());
    return MUCON_OK;
}

int mucon_head_fwd_flush(void) { return head_fwd_flush(); }

int mucon_head_bwd(int32_t B, int32_t Tz, int32_t Tf, int32_t H, int32_t C, const float *enc, const float *w,
                   const float *d_logits, const float *d_logp, float *d_enc, float *d_w, float *d_b,
                   void *workspace, size_t workspace_bytes, void *stream) {
    int rc = head_check(B, Tz, Tf, H, C);
    if (rc != MUCON_OK) return rc;
    if (!enc || !w || !d_enc || !d_w || !d_b || !workspace) return fail(MUCON_E_ARG, "null pointer argument");
    if (workspace_bytes < mucon_head_workspace_bytes(B, Tz, H, C)) return fail(MUCON_E_WORKSPACE, "head workspace too small");
    if (g_head_fwd_pending.pending) {   // a deferred FORWARD nobody took: this pass reads what it saves
        rc = head_fwd_flush();
        if (rc != MUCON_OK) return rc;
    }
    if (g_head_pending.pending) {   // an earlier deferred reduction no encoder_bwd has taken: run it now, on its own stream
        rc = mucon_head_bwd_flush();
        if (rc != MUCON_OK) return rc;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(workspace);
    const int zblocks = H == 128 ? (Tz + HB_Z - 1) / HB_Z : (Tz + HEAD_ZC - 1) / HEAD_ZC;
    const size_t nblk = (size_t)B * zblocks;
    HeadBwdArgs a;
    a.enc = enc;
    a.w = w;
    a.dlogits = d_logits;
    a.dlogp = d_logp;
    a.logp_z = ws;
    a.denc = d_enc;
    a.w_slabs = ws + align64((size_t)B * Tz * C);
    a.b_slabs = a.w_slabs + align64(nblk * C * H);
    a.Tz = Tz;
    a.Tf = Tf;
    a.H = H;
    a.C = C;
    a.scale = (float)Tz / (float)Tf;
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(head_bwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr = true;
    }
    const bool defer = g_head_pending.armed && (C & 3) == 0 && ((C * H) & 3) == 0;   // (the tail sums float4 columns; other shapes reduce here)
    const bool defer_kernel = defer && g_head_kernel_pending.armed && H == 128;
    g_head_pending.armed = false;
    g_head_kernel_pending.armed = false;
    if (defer_kernel) {
        // (mucon_head_bwd_defer bit 1) the kernel itself waits for the next mucon_decoder_bwd on this stream: d_enc and the partial sums exist once THAT call has been enqueued
        HeadKernelPending &kp = g_head_kernel_pending;
        kp.a = a;
        kp.gx = zblocks;
        kp.gy = B;
        kp.stream = s;
        kp.pending = true;
    } else if (H == 128) hipLaunchKernelGGL(head_bwd_z_kernel, dim3(zblocks, B), dim3(256), 0, s, a.enc, a.w, a.dlogits, a.dlogp, a.logp_z, a.Tz, a.Tf, a.C, a.scale, a);
    else hipLaunchKernelGGL(head_bwd_kernel, dim3(zblocks, B), dim3(256), head_smem_bytes(H, C), s, a);
    HIPCHK(hipGetLastError());
    if (defer) {
        HeadPending &hp = g_head_pending;
        hp.pending = true;
        hp.w_slabs = a.w_slabs;
        hp.b_slabs = a.b_slabs;
        hp.d_w = d_w;
        hp.d_b = d_b;
        hp.nblk = (int)nblk;
        hp.C = C;
        hp.H = H;
        hp.stream = s;
        return MUCON_OK;
    }
    Reducer red(s);
    red.add(a.w_slabs, (int)nblk, (long)C * H, C * H, 0, 1, C * H, d_w, 0);
    red.add(a.b_slabs, (int)nblk, C, C, 0, 1, C, d_b, 0);
    HIPCHK(red.run());
    return MUCON_OK;
}

int mucon_head_bwd_defer(int32_t enable) {
    g_head_pending.armed = (enable & 1) != 0;
    g_head_kernel_pending.armed = (enable & 3) == 3;   // (the kernel can only wait if its sums wait too)
    return MUCON_OK;
}

// the z-level backward kernel a deferred mucon_head_bwd left behind and nobody took: launched now, on the stream it was left on
int head_kernel_flush() {
    HeadKernelPending &kp = g_head_kernel_pending;
    if (!kp.pending) return MUCON_OK;
    kp.pending = false;
    const HeadBwdArgs &a = kp.a;
    hipLaunchKernelGGL(head_bwd_z_kernel, dim3(kp.gx, kp.gy), dim3(256), 0, kp.stream, a.enc, a.w, a.dlogits, a.dlogp, a.logp_z, a.Tz, a.Tf, a.C, a.scale, a);
    HIPCHK(hipGetLastError());
    return MUCON_OK;
}

int mucon_head_bwd_flush(void) {
    int rc = head_kernel_flush();
    if (rc != MUCON_OK) return rc;
    HeadPending &hp = g_head_pending;
    if (!hp.pending) return MUCON_OK;
    hp.pending = false;
    Reducer red(hp.stream);
    red.add(hp.w_slabs, hp.nblk, (long)hp.C * hp.H, hp.C * hp.H, 0, 1, hp.C * hp.H, hp.d_w, 0);
    red.add(hp.b_slabs, hp.nblk, hp.C, hp.C, 0, 1, hp.C, hp.d_b, 0);
    HIPCHK(red.run());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ noft
__global__ void split_weights_kernel(const float *W, uint16_t *planes, int D, int img16);
namespace {
struct LinearPlan {
    size_t planes, slabs, bslabs, total;   // floats
    Plan pl;
};
bool linear_plan(int B, int T, int D, LinearPlan &lp) {
    if (B < 1 || T < 1 || D < 128 || D % 128 != 0) return false;
    memset(&lp.pl, 0, sizeof(lp.pl));
    lp.pl.B = B;
    size_t o = 0;
    lp.planes = o;
    o += align64((size_t)3 * 128 * D / 2);
    lp.slabs = o;
    lp.pl.slabs = o;
    one_job_arena(B, T, D, false, lp.pl.slab_floats, lp.pl.bslab_floats);
    o += lp.pl.slab_floats;
    lp.bslabs = o;
    lp.pl.bslabs = o;
    o += lp.pl.bslab_floats;
    lp.total = o;
    return true;
}
}  // namespace

size_t mucon_linear_workspace_bytes(int32_t B, int32_t T, int32_t D) {
    LinearPlan lp;
    if (!linear_plan(B, T, D, lp)) {
        fail(MUCON_E_ARG, "linear: B=%d T=%d D=%d (D must be a positive multiple of 128)", B, T, D);
        return 0;
    }
    return lp.total * sizeof(float);
}

int mucon_linear_fwd(int32_t B, int32_t T, int32_t D, const float *tape, const float *w, const float *b, float *out,
                     void *workspace, size_t workspace_bytes, void *stream) {
    LinearPlan lp;
    if (!linear_plan(B, T, D, lp)) return fail(MUCON_E_ARG, "linear: B=%d T=%d D=%d (D must be a positive multiple of 128)", B, T, D);
    if (!tape || !w || !b || !out || !workspace) return fail(MUCON_E_ARG, "null pointer argument");
    if (workspace_bytes < lp.total * sizeof(float)) return fail(MUCON_E_WORKSPACE, "linear workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(workspace);
    NtParams p = nt_base(tape, (long)T * D, D, T, T, 1, 0, D, w, b, out, 1.f);   // slope 1: the activation slot is the identity
    if (g_first_conv_split && (long)B * T >= g_first_conv_split_rows) {
        uint16_t *P = reinterpret_cast<uint16_t *>(ws + lp.planes);
        hipLaunchKernelGGL(split_weights_kernel, dim3(128), dim3(256), 0, s, w, P, D, g_mfma16 & 1);
        HIPCHK(hipGetLastError());
        HIPCHK((launch_nt_split<false>(p, P, B, s)));
    } else {
        HIPCHK((launch_nt<true, false, false, false, false, false, 0>(p, B, s)));
    }
    return MUCON_OK;
}

int mucon_linear_bwd(int32_t B, int32_t T, int32_t D, const float *tape, const float *d_out, float *d_w, float *d_b,
                     void *workspace, size_t workspace_bytes, void *stream) {
    LinearPlan lp;
    if (!linear_plan(B, T, D, lp)) return fail(MUCON_E_ARG, "linear: B=%d T=%d D=%d (D must be a positive multiple of 128)", B, T, D);
    if (!tape || !d_out || !d_w || !d_b || !workspace) return fail(MUCON_E_ARG, "null pointer argument");
    if (workspace_bytes < lp.total * sizeof(float)) return fail(MUCON_E_WORKSPACE, "linear workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.Y0 = d_out;
    a.X0 = tape;
    a.x_bstride = (long)T * D;
    a.ldx = D;
    a.Tx = T;
    a.taps = 1;
    a.nk0 = D / 128;
    a.out_w0 = d_w;
    a.out_b0 = d_b;
    a.drop = make_drop(0, 0, 0.f, false);
    Reducer red(s);
    size_t arena = 0, barena = 0;
    int rc = wgrad(lp.pl, static_cast<float *>(workspace), arena, barena, T, a, 0.f, red, s);
    if (rc != MUCON_OK) return rc;
    HIPCHK(red.run());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ mstcnpp building block
namespace {
bool conv128_plan(int B, int T, int taps, Plan &pl) {
    if (B < 1 || T < 1 || (taps != 1 && taps != 3)) return false;
    memset(&pl, 0, sizeof(pl));
    pl.B = B;
    pl.slabs = 0;
    one_job_arena(B, T, 128 * taps, false, pl.slab_floats, pl.bslab_floats);
    pl.bslabs = pl.slab_floats;
    return true;
}
}  // namespace

size_t mucon_conv128_workspace_bytes(int32_t B, int32_t T, int32_t taps) {
    Plan pl;
    if (!conv128_plan(B, T, taps, pl)) {
        fail(MUCON_E_ARG, "conv128: B=%d T=%d taps=%d (taps must be 1 or 3)", B, T, taps);
        return 0;
    }
    return (pl.slab_floats + pl.bslab_floats) * sizeof(float);
}

int mucon_conv128_fwd(int32_t B, int32_t T, int32_t taps, int32_t dilation, const float *x, const float *w_fwd, const float *b,
                      float *y, void *stream) {
    if (B < 1 || T < 1 || (taps != 1 && taps != 3) || dilation < 0) return fail(MUCON_E_ARG, "conv128: B=%d T=%d taps=%d dilation=%d", B, T, taps, dilation);
    if (!x || !w_fwd || !y) return fail(MUCON_E_ARG, "null pointer argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // taps that only ever read zero padding (dilation >= T) are dropped: the centre tap's 128 columns of w_fwd
    const bool centre_only = taps == 3 && dilation >= T;
    NtParams p = nt_base(x, (long)T * 128, 128, T, T, centre_only ? 1 : taps, dilation, 128, w_fwd + (centre_only ? 128 : 0), b, y, 1.f);
    p.ldw = taps * 128;
    HIPCHK((launch_nt<true, false, false, false, false, false, 0>(p, B, s)));   // slope 1: the activation slot is the identity
    return MUCON_OK;
}

int mucon_mstcn_fuse_fwd(int32_t B, int32_t T, const float *a, const float *b, const float *w, const float *bias, const float *f,
                         float p_drop, uint64_t seed, int32_t training, int32_t pool, float *y, float *y_pre, float *x_act,
                         void *stream) {
    if (B < 1 || T < 1 || (pool && T < 2)) return fail(MUCON_E_ARG, "mstcn_fuse: B=%d T=%d pool=%d", B, T, pool);
    if (!a || !b || !w || !f || !y || (pool && !y_pre)) return fail(MUCON_E_ARG, "null pointer argument");
    if (p_drop < 0.f || p_drop >= 1.f) return fail(MUCON_E_ARG, "mstcn_fuse: dropout probability %g", (double)p_drop);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the 1x1 convolution over the concatenation [a; b] = two "taps" that differ in their source, not in their row
    NtParams p = nt_base(a, (long)T * 128, 128, T, T, 2, 0, 128, w, bias, y, 0.f);   // slope 0: ReLU
    p.tap_shift = b - a;
    p.ldw = 256;
    p.res = f;
    p.out_act = x_act;
    p.out_pre = y_pre;
    p.drop = make_drop(seed, 0, p_drop, training != 0);
    if (pool) HIPCHK((launch_nt<false, false, true, true, true, false, 1>(p, B, s)));
    else HIPCHK((launch_nt<false, false, true, true, true, false, 0>(p, B, s)));
    return MUCON_OK;
}

int mucon_mstcn_tail_bwd(int32_t B, int32_t T, int32_t pool, const float *d_y, const float *y_pre, const float *x_act, float scale,
                         float *d_sum, float *d_u, void *stream) {
    if (B < 1 || T < 1) return fail(MUCON_E_ARG, "mstcn_tail_bwd: B=%d T=%d", B, T);
    if (!d_y || !x_act || !d_sum || !d_u || (pool && !y_pre)) return fail(MUCON_E_ARG, "null pointer argument");
    const long n4 = (long)B * T * 32;
    const int blocks = (int)std::min<long>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(mstcn_tail_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_y, y_pre, x_act, d_sum, d_u, B, T,
                       pool ? 1 : 0, scale);
    HIPCHK(hipGetLastError());
    return MUCON_OK;
}

int mucon_conv128_dgrad(int32_t B, int32_t T, int32_t taps, int32_t dilation, const float *g, const float *w_bwd, float *d_x,
                        void *stream) {
    if (B < 1 || T < 1 || (taps != 1 && taps != 3) || dilation < 0) return fail(MUCON_E_ARG, "conv128: B=%d T=%d taps=%d dilation=%d", B, T, taps, dilation);
    if (!g || !w_bwd || !d_x) return fail(MUCON_E_ARG, "null pointer argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool centre_only = taps == 3 && dilation >= T;
    NtParams p = nt_base(g, (long)T * 128, 128, T, T, centre_only ? 1 : taps, -dilation, 128, w_bwd + (centre_only ? 128 : 0), nullptr, d_x, 1.f);
    p.ldw = taps * 128;
    HIPCHK((launch_nt<true, false, false, false, false, false, 0>(p, B, s)));
    return MUCON_OK;
}

int mucon_conv128_wgrad(int32_t B, int32_t T, int32_t taps, int32_t dilation, const float *g, const float *x, float *d_w,
                        float *d_b, void *workspace, size_t workspace_bytes, void *stream) {
    Plan pl;
    if (!conv128_plan(B, T, taps, pl) || dilation < 0) return fail(MUCON_E_ARG, "conv128: B=%d T=%d taps=%d dilation=%d", B, T, taps, dilation);
    if (!g || !x || !d_w || !workspace) return fail(MUCON_E_ARG, "null pointer argument");
    if (workspace_bytes < (pl.slab_floats + pl.bslab_floats) * sizeof(float)) return fail(MUCON_E_WORKSPACE, "conv128 workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.Y0 = g;
    a.X0 = x;
    a.x_bstride = (long)T * 128;
    a.ldx = 128;
    a.Tx = T;
    a.taps = taps;
    a.tap_step = dilation;
    a.nk0 = taps;
    a.mode0 = taps == 3 ? 1 : 0;      // [o][i][tap] for the three-tap weight
    a.out_w0 = d_w;
    a.out_b0 = d_b;
    a.drop = make_drop(0, 0, 0.f, false);
    Reducer red(s);
    size_t arena = 0, barena = 0;
    int rc = wgrad(pl, static_cast<float *>(workspace), arena, barena, T, a, 0.f, red, s);
    if (rc != MUCON_OK) return rc;
    HIPCHK(red.run());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ helpers
int mucon_test_gemm_nt(const float *A, const float *W, const float *bias, float *out, int32_t M, int32_t K,
                       int32_t relu, void *stream) {
    if (M < 1 || K < 32 || K % 32 != 0) return fail(MUCON_E_ARG, "test_gemm_nt: M=%d K=%d", M, K);
    NtParams p = nt_base(A, 0, K, M, M, 1, 0, K, W, bias, out, 0.f);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (relu) HIPCHK((launch_nt<false, false, true, false, false, false, 0>(p, 1, s)));
    else HIPCHK((launch_nt<true, false, false, false, false, false, 0>(nt_base(A, 0, K, M, M, 1, 0, K, W, bias, out, 1.f), 1, s)));
    return MUCON_OK;
}

int mucon_test_gemm_tn(const float *Y, const float *X, float *out, int32_t M, int32_t K, void *workspace,
                       size_t workspace_bytes, void *stream) {
    if (M < 1 || K < 128 || K % 128 != 0) return fail(MUCON_E_ARG, "test_gemm_tn: M=%d K=%d", M, K);
    Plan pl;
    memset(&pl, 0, sizeof(pl));
    pl.B = 1;
    pl.slabs = 0;
    one_job_arena(1, M, K, false, pl.slab_floats, pl.bslab_floats);
    pl.bslabs = pl.slab_floats;
    if (workspace_bytes < (pl.slab_floats + pl.bslab_floats) * sizeof(float)) return fail(MUCON_E_WORKSPACE, "test_gemm_tn workspace");
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.Y0 = Y;
    a.X0 = X;
    a.ldx = K;
    a.Tx = M;
    a.taps = 1;
    a.nk0 = K / 128;
    a.out_w0 = out;
    a.drop = make_drop(0, 0, 0.f, false);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Reducer red(s);
    size_t arena = 0, barena = 0;
    int rc = wgrad(pl, static_cast<float *>(workspace), arena, barena, M, a, 0.f, red, s);
    if (rc != MUCON_OK) return rc;
    HIPCHK(red.run());
    return MUCON_OK;
}

int mucon_test_dropout_mask(uint8_t *mask, int64_t n, uint64_t seed, int32_t site, float p, void *stream) {
    DropCfg d = make_drop(seed, site, p, true);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), mask, (long)n, d);
    HIPCHK(hipGetLastError());
    return MUCON_OK;
}

int mucon_profile_begin(int32_t max_records) {
    if (g_prof.on) return fail(MUCON_E_ARG, "profile already active");
    if (max_records < 1 || max_records > 100000) return fail(MUCON_E_ARG, "max_records %d", max_records);
    for (int k = 0; k < 2; ++k) {
        g_prof.ev[k] = new hipEvent_t[2 * (size_t)max_records];
        for (int i = 0; i < 2 * max_records; ++i) HIPCHK(hipEventCreate(&g_prof.ev[k][i]));
        g_prof.n[k] = g_prof.seen[k] = 0;
        g_prof.armed[k] = false;
    }
    g_prof.cap = max_records;
    g_prof.on = true;
    return MUCON_OK;
}

int mucon_profile_stride(int32_t every) {
    if (every < 1) return fail(MUCON_E_ARG, "profile stride %d", every);
    g_prof.stride = every;
    return MUCON_OK;
}

int mucon_profile_end(float *total_ms_host, int32_t *count_host) {
    if (!g_prof.on) return fail(MUCON_E_ARG, "profile not active");
    g_prof.on = false;
    for (int k = 0; k < 2; ++k) {
        float tot = 0.f;
        for (int i = 0; i < g_prof.n[k]; ++i) {
            HIPCHK(hipEventSynchronize(g_prof.ev[k][2 * i + 1]));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[k][2 * i], g_prof.ev[k][2 * i + 1]));
            tot += ms;
        }
        total_ms_host[k] = tot;
        count_host[k] = g_prof.n[k];
        for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[k][i]);
        delete[] g_prof.ev[k];
        g_prof.ev[k] = nullptr;
    }
    return MUCON_OK;
}

__global__ void split_weights_kernel(const float *W, uint16_t *planes, int D, int img16) {
    if (img16) sp_split_weights16(W, planes, D, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
    else sp_split_weights(W, planes, D, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
int mucon_test_first_conv_split(const float *tape, const float *w, const float *b, float *out, int32_t B, int32_t T,
                                int32_t D, int32_t relu, void *planes, size_t planes_bytes, int32_t iters, float *ms_host,
                                void *stream) {
    if (B < 1 || T < 1 || D < 128 || D % 128 != 0 || iters < 1) return fail(MUCON_E_ARG, "test_first_conv_split: B=%d T=%d D=%d", B, T, D);
    if (!planes || planes_bytes < (size_t)3 * 128 * D * 2) return fail(MUCON_E_WORKSPACE, "test_first_conv_split: planes buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint16_t *P = static_cast<uint16_t *>(planes);
    hipLaunchKernelGGL(split_weights_kernel, dim3(128), dim3(256), 0, s, w, P, D, g_mfma16 & 1);
    HIPCHK(hipGetLastError());
    NtParams p = nt_base(tape, (long)T * D, D, T, T, 1, 0, D, w, b, out, 0.f);
    auto go = [&]() { return relu ? launch_nt_split<true>(p, P, B, s) : launch_nt_split<false>(p, P, B, s); };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(go());  // warm-up
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(go());
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_host) *ms_host = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MUCON_OK;
}

int mucon_bench_first_conv(const float *tape, const float *w, const float *b, float *out, int32_t B, int32_t T,
                           int32_t D, int32_t iters, float *ms_host, void *stream) {
    if (D % 32 != 0 || iters < 1) return fail(MUCON_E_ARG, "bench_first_conv: D=%d iters=%d", D, iters);
    hipStream_t s = static_cast<hipStream_t>(stream);
    NtParams p = nt_base(tape, (long)T * D, D, T, T, 1, 0, D, w, b, out, 0.f);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK((launch_nt<false, false, true, false, false, false, 0, 1>(p, B, s)));  // warm-up
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) HIPCHK((launch_nt<false, false, true, false, false, false, 0, 1>(p, B, s)));
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_host = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MUCON_OK;
}

}  // extern "C"
