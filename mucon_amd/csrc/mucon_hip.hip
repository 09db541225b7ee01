// libmucon_hip.so -- host side of the C ABI (include/mucon_hip.h): sequences the gfx950 kernels of
// the encoder / y-head forward and backward on the caller's stream.  No allocation, no
// synchronisation (except the explicit bench helper); all state lives in the caller's workspace.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mucon_hip.h"
#include "common.hpp"
#include "gemm_nt.hpp"
#include "gemm_tn.hpp"
#include "small_kernels.hpp"

int g_nt_force_bm = 0;

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

// time-chunk length for a weight-gradient launch: aim at ~512 workgroups
inline int pick_mc(int B, int Trows, int kchunks) {
    long want = ((long)B * Trows * kchunks + 511) / 512;
    long mc = ((want + 31) / 32) * 32;
    if (mc < 32) mc = 32;
    if (mc > 1024) mc = 1024;
    return (int)mc;
}

struct Plan {
    int L, B, T, D, Tz;
    int Tl[MUCON_MAX_LAYERS + 1];
    size_t W1f, W1b, W2t, Wlt;
    size_t x[MUCON_MAX_LAYERS + 1], h[MUCON_MAX_LAYERS], ypre[MUCON_MAX_LAYERS];
    size_t z, gnstat, gnpart;
    size_t gA, gB, dpre, dyd;
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
    for (int l = 0; l <= p.L; ++l) p.x[l] = take((size_t)p.B * p.Tl[l] * 128);
    for (int l = 0; l < p.L; ++l) {
        p.h[l] = take((size_t)p.B * p.Tl[l] * 128);
        p.ypre[l] = (c->pool_after[l] && c->pool_type == 0) ? take((size_t)p.B * p.Tl[l] * 128) : 0;
    }
    p.z = take((size_t)p.B * p.Tz * 128);
    p.gnstat = take((size_t)p.B * 128 * 2);
    p.gnpart = take((size_t)p.B * 256);
    const size_t full = (size_t)p.B * p.T * 128;
    p.gA = take(full);
    p.gB = take(full);
    p.dpre = take(full);
    p.dyd = take(full);
    // slabs: the largest weight-gradient launch
    size_t sf = 0, bf = 0;
    auto consider = [&](int Trows, int Ktot) {
        const int mc = pick_mc(p.B, Trows, Ktot / 128);
        const size_t nmc = (size_t)p.B * ((Trows + mc - 1) / mc);
        sf = sf > nmc * 128 * Ktot ? sf : nmc * 128 * Ktot;
        bf = bf > nmc * 128 ? bf : nmc * 128;
    };
    consider(p.T, p.D);
    for (int l = 0; l < p.L; ++l) {
        consider(p.Tl[l], 384);
        consider(p.Tl[l], 128);
    }
    consider(p.Tz, 128);
    p.slab_floats = sf;
    p.bslab_floats = bf;
    p.slabs = take(sf);
    p.bslabs = take(bf);
    p.total = o;
}

void prof_mark(int slot, bool stop, hipStream_t s);

hipError_t reduce_slabs(const float *slabs, int nslabs, long stride, float *out, int n, int mode, hipStream_t s) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((n + 63) / 64), dim3(1024), 0, s, slabs, nslabs, stride, out, n, mode);
    return hipGetLastError();
}

// weight gradient through the TN core: out_w (mode 0: [128][Ktot], mode 1: conv3 layout), out_b [128]
template <bool Y_DROP, bool X_ACT>
int wgrad(const Plan &pl, float *ws, const float *Y, int Trows, const float *X, long x_bstride, int ldx, int Tx,
          int taps, int tap_step, int Ktot, float slope, DropCfg drop, float *out_w, int mode, float *out_b,
          hipStream_t s, int prof_slot = -1) {
    TnParams t;
    t.Y = Y;
    t.Trows = Trows;
    t.X = X;
    t.x_bstride = x_bstride;
    t.ldx = ldx;
    t.Tx = Tx;
    t.taps = taps;
    t.tap_step = tap_step;
    t.Ktot = Ktot;
    t.slabs = ws + pl.slabs;
    t.bias_slabs = out_b ? ws + pl.bslabs : nullptr;
    t.MC = pick_mc(pl.B, Trows, Ktot / 128);
    t.chunks_per_video = (Trows + t.MC - 1) / t.MC;
    t.slope = slope;
    t.drop = drop;
    const int nmc = pl.B * t.chunks_per_video;
    if ((size_t)nmc * 128 * Ktot > pl.slab_floats) return fail(MUCON_E_WORKSPACE, "internal: slab region too small");
    if (prof_slot >= 0) prof_mark(prof_slot, false, s);
    HIPCHK((launch_tn<Y_DROP, X_ACT>(t, pl.B, s)));
    if (prof_slot >= 0) prof_mark(prof_slot, true, s);
    HIPCHK(reduce_slabs(t.slabs, nmc, (long)128 * Ktot, out_w, 128 * Ktot, mode, s));
    if (out_b) HIPCHK(reduce_slabs(t.bias_slabs, nmc, 128, out_b, 128, 0, s));
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
    p.bias = bias;
    p.out = out;
    p.slope = slope;
    p.drop.thresh = 0;
    p.drop.scale = 1.f;
    return p;
}

// --- optional in-library timing of the two tape-streaming kernels (bench.py's roofline leg) ---
// slot 0: first_conv forward (NT core on the tape); slot 1: first_conv weight gradient (TN core).
struct ProfState {
    bool on = false;
    int cap = 0;
    int n[2] = {0, 0};
    hipEvent_t *ev[2] = {nullptr, nullptr};  // pairs (start, stop)
} g_prof;

void prof_mark(int slot, bool stop, hipStream_t s) {
    if (!g_prof.on || g_prof.n[slot] >= g_prof.cap) return;
    (void)hipEventRecord(g_prof.ev[slot][2 * g_prof.n[slot] + (stop ? 1 : 0)], s);
    if (stop) ++g_prof.n[slot];
}

__global__ void dropout_mask_kernel(uint8_t *mask, long n, DropCfg d) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        mask[e] = drop_mul(d, (uint32_t)e) != 0.f ? 1 : 0;
}

}  // namespace

void mucon_internal_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

extern "C" {

int mucon_abi_version(void) {
    static bool once = false;
    if (!once) {  // tuning hook: force the NT tile height (32 / 64 / 128)
        const char *e = getenv("MUCON_NT_BM");
        if (e) g_nt_force_bm = atoi(e);
        once = true;
    }
    return MUCON_ABI_VERSION;
}
const char *mucon_last_error(void) { return g_err; }

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
    pa.W1f = ws + pl.W1f;
    pa.W1b = ws + pl.W1b;
    pa.W2t = ws + pl.W2t;
    pa.Wlt = ws + pl.Wlt;
    pa.L = L;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(192, L + 1), dim3(256), 0, s, pa);
    HIPCHK(hipGetLastError());

    // first_conv + non-linearity (temporal.py:133); the tape is consumed row-major, no permute
    {
        NtParams p = nt_base(tape, (long)pl.T * pl.D, pl.D, pl.T, pl.T, 1, 0, pl.D, prm->first_w, prm->first_b,
                             ws + pl.x[0], slope);
        prof_mark(0, false, s);
        HIPCHK((launch_nt<false, false, true, false, false, false, 0>(p, B, s)));
        prof_mark(0, true, s);
    }
    for (int l = 0; l < L; ++l) {
        const int Tl = pl.Tl[l];
        {   // dilated_conv + non-linearity (temporal.py:48-49)
            NtParams p = nt_base(ws + pl.x[l], (long)Tl * 128, 128, Tl, Tl, 3, cfg->dilation[l], 128,
                                 ws + pl.W1f + (size_t)l * 49152, prm->dil_b[l], ws + pl.h[l], slope);
            HIPCHK((launch_nt<false, false, true, false, false, false, 0>(p, B, s)));
        }
        {   // conv_1x1, dropout, residual (temporal.py:50-52) and the pooling of WaveNetBlock (:137-142)
            NtParams p = nt_base(ws + pl.h[l], (long)Tl * 128, 128, Tl, Tl, 1, 0, 128, prm->pw_w[l], prm->pw_b[l],
                                 ws + pl.x[l + 1], slope);
            p.res = ws + pl.x[l];
            p.drop = make_drop(cfg->seed, l, cfg->p_drop_layer, cfg->training != 0);
            if (!cfg->pool_after[l]) {
                HIPCHK((launch_nt<false, false, false, true, true, false, 0>(p, B, s)));
            } else if (cfg->pool_type == 0) {
                p.out_pre = ws + pl.ypre[l];
                HIPCHK((launch_nt<false, false, false, true, true, false, 1>(p, B, s)));
            } else {
                HIPCHK((launch_nt<false, false, false, true, true, false, 2>(p, B, s)));
            }
        }
    }
    {   // non-linearity + last_conv (temporal.py:144-145)
        const int Tz = pl.Tz;
        NtParams p = nt_base(ws + pl.x[L], (long)Tz * 128, 128, Tz, Tz, 1, 0, 128, prm->last_w, prm->last_b,
                             ws + pl.z, slope);
        HIPCHK((launch_nt<true, false, false, false, false, false, 0>(p, B, s)));
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
        hipLaunchKernelGGL(gn_fwd_kernel, dim3(B), dim3(256), 0, s, g);
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

    float *cur = ws + pl.gA;   // gradient w.r.t. the current activation
    float *other = ws + pl.gB;

    {   // GroupNorm / ReLU / Dropout backward -> dz in `cur`
        GnBwdArgs g;
        g.z = ws + pl.z;
        g.denc = d_enc;
        g.dz = cur;
        g.gamma = prm->gn_w;
        g.beta = prm->gn_b;
        g.stats = ws + pl.gnstat;
        g.part = ws + pl.gnpart;
        g.Tz = Tz;
        g.G = cfg->last_gn ? cfg->gn_groups : 32;
        g.use_gn = cfg->last_gn;
        g.use_relu = cfg->last_relu;
        g.drop = make_drop(cfg->seed, L, cfg->p_drop_last, cfg->training != 0);
        hipLaunchKernelGGL(gn_bwd_kernel, dim3(B), dim3(256), 0, s, g);
        HIPCHK(hipGetLastError());
        if (cfg->last_gn) {
            HIPCHK(reduce_slabs(ws + pl.gnpart, B, 256, gr->gn_w, 128, 0, s));
            HIPCHK(reduce_slabs(ws + pl.gnpart + 128, B, 256, gr->gn_b, 128, 0, s));
        } else {
            HIPCHK(hipMemsetAsync(gr->gn_w, 0, 128 * sizeof(float), s));
            HIPCHK(hipMemsetAsync(gr->gn_b, 0, 128 * sizeof(float), s));
        }
    }
    {   // last_conv backward
        rc = wgrad<false, true>(pl, ws, cur, Tz, ws + pl.x[L], (long)Tz * 128, 128, Tz, 1, 0, 128, slope, nodrop,
                                gr->last_w, 0, gr->last_b, s);
        if (rc != MUCON_OK) return rc;
        NtParams p = nt_base(cur, (long)Tz * 128, 128, Tz, Tz, 1, 0, 128, ws + pl.Wlt, nullptr, other, slope);
        p.mask = ws + pl.x[L];
        HIPCHK((launch_nt<false, false, false, false, false, true, 0>(p, B, s)));
        float *t = cur;
        cur = other;
        other = t;
    }
    for (int l = L - 1; l >= 0; --l) {
        const int Tl = pl.Tl[l];
        const DropCfg dl = make_drop(cfg->seed, l, cfg->p_drop_layer, cfg->training != 0);
        const float *dyd = cur;
        if (cfg->pool_after[l]) {
            float *u = ws + pl.dyd;
            const long n4 = (long)B * Tl * 32;
            const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
            hipLaunchKernelGGL(unpool_kernel, dim3(blocks), dim3(256), 0, s, cur,
                               cfg->pool_type == 0 ? ws + pl.ypre[l] : nullptr, u, B, Tl, cfg->pool_type);
            HIPCHK(hipGetLastError());
            dyd = u;
        }
        // conv_1x1: weight/bias gradient, then data gradient through the dilated conv's non-linearity
        rc = wgrad<true, false>(pl, ws, dyd, Tl, ws + pl.h[l], (long)Tl * 128, 128, Tl, 1, 0, 128, slope, dl,
                                gr->pw_w[l], 0, gr->pw_b[l], s);
        if (rc != MUCON_OK) return rc;
        {
            NtParams p = nt_base(dyd, (long)Tl * 128, 128, Tl, Tl, 1, 0, 128, ws + pl.W2t + (size_t)l * 16384, nullptr,
                                 ws + pl.dpre, slope);
            p.mask = ws + pl.h[l];
            p.drop = dl;
            HIPCHK((launch_nt<false, true, false, false, false, true, 0>(p, B, s)));
        }
        // dilated_conv: weight/bias gradient, then data gradient + residual
        rc = wgrad<false, false>(pl, ws, ws + pl.dpre, Tl, ws + pl.x[l], (long)Tl * 128, 128, Tl, 3, cfg->dilation[l],
                                 384, slope, nodrop, gr->dil_w[l], 1, gr->dil_b[l], s);
        if (rc != MUCON_OK) return rc;
        {
            float *dst = (dyd == cur) ? other : cur;
            NtParams p = nt_base(ws + pl.dpre, (long)Tl * 128, 128, Tl, Tl, 3, -cfg->dilation[l], 128,
                                 ws + pl.W1b + (size_t)l * 49152, nullptr, dst, slope);
            p.res = dyd;
            p.mask = (l == 0) ? ws + pl.x[0] : nullptr;  // through first_conv's non-linearity
            HIPCHK((launch_nt<false, false, false, false, true, true, 0>(p, B, s)));
            if (dst == other) {
                other = cur;
                cur = dst;
            }
        }
    }
    // first_conv: the tape needs no gradient; weight gradient streams the tape once more
    rc = wgrad<false, false>(pl, ws, cur, pl.T, tape, (long)pl.T * pl.D, pl.D, pl.T, 1, 0, pl.D, slope, nodrop,
                             gr->first_w, 0, gr->first_b, s, 1);
    return rc;
}

// ------------------------------------------------------------------------------------------ head
size_t mucon_head_workspace_bytes(int32_t B, int32_t Tz, int32_t H, int32_t C) {
    const size_t nblk = (size_t)B * ((Tz + HEAD_ZC - 1) / HEAD_ZC);
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
    hipLaunchKernelGGL(head_fwd_kernel, dim3((Tf + HEAD_FB - 1) / HEAD_FB, B), dim3(256), smem, s, a);
    HIPCHK(hipGetLastError());
    return MUCON_OK;
}

int mucon_head_bwd(int32_t B, int32_t Tz, int32_t Tf, int32_t H, int32_t C, const float *enc, const float *w,
                   const float *d_logits, const float *d_logp, float *d_enc, float *d_w, float *d_b,
                   void *workspace, size_t workspace_bytes, void *stream) {
    int rc = head_check(B, Tz, Tf, H, C);
    if (rc != MUCON_OK) return rc;
    if (!enc || !w || !d_enc || !d_w || !d_b || !workspace) return fail(MUCON_E_ARG, "null pointer argument");
    if (workspace_bytes < mucon_head_workspace_bytes(B, Tz, H, C)) return fail(MUCON_E_WORKSPACE, "head workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(workspace);
    const int zblocks = (Tz + HEAD_ZC - 1) / HEAD_ZC;
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
    hipLaunchKernelGGL(head_bwd_kernel, dim3(zblocks, B), dim3(256), head_smem_bytes(H, C), s, a);
    HIPCHK(hipGetLastError());
    HIPCHK(reduce_slabs(a.w_slabs, (int)nblk, (long)C * H, d_w, C * H, 0, s));
    HIPCHK(reduce_slabs(a.b_slabs, (int)nblk, C, d_b, C, 0, s));
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
    const int mc = pick_mc(1, M, K / 128);
    const size_t nmc = (M + mc - 1) / mc;
    pl.slabs = 0;
    pl.slab_floats = nmc * 128 * K;
    pl.bslabs = align64(pl.slab_floats);
    if (workspace_bytes < (pl.bslabs + nmc * 128) * sizeof(float)) return fail(MUCON_E_WORKSPACE, "test_gemm_tn workspace");
    return wgrad<false, false>(pl, static_cast<float *>(workspace), Y, M, X, 0, K, M, 1, 0, K, 0.f,
                               make_drop(0, 0, 0.f, false), out, 0, nullptr, static_cast<hipStream_t>(stream));
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
        g_prof.n[k] = 0;
    }
    g_prof.cap = max_records;
    g_prof.on = true;
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

int mucon_bench_first_conv(const float *tape, const float *w, const float *b, float *out, int32_t B, int32_t T,
                           int32_t D, int32_t iters, float *ms_host, void *stream) {
    if (D % 32 != 0 || iters < 1) return fail(MUCON_E_ARG, "bench_first_conv: D=%d iters=%d", D, iters);
    hipStream_t s = static_cast<hipStream_t>(stream);
    NtParams p = nt_base(tape, (long)T * D, D, T, T, 1, 0, D, w, b, out, 0.f);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK((launch_nt<false, false, true, false, false, false, 0>(p, B, s)));  // warm-up
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) HIPCHK((launch_nt<false, false, true, false, false, false, 0>(p, B, s)));
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
