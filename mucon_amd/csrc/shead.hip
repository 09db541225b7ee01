// C-ABI entry points of the s-head: the sequence encoder (bidirectional LSTM, lstm.hpp)
// and the attention decoder (decoder.hpp); of the fused losses (8f row 2, loss.hpp) and of the
// clip + SGD step (optim.hpp).  Declared in include/mucon_hip.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mucon_hip.h"
#include "head_body.hpp"
#include "lstm.hpp"
#include "decoder.hpp"
#include "decoder_mw.hpp"
#include "loss.hpp"
#include "optim.hpp"
#include <vector>


void mucon_internal_set_error(const char *msg);  // mucon_hip.hip: feeds mucon_last_error()

static int sfail(int code, const char *fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mucon_internal_set_error(buf);
    return code;
}
#define SHIPCHK(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return sfail(MUCON_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

// workspace (floats): Gx / dG [ndir][T][512] (the input projections forward, the pre-activation gradients
// backward: same slot), gates [ndir][T][512], cells [ndir][T][128]
extern "C" size_t mucon_lstm_workspace_bytes(int32_t T, int32_t ndir) {
    if (T < 1 || ndir < 1 || ndir > 2) return 0;
    const size_t per = (size_t)ndir * T;
    return sizeof(float) * (al64(per * LSTM_G) * 2 + al64(per * LSTM_H));
}

static int lstm_check(int T, int I, int H, int ndir) {
    if (T < 1) return sfail(MUCON_E_ARG, "lstm: T=%d", T);
    if (I != LSTM_H || H != LSTM_H) return sfail(MUCON_E_ARG, "lstm: input %d / hidden %d unsupported (both must be %d)", I, H, LSTM_H);
    if (ndir < 1 || ndir > 2) return sfail(MUCON_E_ARG, "lstm: %d directions", ndir);
    return MUCON_OK;
}

static int fill_weights(LstmWeights &w, const mucon_lstm_params *p, int ndir) {
    for (int d = 0; d < ndir; ++d) {
        w.w_ih[d] = p->w_ih[d];
        w.w_hh[d] = p->w_hh[d];
        w.b_ih[d] = p->b_ih[d];
        w.b_hh[d] = p->b_hh[d];
        if (!w.w_ih[d] || !w.w_hh[d] || !w.b_ih[d] || !w.b_hh[d]) return sfail(MUCON_E_ARG, "lstm: null weight pointer (direction %d)", d);
    }
    for (int d = ndir; d < 2; ++d) {
        w.w_ih[d] = w.w_ih[0];
        w.w_hh[d] = w.w_hh[0];
        w.b_ih[d] = w.b_ih[0];
        w.b_hh[d] = w.b_hh[0];
    }
    return MUCON_OK;
}

extern "C" int mucon_lstm_fwd(int32_t T, int32_t I, int32_t H, int32_t ndir, const float *x, const mucon_lstm_params *params,
                              float *out, float *hn, float *cn, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = lstm_check(T, I, H, ndir);
    if (rc != MUCON_OK) return rc;
    if (!x || !params || !out || !hn || !cn || !workspace) return sfail(MUCON_E_ARG, "lstm: null pointer argument");
    if (workspace_bytes < mucon_lstm_workspace_bytes(T, ndir)) return sfail(MUCON_E_WORKSPACE, "lstm workspace too small");
    LstmWeights w;
    if ((rc = fill_weights(w, params, ndir)) != MUCON_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t per = (size_t)ndir * T;
    float *Gx = static_cast<float *>(workspace);
    float *gates = Gx + al64(per * LSTM_G);
    float *cells = gates + al64(per * LSTM_G);
    hipLaunchKernelGGL(lstm_inproj_kernel, dim3((T + LSTM_IP_T - 1) / LSTM_IP_T, ndir), dim3(512), 0, s, x, w, Gx, T);
    // (r6) a y-head forward left pending on this stream (mucon_head_fwd_defer) rides in extra workgroups of the recurrence launch: two of its 256-thread blocks each
    HeadFwdArgs ha;
    memset(&ha, 0, sizeof(ha));
    int hgx = 1, hblocks = 0;
    if (g_head_fwd_pending.pending && g_head_fwd_pending.stream == s) {
        ha = g_head_fwd_pending.a;
        hgx = g_head_fwd_pending.gx;
        hblocks = g_head_fwd_pending.gx * g_head_fwd_pending.gy;
        g_head_fwd_pending.pending = false;
    } else if (g_head_fwd_pending.pending) {
        if ((rc = head_fwd_flush()) != MUCON_OK) return rc;   // left on another stream: finished there
    }
    hipLaunchKernelGGL(lstm_recur_fwd_kernel, dim3(ndir + (hblocks + 1) / 2), dim3(512), 0, s, Gx, w, out, gates, cells, hn, cn, T, ndir, ha, hgx, hblocks);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

// One-shot option of the NEXT mucon_decoder_bwd call (mucon_decoder_bwd_defer), and what such a call leaves behind: the batch of outer products that are
// the decoder's weight gradients, to ride in extra workgroups of the next mucon_lstm_bwd's recurrence launch on the same stream -- or taken by
// mucon_decoder_bwd_flush (a launch of their own on the stream they were left on)
static struct OuterPending {
    bool armed = false, pending = false;
    OuterBatch ob;
    hipStream_t stream = nullptr;
} g_outer_pending;

extern "C" int mucon_decoder_bwd_defer(int32_t enable) {
    g_outer_pending.armed = enable != 0;
    return MUCON_OK;
}

extern "C" int mucon_decoder_bwd_flush(void) {
    OuterPending &op = g_outer_pending;
    if (!op.pending) return MUCON_OK;
    op.pending = false;
    if (op.ob.nblocks > 0) hipLaunchKernelGGL(dec_outer_kernel, dim3(op.ob.nblocks), dim3(256), 0, op.stream, op.ob);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

extern "C" int mucon_lstm_bwd(int32_t T, int32_t I, int32_t H, int32_t ndir, const float *x, const mucon_lstm_params *params,
                              const float *out, const float *d_out, const float *d_hn, const float *d_cn, float *d_x,
                              const float *d_x_add, const mucon_lstm_params *d_params, void *workspace, size_t workspace_bytes,
                              void *stream) {
    int rc = lstm_check(T, I, H, ndir);
    if (rc != MUCON_OK) return rc;
    if (!x || !params || !out || !d_x || !d_params || !workspace) return sfail(MUCON_E_ARG, "lstm: null pointer argument");
    if (workspace_bytes < mucon_lstm_workspace_bytes(T, ndir)) return sfail(MUCON_E_WORKSPACE, "lstm workspace too small");
    LstmWeights w;
    if ((rc = fill_weights(w, params, ndir)) != MUCON_OK) return rc;
    LstmGrads g;
    for (int d = 0; d < 2; ++d) {
        const int e = d < ndir ? d : 0;
        g.w_ih[d] = const_cast<float *>(d_params->w_ih[e]);
        g.w_hh[d] = const_cast<float *>(d_params->w_hh[e]);
        g.b_ih[d] = const_cast<float *>(d_params->b_ih[e]);
        g.b_hh[d] = const_cast<float *>(d_params->b_hh[e]);
        if (!g.w_ih[d] || !g.w_hh[d] || !g.b_ih[d] || !g.b_hh[d]) return sfail(MUCON_E_ARG, "lstm: null gradient pointer");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t per = (size_t)ndir * T;
    float *dG = static_cast<float *>(workspace);
    float *gates = dG + al64(per * LSTM_G);
    float *cells = gates + al64(per * LSTM_G);
    // (r6) a batch of the decoder's outer products left pending on this stream rides in extra workgroups of the recurrence launch (two CUs busy for ~89 us)
    OuterBatch ob;
    ob.njobs = 0;
    ob.nblocks = 0;
    int extra = 0;
    if (g_outer_pending.pending && g_outer_pending.stream == s) {
        ob = g_outer_pending.ob;
        extra = (ob.nblocks + 1) / 2;
        g_outer_pending.pending = false;
    } else if (g_outer_pending.pending) {
        if ((rc = mucon_decoder_bwd_flush()) != MUCON_OK) return rc;   // left on another stream: finished there
    }
    hipLaunchKernelGGL(lstm_recur_bwd_kernel, dim3(ndir + extra), dim3(512), 0, s, w, out, gates, cells, d_out, d_hn, d_cn, dG, T, ndir, ob);
    static bool wg_attr = false;
    if (!wg_attr) {
        SHIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_wgrad_dx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_WG_LDS_BYTES));
        wg_attr = true;
    }
    hipLaunchKernelGGL(lstm_wgrad_dx_kernel, dim3(T + (LSTM_G / 4) * ndir), dim3(512), LSTM_WG_LDS_BYTES, s, dG, x, out, g, w, d_x, d_x_add, T, ndir);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ decoder
constexpr int DEC_MAXTZ = 4096;   // 2 x 4 x Tz bytes of LDS for the attention weights; 65,536 frames at the default pooling

static int dec_check(const mucon_decoder_cfg *c) {
    if (!c) return sfail(MUCON_E_ARG, "decoder: null cfg");
    if (c->Tz < 1 || c->Tz > DEC_MAXTZ) return sfail(MUCON_E_ARG, "decoder: Tz=%d unsupported (1..%d)", c->Tz, DEC_MAXTZ);
    if (c->D != DEC_D) return sfail(MUCON_E_ARG, "decoder: width %d unsupported (must be %d)", c->D, DEC_D);
    if (c->ME < 1 || c->ME > DEC_MAXME) return sfail(MUCON_E_ARG, "decoder: memory width %d unsupported (max %d)", c->ME, DEC_MAXME);
    if (c->NC < 1 || c->NC > DEC_MAXNC) return sfail(MUCON_E_ARG, "decoder: %d output classes unsupported (max %d)", c->NC, DEC_MAXNC);
    if (c->max_steps < 1) return sfail(MUCON_E_ARG, "decoder: max_steps=%d", c->max_steps);
    if (c->n_emb < 1) return sfail(MUCON_E_ARG, "decoder: n_emb=%d", c->n_emb);
    return MUCON_OK;
}

struct DecLayout {
    DecSaved sv;
    DecDeltas dl;
    unsigned long long *xbuf;   // exchange granules of decoder_fwd_mw_kernel
    size_t floats;
};
int g_dec_mw = 1;   // MUCON_DEC_MW=0: the decoder's forward step loop on one workgroup (decoder_fwd_kernel) also where the eight-workgroup kernel applies (tests, A/B)
static DecLayout dec_layout(const mucon_decoder_cfg *c, float *base) {
    DecLayout L;
    size_t off = 0;
    auto take = [&](size_t n) {
        float *p = base ? base + off : nullptr;
        off += al64(n);
        return p;
    };
    const size_t S = c->max_steps, Tz = c->Tz, CW = DEC_D + c->ME, LW = DEC_D + c->NC;
    L.sv.mp = take(Tz * DEC_D);
    L.sv.h = take((S + 1) * DEC_D);
    L.sv.c = take((S + 1) * DEC_D);
    L.sv.q = take(S * DEC_D);
    L.sv.cat = take(S * CW);
    L.sv.attn = take(S * Tz);
    L.sv.mixed = take(S * DEC_D);
    L.sv.gates = take(S * 4 * DEC_D);
    L.sv.t1 = take(S * DEC_D);
    L.sv.lencat = take(S * LW);
    L.sv.l1 = take(S * DEC_NL);
    L.sv.toks = reinterpret_cast<int *>(take(S));
    L.dl.ctx = take(S * c->ME);
    L.dl.score = take(S * Tz);
    L.dl.mp = take(Tz * DEC_D);
    L.dl.q = take(S * DEC_D);
    L.dl.mixed = take(S * DEC_D);
    L.dl.gates = take(S * 4 * DEC_D);
    L.dl.t1 = take(S * DEC_D);
    L.dl.logits = take(S * c->NC);
    L.dl.l1 = take(S * DEC_NL);
    L.dl.len = take(S);
    L.dl.h0 = take(DEC_D);
    L.dl.c0 = take(DEC_D);
    L.xbuf = reinterpret_cast<unsigned long long *>(take(2 * 2 * MW_X_WORDS));   // exchange granules of the eight-workgroup step kernels (8 bytes each; the backward's set is the larger, + its failure word)
    L.floats = off;
    return L;
}

extern "C" size_t mucon_decoder_workspace_bytes(const mucon_decoder_cfg *cfg) {
    if (dec_check(cfg) != MUCON_OK) return 0;
    return dec_layout(cfg, nullptr).floats * sizeof(float);
}

static int dec_params(DecParams &p, const mucon_decoder_params *q, const char *what) {
    const float *const *src = reinterpret_cast<const float *const *>(q);
    const float **dst = reinterpret_cast<const float **>(&p);
    static_assert(sizeof(DecParams) == DEC_NPARAMS * sizeof(float *), "DecParams layout");
    static_assert(sizeof(mucon_decoder_params) == DEC_NPARAMS * sizeof(float *), "mucon_decoder_params layout");
    for (int i = 0; i < DEC_NPARAMS; ++i) {
        if (!src[i]) return sfail(MUCON_E_ARG, "decoder: null %s pointer (#%d)", what, i);
        dst[i] = src[i];
    }
    return MUCON_OK;
}

// the step kernels keep attention_l2's weight (and the forward, when it fits, the attention projection) in dynamic LDS: more
// than the 64 KB a kernel gets by default
static int dec_lds_attr() {
    static int done_dev = -1;      // (the > 64 KB opt-ins are per device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return sfail(MUCON_E_HIP, "decoder: hipGetDevice failed");
    if (done_dev == dev) return MUCON_OK;
    const bool bad = hipFuncSetAttribute(reinterpret_cast<const void *>(decoder_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         DEC_MAX_DYN_LDS) != hipSuccess ||
                     hipFuncSetAttribute(reinterpret_cast<const void *>(decoder_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         DEC_MAX_DYN_LDS_BWD) != hipSuccess ||
                     hipFuncSetAttribute(reinterpret_cast<const void *>(decoder_fwd_mw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)mw_fwd_lds_bytes(MW_TZ)) != hipSuccess ||
                     hipFuncSetAttribute(reinterpret_cast<const void *>(decoder_bwd_mw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)mw_bwd_lds_bytes(MW_TZ)) != hipSuccess;
    if (bad) return sfail(MUCON_E_HIP, "decoder: hipFuncSetAttribute failed");
    done_dev = dev;
    return MUCON_OK;
}

static DecDims dec_dims(const mucon_decoder_cfg *c, int S) {
    DecDims d;
    d.Tz = c->Tz;
    d.ME = c->ME;
    d.NC = c->NC;
    d.S = S;
    d.n_emb = c->n_emb;
    d.mp_lds = 0;
    d.teacher_forcing = c->teacher_forcing;
    d.stop_on_eos = c->stop_on_eos;
    d.eos = c->eos;
    return d;
}

extern "C" int mucon_decoder_fwd(const mucon_decoder_cfg *cfg, const mucon_decoder_params *params, const float *memory,
                                 const float *hn, const float *cn, const int64_t *tf_input, const float *dropmask,
                                 float *logp, float *lengths, int32_t *n_steps, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    int rc = dec_check(cfg);
    if (rc != MUCON_OK) return rc;
    if (!params || !memory || !hn || !cn || !tf_input || !logp || !lengths || !n_steps || !workspace)
        return sfail(MUCON_E_ARG, "decoder: null pointer argument");
    if (workspace_bytes < mucon_decoder_workspace_bytes(cfg)) return sfail(MUCON_E_WORKSPACE, "decoder workspace too small");
    DecParams p;
    if ((rc = dec_params(p, params, "parameter")) != MUCON_OK) return rc;
    if ((rc = dec_lds_attr()) != MUCON_OK) return rc;
    const DecLayout L = dec_layout(cfg, static_cast<float *>(workspace));
    hipStream_t s = static_cast<hipStream_t>(stream);
    DecDims dm = dec_dims(cfg, cfg->max_steps);
    if (g_dec_mw && cfg->ME == DEC_MAXME && cfg->Tz <= MW_TZ) {
        // eight workgroups, every per-step operand resident in their LDS, three exchanges per step (decoder_mw.hpp); the memory
        // projection launch in front clears the exchange granules and the step counter
        hipLaunchKernelGGL(dec_memproj_kernel, dim3((cfg->Tz + 3) / 4), dim3(512), 0, s, memory, p.w1, L.sv.mp, cfg->Tz, cfg->ME, L.xbuf,
                           (int)MW_X_WORDS, n_steps);
        hipLaunchKernelGGL(decoder_fwd_mw_kernel, dim3(MW_G), dim3(MW_T), mw_fwd_lds_bytes(cfg->Tz), s, dm, p, L.sv, memory, hn, cn,
                           reinterpret_cast<const long *>(tf_input), dropmask, logp, L.xbuf, n_steps);
        const int greedy = !(cfg->teacher_forcing && !cfg->stop_on_eos);
        hipLaunchKernelGGL(decoder_heads_kernel, dim3(cfg->max_steps), dim3(DEC_THREADS), 0, s, dm, p, L.sv, logp, lengths, n_steps, greedy);
        SHIPCHK(hipGetLastError());
        return MUCON_OK;
    }
    hipLaunchKernelGGL(dec_memproj_kernel, dim3((cfg->Tz + 3) / 4), dim3(512), 0, s, memory, p.w1, L.sv.mp, cfg->Tz, cfg->ME, nullptr, 0, nullptr);
    const size_t lds = dec_fwd_lds_bytes(cfg->Tz, &dm.mp_lds);
    hipLaunchKernelGGL(decoder_fwd_kernel, dim3(1), dim3(DEC_THREADS), lds, s, dm, p,
                       L.sv, memory, hn, cn, reinterpret_cast<const long *>(tf_input), dropmask, logp, lengths, n_steps);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

extern "C" int mucon_decoder_bwd(const mucon_decoder_cfg *cfg, int32_t n_steps, const mucon_decoder_params *params,
                                 const float *memory, const float *hn, const float *cn, const float *logp, const float *d_logp,
                                 const float *d_lengths, const float *dropmask, float *d_memory, float *d_hn, float *d_cn,
                                 const mucon_decoder_params *d_params, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = dec_check(cfg);
    if (rc != MUCON_OK) return rc;
    if (n_steps < 1 || n_steps > cfg->max_steps) return sfail(MUCON_E_ARG, "decoder: n_steps=%d outside 1..%d", n_steps, cfg->max_steps);
    if (!params || !memory || !hn || !cn || !logp || !d_memory || !d_hn || !d_cn || !d_params || !workspace)
        return sfail(MUCON_E_ARG, "decoder: null pointer argument");
    if (workspace_bytes < mucon_decoder_workspace_bytes(cfg)) return sfail(MUCON_E_WORKSPACE, "decoder workspace too small");
    if (g_outer_pending.pending && (rc = mucon_decoder_bwd_flush()) != MUCON_OK) return rc;   // an earlier deferred batch nobody took
    DecParams p, g;
    if ((rc = dec_params(p, params, "parameter")) != MUCON_OK) return rc;
    if ((rc = dec_params(g, d_params, "gradient")) != MUCON_OK) return rc;
    if ((rc = dec_lds_attr()) != MUCON_OK) return rc;
    auto W = [](const float *q) { return const_cast<float *>(q); };
    const DecLayout L = dec_layout(cfg, static_cast<float *>(workspace));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int S = n_steps, Tz = cfg->Tz, ME = cfg->ME, NC = cfg->NC, CW = DEC_D + ME, LW = DEC_D + NC;
    if (g_dec_mw && cfg->ME == DEC_MAXME && Tz <= MW_TZ) {
        // the heads for all steps (one workgroup; clears the exchange granules), then the step loop on eight workgroups (decoder_mw.hpp)
        hipLaunchKernelGGL(decoder_heads_bwd_kernel, dim3(S), dim3(DEC_THREADS), 0, s, dec_dims(cfg, S), p, L.sv, L.dl, logp, d_logp, d_lengths,
                           W(g.emb), L.xbuf, (int)MWB_X_WORDS);
        // (r6) a y-head backward kernel left pending on this stream (mucon_head_bwd_defer, bit 1) rides in extra workgroups of the step loop's launch: same 256 threads
        HeadBwdArgs ha;
        memset(&ha, 0, sizeof(ha));
        int hgx = 1, hblocks = 0;
        if (g_head_kernel_pending.pending && g_head_kernel_pending.stream == s) {
            ha = g_head_kernel_pending.a;
            hgx = g_head_kernel_pending.gx;
            hblocks = g_head_kernel_pending.gx * g_head_kernel_pending.gy;
            g_head_kernel_pending.pending = false;
        } else if (g_head_kernel_pending.pending) {
            if ((rc = head_kernel_flush()) != MUCON_OK) return rc;
        }
        hipLaunchKernelGGL(decoder_bwd_mw_kernel, dim3(MW_G + hblocks), dim3(MW_T), mw_bwd_lds_bytes(Tz), s, dec_dims(cfg, S), p, L.sv, L.dl, memory,
                           dropmask, W(g.emb), W(g.v), d_hn, d_cn, L.xbuf, ha, hgx, hblocks);
    } else {
        if (g_head_kernel_pending.pending && (rc = head_kernel_flush()) != MUCON_OK) return rc;   // (the one-workgroup kernel has 1,024 threads: the pending kernel runs on its own)
        hipLaunchKernelGGL(decoder_bwd_kernel, dim3(1), dim3(DEC_THREADS), dec_bwd_lds_bytes(Tz), s, dec_dims(cfg, S), p, L.sv, L.dl, memory,
                           logp, d_logp, d_lengths, dropmask, d_memory, W(g.emb), W(g.v), d_hn, d_cn);
    }
    OuterBatch ob;
    int nj = 0, blocks = 0;
    auto job = [&](const float *A, int lda, int ra, const float *B, int ldb, int cb, int n, const float *out, const float *bias,
                   const float *bias2) {
        OuterJob &j = ob.job[nj++];
        j.A = A;
        j.B = B;
        j.out = W(out);
        j.bias = W(bias);
        j.bias2 = W(bias2);
        j.lda = lda;
        j.ldb = ldb;
        j.ra = ra;
        j.cb = cb;
        j.n = n;
        j.block0 = blocks;
        blocks += (ra * cb + 255) / 256;
    };
    job(L.dl.q, DEC_D, DEC_D, L.sv.h, DEC_D, DEC_D, S, g.l2_w, g.l2_b, nullptr);
    job(L.dl.mixed, DEC_D, DEC_D, L.sv.cat, CW, CW, S, g.cmb_w, g.cmb_b, nullptr);
    job(L.dl.gates, 4 * DEC_D, 4 * DEC_D, L.sv.mixed, DEC_D, DEC_D, S, g.w_ih, g.b_ih, g.b_hh);
    job(L.dl.gates, 4 * DEC_D, 4 * DEC_D, L.sv.h, DEC_D, DEC_D, S, g.w_hh, nullptr, nullptr);
    job(L.dl.t1, DEC_D, DEC_D, L.sv.h + DEC_D, DEC_D, DEC_D, S, g.t1_w, g.t1_b, nullptr);
    job(L.dl.logits, NC, NC, L.sv.t1, DEC_D, DEC_D, S, g.t2_w, g.t2_b, nullptr);
    job(L.dl.l1, DEC_NL, DEC_NL, L.sv.lencat, LW, LW, S, g.n1_w, g.n1_b, nullptr);
    job(L.dl.len, 1, 1, L.sv.l1, DEC_NL, DEC_NL, S, g.n2_w, g.n2_b, nullptr);
    job(L.dl.h0, DEC_D, DEC_D, hn, ME, ME, 1, g.ho_w, g.ho_b, nullptr);
    job(L.dl.c0, DEC_D, DEC_D, cn, ME, ME, 1, g.co_w, g.co_b, nullptr);
    job(memory, ME, ME, L.dl.mp, DEC_D, DEC_D, Tz, g.w1, nullptr, nullptr);
    ob.njobs = nj;
    ob.nblocks = blocks;
    hipLaunchKernelGGL(dec_attn_grad_kernel, dim3(Tz), dim3(256), 0, s, L.sv, L.dl, p.w1, p.v, d_memory, S, Tz, ME);   // before dW1's job
    const bool defer = g_outer_pending.armed;
    g_outer_pending.armed = false;
    if (defer) {   // (mucon_decoder_bwd_defer) the weight gradients are left to the next mucon_lstm_bwd on this stream
        g_outer_pending.ob = ob;
        g_outer_pending.stream = s;
        g_outer_pending.pending = true;
        SHIPCHK(hipGetLastError());
        return MUCON_OK;
    }
    hipLaunchKernelGGL(dec_outer_kernel, dim3(blocks), dim3(256), 0, s, ob);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ losses
static int loss_check(const mucon_loss_cfg *c) {
    if (!c) return sfail(MUCON_E_ARG, "loss: null cfg");
    if (c->T < 2) return sfail(MUCON_E_ARG, "loss: T=%d (at least 2 frames)", c->T);
    if (c->M < 1 || c->M > LOSS_MAXM) return sfail(MUCON_E_ARG, "loss: %d classes unsupported (max %d)", c->M, LOSS_MAXM);
    if (c->N < 1 || c->N > LOSS_MAXN) return sfail(MUCON_E_ARG, "loss: %d segments unsupported (1..%d)", c->N, LOSS_MAXN);
    if (c->S < 1 || c->NC < 1) return sfail(MUCON_E_ARG, "loss: S=%d NC=%d", c->S, c->NC);
    if (c->mucon_type != 0 && c->mucon_type != 1) return sfail(MUCON_E_ARG, "loss: mucon_type %d (0 flint, 1 arithmetic)", c->mucon_type);
    return MUCON_OK;
}

static size_t loss_layout(const mucon_loss_cfg *c, float *base, LossBufs *b) {
    size_t off = 0;
    auto take = [&](size_t n) {
        float *p = base ? base + off : nullptr;
        off += al64(n);
        return p;
    };
    const size_t chunks = (c->T + LOSS_FB - 1) / LOSS_FB, NM = (size_t)c->N * c->M;
    float *geo = take(6 * LOSS_MAXN), *small = take(8), *slab = take(chunks * (NM + 2)), *gwin = take(NM);
    float *glwin = take(LOSS_MAXN), *gsm = take(1), *gslab = take(chunks * c->N * 2), *geod = take(4 * LOSS_MAXN);
    if (b) {
        b->geo = geo;
        b->geod = reinterpret_cast<double *>(geod);
        b->small = small;
        b->slab = slab;
        b->gwin = gwin;
        b->glwin = glwin;
        b->gsm = gsm;
        b->gslab = gslab;
    }
    return off;
}

extern "C" size_t mucon_loss_workspace_bytes(const mucon_loss_cfg *cfg) {
    if (loss_check(cfg) != MUCON_OK) return 0;
    return loss_layout(cfg, nullptr, nullptr) * sizeof(float);
}

extern "C" int mucon_loss_fwd_bwd(const mucon_loss_cfg *cfg, const float *segmentation, const float *smoothing_input,
                                  const float *transcript_logp, const float *lengths, const int64_t *mucon_target,
                                  const int64_t *transcript_target, const float *mask_template, const float *mucon_class_weight,
                                  const float *transcript_class_weight, float *losses, float *d_segmentation,
                                  float *d_smoothing_input, float *d_transcript_logp, float *d_lengths, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    int rc = loss_check(cfg);
    if (rc != MUCON_OK) return rc;
    if (!segmentation || !smoothing_input || !transcript_logp || !lengths || !mucon_target || !transcript_target || !mask_template ||
        !losses || !d_segmentation || !d_smoothing_input || !d_transcript_logp || !d_lengths || !workspace)
        return sfail(MUCON_E_ARG, "loss: null pointer argument");
    if (workspace_bytes < mucon_loss_workspace_bytes(cfg)) return sfail(MUCON_E_WORKSPACE, "loss workspace too small");
    LossDims d;
    d.T = cfg->T;
    d.M = cfg->M;
    d.N = cfg->N;
    d.S = cfg->S;
    d.NC = cfg->NC;
    d.mucon_type = cfg->mucon_type;
    d.smoothing_clamp = cfg->smoothing_clamp;
    d.transcript_average = cfg->transcript_average;
    d.overlap = cfg->overlap;
    d.clamp_min = cfg->clamp_min;
    d.clamp_max = cfg->clamp_max;
    d.length_width = cfg->length_width;
    d.mul_transcript = cfg->mul_transcript;
    d.mul_length = cfg->mul_length;
    d.mul_mucon = cfg->mul_mucon;
    d.mul_smoothing = cfg->mul_smoothing;
    d.align_corners = cfg->align_corners ? 1 : 0;
    LossBufs b;
    loss_layout(cfg, static_cast<float *>(workspace), &b);
    b.seg = segmentation;
    b.sx = smoothing_input;
    b.tlogp = transcript_logp;
    b.lengths = lengths;
    b.mtarget = reinterpret_cast<const long *>(mucon_target);
    b.ttarget = reinterpret_cast<const long *>(transcript_target);
    b.tmpl = mask_template;
    b.mweight = mucon_class_weight;
    b.tweight = transcript_class_weight;
    b.losses = losses;
    b.d_seg = d_segmentation;
    b.d_sx = d_smoothing_input;
    b.d_tlogp = d_transcript_logp;
    b.d_lengths = d_lengths;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_head_fwd_pending.pending) {   // (a deferred y-head forward nobody took: the losses read its outputs)
        const int frc = head_fwd_flush();
        if (frc != MUCON_OK) return frc;
    }
    const int chunks = (cfg->T + LOSS_FB - 1) / LOSS_FB;
    hipLaunchKernelGGL(loss_acc_kernel, dim3(chunks), dim3(256), 0, s, d, b);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(chunks), dim3(256), 0, s, d, b, chunks);   // (r6: every workgroup does the step between the two passes for itself, loss_mid_body)
    hipLaunchKernelGGL(loss_fin_kernel, dim3(1), dim3(64), 0, s, d, b, chunks);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ clip + SGD
extern "C" size_t mucon_sgd_workspace_bytes(int32_t n_tensors, int64_t total_elements) {
    if (n_tensors < 1 || total_elements < 1) return 0;
    const size_t blocks = (size_t)(total_elements / SGD_CHUNK) + n_tensors;   // upper bound on the chunk count
    return al64(sizeof(SgdTensor) * n_tensors) + sizeof(float) * al64(blocks);
}

extern "C" int mucon_sgd_clip_step(int32_t n_tensors, const mucon_sgd_tensor *tensors, int32_t n_groups, const float *max_norm,
                                   float lr, float weight_decay, float momentum, float *group_norms, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (n_tensors < 1 || !tensors || !workspace) return sfail(MUCON_E_ARG, "sgd: no tensors");
    if (n_groups < 1 || n_groups > SGD_MAXGROUPS || !max_norm) return sfail(MUCON_E_ARG, "sgd: %d clipping groups (1..%d)", n_groups, SGD_MAXGROUPS);
    std::vector<SgdTensor> tab(n_tensors);
    long total = 0;
    int blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const mucon_sgd_tensor &t = tensors[i];
        if (!t.param || !t.grad || t.n < 1) return sfail(MUCON_E_ARG, "sgd: tensor %d: null pointer or empty", i);
        if (t.group < 0 || t.group >= n_groups) return sfail(MUCON_E_ARG, "sgd: tensor %d: group %d outside 0..%d", i, t.group, n_groups - 1);
        if (momentum != 0.f && !t.momentum_buf) return sfail(MUCON_E_ARG, "sgd: tensor %d: momentum %g needs a momentum buffer", i, momentum);
        tab[i].p = t.param;
        tab[i].g = t.grad;
        tab[i].mom = momentum != 0.f ? t.momentum_buf : nullptr;
        tab[i].n = t.n;
        tab[i].group = t.group;
        tab[i].block0 = blocks;
        blocks += (int)((t.n + SGD_CHUNK - 1) / SGD_CHUNK);
        total += t.n;
    }
    if (workspace_bytes < mucon_sgd_workspace_bytes(n_tensors, total)) return sfail(MUCON_E_WORKSPACE, "sgd workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    SgdTensor *dtab = static_cast<SgdTensor *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + al64(sizeof(SgdTensor) * n_tensors));
    // The table is the same step after step (torch's caching allocator hands the same gradient buffers back).  Uploading it
    // puts a 5 us copy on the stream in front of every optimizer step, so a copy is kept in a buffer the LIBRARY owns (the
    // caller cannot overwrite it) and reused while table, device and stream are the ones it was filled for.  Another stream
    // or device never touches that buffer: it takes the upload into the caller's workspace, as every call did before.
    // Two slots: a training loop that still holds step i-1's gradients while step i's are allocated alternates between two
    // sets of buffers, i.e. between two tables.
    constexpr size_t CACHE_TENSORS = 1024;
    constexpr int SLOTS = 2;
    static std::vector<SgdTensor> shadow[SLOTS];
    static SgdTensor *cache_dev[SLOTS] = {nullptr, nullptr};
    static hipStream_t cache_stream = nullptr;
    static int cache_device = -1, next_slot = 0;
    static bool cache_owned = false;
    int device = -1;
    SHIPCHK(hipGetDevice(&device));
    const SgdTensor *use_tab = dtab;
    const bool cacheable = (size_t)n_tensors <= CACHE_TENSORS && (!cache_owned || (cache_device == device && cache_stream == s));
    if (cacheable) {
        int hit = -1;
        for (int k = 0; k < SLOTS && cache_owned; ++k)
            if (shadow[k].size() == tab.size() && memcmp(shadow[k].data(), tab.data(), sizeof(SgdTensor) * n_tensors) == 0) hit = k;
        if (hit < 0) {
            hit = next_slot;
            next_slot = (next_slot + 1) % SLOTS;
            if (!cache_dev[hit]) SHIPCHK(hipMalloc(&cache_dev[hit], sizeof(SgdTensor) * CACHE_TENSORS));
            shadow[hit] = tab;   // the source of the async copy must outlive it: the shadow does
            SHIPCHK(hipMemcpyAsync(cache_dev[hit], shadow[hit].data(), sizeof(SgdTensor) * n_tensors, hipMemcpyHostToDevice, s));
            cache_owned = true;
            cache_device = device;
            cache_stream = s;
        }
        use_tab = cache_dev[hit];
    } else {
        SHIPCHK(hipMemcpyAsync(dtab, tab.data(), sizeof(SgdTensor) * n_tensors, hipMemcpyHostToDevice, s));
    }
    SgdHyper h;
    for (int g = 0; g < SGD_MAXGROUPS; ++g) h.max_norm[g] = g < n_groups ? max_norm[g] : 0.f;
    h.lr = lr;
    h.weight_decay = weight_decay;
    h.momentum = momentum;
    h.ngroups = n_groups;
    h.nblocks = blocks;
    h.any_clip = 0;
    for (int g = 0; g < n_groups; ++g) h.any_clip |= max_norm[g] > 0.f;
    if (h.any_clip) hipLaunchKernelGGL(sgd_norm_kernel<SgdTensor>, dim3(blocks), dim3(256), 0, s, use_tab, n_tensors, partial);
    hipLaunchKernelGGL(sgd_apply_kernel, dim3(blocks), dim3(256), 0, s, use_tab, n_tensors, partial, h, group_norms);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ clip alone
extern "C" int mucon_clip_grads(int32_t n_tensors, const mucon_sgd_tensor *tensors, int32_t n_groups, const float *max_norm,
                                float *group_norms, void *workspace, size_t workspace_bytes, void *stream) {
    if (n_tensors < 1 || !tensors || !workspace) return sfail(MUCON_E_ARG, "clip: no tensors");
    if (n_groups < 1 || n_groups > SGD_MAXGROUPS || !max_norm) return sfail(MUCON_E_ARG, "clip: %d clipping groups (1..%d)", n_groups, SGD_MAXGROUPS);
    std::vector<SgdTensor> tab(n_tensors);
    long total = 0;
    int blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const mucon_sgd_tensor &t = tensors[i];
        if (!t.grad || t.n < 1) return sfail(MUCON_E_ARG, "clip: tensor %d: null pointer or empty", i);
        if (t.group < 0 || t.group >= n_groups) return sfail(MUCON_E_ARG, "clip: tensor %d: group %d outside 0..%d", i, t.group, n_groups - 1);
        tab[i].p = t.param;
        tab[i].g = t.grad;
        tab[i].mom = nullptr;
        tab[i].n = t.n;
        tab[i].group = t.group;
        tab[i].block0 = blocks;
        blocks += (int)((t.n + SGD_CHUNK - 1) / SGD_CHUNK);
        total += t.n;
    }
    if (workspace_bytes < mucon_sgd_workspace_bytes(n_tensors, total)) return sfail(MUCON_E_WORKSPACE, "clip workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    SgdTensor *dtab = static_cast<SgdTensor *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + al64(sizeof(SgdTensor) * n_tensors));
    SHIPCHK(hipMemcpyAsync(dtab, tab.data(), sizeof(SgdTensor) * n_tensors, hipMemcpyHostToDevice, s));
    SgdHyper h;
    memset(&h, 0, sizeof(h));
    for (int g = 0; g < SGD_MAXGROUPS; ++g) h.max_norm[g] = g < n_groups ? max_norm[g] : 0.f;
    h.ngroups = n_groups;
    h.nblocks = blocks;
    hipLaunchKernelGGL(sgd_norm_kernel<SgdTensor>, dim3(blocks), dim3(256), 0, s, dtab, n_tensors, partial);
    hipLaunchKernelGGL(clip_apply_kernel, dim3(blocks), dim3(256), 0, s, dtab, n_tensors, partial, h, group_norms);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

// ------------------------------------------------------------------------------------------ clip + Adam
extern "C" size_t mucon_adam_workspace_bytes(int32_t n_tensors, int64_t total_elements) {
    if (n_tensors < 1 || total_elements < 1) return 0;
    const size_t blocks = (size_t)(total_elements / SGD_CHUNK) + n_tensors;
    return al64(sizeof(AdamTensor) * n_tensors) + sizeof(float) * al64(blocks);
}

extern "C" int mucon_adam_clip_step(int32_t n_tensors, const mucon_adam_tensor *tensors, int32_t n_groups, const float *max_norm,
                                    double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                                    float *group_norms, void *workspace, size_t workspace_bytes, void *stream) {
    if (n_tensors < 1 || !tensors || !workspace) return sfail(MUCON_E_ARG, "adam: no tensors");
    if (n_groups < 1 || n_groups > SGD_MAXGROUPS || !max_norm) return sfail(MUCON_E_ARG, "adam: %d clipping groups (1..%d)", n_groups, SGD_MAXGROUPS);
    if (step < 1 || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1)) return sfail(MUCON_E_ARG, "adam: step %lld, betas %g %g", (long long)step, beta1, beta2);
    std::vector<AdamTensor> tab(n_tensors);
    long total = 0;
    int blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const mucon_adam_tensor &t = tensors[i];
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || t.n < 1) return sfail(MUCON_E_ARG, "adam: tensor %d: null pointer or empty", i);
        if (t.group < 0 || t.group >= n_groups) return sfail(MUCON_E_ARG, "adam: tensor %d: group %d outside 0..%d", i, t.group, n_groups - 1);
        tab[i].p = t.param;
        tab[i].g = t.grad;
        tab[i].m = t.exp_avg;
        tab[i].v = t.exp_avg_sq;
        tab[i].vmax = t.max_exp_avg_sq;
        tab[i].n = t.n;
        tab[i].group = t.group;
        tab[i].block0 = blocks;
        blocks += (int)((t.n + SGD_CHUNK - 1) / SGD_CHUNK);
        total += t.n;
    }
    if (workspace_bytes < mucon_adam_workspace_bytes(n_tensors, total)) return sfail(MUCON_E_WORKSPACE, "adam workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    AdamTensor *dtab = static_cast<AdamTensor *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + al64(sizeof(AdamTensor) * n_tensors));
    // (the table is uploaded on every call: a pageable source is staged by the runtime before the call returns)
    SHIPCHK(hipMemcpyAsync(dtab, tab.data(), sizeof(AdamTensor) * n_tensors, hipMemcpyHostToDevice, s));
    AdamHyper h;
    for (int g = 0; g < SGD_MAXGROUPS; ++g) h.max_norm[g] = g < n_groups ? max_norm[g] : 0.f;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    h.step_size = (float)(lr / bc1);
    h.one_minus_beta1 = (float)(1.0 - beta1);
    h.beta2 = (float)beta2;
    h.one_minus_beta2 = (float)(1.0 - beta2);
    h.sqrt_bc2 = (float)sqrt(bc2);
    h.eps = (float)eps;
    h.weight_decay = (float)weight_decay;
    h.ngroups = n_groups;
    h.nblocks = blocks;
    h.any_clip = 0;
    for (int g = 0; g < n_groups; ++g) h.any_clip |= max_norm[g] > 0.f;
    if (h.any_clip) hipLaunchKernelGGL(sgd_norm_kernel<AdamTensor>, dim3(blocks), dim3(256), 0, s, dtab, n_tensors, partial);
    hipLaunchKernelGGL(adam_apply_kernel, dim3(blocks), dim3(256), 0, s, dtab, n_tensors, partial, h, group_norms);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}
