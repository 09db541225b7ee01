// C-ABI entry points of the s-head's sequence encoder (bidirectional LSTM), SURVEY.md 8f row 1.
// Kernels in lstm.hpp; declared in include/mucon_hip.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mucon_hip.h"
#include "lstm.hpp"

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
    hipLaunchKernelGGL(lstm_inproj_kernel, dim3((T + 7) / 8, ndir), dim3(512), 0, s, x, w, Gx, T);
    hipLaunchKernelGGL(lstm_recur_fwd_kernel, dim3(ndir), dim3(512), 0, s, Gx, w, out, gates, cells, hn, cn, T, ndir);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}

extern "C" int mucon_lstm_bwd(int32_t T, int32_t I, int32_t H, int32_t ndir, const float *x, const mucon_lstm_params *params,
                              const float *out, const float *d_out, const float *d_hn, const float *d_cn, float *d_x,
                              const mucon_lstm_params *d_params, void *workspace, size_t workspace_bytes, void *stream) {
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
    hipLaunchKernelGGL(lstm_recur_bwd_kernel, dim3(ndir), dim3(512), 0, s, w, out, gates, cells, d_out, d_hn, d_cn, dG, T, ndir);
    hipLaunchKernelGGL(lstm_wgrad_kernel, dim3(LSTM_G / 4, ndir), dim3(512), 0, s, dG, x, out, g, T, ndir);
    hipLaunchKernelGGL(lstm_dx_kernel, dim3(T), dim3(128), 0, s, dG, w, d_x, T, ndir);
    SHIPCHK(hipGetLastError());
    return MUCON_OK;
}
