// Frame accuracy and segment-overlap scores on the device -- the counters of the reference's MoFAccuracyMetric / IoDMetric /
// IoUMetric (src/core/metrics/segmentation.py:16-91, which calls isba_code.py:23-109) for labellings that already live in HBM
// (the y-head's arg max, the Viterbi labels): only a few numbers per video travel to the host instead of T labels.
//
// One 256-thread workgroup per video:
//   1. frame accuracy: frames whose target is not ignored / of those, the ones where target == prediction (integer sums);
//   2. run-length encoding of both labellings (a run starts where the label changes; ballot + popcount scan, 256 frames a round);
//   3. thread i = target run i: the best score over the predicted runs with the same label,
//        IoD  intersection / predicted run's length          IoU  intersection / span of the union
//      as float64 quotients of integers -- single IEEE divisions, so the host metric classes reproduce their NumPy results bit for
//      bit: the kernel hands back the per-run maxima (and the run labels), the host applies `ignore`, max(., 0) and the mean.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mucon_hip.h"

void mucon_internal_set_error(const char *msg);

namespace {
constexpr int MT_THREADS = 256;
constexpr int MT_MAX_RUNS = MUCON_METRICS_MAX_RUNS;

// starts[] / labels[] of the runs of y[0..T): returns the number of runs (capped: further runs are dropped, the caller checks)
__device__ int encode_runs(const int32_t *y, int T, int *starts, int *labels, int *wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int running = 0;
    for (int base = 0; base < T; base += MT_THREADS) {
        const int t = base + tid;
        const bool flag = t < T && (t == 0 || y[t] != y[t - 1]);
        const unsigned long long m = __ballot(flag);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        const int tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        if (flag && off + before < MT_MAX_RUNS) {
            starts[off + before] = t;
            labels[off + before] = y[t];
        }
        running += tot;
        __syncthreads();
    }
    return running;
}

__global__ __launch_bounds__(MT_THREADS) void metrics_overlap_kernel(const int64_t *offsets, const int32_t *targets, const int32_t *predictions,
                                                                     const int32_t *ignore, int n_ignore, int64_t *mof, int32_t *n_runs,
                                                                     int32_t *run_label, double *iod, double *iou) {
    __shared__ int ts[MT_MAX_RUNS], tl[MT_MAX_RUNS], ps[MT_MAX_RUNS], pl[MT_MAX_RUNS];
    __shared__ int wave_tot[4];
    __shared__ long long red[2][4];
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long o = offsets[v];
    const int T = (int)(offsets[v + 1] - o);
    const int32_t *tg = targets + o, *pr = predictions + o;
    auto ignored = [&](int label) {
        bool ig = false;
        for (int i = 0; i < n_ignore; ++i) ig |= label == ignore[i];
        return ig;
    };
    // 1. frame accuracy
    long long tot = 0, cor = 0;
    for (int t = tid; t < T; t += MT_THREADS) {
        const int a = tg[t];
        const bool keep = !ignored(a);
        tot += keep;
        cor += keep && a == pr[t];
    }
    for (int s = 32; s >= 1; s >>= 1) {
        tot += __shfl_xor(tot, s);
        cor += __shfl_xor(cor, s);
    }
    if (lane == 0) {
        red[0][wave] = cor;
        red[1][wave] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        mof[2 * v] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        mof[2 * v + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
    // 2. runs
    const int nt = encode_runs(tg, T, ts, tl, wave_tot);
    const int np = encode_runs(pr, T, ps, pl, wave_tot);
    if (tid == 0) {
        n_runs[3 * v] = nt;
        n_runs[3 * v + 1] = np;
    }
    if (nt > MT_MAX_RUNS || np > MT_MAX_RUNS) return;   // the host sees the counts and takes its own path
    {   // predicted runs whose label is not ignored (none left: the score is 0 by definition)
        int kept = 0;
        for (int j = tid; j < np; j += MT_THREADS) kept += !ignored(pl[j]);
        for (int s = 32; s >= 1; s >>= 1) kept += __shfl_xor(kept, s);
        __syncthreads();
        if (lane == 0) wave_tot[wave] = kept;
        __syncthreads();
        if (tid == 0) n_runs[3 * v + 2] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    }
    // 3. best same-label overlap per target run (predicted runs with an ignored label never match a kept target run)
    for (int i = tid; i < nt; i += MT_THREADS) {
        const long long s0 = ts[i], e0 = i + 1 < nt ? ts[i + 1] : T;
        const int lab = tl[i];
        double bd = -INFINITY, bu = -INFINITY;
        for (int j = 0; j < np; ++j) {
            if (pl[j] != lab) continue;
            const long long s1 = ps[j], e1 = j + 1 < np ? ps[j + 1] : T;
            const long long inter = min(e1, e0) - max(s1, s0);
            bd = fmax(bd, (double)inter / (double)(e1 - s1));
            bu = fmax(bu, (double)inter / (double)(max(e1, e0) - min(s1, s0)));
        }
        run_label[(size_t)v * MT_MAX_RUNS + i] = lab;
        iod[(size_t)v * MT_MAX_RUNS + i] = bd;
        iou[(size_t)v * MT_MAX_RUNS + i] = bu;
    }
}
}  // namespace

extern "C" int mucon_metrics_overlap(int32_t n_videos, const int64_t *offsets, const int32_t *targets, const int32_t *predictions,
                                     const int32_t *ignore_ids, int32_t n_ignore, int64_t *mof, int32_t *n_runs, int32_t *run_label,
                                     double *iod, double *iou, void *stream) {
    if (n_videos <= 0) return MUCON_OK;
    if (!offsets || !targets || !predictions || !mof || !n_runs || !run_label || !iod || !iou || n_ignore < 0 || (n_ignore > 0 && !ignore_ids)) {
        mucon_internal_set_error("metrics_overlap: null pointer argument");
        return MUCON_E_ARG;
    }
    hipLaunchKernelGGL(metrics_overlap_kernel, dim3(n_videos), dim3(MT_THREADS), 0, static_cast<hipStream_t>(stream), offsets, targets,
                       predictions, ignore_ids, n_ignore, mof, n_runs, run_label, iod, iou);
    if (hipGetLastError() != hipSuccess) {
        mucon_internal_set_error("metrics_overlap: kernel launch failed");
        return MUCON_E_HIP;
    }
    return MUCON_OK;
}
