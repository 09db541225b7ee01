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

// Everything the reference's evaluator accumulates per (target, prediction) pair, in one workgroup (SURVEY.md 8f row 4:
// src/core/metrics/segmentation.py:16-91, mstcn_code.py:6-81 / fully_supervised.py:9-94):
//   * frame accuracy with and without the ignored labels;
//   * per target run the best same-label IoD / IoU (as metrics_overlap_kernel);
//   * the segmental edit distance: unit-cost Levenshtein between the two run-label sequences (an integer; the host turns it into
//     the reference's (1 - d / max(len)) * 100), as a wavefront over anti-diagonals, three diagonals in LDS;
//   * segmental F1 counts at up to 4 IoU thresholds: every predicted run claims, in order, the target run of the same label it
//     overlaps most -- numpy's argmax over `(1.0 * inter / union) * (label equal)`, the first maximum -- once per target run.
// No labels are ignored for edit / F1 (the evaluator builds Edit() and F1Score() without ignore_ids).
constexpr int MS_MAX_OV = 4;
__global__ __launch_bounds__(MT_THREADS) void metrics_segmental_kernel(const int64_t *offsets, const int32_t *targets, const int32_t *predictions,
                                                                       const int32_t *ignore, int n_ignore, const double *thresholds,
                                                                       int n_thr, int64_t *mof, int32_t *n_runs, int32_t *run_label,
                                                                       double *iod, double *iou, int32_t *seg) {
    __shared__ int ts[MT_MAX_RUNS], tl[MT_MAX_RUNS], ps[MT_MAX_RUNS], pl[MT_MAX_RUNS];
    __shared__ int best_i[MT_MAX_RUNS];
    __shared__ double best_v[MT_MAX_RUNS];
    __shared__ int diag[3][MT_MAX_RUNS + 1];
    __shared__ unsigned hits[MS_MAX_OV][MT_MAX_RUNS / 32];
    __shared__ int wave_tot[4];
    __shared__ long long red[4][4];
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long o = offsets[v];
    const int T = (int)(offsets[v + 1] - o);
    const int32_t *tg = targets + o, *pr = predictions + o;
    auto ignored = [&](int label) {
        bool ig = false;
        for (int i = 0; i < n_ignore; ++i) ig |= label == ignore[i];
        return ig;
    };
    // 1. frame accuracy: all frames, and the frames whose target is not ignored
    long long tot = 0, cor = 0, tot_i = 0, cor_i = 0;
    for (int t = tid; t < T; t += MT_THREADS) {
        const int a = tg[t];
        const bool hit = a == pr[t], keep = !ignored(a);
        tot += 1;
        cor += hit;
        tot_i += keep;
        cor_i += keep && hit;
    }
    for (int s = 32; s >= 1; s >>= 1) {
        tot += __shfl_xor(tot, s);
        cor += __shfl_xor(cor, s);
        tot_i += __shfl_xor(tot_i, s);
        cor_i += __shfl_xor(cor_i, s);
    }
    if (lane == 0) {
        red[0][wave] = cor;
        red[1][wave] = tot;
        red[2][wave] = cor_i;
        red[3][wave] = tot_i;
    }
    __syncthreads();
    if (tid < 4) mof[4 * v + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    // 2. runs
    const int nt = encode_runs(tg, T, ts, tl, wave_tot);
    const int np = encode_runs(pr, T, ps, pl, wave_tot);
    if (tid == 0) {
        n_runs[3 * v] = nt;
        n_runs[3 * v + 1] = np;
    }
    if (nt > MT_MAX_RUNS || np > MT_MAX_RUNS) return;   // the host sees the counts and takes its own path
    {   // predicted runs whose label is not ignored (none left: the no-background overlap score is 0 by definition)
        int kept = 0;
        for (int j = tid; j < np; j += MT_THREADS) kept += !ignored(pl[j]);
        for (int s = 32; s >= 1; s >>= 1) kept += __shfl_xor(kept, s);
        __syncthreads();
        if (lane == 0) wave_tot[wave] = kept;
        __syncthreads();
        if (tid == 0) n_runs[3 * v + 2] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    }
    // 3. best same-label overlap per target run
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
    // 4. F1: per predicted run the target run of numpy's argmax (first of the maxima; strict > while scanning), then the greedy claims
    for (int j = tid; j < np; j += MT_THREADS) {
        const long long s1 = ps[j], e1 = j + 1 < np ? ps[j + 1] : T;
        int bi = 0;
        double bv = 0.0;
        for (int i = 0; i < nt; ++i) {
            const long long s0 = ts[i], e0 = i + 1 < nt ? ts[i + 1] : T;
            const long long inter = min(e1, e0) - max(s1, s0), uni = max(e1, e0) - min(s1, s0);
            const double val = (1.0 * (double)inter / (double)uni) * (pl[j] == tl[i] ? 1.0 : 0.0);
            if (i == 0 || val > bv) {
                bv = val;
                bi = i;
            }
        }
        best_i[j] = bi;
        best_v[j] = bv;
    }
    for (int e = tid; e < MS_MAX_OV * (MT_MAX_RUNS / 32); e += MT_THREADS) (&hits[0][0])[e] = 0u;
    __syncthreads();
    if (tid < n_thr) {
        const double ov = thresholds[tid];
        int tp = 0;
        if (nt > 0)
            for (int j = 0; j < np; ++j) {
                const int k = best_i[j];
                if (best_v[j] >= ov && !((hits[tid][k >> 5] >> (k & 31)) & 1u)) {
                    ++tp;
                    hits[tid][k >> 5] |= 1u << (k & 31);
                }
            }
        // (no target run: the predicted runs are false positives; no predicted run: the target runs are misses)
        seg[(size_t)v * (1 + 3 * MS_MAX_OV) + 1 + 3 * tid] = tp;
        seg[(size_t)v * (1 + 3 * MS_MAX_OV) + 2 + 3 * tid] = np - tp;
        seg[(size_t)v * (1 + 3 * MS_MAX_OV) + 3 + 3 * tid] = nt == 0 ? 0 : nt - tp;
    }
    // 5. edit distance between the run-label sequences pl[0..np) and tl[0..nt): cell (i, j) on anti-diagonal d = i + j
    //    D(i, 0) = i, D(0, j) = j, D(i, j) = min(D(i-1, j-1) + (pl[i-1] != tl[j-1]), D(i-1, j) + 1, D(i, j-1) + 1); diag[d % 3][i]
    for (int d = 0; d <= np + nt; ++d) {
        int *cur = diag[d % 3];
        const int *p1 = diag[(d + 2) % 3], *p2 = diag[(d + 1) % 3];   // diagonals d - 1 and d - 2
        const int ilo = max(0, d - nt), ihi = min(np, d);
        for (int i = ilo + tid; i <= ihi; i += MT_THREADS) {
            const int j = d - i;
            int val;
            if (i == 0) val = j;
            else if (j == 0) val = i;
            else val = min(min(p2[i - 1] + (pl[i - 1] != tl[j - 1] ? 1 : 0), p1[i - 1] + 1), p1[i] + 1);
            cur[i] = val;
        }
        __syncthreads();
    }
    if (tid == 0) seg[(size_t)v * (1 + 3 * MS_MAX_OV)] = diag[(np + nt) % 3][np];
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

extern "C" int mucon_metrics_segmental(int32_t n_pairs, const int64_t *offsets, const int32_t *targets, const int32_t *predictions,
                                       const int32_t *ignore_ids, int32_t n_ignore, const double *thresholds, int32_t n_thresholds,
                                       int64_t *mof, int32_t *n_runs, int32_t *run_label, double *iod, double *iou, int32_t *seg,
                                       void *stream) {
    if (n_pairs <= 0) return MUCON_OK;
    if (!offsets || !targets || !predictions || !mof || !n_runs || !run_label || !iod || !iou || !seg || n_ignore < 0 ||
        (n_ignore > 0 && !ignore_ids) || n_thresholds < 0 || n_thresholds > MS_MAX_OV || (n_thresholds > 0 && !thresholds)) {
        mucon_internal_set_error("metrics_segmental: null pointer argument / more than 4 thresholds");
        return MUCON_E_ARG;
    }
    hipLaunchKernelGGL(metrics_segmental_kernel, dim3(n_pairs), dim3(MT_THREADS), 0, static_cast<hipStream_t>(stream), offsets, targets,
                       predictions, ignore_ids, n_ignore, thresholds, n_thresholds, mof, n_runs, run_label, iod, iou, seg);
    if (hipGetLastError() != hipSuccess) {
        mucon_internal_set_error("metrics_segmental: kernel launch failed");
        return MUCON_E_HIP;
    }
    return MUCON_OK;
}
