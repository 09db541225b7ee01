/* mucon_hip.h -- C ABI of libmucon_hip.so: the MI355X (gfx950) hot path of MuCon.
 *
 * The reference (yassersouri/MuCon) is pure Python and has no FFI seam; its seam for this path is
 * a Python class surface.  Each entry point below replaces the arithmetic behind
 * one reference call site; mucon_amd/ mirrors the reference's classes on top of it and
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every data pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *     the caller owns all buffers; the library allocates no device memory beyond one small table for the
 *     fused SGD step.  Everything a call launches runs on `stream`, in order (the one exception is the
 *     regression schedule MUCON_TN_BATCH=0, which runs coarse-level weight gradients on a library-owned
 *     side stream and joins it back into `stream` before the call's last kernel).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *     enqueues work on it and returns (no host synchronisation), except where stated.
 *   - return value: 0 on success, negative MUCON_E_* on error; mucon_last_error() gives the text.
 *   - activations are time-major [B][T][channels] float32; weights keep the reference's
 *     nn.Conv1d layout [out][in][k] so a reference state_dict can be passed unchanged.
 */
#ifndef MUCON_HIP_H
#define MUCON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUCON_ABI_VERSION 8   /* 8: mucon_head_bwd_defer / _flush (the y-head's slab reduction inside the encoder backward's first launch), mucon_decoder_bwd_defer / _flush (the decoder's weight-gradient outer products inside the LSTM backward's recurrence launch), mucon_head_fwd_defer / _flush (the y-head's forward inside the LSTM forward's recurrence launch); 7: the *_poisson Viterbi entries (length scores built on the device), group_norms [2 * n_groups] with sticky skipped-step counts; 5: label_format of the Viterbi entry points; 4: mucon_viterbi_job carries the emission pointer; mucon_viterbi_decode_host */
#define MUCON_MAX_LAYERS 16

#define MUCON_OK 0
#define MUCON_E_ARG (-1)        /* unsupported / inconsistent argument            */
#define MUCON_E_WORKSPACE (-2)  /* workspace too small                            */
#define MUCON_E_HIP (-3)        /* a HIP runtime call failed                      */

/* Per-video status written by mucon_viterbi_decode_batch (mirrors the reference's behaviour) */
#define MUCON_VIT_OK 0
#define MUCON_VIT_INDEX_ERROR 1    /* T < frame_sampling: reference raises IndexError     (viterbi.py:87)  */
#define MUCON_VIT_NO_HYPOTHESIS 2  /* hypothesis set empty / all NaN: AttributeError      (viterbi.py:147) */
#define MUCON_VIT_TRUNCATED 3      /* no final-state hypothesis: score = -inf, transcript truncated        */

int mucon_abi_version(void);
const char *mucon_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Encoder: cfg.model.ft.* (reference src/configs/mucon/default.py:81-96)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, T, D, H;                      /* batch, frames, input dim (2048), hidden (128)       */
    int32_t n_layers;                        /* len(cfg.model.ft.stages)                            */
    int32_t dilation[MUCON_MAX_LAYERS];      /* cfg.model.ft.stages                                 */
    int32_t pool_after[MUCON_MAX_LAYERS];    /* 1 if i in cfg.model.ft.pooling_layers and pooling   */
    int32_t pool_type;                       /* 0 = "max", 1 = "sum" (avg_pool1d * 2)               */
    int32_t leaky;                           /* cfg.model.ft.leaky_relu (slope 0.01)                */
    int32_t last_gn, gn_groups;              /* cfg.model.ft.last_gn, last_gn_num_groups            */
    float gn_eps;                            /* nn.GroupNorm default 1e-5                           */
    int32_t last_relu;                       /* cfg.model.ft.last_relu                              */
    int32_t training;                        /* nn.Module.training: dropout active                  */
    float p_drop_layer;                      /* cfg.model.ft.dropout_rate (WaveNetLayer.drop)       */
    float p_drop_last;                       /* cfg.model.ft.last_dropout_rate if last_dropout else 0 */
    uint64_t seed;                           /* dropout stream for this step                        */
} mucon_encoder_cfg;

/* Parameters (and, with the same shapes, their gradients).  Names = reference state_dict keys. */
typedef struct {
    float *first_w, *first_b;                               /* ft.first_conv.{weight[H,D,1],bias[H]}       */
    float *dil_w[MUCON_MAX_LAYERS], *dil_b[MUCON_MAX_LAYERS]; /* ft.l_i.dilated_conv.{weight[H,H,3],bias} */
    float *pw_w[MUCON_MAX_LAYERS], *pw_b[MUCON_MAX_LAYERS];   /* ft.l_i.conv_1x1.{weight[H,H,1],bias}     */
    float *last_w, *last_b;                                 /* ft.last_conv.{weight[H,H,1],bias[H]}        */
    float *gn_w, *gn_b;                                     /* ft_last_gn.{weight,bias}[H]                 */
} mucon_encoder_params;

/* Output length Tz after the pooling schedule (reference temporal.py:137-142: floor halving). */
int32_t mucon_encoder_out_length(const mucon_encoder_cfg *cfg);

/* Bytes of caller-provided scratch that fwd fills and bwd reads (saved activations, packed
 * weights, gradient slabs).  Depends on B, T and the layer schedule only. */
size_t mucon_encoder_workspace_bytes(const mucon_encoder_cfg *cfg);

/* Replaces MuCon.temporal_modeling_forward (reference src/mucon/models.py:746-773), i.e.
 * permute -> WaveNetBlock.forward (src/core/modules/temporal.py:128-147, :43-53) -> GroupNorm ->
 * ReLU -> Dropout -> permute.   tape [B][T][D] -> enc [B][Tz][H]. */
int mucon_encoder_fwd(const mucon_encoder_cfg *cfg, const mucon_encoder_params *params,
                      const float *tape, float *enc, void *workspace, size_t workspace_bytes,
                      void *stream);

/* Replaces autograd through the above (triggered at reference src/mucon/trainers.py:131).
 * d_enc [B][Tz][H] -> grads (every member written, not accumulated).  Must follow a fwd call
 * with training-independent identical cfg on the same workspace. */
int mucon_encoder_bwd(const mucon_encoder_cfg *cfg, const mucon_encoder_params *params,
                      const float *tape, const float *d_enc, void *workspace,
                      size_t workspace_bytes, const mucon_encoder_params *grads, void *stream);

/* Data-parallel training (no reference counterpart: reference src/core/config.py:16 has one device string; BASELINE.json's north_star names the
 * 1/2/4/8-GPU curve).  One-shot options of the NEXT mucon_encoder_bwd call on this thread:
 *   event_after_layers  a hipEvent_t (NULL: none) the pass records on its stream once every gradient EXCEPT first_conv.weight / .bias is final
 *                       (the residual layers', last_conv's and GroupNorm's: 3 of the hot path's 4 MB).  The caller's all-reduce of that part of
 *                       its flat gradient buffer, issued on another stream behind the event, then travels UNDER first_conv's weight-gradient
 *                       launch (the tape's second pass, ~85 us at B = 8 x T = 4096) instead of behind the whole backward;
 *   max_workgroups      > 0: the weight-gradient launches of that pass use at most this many persistent workgroups (one per CU by default):
 *                       the CUs left free are where RCCL's kernel runs meanwhile.  0: no cap.
 * Both are cleared when the pass has been enqueued.  The gradients are the same sums cut into other shares: equal to fp32 rounding. */
int mucon_encoder_bwd_overlap(void *event_after_layers, int32_t max_workgroups);

/* ------------------------------------------------------------------------------------------
 * Encoder variant "noft": no temporal modelling, one position-wise linear map of the tape
 * ---------------------------------------------------------------------------------------- */
/* Replaces NoFt.forward (reference src/core/modules/temporal.py:56-74: a single kernel-size-1 nn.Conv1d, selected by
 * cfg.model.ft.type = "noft", src/mucon/models.py:181-185):  out [B][T][128] = tape [B][T][D] * w [128][D]^T + b.
 * It is first_conv without its non-linearity and runs on the same kernels (f32 MFMA; the split-bf16 kernel from 8,192
 * frames per launch); the backward is first_conv's weight-gradient job (the tape needs no gradient).
 * D a positive multiple of 128, 128 output channels. */
size_t mucon_linear_workspace_bytes(int32_t B, int32_t T, int32_t D);
int mucon_linear_fwd(int32_t B, int32_t T, int32_t D, const float *tape, const float *w, const float *b, float *out,
                     void *workspace, size_t workspace_bytes, void *stream);
/* d_out [B][T][128] -> d_w [128][D], d_b [128] */
int mucon_linear_bwd(int32_t B, int32_t T, int32_t D, const float *tape, const float *d_out, float *d_w, float *d_b,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Encoder variant "mstcnpp": the 128-channel temporal convolutions of MSTCNPPFirstStage
 * ---------------------------------------------------------------------------------------- */
/* The building block of MSTCNPPFirstStage (reference src/core/modules/temporal.py:150-204, selected by
 * cfg.model.ft.type = "mstcnpp", src/mucon/models.py:172-179): nn.Conv1d(128, 128, taps, padding = dilation, dilation) with
 * taps = 3 (conv_dilated_1 / conv_dilated_2) or taps = 1 (the halves of conv_fusion, conv_out), time-major:
 *     y [B][T][128] = sum_tap x [B][t + (tap - taps/2) * dilation][128] * w[:, :, tap]^T + b        (zero rows outside [0, T))
 * on the encoder's f32-MFMA kernels (gemm_nt.hpp / gemm_tn.hpp).  The three entry points are the three linear maps autograd needs;
 * non-linearity, dropout, the residual and the max-pooling of a layer stay element-wise operations of the caller.
 *   fwd    w_fwd [128][taps * 128] = w[o][i][tap] at column tap * 128 + i;  b [128] or NULL
 *   dgrad  g [B][T][128] -> d_x [B][T][128];  w_bwd [128][taps * 128] = w[o][i][tap] at row i, column tap * 128 + o
 *   wgrad  g, x -> d_w [128][128][taps] (the nn.Conv1d layout), d_b [128] (may be NULL) */
size_t mucon_conv128_workspace_bytes(int32_t B, int32_t T, int32_t taps);
int mucon_conv128_fwd(int32_t B, int32_t T, int32_t taps, int32_t dilation, const float *x, const float *w_fwd, const float *b,
                      float *y, void *stream);
/* The tail of an MS-TCN++ layer (reference src/core/modules/temporal.py:196-201) as ONE launch:
 *   y = [max_pool1d(2)] ( f + dropout( relu( conv_fusion(cat(a, b)) ) ) )
 * a, b [B][T][128]: the two dilated convolutions' outputs; w = conv_fusion.weight [128][256] as nn.Conv1d stores it ([out][in][1]);
 * bias [128] or NULL; f [B][T][128] the layer input (residual).  Dropout is counter-based (seed, element index), active when
 * training != 0.  pool != 0: y is [B][T/2][128] and y_pre receives the un-pooled rows [B][T][128] (the backward's arg-max).
 * x_act (NULL, or [B][T][128]) receives the branch value dropout(relu(.)) before the residual is added: x > 0 marks the elements
 * whose gradient passes (mucon_mstcn_tail_bwd). */
int mucon_mstcn_fuse_fwd(int32_t B, int32_t T, const float *a, const float *b, const float *w, const float *bias, const float *f,
                         float p_drop, uint64_t seed, int32_t training, int32_t pool, float *y, float *y_pre, float *x_act,
                         void *stream);
/* Its backward up to the fusion convolution's output u: d_sum [B][T][128] = the gradient w.r.t. the un-pooled sum f + x (d_y routed
 * to the arg-max row of each pair when pooled; this is also the residual path's gradient w.r.t. f), d_u = d_sum * scale where
 * x_act > 0, else 0 (scale = 1 / (1 - p) in training, 1 otherwise).  The convolutions' own gradients are mucon_conv128_dgrad / _wgrad. */
int mucon_mstcn_tail_bwd(int32_t B, int32_t T, int32_t pool, const float *d_y, const float *y_pre, const float *x_act, float scale,
                         float *d_sum, float *d_u, void *stream);
int mucon_conv128_dgrad(int32_t B, int32_t T, int32_t taps, int32_t dilation, const float *g, const float *w_bwd, float *d_x,
                        void *stream);
int mucon_conv128_wgrad(int32_t B, int32_t T, int32_t taps, int32_t dilation, const float *g, const float *x, float *d_w,
                        float *d_b, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation counters on the device: frame accuracy and segment overlap (MoF / IoD / IoU)
 * ---------------------------------------------------------------------------------------- */
/* The counters of MoFAccuracyMetric, IoDMetric and IoUMetric (reference src/core/metrics/segmentation.py:16-91 and the
 * isba_code.py:23-109 they call) for labellings that live on the device.  Video v owns frames offsets[v] .. offsets[v+1] of
 * targets / predictions (int32 labels; offsets has n_videos + 1 entries, all device pointers).
 *   mof [n_videos][2]                 frames whose target is not in ignore_ids: those with target == prediction, and their number
 *   n_runs [n_videos][3]              runs (maximal constant stretches) of the targets, of the predictions, and the predicted runs
 *                                     whose label is not in ignore_ids
 *   run_label, iod, iou [n_videos][MUCON_METRICS_MAX_RUNS]
 *                                     per TARGET run: its label and the largest intersection / predicted-run length (iod) and
 *                                     intersection / union span (iou) over the predicted runs with the same label (-inf if none);
 *                                     valid when both run counts are <= MUCON_METRICS_MAX_RUNS
 * The quotients are single float64 divisions of integers, so mean(max(., 0)) over the runs whose label is not ignored reproduces the
 * host metric's value bit for bit (mucon_amd/core/metrics/device.py). */
#define MUCON_METRICS_MAX_RUNS 1024
int mucon_metrics_overlap(int32_t n_videos, const int64_t *offsets, const int32_t *targets, const int32_t *predictions,
                          const int32_t *ignore_ids, int32_t n_ignore, int64_t *mof, int32_t *n_runs, int32_t *run_label,
                          double *iod, double *iou, void *stream);

/* The whole per-video record of the reference's evaluator for a list of (target, prediction) pairs in ONE launch: the counters
 * above plus the segmental edit distance and F1 counts (reference src/core/metrics/mstcn_code.py:6-81, fully_supervised.py:9-94).
 *   mof [n_pairs][4]                  correct, total over all frames; correct, total over the frames whose target is not ignored
 *   n_runs, run_label, iod, iou       as above
 *   seg [n_pairs][1 + 3 * 4]          unit-cost Levenshtein distance between the two run-label sequences, then (tp, fp, fn) of the
 *                                     segmental F1 at thresholds[0 .. n_thresholds) (n_thresholds <= 4): every predicted run claims,
 *                                     in order, the target run numpy's argmax over (1.0 * inter / union) * (labels equal) names
 * Integers and single float64 quotients only: the host classes' float results follow from them bit for bit. */
int mucon_metrics_segmental(int32_t n_pairs, const int64_t *offsets, const int32_t *targets, const int32_t *predictions,
                            const int32_t *ignore_ids, int32_t n_ignore, const double *thresholds, int32_t n_thresholds,
                            int64_t *mof, int32_t *n_runs, int32_t *run_label, double *iod, double *iou, int32_t *seg,
                            void *stream);

/* ------------------------------------------------------------------------------------------
 * y-head: nearest upsample Tz -> Tf, 1x1 conv H -> C, log-softmax over C
 * ---------------------------------------------------------------------------------------- */
size_t mucon_head_workspace_bytes(int32_t B, int32_t Tz, int32_t H, int32_t C);

/* Replaces MuCon.frame_classifier_forward (reference src/mucon/models.py:567-582) and the
 * F.log_softmax of MuCon.predict (:367-368) / the smoothing loss (:403-405).
 * enc [B][Tz][H], w [C][H][1], b [C] -> logits [B][Tf][C] (may be NULL), logp [B][Tf][C] (may be
 * NULL).  workspace keeps the z-level log-probs for the backward. */
int mucon_head_fwd(int32_t B, int32_t Tz, int32_t Tf, int32_t H, int32_t C, const float *enc,
                   const float *w, const float *b, float *logits, float *logp, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ABI 8 (no reference counterpart).  One-shot option of the NEXT mucon_head_fwd on this thread (enable != 0, H = 128): the call enqueues nothing; the NEXT
 * mucon_lstm_fwd on the same stream runs the classifier in extra workgroups of its recurrence launch (two CUs busy for ~73 us at Tz = 125: ~9 us of launch off the
 * step's critical path).  logits / logp and the workspace's saved rows exist once that call has been enqueued; enc, w, b and the outputs must stay valid until then.
 * Another mucon_head_fwd, mucon_head_bwd, mucon_loss_fwd_bwd, a mucon_lstm_fwd on another stream, or mucon_head_fwd_flush launch a pending forward on its own.  Bitwise
 * the plain call's results. */
int mucon_head_fwd_defer(int32_t enable);
int mucon_head_fwd_flush(void);

/* d_logits / d_logp [B][Tf][C] (either may be NULL) -> d_enc [B][Tz][H], d_w [C][H], d_b [C]. */
int mucon_head_bwd(int32_t B, int32_t Tz, int32_t Tf, int32_t H, int32_t C, const float *enc,
                   const float *w, const float *d_logits, const float *d_logp, float *d_enc,
                   float *d_w, float *d_b, void *workspace, size_t workspace_bytes, void *stream);

/* ABI 8 (no reference counterpart: autograd runs the classifier's weight gradient as a launch of its own, reference src/mucon/trainers.py:131).
 * One-shot option of the NEXT mucon_head_bwd call on this thread (enable != 0): that call writes d_enc as always but leaves d_w / d_b as the
 * per-workgroup partial sums in its workspace; the NEXT mucon_encoder_bwd on the same stream adds them up in extra workgroups of its first launch
 * (beside the GroupNorm backward: one launch fewer on the step's critical path, ~6 us at B = 8 x T = 4096) -- the same sums in the same order, bitwise the
 * d_w / d_b of the plain call.  d_w / d_b and the workspace must stay valid until then.  If another mucon_head_bwd comes first, or
 * mucon_head_bwd_flush is called, the pending sums are taken by a launch of their own on the stream they were left on.  Shapes the float4 sums do not
 * cover (C not a multiple of 4) are reduced at once as if the option were off.
 * enable bit 1 (enable = 3, H = 128 only): the z-level backward KERNEL waits as well -- for the next mucon_decoder_bwd on the same stream, whose eight-workgroup step
 * loop (~65 us on eight CUs) runs it in extra workgroups; d_enc exists once that call has been enqueued (the LSTM's backward, which adds onto it, follows).  If no
 * mucon_decoder_bwd comes, mucon_encoder_bwd / mucon_head_bwd_flush / the next mucon_head_bwd launch the kernel on its own.  Bitwise the plain call's results. */
int mucon_head_bwd_defer(int32_t enable);
int mucon_head_bwd_flush(void);

/* ------------------------------------------------------------------------------------------
 * Viterbi: transcript-constrained decode with a length model
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float *lp;    /* DEVICE pointer to this video's emissions [T][C] f32, row-major, 16-byte
                           aligned when C is a multiple of 4 (ABI 4: one pointer per video -- a batch
                           of videos is decoded where its emissions lie, nothing is concatenated)  */
    int64_t tr_off;     /* int32 offset of its transcript [N] inside `transcripts`              */
    int64_t p_off;      /* double offset of its length table [J][N] inside `length_tables`      */
    int64_t label_off;  /* ELEMENT offset of its output labels [T] inside `labels` (int32 or uint8
                           elements, by label_format; unused with MUCON_VIT_LABELS_NONE)          */
    int64_t seg_off;    /* int32 offset of its output segment lengths [N] inside `seg_len`      */
    int64_t ws_off;     /* byte offset of its scratch inside `workspace` (16-byte aligned)      */
    int32_t T, N;
    int32_t force_n;    /* >= 0: finalize on hypothesis (force_n, force_j) with score -inf      */
    int32_t force_j;    /*       (the reference's degenerate outcomes, decided by the host)     */
} mucon_viterbi_job;

/* scratch bytes one video needs (frame scores [K][C] f32 + back-pointers [K][N] u8, aligned) */
size_t mucon_viterbi_job_workspace_bytes(int32_t T, int32_t C, int32_t N, int32_t fs);

/* The form the per-frame labels leave the decode in (ABI 5).  The decode's result proper is the segmentation
 * (seg_len[N], n_seg) -- the reference's traceback (src/core/viterbi/viterbi.py:140-158) builds its `labels` list from exactly
 * that: `T - K*fs` leftover frames with the LAST segment's label at the start of the video, then every segment's label
 * `length` times (the leftover frames are included in the last segment's length, :154-157).
 *   I32   int32 [T] per video: the reference's list of ints, 4 T bytes
 *   U8    uint8 [T] per video (C <= 64 is an ABI limit): a quarter of the bytes
 *   NONE  nothing is written, `labels` may be NULL: the caller expands (transcript, seg_len, n_seg, T, fs) itself when it needs
 *         per-frame labels (mucon_amd/ops.py: ViterbiResult.labels does it on first access) */
#define MUCON_VIT_LABELS_I32 0
#define MUCON_VIT_LABELS_U8 1
#define MUCON_VIT_LABELS_NONE 2

/* mucon_viterbi_decode_host: calls of up to this many videos are latency calls (one launch per call, results through the library's
 * pinned staging buffer); from this many on, a `labels` array that is itself pinned host memory is written in place. */
#define MUCON_VIT_LATENCY_VIDEOS 8

/* Replaces Viterbi.decode (reference src/core/viterbi/viterbi.py:49-158) with
 * SingleTranscriptGrammar (src/core/viterbi/grammar.py:196-217) and an f64 length table
 * P[j][n] = length_model.score((j+1)*fs, a_n), J = max_len / fs rows
 * (PoissonModel, src/core/viterbi/length_model.py:76-80), as driven by
 * src/mucon/evaluators.py:147-180.  One workgroup per video; bit-exact (score, labels, segments).
 * jobs: DEVICE array [n_videos].  Outputs (all DEVICE memory; the call is asynchronous on `stream`): labels in
 * `label_format` (MUCON_VIT_LABELS_*), seg_len (per job offsets), n_seg[n_videos], score[n_videos] (f64),
 * status[n_videos] (MUCON_VIT_*).  Always the throughput schedule: one frame-score launch + one DP launch for the
 * whole batch, back-pointers in `workspace`. */
int mucon_viterbi_decode_batch(int32_t n_videos, const mucon_viterbi_job *jobs, int32_t C,
                               int32_t fs, int32_t max_len, int32_t max_N,
                               const int32_t *transcripts, const double *length_tables,
                               void *labels, int32_t label_format, int32_t *seg_len, int32_t *n_seg,
                               double *score, int32_t *status, void *workspace, void *stream);

/* The same decode with HOST-side inputs and outputs -- what the reference's call site is
 * (src/mucon/evaluators.py:178-180: numpy in, Python lists out): emissions stay on the device,
 * transcript and length table are host arrays, the results arrive in host arrays when the call returns
 * (it synchronises with the work it enqueued on `stream`; everything queued before it on that stream is
 * ordered in front of the decode).  One launch for a single short video (<= 16 transcript states, <= 640
 * columns, T / fs * (C * 4 + N) bytes of frame scores and back-pointers beside the chain's LDS buffers: the
 * DP runs under the frame-score chain); inputs are read from and results written to library-owned pinned
 * host buffers, no copy calls.  labels (in `label_format`; NULL with MUCON_VIT_LABELS_NONE): video v's T labels at the
 * ELEMENT offset sum of max(T, 1) of the videos before it; seg_len: its N entries at the sum of N before it (n_seg[v] valid).
 * If `labels` is itself pinned host memory (hipHostMalloc / torch pin_memory) and the call has >= MUCON_VIT_LATENCY_VIDEOS
 * videos, the kernels write it in place (no staging copy of the largest output). */
typedef struct {
    const float *lp;            /* DEVICE: emissions [T][C] f32 */
    const int32_t *transcript;  /* HOST [N] */
    const double *table;        /* HOST [J][N], J = max_len / fs */
    int32_t T, N;
    int32_t force_n, force_j;   /* as in mucon_viterbi_job (-1, -1: none) */
} mucon_viterbi_video;
int mucon_viterbi_decode_host(int32_t n_videos, const mucon_viterbi_video *videos, int32_t C, int32_t fs,
                              int32_t max_len, double *score, int32_t *n_seg, int32_t *status,
                              void *labels, int32_t label_format, int32_t *seg_len, void *stream);

/* ABI 7: the same two decodes with the PoissonModel's length scores BUILT ON THE DEVICE (reference src/core/viterbi/length_model.py:65-71:
 * `self.poisson[l, :] = l * np.log(self.mean_lengths) - self.mean_lengths - logFak - self.norms`, read at l = (j + 1) fs; >= max_len: -inf, :76-80).
 * Per transcript state n the caller passes the three numbers of that expression that depend on the class c = a_n -- the rows
 * [0][n] = np.log(mu_c), [1][n] = mu_c, [2][n] = norms_c of a [3][N] block (NumPy's log and the norms' sums stay on the host: length_model.py:54-63)
 * -- and ONE shared row log_fact [J], log_fact[j] = the reference's running `logFak` at l = (j + 1) fs.  The kernels evaluate
 * ((l * [0][n] - [1][n]) - log_fact[j]) - [2][n] left to right as four single IEEE double operations (csrc/viterbi.hip is built with
 * -ffp-contract=off): bit for bit the table the host would have sent (tests/test_gpu_viterbi.py::test_device_built_length_rows, incl.
 * mu < 0.5 -> NaN).  3 doubles per state cross PCIe instead of J = 66: 8.6 MB -> 0.4 MB for a 256-video call of BASELINE config 5.
 *   mucon_viterbi_decode_host_poisson:  videos[v].table = HOST [3][N] parameter block (instead of [J][N]); log_fact HOST [J].
 *   mucon_viterbi_decode_batch_poisson: jobs[v].p_off = double offset of the video's [3][N] block inside poisson_params (DEVICE); log_fact DEVICE [J].
 * Everything else as mucon_viterbi_decode_host / mucon_viterbi_decode_batch. */
int mucon_viterbi_decode_host_poisson(int32_t n_videos, const mucon_viterbi_video *videos, const double *log_fact, int32_t C, int32_t fs,
                                      int32_t max_len, double *score, int32_t *n_seg, int32_t *status,
                                      void *labels, int32_t label_format, int32_t *seg_len, void *stream);
int mucon_viterbi_decode_batch_poisson(int32_t n_videos, const mucon_viterbi_job *jobs, int32_t C,
                                       int32_t fs, int32_t max_len, int32_t max_N,
                                       const int32_t *transcripts, const double *poisson_params, const double *log_fact,
                                       void *labels, int32_t label_format, int32_t *seg_len, int32_t *n_seg,
                                       double *score, int32_t *status, void *workspace, void *stream);

/* Viterbi.decode of a decoder built with a finite max_hypotheses (reference src/core/viterbi/viterbi.py:34; prune(), :74-79, after
 * every column :57-61): the beam search, bit for bit -- which hypothesis wins a tie depends on the iteration order of the reference's
 * hypothesis dict and on Python's tuple order of prune()'s (score, key) pairs; both are reproduced (csrc/viterbi_beam.hip).
 * The reference's callers never pass max_hypotheses (evaluators.py:80): this entry exists for those who do.  Same inputs as
 * mucon_viterbi_decode_host (force_n / force_j are ignored: the degenerate outcomes they encode fall out of the list the kernel keeps);
 * 1 <= max_hypotheses, max_hypotheses + N <= MUCON_VIT_BEAM_MAX_ITEMS, N <= 128, max_len / fs <= 128, C <= 64.  A max_hypotheses of at least
 * N * (max_len / fs) never prunes: mucon_viterbi_decode_host is the entry for that (and for inf).  Outputs are HOST arrays: score, n_seg,
 * status [n_videos], seg_len (video v's entries at the sum of N before it); the per-frame labels are the expansion of the segments
 * (viterbi.py:140-158: the frames beyond T // fs * fs carry the last label and come first).  status: MUCON_VIT_OK, MUCON_VIT_INDEX_ERROR,
 * MUCON_VIT_NO_HYPOTHESIS (every hypothesis outlived max_length or was pruned), MUCON_VIT_TRUNCATED (the beam lost every path into the last
 * transcript state: score -inf and n_seg < N segments, what the reference returns then).  Synchronous. */
#define MUCON_VIT_BEAM_MAX_ITEMS 4096
int mucon_viterbi_decode_beam(int32_t n_videos, const mucon_viterbi_video *videos, int32_t C, int32_t fs, int32_t max_len,
                              int64_t max_hypotheses, double *score, int32_t *n_seg, int32_t *status, int32_t *seg_len, void *stream);

/* ---- s-head sequence encoder: bidirectional LSTM --------------------------------
 * Replaces torch.nn.LSTM(128, 128, batch_first=True, bidirectional=True) as the reference's s-head calls
 * it (reference src/mucon/models.py:195-201 construction, :605-611 call: batch 1, zero initial state).
 * Parameter layout is torch's: w_ih [4H][I], w_hh [4H][H], b_ih [4H], b_hh [4H], gate order i,f,g,o;
 * index 0 = forward direction, 1 = reverse (weight_*_l0_reverse).  x [T][I], out [T][ndir*H]
 * (forward half first), hn / cn [ndir][H].  I = H = 128 only.  All pointers are device pointers. */
typedef struct mucon_lstm_params {
    const float *w_ih[2];
    const float *w_hh[2];
    const float *b_ih[2];
    const float *b_hh[2];
} mucon_lstm_params;
size_t mucon_lstm_workspace_bytes(int32_t T, int32_t ndir);
/* Forward; keeps the gate activations and cell states in `workspace` for mucon_lstm_bwd. */
int mucon_lstm_fwd(int32_t T, int32_t I, int32_t H, int32_t ndir, const float *x, const mucon_lstm_params *params,
                   float *out, float *hn, float *cn, void *workspace, size_t workspace_bytes, void *stream);
/* Backward through time from d_out [T][ndir*H], d_hn, d_cn [ndir][H] (each may be NULL = zero) with the
 * workspace the forward filled.  Writes d_x [T][I] = (d_x_add [T][I] when non-NULL, may alias d_x: the gradient another
 * consumer of x already produced) + the LSTM's own, and every tensor of d_params. */
int mucon_lstm_bwd(int32_t T, int32_t I, int32_t H, int32_t ndir, const float *x, const mucon_lstm_params *params,
                   const float *out, const float *d_out, const float *d_hn, const float *d_cn, float *d_x,
                   const float *d_x_add, const mucon_lstm_params *d_params, void *workspace, size_t workspace_bytes, void *stream);

/* ---- s-head attention decoder ---------------------------------------------------
 * Replaces the decoding loop of reference src/mucon/models.py:612-728 (sequence_generation_forward after
 * the encoder LSTM) and :730-744 (_calculate_attention): initial state from h_n / c_n through
 * fs_encoder_hidden_out / fs_encoder_cn_out, then per step embedding -> additive attention over `memory`
 * -> attn_combine -> one LSTM cell -> transcript MLP + length MLP -> log-softmax -> arg-max feedback.
 * One persistent workgroup walks all steps.  Tensors keep torch's layouts (Linear weight [out][in]);
 * every pointer is a device pointer. */
typedef struct mucon_decoder_cfg {
    int32_t Tz;              /* encoder states attended over */
    int32_t ME;              /* memory width = directions * encoder hidden (256) */
    int32_t D;               /* decoder width: embedding = hidden = attention size; must be 128 */
    int32_t NC;              /* transcript outputs = num_classes + 1 (EOS) */
    int32_t n_emb;           /* embedding rows = num_classes + 2 */
    int32_t max_steps;       /* steps to run (and the workspace layout) */
    int32_t teacher_forcing; /* 1: step input = tf_input[step]; 0: tf_input[0], then the previous arg-max */
    int32_t stop_on_eos;     /* 1: stop after the step whose arg-max is `eos` (models.py:718-721) */
    int32_t eos;
} mucon_decoder_cfg;
typedef struct mucon_decoder_params {
    const float *hidden_out_w, *hidden_out_b;   /* fs_encoder_hidden_out   [D][ME], [D] */
    const float *cn_out_w, *cn_out_b;           /* fs_encoder_cn_out       [D][ME], [D] */
    const float *attention_W1;                  /* fs_decoder_attention_W1 [ME][D] */
    const float *attention_l2_w, *attention_l2_b; /* fs_decoder_attention_l2 [D][D], [D] */
    const float *attention_V;                   /* fs_decoder_attention_V  [D] */
    const float *embedding;                     /* fs_decoder_embedding    [n_emb][D] */
    const float *attn_combine_w, *attn_combine_b; /* fs_decoder_attn_combine [D][D+ME], [D] */
    const float *lstm_w_ih, *lstm_w_hh, *lstm_b_ih, *lstm_b_hh; /* fs_decoder_lstm [4D][D] x2, [4D] x2 */
    const float *transcript0_w, *transcript0_b; /* fs_decoder_transcript[0] [D][D], [D] */
    const float *transcript2_w, *transcript2_b; /* fs_decoder_transcript[2] [NC][D], [NC] */
    const float *length0_w, *length0_b;         /* fs_decoder_length[0]     [D/2][D+NC], [D/2] */
    const float *length2_w, *length2_b;         /* fs_decoder_length[2]     [1][D/2], [1] */
} mucon_decoder_params;
size_t mucon_decoder_workspace_bytes(const mucon_decoder_cfg *cfg);
/* memory [Tz][ME] (encoder LSTM output), hn / cn [ME] (h_n.view(1,-1)), tf_input int64 [>= max_steps when
 * teacher forcing, >= 1 otherwise], dropmask [max_steps][D] = the embedding-dropout keep mask already scaled
 * by 1/(1-p), or NULL.  Writes logp [max_steps][NC], lengths [max_steps] and *n_steps (device int32: rows
 * valid).  Keeps the activations in `workspace` for mucon_decoder_bwd. */
int mucon_decoder_fwd(const mucon_decoder_cfg *cfg, const mucon_decoder_params *params, const float *memory,
                      const float *hn, const float *cn, const int64_t *tf_input, const float *dropmask,
                      float *logp, float *lengths, int32_t *n_steps, void *workspace, size_t workspace_bytes,
                      void *stream);
/* Backward over the n_steps the forward ran, from d_logp [n_steps][NC] and d_lengths [n_steps] (each may be
 * NULL = zero).  Writes d_memory [Tz][ME], d_hn / d_cn [ME] and every tensor of d_params. */
int mucon_decoder_bwd(const mucon_decoder_cfg *cfg, int32_t n_steps, const mucon_decoder_params *params,
                      const float *memory, const float *hn, const float *cn, const float *logp, const float *d_logp,
                      const float *d_lengths, const float *dropmask, float *d_memory, float *d_hn, float *d_cn,
                      const mucon_decoder_params *d_params, void *workspace, size_t workspace_bytes, void *stream);
/* ABI 8 (no reference counterpart: autograd runs each weight gradient as a launch of its own, reference src/mucon/trainers.py:131).  One-shot option of the
 * NEXT mucon_decoder_bwd on this thread (enable != 0): that call writes d_memory / d_hn / d_cn and the two gradients its step loop produces (embedding, v) as
 * always, but leaves the outer products that are every other tensor of d_params to the NEXT mucon_lstm_bwd on the same stream, which computes them in extra
 * workgroups of its backward recurrence launch (two workgroups busy for ~89 us at Tz = 125: the sums cost the step nothing there; a launch of ~13 us otherwise).
 * The same sums in the same order: bitwise the plain call's gradients.  d_params, the decoder's workspace and memory / hn / cn must stay valid until then.  Another
 * mucon_decoder_bwd, a mucon_lstm_bwd on another stream, or mucon_decoder_bwd_flush finish a pending batch in a launch of its own. */
int mucon_decoder_bwd_defer(int32_t enable);
int mucon_decoder_bwd_flush(void);

/* ---- the four MuCon losses, forward and gradients --------------------------------
 * Replaces MuCon.loss (reference src/mucon/models.py:376-565) together with project_lengths_softmax and
 * create_masks (reference src/mucon/masks.py:8-74) for one video:
 *   main = mul_transcript * transcript + mul_length * length + mul_mucon * mucon + mul_smoothing * smoothing.
 * The loss is a scalar, so one call returns the five values AND d main / d input for every input. */
typedef struct mucon_loss_cfg {
    int32_t T;                  /* frames (rows of segmentation) */
    int32_t M;                  /* classes (<= 64) */
    int32_t N;                  /* segments = number of length logits (<= 64) */
    int32_t S;                  /* transcript steps (rows of transcript_logp) */
    int32_t NC;                 /* transcript outputs (M + 1) */
    int32_t mucon_type;         /* 0 "flint", 1 "arithmetic"  (cfg.model.loss.mucon.type) */
    int32_t smoothing_clamp;    /* cfg.model.loss.smoothing.clamp */
    int32_t transcript_average; /* cfg.model.loss.transcript_average */
    float overlap;              /* cfg.model.loss.mucon.overlap */
    float clamp_min, clamp_max; /* cfg.model.loss.smoothing.clamp_min / clamp_max */
    float length_width;         /* cfg.model.loss.length_width */
    float mul_transcript, mul_length, mul_mucon, mul_smoothing; /* cfg.model.loss.mul_* */
    int32_t align_corners;      /* convention of masks.py's affine_grid / grid_sample calls (masks.py:72-73 pass none):
                                 * 1 = PyTorch <= 1.2 behaviour, i.e. the 1.1 the reference pins (docker/pytorch1.1/Dockerfile:25);
                                 * 0 = the default since PyTorch 1.3 (what the same code does under a current torch) */
} mucon_loss_cfg;
size_t mucon_loss_workspace_bytes(const mucon_loss_cfg *cfg);
/* segmentation [T][M] logits; smoothing_input [T][M] = what the smoothing loss runs on (log_softmax(segmentation)
 * when cfg.model.loss.smoothing.log_softmax_before, else the logits themselves; may alias segmentation);
 * transcript_logp [S][NC]; lengths [N]; mucon_target int64 [N]; transcript_target int64 [S]; mask_template [100]
 * (masks.py:26-41: box / gaussian / trapezoid); the class-weight vectors [M] / [NC] may be NULL (= ones).
 * Writes losses[5] = {main, transcript, length, mucon, smoothing} and the gradients of `main`:
 * d_segmentation [T][M] (mucon part), d_smoothing_input [T][M] (smoothing part; never aliases d_segmentation),
 * d_transcript_logp [S][NC], d_lengths [N + 1] (entry N is written 0: the decoder's backward takes the gradient of all S = N + 1
 * step lengths, and the last step's -- EOS -- is read by no loss). */
int mucon_loss_fwd_bwd(const mucon_loss_cfg *cfg, const float *segmentation, const float *smoothing_input,
                       const float *transcript_logp, const float *lengths, const int64_t *mucon_target,
                       const int64_t *transcript_target, const float *mask_template, const float *mucon_class_weight,
                       const float *transcript_class_weight, float *losses, float *d_segmentation,
                       float *d_smoothing_input, float *d_transcript_logp, float *d_lengths, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---- gradient clipping + SGD step (the tail of the training step) ------------------------------------
 * Replaces, for one optimizer step, reference src/mucon/trainers.py:137-140:
 *   clip_grad_norm_(model.encode_params, max_norm); clip_grad_norm_(model.decode_params, max_norm);
 *   optimizer.step()     with torch.optim.SGD(lr, momentum, weight_decay) (trainers.py:18-30)
 * Per clipping group g: norm_g = ||all gradients of the group||_2, coef = min(max_norm[g] / (norm_g + 1e-6), 1)
 * (max_norm[g] <= 0: no clipping); then per element  grad *= coef (written back),  u = grad + weight_decay * param,
 * buf = momentum * buf + u and u = buf (when momentum != 0),  param -= lr * u.   `tensors` is a HOST array;
 * the pointers in it are device pointers.  group_norms [2 * n_groups] (device, may be NULL): [g] receives the norm of group g; a clipped
 * group whose norm is not finite is NOT applied (parameters, gradients, momentum untouched -- torch would write NaN into every parameter)
 * and [n_groups + g] is incremented: a STICKY count of skipped steps the library never clears (the caller zeroes the buffer once and
 * reads / resets the counts where it synchronises anyway: ops.check_health).  (ABI 7: the buffer was [n_groups] and the only trace of a
 * skipped step was its norm, which the next healthy step overwrote.)  The same holds for mucon_clip_grads and mucon_adam_clip_step. */
typedef struct mucon_sgd_tensor {
    float *param;
    float *grad;
    float *momentum_buf; /* NULL when momentum == 0 */
    int64_t n;           /* elements */
    int32_t group;       /* clipping group, 0 .. n_groups-1 */
    int32_t reserved;
} mucon_sgd_tensor;
size_t mucon_sgd_workspace_bytes(int32_t n_tensors, int64_t total_elements);
int mucon_sgd_clip_step(int32_t n_tensors, const mucon_sgd_tensor *tensors, int32_t n_groups, const float *max_norm,
                        float lr, float weight_decay, float momentum, float *group_norms, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The clipping alone -- an iteration of a gradient-accumulation group that does not step (reference trainers.py:131-147 clips the
 * accumulated gradient at EVERY iteration and steps at the last of the group): grad *= min(max_norm[g] / (norm_g + 1e-6), 1) per group.
 * Same table and workspace as mucon_sgd_clip_step (param and momentum_buf are not touched). */
int mucon_clip_grads(int32_t n_tensors, const mucon_sgd_tensor *tensors, int32_t n_groups, const float *max_norm,
                     float *group_norms, void *workspace, size_t workspace_bytes, void *stream);

/* The same step tail with torch.optim.Adam (reference src/mucon/trainers.py:31-36: optimizer "Adam", amsgrad=True): group-wise
 * clipping as above, then per element the sequence of torch's Adam -- u = grad + weight_decay * param, exp_avg += (u - exp_avg)(1 - beta1),
 * exp_avg_sq = exp_avg_sq beta2 + (1 - beta2) u u, max_exp_avg_sq = max(., exp_avg_sq) (when the pointer is non-NULL: AMSGrad),
 * param -= lr / (1 - beta1^step) * exp_avg / (sqrt(max_exp_avg_sq or exp_avg_sq) / sqrt(1 - beta2^step) + eps).  `step` is the
 * 1-based step count AFTER this call; the state buffers are torch's optimizer.state tensors, so checkpoints stay interchangeable. */
typedef struct mucon_adam_tensor {
    float *param;
    float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    float *max_exp_avg_sq; /* NULL: plain Adam */
    int64_t n;
    int32_t group;
    int32_t reserved;
} mucon_adam_tensor;
size_t mucon_adam_workspace_bytes(int32_t n_tensors, int64_t total_elements);
int mucon_adam_clip_step(int32_t n_tensors, const mucon_adam_tensor *tensors, int32_t n_groups, const float *max_norm,
                         double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                         float *group_norms, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MUCON_HIP_H */
