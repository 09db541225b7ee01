/* mucon_hip_test.h -- test, bench and tuning hooks of libmucon_hip.so.
 *
 * NOT part of the surface that replaces the reference (that is include/mucon_hip.h): these entry points exist so
 * that tests/ can drive single kernels through the C ABI, bench.py can time the tape-streaming launches with HIP
 * events on their own stream, and regression tests can switch code paths inside one process.
 */
#ifndef MUCON_HIP_TEST_H
#define MUCON_HIP_TEST_H

#include "mucon_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Where mucon_encoder_fwd left an intermediate inside the workspace ([B][rows][128] float32):
 * kind 0 = x[layer] (input of layer `layer`; x[0] = activated first_conv output, x[n_layers] =
 * last_conv input), 1 = h[layer] (activated dilated_conv output), 2 = ypre[layer] (un-pooled output
 * of a max-pooled layer), 3 = z (last_conv output).  Lets the parity tests compare gradients on the
 * activation pattern (ReLU masks, max-pool arg-max) the kernels actually took. */
int mucon_encoder_saved_view(const mucon_encoder_cfg *cfg, int32_t kind, int32_t layer, size_t *byte_offset,
                             int32_t *rows_per_video);
/* Plain GEMM on the same MFMA core the encoder uses: out[M][128] = A[M][K] * W[128][K]^T (+bias, relu) */
int mucon_test_gemm_nt(const float *A, const float *W, const float *bias, float *out, int32_t M,
                       int32_t K, int32_t relu, void *stream);
/* out[128][K] = Y[M][128]^T * X[M][K] through the weight-gradient core (slabs + reduce). */
int mucon_test_gemm_tn(const float *Y, const float *X, float *out, int32_t M, int32_t K,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The dropout keep-mask (1 = kept) for `site`, written as uint8 [n]. */
int mucon_test_dropout_mask(uint8_t *mask, int64_t n, uint64_t seed, int32_t site, float p, void *stream);
/* Times `iters` launches of the first-conv forward kernel with HIP events on `stream`;
 * returns the average milliseconds per launch in *ms (synchronises the stream). */
int mucon_bench_first_conv(const float *tape, const float *w, const float *b, float *out, int32_t B,
                           int32_t T, int32_t D, int32_t iters, float *ms_host, void *stream);

/* first_conv forward on the bf16 MFMA with exactly split operands (csrc/gemm_split.hpp), whatever the size
 * threshold of mucon_encoder_fwd says: out[B][T][128] = act(tape[B][T][D] * w[128][D]^T + b).  `planes` receives
 * the three bf16 planes of w (3*128*D*2 bytes).  Runs 1 + iters launches; *ms_host (may be null) = average
 * milliseconds of the timed ones (HIP events on `stream`; synchronises). */
int mucon_test_first_conv_split(const float *tape, const float *w, const float *b, float *out, int32_t B,
                                int32_t T, int32_t D, int32_t relu, void *planes, size_t planes_bytes,
                                int32_t iters, float *ms_host, void *stream);

/* Per-launch timing of the two kernels that stream the tape, taken with HIP events on the stream
 * the kernels run on, while the normal fwd/bwd calls execute (bench.py's roofline leg):
 * slot 0 = first_conv forward, slot 1 = the weight-gradient launch (the one batched launch of every layer's and
 * first_conv's weight gradients; first_conv's alone with MUCON_TN_BATCH<2).  begin() arms up to
 * max_records launches per slot; end() synchronises on the recorded events and returns the
 * summed milliseconds and the launch count per slot (arrays of 2). */
int mucon_profile_begin(int32_t max_records);
int mucon_profile_end(float *total_ms_host, int32_t *count_host);
/* Time only every `every`-th launch of a slot (default 1: all).  A recorded event pair costs two ~6 us bubbles on the stream, which
 * a bench that times EVERY launch adds to every step it reports; sampled launches are timed exactly as before. */
int mucon_profile_stride(int32_t every);

/* Sets one tuning / regression knob by its environment name (e.g. "MUCON_TN_SPLIT", "0"), with the parsing the
 * environment gets when the library is first used.  Affects later calls; workspaces sized before a change that
 * needs more slab space make those calls fail with MUCON_E_WORKSPACE, never overrun. */
int mucon_test_set_knob(const char *name, const char *value);
/* The current value of MUCON_MFMA16, MUCON_TN_SPLIT or MUCON_FIRST_CONV_SPLIT (what a test restores after forcing another); -1: not readable. */
int mucon_test_get_knob(const char *name);

/* Timing builds only (MUCON_HIPCC_FLAGS=-DFS_STAMP=1): the s_memtime sums fs_kernel's block 0 left behind, [64 variants][8 waves][8
 * phases] (gemm_fused_split.hpp); returns MUCON_E_ARG in a normal build. */
int mucon_test_read_stamps(long long *out, int32_t n);

/* Box calibration (bench.py `box_calibration`; csrc/probe.hpp): 1 + `launches` launches of a bare bf16 MFMA loop (1,024 workgroups of four waves,
 * `iters` x 16 v_mfma_f32_32x32x16_bf16 -- or, with shape16 = 1, the same FLOPs as 16x16x32; shape16 = 2: the 32x32x16 loop with one operand of every MFMA
 * re-read from LDS by ds_read_b128 -- per wave, pseudo-random operands in registers) on
 * `stream`: *tflops_host = sustained dense bf16 TFLOP/s over the timed launches (HIP events), *clock_ghz_host = the in-kernel shader clock of the
 * last launch (delta s_memtime / delta s_memrealtime, median over workgroups), *ms_host = the timed launches' milliseconds.  `scratch`: >= 16,448
 * bytes of device memory.  Synchronises the stream. */
int mucon_test_mfma_probe(int32_t shape16, int32_t launches, int32_t iters, void *scratch, size_t scratch_bytes, float *tflops_host,
                          float *clock_ghz_host, float *ms_host, void *stream);

/* Diagnostic builds only (MUCON_HIPCC_FLAGS=-DCLK_STAMP=1): per workgroup of the last launches of slot 0 (first_conv's split-bf16 kernel) or
 * slot 1 (the split-bf16 weight-gradient launch) the pair (shader cycles, 100 MHz ticks) spent in the kernel's main loop, [4096][2]; returns
 * the number of workgroup records (0 in a normal build, -1 on a bad argument).  In-kernel clock = cycles / ticks x 100 MHz.
 * slot 2: per workgroup of the last weight-gradient launch the absolute s_memrealtime ticks at its entry and exit; slot 3: behind its job lookup and
 * at its first tile; slot 4: behind its last tile (second word: 1). */
int mucon_test_read_clock(int32_t slot, long long *out, int32_t n);

/* Timing builds only (MUCON_HIPCC_FLAGS=-DCS_STAMP=1): the phase stamps of the cs_kernel launches since the last call (at most 64; csrc/gemm_coarse_split.hpp):
 * stamps [64 launches][2 workgroups: first, middle of the grid][4 waves][12] = nine s_memtime values, then s_memrealtime at entry and at the end; info
 * [64][8] = BWD, POOL, TAPS, ONE, row blocks, grid x, grid y, rows per video.  Returns the number of launches recorded (0 in a normal build). */
int mucon_test_read_cs_stamps(long long *stamps, int32_t *info, int32_t n_slots);

/* Host-side phases of the LAST mucon_viterbi_decode_host call, in microseconds (steady_clock): [0] argument scan + staging set-up
 * (the job table, and the memcpy of every video's transcript and length table into the pinned input buffer), [1] the launches,
 * [2] waiting for the device (flag spin or stream synchronisation), [3] copying the results out of the pinned output buffer. */
int mucon_test_vit_host_phases(double *us4);

/* The length rows the Viterbi kernels build from a [3][N] PoissonModel parameter block and the shared log-factorial row [J]
 * (include/mucon_hip.h, mucon_viterbi_decode_host_poisson): out [J][N] f64.  All pointers DEVICE. */
int mucon_test_vit_rows(const double *poisson_params, const double *log_fact, int32_t N, int32_t J, int32_t fs, int32_t max_len,
                        double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MUCON_HIP_TEST_H */
