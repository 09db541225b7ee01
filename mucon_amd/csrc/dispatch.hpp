// Which kernel a launch takes, in one place: the fixed dispatch constants of the host code (mucon_hip.hip) and of the launch
// helpers in the kernel headers.  The table itself -- shape -> kernel, and why -- is DESIGN.md section 3.1.
//
// Two kinds of values:
//   * `constexpr`: measured once, not switchable (round 3 turned the knobs no test or tool sets into these);
//   * `extern int g_*` (defined in mucon_hip.hip): the few run-time switches the TESTS use to force a kernel family onto shapes it
//     would not take by default (tests/test_gpu_dense.py, test_gpu_tn_split.py, test_gpu_viterbi.py), read from the
//     environment once (MUCON_*) or set through mucon_test_set_knob.
#pragma once

// ---- weight gradients (gemm_tn_split.hpp / gemm_tn.hpp) -----------------------------------------------------------------
constexpr int kTnBatchTarget = 64;      // time chunks are sized for about this many workgroups per job of the batched launch (r3: 128 -> 64
                                        // halves the layer jobs' slabs; the slab reduction 20 -> 12 us, the launch itself unchanged)
constexpr int kTnTarget = 256;          // ... and per stand-alone launch (mucon_linear_bwd, mucon_conv128_wgrad)
constexpr int kTnMcCap = 2048;          // longest time chunk of an f32 weight-gradient workgroup
constexpr int kTnBatchKs = 2;           // the f32 batched launch runs 8-wave workgroups (waves 4-7: second half of every 32-step tile)
constexpr int kTnKs = 0;                // stand-alone f32 launches: 0 = k-split 2 for layer jobs, 1 for first_conv's
constexpr int kTsMaxWorkgroups = 256;    // static-runs launch: persistent workgroups = min(CUs, this); the slab arena is sized for it
constexpr int kReduceLanes = 4;         // slab lanes per workgroup of the batched slab reduction
constexpr int kReduceChunks = 4;        // ... and 256-element chunks per workgroup of a pass of >= 1,024 chunks (r6)

// ---- NT / two-stage layer kernels ---------------------------------------------------------------------------------------
constexpr int kFirstConv8w = 1;         // first_conv forward on the f32 MFMA: 128-row tiles with 8 waves (0.160 -> 0.153 ms)
constexpr int kFusedBm = 0;             // f32 two-stage kernel: 0 = tile height by level size
constexpr int kFusedKs = 1;             // ... without the in-workgroup k-split (measured: not faster; changes the summation order)
constexpr int kCsRb2Workgroups = 256;   // coarse kernel: 32 rows per workgroup where 16-row workgroups would number more than this (one per CU)
constexpr int kFsNw = 0;                // split two-stage kernel: 0 = 8 waves where that gives >= 256 workgroups, else 4
constexpr long kFuseMaxRows = 1L << 40; // the two-stage kernels take every level (no row limit)
constexpr int kNtSplitDgrad0 = 1;       // layer 0's dilated-conv data gradient on gemm_split.hpp (50.6 -> 39.8 us)
constexpr int kPoolFuse = 1;            // pooled boundaries of the backward inside the two-stage launch (POOL = 3 / 4)
constexpr int kUnpoolFuse = 1;          // the max-pool backward as the epilogue of the launch that produces the pooled level's gradient
