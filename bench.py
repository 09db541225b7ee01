#!/usr/bin/env python3
"""Benchmark of the MuCon temporal hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) this process IS one rank.  Started bare (`python bench.py --gpus N`, WORLD_SIZE unset) it
is the LAUNCHER: before anything touches the GPU it starts N fresh child processes of itself with those variables set (rendezvous
on 127.0.0.1, a free port), waits for them, lets rank 0's single JSON line through on stdout and exits non-zero if any child did.
The launcher never initialises the GPU and never replaces itself (no exec).

Metric (BASELINE.json): frames/sec of encoder + y-head forward+backward on Breakfast-I3D-shaped
tapes (T x 2048), plus Viterbi ms/video as extra fields.  One step = one pass of the hot path over
one batch of synthetic tapes that are already resident in HBM:
    encoder fwd (training mode, dropout on) -> y-head fwd (log-softmax) -> dL/dlogp given ->
    y-head bwd -> encoder bwd -> [N>1: RCCL all-reduce of the flat gradient] -> SGD update.
Workload: BASELINE config 3 shape per GPU (B=8 videos x T=4096 frames x D=2048, 48 classes).  FOUR distinct
tape batches (4 x 256 MiB = 1 GiB) are rotated, one per step: between two uses of a batch more than 768 MiB of
other tape (plus every activation) has streamed through the 256 MiB Infinity Cache, so every tape read comes
from HBM.  Weak scaling: every rank has its own batches; videos are independent, the only exchange is the
gradient all-reduce.

Timing: W warm-up steps, then the region of EXACTLY K steps (barrier + synchronize on both sides, max over ranks)
is timed `--repeats` times back to back (default 5); `value` / `ms_per_step` are the MEDIAN region's, every region
is listed in `ms_per_step_repeats` (a single 20-step region is 18 ms: too short to be stable against clock ramps).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline          dominant kernel (the one batched weight-gradient launch), timed per launch with HIP events on its stream
  roofline_viterbi  Viterbi decode against its algorithmic bytes (single video and 256 videos in flight)
  cpu_baseline      the dense path as the torch module graph the reference runs, fp32, physical host cores, best of 3
  viterbi           decode latency vs the C oracle on one host core and on all of them
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense f32-input MFMA peak (= f32 vector peak), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0          # HBM3E spec
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (first_conv forward issues six bf16 MFMAs per fp32 product block)
BYTES_PER_FRAME_FWD_BWD = 16768  # tape read twice (fwd + first-conv dW) + log-probs written + their grad read


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps; the median is reported")
    ap.add_argument("--tapes", type=int, default=4, help="distinct tape batches rotated through the steps (4 x 256 MiB)")
    ap.add_argument("--batch", type=int, default=8, help="videos per GPU per step")
    ap.add_argument("--frames", type=int, default=4096, help="frames per video")
    ap.add_argument("--time-every", type=int, default=8, help="HIP-event timing of the two tape-streaming kernels on every n-th step "
                    "(an event pair opens two ~6 us bubbles on the stream: timing every step taxes the step it reports)")
    ap.add_argument("--prewarm-steps", type=int, default=100, help="untimed steps in front of the --warmup steps (the first ~50 steps after a cold start run "
                    "slower); the random-init weights are restored behind them")
    ap.add_argument("--keep-drift", action="store_true", help="EXPERIMENT: do NOT restore the random-init weights behind the pre-warm steps and in front of "
                    "every timed region.  The step's speed depends on the VALUES flowing through it (the MFMA kernels run against the power cap; the synthetic objective has no minimum and drives the weights to NaN): after ~500 SGD "
                    "steps on synthetic noise the same launches are up to 9 %% faster (0.77 -> 0.70 ms per step on a power-limited box); with MUCON_BENCH_LR=0 "
                    "they are not, and GEMMs of other data (--prewarm-seconds) change nothing")
    ap.add_argument("--prewarm-seconds", type=float, default=0.0, help="EXPERIMENT: seconds of other work (bf16 GEMMs + a tape-sized copy) in front of the warm-up steps")
    ap.add_argument("--drain-every", type=int, default=0, help="synchronise the stream every n steps INSIDE the timed regions (0: never): bounds how far "
                    "the host runs ahead of the GPU -- with several hundred launches queued the HIP runtime stalls for milliseconds at a time")
    ap.add_argument("--no-calibration", action="store_true", help="skip the box calibration probes behind the timed regions (~1 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two short rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic on this box; "
                                                                   "the committed profile's figure is reported instead, labelled as such")
    ap.add_argument("--no-viterbi", action="store_true")
    return ap.parse_args()


def make_params(spec, C, dev):
    """Random-init parameters with the reference's initialisation (nn.Conv1d / nn.GroupNorm defaults),
    keyed by the reference's state_dict names."""
    import torch.nn as nn
    from mucon_amd import ops

    torch.manual_seed(1234)
    mods = {"ft.first_conv": nn.Conv1d(spec.in_dim, spec.hidden, 1)}
    for i, d in enumerate(spec.stages):
        mods[f"ft.l_{i}.dilated_conv"] = nn.Conv1d(spec.hidden, spec.hidden, 3, dilation=d, padding=d)
        mods[f"ft.l_{i}.conv_1x1"] = nn.Conv1d(spec.hidden, spec.hidden, 1)
    mods["ft.last_conv"] = nn.Conv1d(spec.hidden, spec.hidden, 1)
    mods["ft_last_gn"] = nn.GroupNorm(spec.last_gn_num_groups, spec.hidden)
    mods["conv_classifier"] = nn.Conv1d(spec.hidden, C, 1)
    sd = {}
    for k, m in mods.items():
        sd[k + ".weight"], sd[k + ".bias"] = m.weight.detach(), m.bias.detach()
    names = ops.param_names(spec) + ["conv_classifier.weight", "conv_classifier.bias"]
    return names, [sd[k].to(dev).contiguous().requires_grad_(True) for k in names]




def box_calibration(lib, dev):
    """What THIS box sustains, measured outside the timed regions (about 0.3 s per figure): a bare all-CU bf16 MFMA loop on
    pseudo-random operands (csrc/probe.hpp; both MFMA shapes) with its in-kernel shader clock, and a 1 GiB device-to-device copy.
    The pool's boards differ by up to ~10 % on MFMA-dense kernels (power management); these figures let two bench lines from two
    boxes be compared."""
    from mucon_amd import _lib

    scratch = torch.empty(32768, dtype=torch.uint8, device=dev)
    tf, ghz, ms = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
    s = _lib.current_stream_ptr()
    out = {}
    for name, shape16 in (("32x32x16", 0), ("16x16x32", 1), ("32x32x16_lds", 2)):
        _lib.check(lib.mucon_test_mfma_probe(shape16, 10, 2000, _lib.ptr(scratch), scratch.numel(), ctypes.byref(tf), ctypes.byref(ghz),
                                             ctypes.byref(ms), s), "mfma_probe")            # ramp
        _lib.check(lib.mucon_test_mfma_probe(shape16, 60, 2000, _lib.ptr(scratch), scratch.numel(), ctypes.byref(tf), ctypes.byref(ghz),
                                             ctypes.byref(ms), s), "mfma_probe")
        out[f"mfma_tflops_{name}"] = round(tf.value, 1)
        out[f"clock_ghz_{name}"] = round(ghz.value, 3)
        out[f"probe_seconds_{name}"] = round(ms.value * 1e-3, 3)
    out["mfma_tflops"], out["clock_ghz"] = out["mfma_tflops_32x32x16"], out["clock_ghz_32x32x16"]
    # A yardstick that loads the box the way the path does (tape from HBM + fragments from LDS + split + MFMAs, all CUs): round 2's first_conv
    # kernel (gemm_split.hpp: nt_split_kernel, the 32x32x16 body -- unchanged since round 2, not the kernel the path runs by default) on one
    # 268 MB tape, 300 back-to-back launches.  The bare loops above read within 1 % on boxes whose steps differ by 7 %; this one follows them.
    g = torch.Generator(device=dev).manual_seed(77)
    tape = torch.randn(8, 4096, 2048, device=dev, generator=g)
    W = torch.randn(128, 2048, device=dev, generator=g) * 0.02
    b = torch.randn(128, device=dev, generator=g)
    o = torch.empty(8, 4096, 128, device=dev)
    planes = torch.empty(3 * 128 * 2048 * 2, dtype=torch.uint8, device=dev)
    keep = int(lib.mucon_test_get_knob(b"MUCON_MFMA16"))
    try:
        _lib.set_knob("MUCON_MFMA16", keep & ~1)
        for iters in (100, 300):
            _lib.check(lib.mucon_test_first_conv_split(_lib.ptr(tape), _lib.ptr(W), _lib.ptr(b), _lib.ptr(o), 8, 4096, 2048, 1, _lib.ptr(planes),
                                                       planes.numel(), iters, ctypes.byref(ms), s), "first_conv_split")
    finally:
        _lib.set_knob("MUCON_MFMA16", keep)
    out["yardstick_first_conv_r2_us"] = round(ms.value * 1e3, 2)
    del tape, W, b, o, planes
    n = 1 << 28                                      # 1 GiB of float32
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    out["copy_gbs"] = round(20 * 2.0 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)     # bytes read + bytes written
    out["note"] = ("mfma_tflops / clock_ghz: 60 launches of 1,024 workgroups x 4 waves x 2,000 x 16 v_mfma_f32_32x32x16_bf16 on pseudo-random operands in registers "
                   "(the *_16x16x32 pair: the same FLOPs on the narrower shape; *_32x32x16_lds: one operand of every MFMA re-read from LDS by ds_read_b128), HIP events; clock = delta s_memtime / delta s_memrealtime inside the last launch, "
                   "median over workgroups; copy_gbs: 20 x 1 GiB device-to-device torch copy, read + written bytes")
    del src, dst
    return out


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(spec, C, T, B=8):
    """The dense hot path as the torch MODULE graph the reference runs (nn.Conv1d on [B, C, T], max_pool1d, GroupNorm,
    interpolate, log_softmax: oracle/dense.py:module_graph), fp32, fwd+bwd on one batch of the bench's own
    shape (B=8 x T=4096), one thread per PHYSICAL core, best of 3 passes after a warm-up pass."""
    from oracle import dense as od

    ocfg = od.EncoderConfig(in_dim=spec.in_dim, hidden=spec.hidden, num_classes=C, stages=list(spec.stages),
                            pooling=spec.pooling, pooling_type=spec.pooling_type, pooling_layers=list(spec.pooling_layers),
                            leaky_relu=spec.leaky_relu, last_gn=spec.last_gn, last_gn_num_groups=spec.last_gn_num_groups,
                            last_relu=spec.last_relu)
    cores = physical_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        torch.manual_seed(5)
        model = od.module_graph(ocfg)
        tape = torch.randn(B, T, spec.in_dim)
        w = torch.randn(B, C, T)
        times = []
        for it in range(4):
            model.zero_grad()
            t0 = time.perf_counter()
            (w * model(tape)).sum().backward()
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
    finally:
        torch.set_num_threads(prev)
    return {"value": round(B * T / best, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/dense.py:module_graph (nn.Conv1d / max_pool1d / GroupNorm / interpolate / log_softmax) fwd+bwd fp32, "
                      f"B={B} x T={T} x D={spec.in_dim}, best of 3 passes ({', '.join(f'{t:.2f}' for t in times[1:])} s) after one warm-up, "
                      f"{cores} threads = physical cores; the module graph does not scale with cores (8 cores give about the same rate: "
                      f"SURVEY.md 6) -- a stated baseline, not a tuned CPU implementation"}


def _cpu_viterbi_all_cores(lp_h, tr, P, fs, max_len, n_videos, cores):
    """ms per video with `cores` host threads decoding `n_videos` videos (the C oracle is a ctypes call: no GIL held)."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor

    lps = [lp_h.copy() for _ in range(min(n_videos, 32))]     # distinct buffers (the decode only reads them)
    with ThreadPoolExecutor(max_workers=cores) as pool:
        list(pool.map(lambda i: oracle.viterbi_decode_table(lps[i % len(lps)], tr, P, fs, max_len), range(cores)))   # warm the pool
        t0 = time.perf_counter()
        list(pool.map(lambda i: oracle.viterbi_decode_table(lps[i % len(lps)], tr, P, fs, max_len), range(n_videos)))
        return (time.perf_counter() - t0) / n_videos * 1e3


def viterbi_bench(dev, C=48):
    """Breakfast-typical (T=2000, N=6) and BASELINE config 5 (T=16384, 64-state transcript): single-stream latency and amortised
    latency with many videos per launch; beside them the C oracle on ONE host core and on ALL physical cores (one video per
    thread), which is the CPU figure the amortised GPU number has to be read against."""
    import oracle
    from mucon_amd import ops
    from mucon_amd.core.viterbi import PoissonModel

    fs, max_len = 30, 2000
    cores = physical_cores()
    g = torch.Generator(device="cpu").manual_seed(7)
    out = {"cpu_cores": cores}

    def timed(fn, reps, rounds=5):
        """seconds per call: the median over `rounds` of the mean over `reps` calls (one allocator or host hiccup of tens of
        milliseconds inside a single round of 50 calls otherwise multiplies a 0.1 ms figure by eight)"""
        per = []
        for _ in range(rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / reps)
        return sorted(per)[len(per) // 2]

    from mucon_amd.core.viterbi import poisson_params_for_many

    def decoder(tr, mu):
        """ops.viterbi_decode_batch as the evaluator calls it since round 6 (mucon/evaluators.py): the PoissonModel's [3, N] parameter block per
        video + the shared log-factorial row, the length scores built on the device (ABI 7) -- and the host-built [J, N] table form beside it."""
        pp = poisson_params_for_many([mu], [tr], fs, max_len)[0]
        return (lambda lps, **kw: ops.viterbi_decode_batch(lps, [tr] * len(lps), [pp.params] * len(lps), fs, max_len, log_fact=pp.log_fact, **kw),
                PoissonModel(mu).rows_for(tr, fs))

    # a Breakfast-typical video first (T ~ 2000 frames, 6 actions), single stream
    T, N = 2000, 6
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    dec, P = decoder(tr, mu)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for _ in range(3):
        dec([lp])
    out["ms_per_video_T2000_N6_single"] = round(timed(lambda: dec([lp]), 20) * 1e3, 4)
    lps256 = [torch.log_softmax(3 * torch.randn(T, C, device=dev), dim=1) for _ in range(256)]   # 256 videos in flight
    dec(lps256)
    out["ms_per_video_T2000_N6_batch256"] = round(timed(lambda: dec(lps256), 2) / 256 * 1e3, 5)
    del lps256
    lp_h = lp.cpu().numpy()
    best = float("inf")
    for _ in range(20):
        t0 = time.perf_counter()
        oracle.viterbi_decode_table(lp_h, tr, P, fs, max_len)
        best = min(best, time.perf_counter() - t0)
    out["cpu_oracle_ms_per_video_T2000_N6"] = round(best * 1e3, 4)
    out["cpu_oracle_all_cores_ms_per_video_T2000_N6"] = round(_cpu_viterbi_all_cores(lp_h, tr, P, fs, max_len, 16 * cores, cores), 5)
    out["reference_python_ms_per_video_T2000_N6"] = 68.0   # measured in the build container, context only
    bytes_small = T * C * 4 + T * 4

    # BASELINE.md section 3's two other planned decodes, 64 in flight each
    for (Tq, Nq) in ((6000, 10), (10000, 30)):
        trq = torch.randint(0, C, (Nq,), generator=g).numpy().astype(np.int32)
        muq = np.ones(C)
        muq[np.unique(trq)] = Tq / Nq
        decq, _ = decoder(trq, muq)
        lpq = [torch.log_softmax(3 * torch.randn(Tq, C, device=dev), dim=1) for _ in range(64)]
        decq(lpq)
        out[f"ms_per_video_T{Tq}_N{Nq}_batch64"] = round(timed(lambda: decq(lpq), 2, 3) / 64 * 1e3, 5)
        decq(lpq[:1])
        out[f"ms_per_video_T{Tq}_N{Nq}_single"] = round(timed(lambda: decq(lpq[:1]), 10, 3) * 1e3, 4)
        del lpq

    T, N = 16384, 64
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    dec, P = decoder(tr, mu)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for label, nv, reps in (("single", 1, 5), ("batch64", 64, 2), ("batch256", 256, 1)):
        if nv == 1:
            lps = [lp]
        elif nv == 64:
            lps = base64 = [torch.log_softmax(3 * torch.randn(T, C, device=dev), dim=1) for _ in range(64)]
        else:
            lps = base64 * (nv // 64)          # 256 videos in flight (the 64 emission tensors four times over)
        dec(lps)  # warm-up
        out[f"ms_per_video_{label}"] = round(timed(lambda: dec(lps), reps, 3) / nv * 1e3, 4)
        if nv > 1:
            # the same call with host-built [J, N] length tables (rounds 1-5: 66 doubles per transcript state over PCIe instead of 3)
            ops.viterbi_decode_batch(lps, [tr] * nv, [P] * nv, fs, max_len)
            out[f"ms_per_video_{label}_host_tables"] = round(timed(lambda: ops.viterbi_decode_batch(lps, [tr] * nv, [P] * nv, fs, max_len), reps, 3) / nv * 1e3, 4)
            # ... with the per-frame labels leaving the GPU (uint8 / the reference's int32, written by the kernels into pinned
            # host memory), and with the segments expanded to int32 labels on the host for EVERY video (ViterbiResult.labels)
            for fmt in ("uint8", "int32"):
                dec(lps, labels=fmt)
                out[f"ms_per_video_{label}_{fmt}_labels"] = round(timed(lambda: dec(lps, labels=fmt), reps, 3) / nv * 1e3, 4)
            out[f"ms_per_video_{label}_expanded_on_host"] = round(timed(lambda: [r.labels for r in dec(lps)], reps, 3) / nv * 1e3, 4)
    lp_h = lp.cpu().numpy()
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.viterbi_decode_table(lp_h, tr, P, fs, max_len)
        best = min(best, time.perf_counter() - t0)
    out["cpu_oracle_ms_per_video"] = round(best * 1e3, 3)
    out["cpu_oracle_all_cores_ms_per_video"] = round(_cpu_viterbi_all_cores(lp_h, tr, P, fs, max_len, 4 * cores, cores), 4)
    out["config"] = (f"ms_per_video_single/batch64/batch256 and cpu_oracle_*: BASELINE config 5, T={T}, N={N}, C={C}, fs={fs} "
                     f"(K=546 columns, 64x66 hypotheses); every GPU timing is the whole ops.viterbi_decode_batch call: job table / transcripts / the PoissonModel's [3, N] parameter blocks read by the kernels from pinned host memory (the length scores are built on the device, bit for bit: ABI 7; *_host_tables: the [J, N] tables built on the host as in rounds 1-5), results written there (no copy calls).  "
                     f"The decode's result is the segmentation (score, segment lengths): ms_per_video_* without a suffix is that call (label format 'lazy': no per-frame labels leave the GPU, ViterbiResult.labels expands the segments on access); "
                     f"*_uint8_labels / *_int32_labels: the kernels also write the per-frame labels (T or 4 T bytes per video over PCIe); *_expanded_on_host: the lazy call plus the int32 expansion of every video on the host; "
                     f"cpu_oracle_all_cores_*: {cores} threads, one video each, {cores} physical cores")
    out["algorithmic_bytes_per_video"] = T * C * 4 + T * 4
    out["algorithmic_bytes_per_video_T2000_N6"] = bytes_small
    return out


def viterbi_roofline(v):
    """Viterbi against its algorithmic bytes (emissions read once + labels written: SURVEY.md 8d): a latency-bound DP, so the
    fraction of the HBM peak is tiny by construction; it is reported, not hidden."""
    def leg(ms, nbytes):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"ms_per_video": ms, "achieved": round(gbs, 3), "frac": round(gbs / PEAK_HBM_GBS, 6)}
    b5, b2 = v["algorithmic_bytes_per_video"], v["algorithmic_bytes_per_video_T2000_N6"]
    return {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS, "kernel": "single video: viterbi_fused_kernel (T=2000/N=6: one launch, the DP under the frame-score chain) / viterbi_pair_kernel (config 5: both phases as two workgroups of one launch); batches: viterbi_framescore_cols_kernel + viterbi_dp_lanes_kernel",
            "config5_T16384_N64": {"algorithmic_bytes_per_video": b5, "single": leg(v["ms_per_video_single"], b5),
                                   "batch256": leg(v["ms_per_video_batch256"], b5)},
            "T2000_N6": {"algorithmic_bytes_per_video": b2, "single": leg(v["ms_per_video_T2000_N6_single"], b2),
                         "batch256": leg(v["ms_per_video_T2000_N6_batch256"], b2)}}


def end_to_end_bench(dev, steps=40):
    """Second timed scope of SURVEY.md 8d: the reference's whole training step (_train_1_batch: forward incl. the
    s-head, losses, backward, two clip_grad_norm_, SGD) on one Breakfast-typical synthetic video,
    batch size 1 as in the reference (README: 14.67-16.23 it/s on the authors' GPU)."""
    from mucon_amd import synth
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.core.datasets import Batch
    from mucon_amd.mucon.models import create_model
    from mucon_amd.mucon.trainers import SimpleTrainer

    T, N, C = 2000, 6, 48
    cfg = update_config(get_cfg_defaults(), [], [])
    torch.manual_seed(0)
    model = create_model(cfg, C, 31, 2048).to(dev)
    trainer = SimpleTrainer(cfg, model, dev)
    trainer.on_start_epoch(0)
    model.train()
    tr = synth.transcript(3, N, C, allow_repeats=False)
    batch = Batch(feats=torch.randn(1, T, 2048), gt_label=torch.from_numpy(synth.segment_labels(4, T, tr)),
                  transcript=torch.from_numpy(tr), transcript_tf_input=torch.tensor([C + 1] + tr.tolist()),
                  transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name="synthetic").to(dev)
    for i in range(20):
        trainer._train_1_batch(i, batch)
    rounds = []                      # median of five rounds: one host hiccup does not decide the figure
    for r in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            trainer._train_1_batch(20 + r * steps + i, batch)
        torch.cuda.synchronize()
        rounds.append((time.perf_counter() - t0) / steps)
    dt = sorted(rounds)[2]
    trainer.check_health()          # raises on a decoder hand-over failure / a non-finite gradient norm met during the timed steps
    finite = bool(all(torch.isfinite(p_).all().item() for p_ in model.parameters()))     # (NaN operands would have run faster: see the hot-path leg)
    return {"videos_per_s": round(1.0 / dt, 1), "frames_per_s": round(T / dt, 1), "ms_per_video": round(dt * 1e3, 3),
            "ms_per_video_rounds": [round(r * 1e3, 3) for r in rounds], "weights_finite_after_last_round": finite,
            "config": f"full MuCon train step, batch 1, T={T}, N={N}: all-HIP, graph-free: encoder, s-head (persistent LSTM / decoder), y-head, fused losses, backward, fused clip+SGD as one straight line of launches (MuCon.fused_train_step)",
            "reference_readme_it_per_s": [14.67, 16.23]}


def eval_bench(dev, n_videos=32, rank=0, world=1, sync=None, allmax=None, T=2000, max_words=8):
    """Evaluation scope: MuConEvaluator.evaluate() per test video = eval-mode forward (greedy s-head decode, at most 8
    words here), predict, Viterbi decode on the device-resident log-probs, and the reference's full metric set
    (reference src/mucon/evaluators.py:121-257) -- batched: one chunk of 32 videos = every forward enqueued, ONE copy of the
    transcripts back, ONE Viterbi launch, ONE metrics launch (MoF / IoD / IoU / edit / F1 on the device); the record equals the
    one-video-at-a-time path's bit for bit (tests/test_gpu_eval_batched.py).  Random-init weights; the EOS logit is biased down so that the greedy
    decode emits a transcript (an untrained s-head emits EOS first, on which the reference's evaluator fails as well)."""
    from mucon_amd import synth
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.core.datasets import Batch
    from mucon_amd.mucon.evaluators import MuConEvaluator
    from mucon_amd.mucon.models import create_model

    N, C = 6, 48
    cfg = update_config(get_cfg_defaults(), [], [])
    if str(dev) == "cpu":
        cfg.defrost()
        cfg.system.device = "cpu"        # (the launcher's self-test: the reference's own device switch, mucon_amd/cpu_plumbing.py)
        cfg.freeze()
    torch.manual_seed(0)                 # every rank builds the same model
    model = create_model(cfg, C, max_words, 2048).to(dev)
    with torch.no_grad():
        model.fs_decoder_transcript[2].bias[C] = -20.0
    total = n_videos * world             # N > 1: the evaluator shards the test videos (video i on rank i mod world); n_videos per rank

    class Videos:
        background_class_ids = [0]

        def __init__(self):
            self.items = []
            for v in range(total):
                tr = synth.transcript(100 + v, N, C, allow_repeats=False)
                self.items.append(Batch(feats=torch.randn(1, T, 2048), gt_label=torch.from_numpy(synth.segment_labels(200 + v, T, tr)),
                                        transcript=torch.from_numpy(tr), transcript_tf_input=torch.tensor([C + 1] + tr.tolist()),
                                        transcript_tf_target=torch.tensor(tr.tolist() + [C]), video_name=f"v{v}").to(dev))

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

        def get_num_classes(self):
            return C

    db = Videos()
    ev = MuConEvaluator(cfg, db, model, dev)
    ev.viterbi_mode(True)
    if sync is None:
        sync = torch.cuda.synchronize
    ev.evaluate(rank, world)
    sync()
    # (r6) three timed passes, the median reported: a pass over these 32 videos is ~10 ms, and the second pass after process start still pays for the caching
    # allocator's pools of the side streams growing (0.37 ms per video against 0.27 - 0.30 from the third pass on: a real test set is hundreds of videos)
    passes = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = ev.evaluate(rank, world)
        sync()
        dt = time.perf_counter() - t0
        if allmax is not None:
            dt = allmax(dt)                  # the slowest rank's time for the whole sharded pass
        passes.append(dt / len(db))
    dt = sorted(passes)[1]
    out = {"videos_per_s": round(1.0 / dt, 1), "ms_per_video": round(dt * 1e3, 3), "ms_per_video_passes": [round(x * 1e3, 3) for x in passes]}
    if world > 1 or allmax is not None:
        out.update({"n_gpus": world, "videos": len(db), "videos_per_rank": n_videos, "skipped_videos": int(res.get("skipped_videos", 0)),
                    "sharding": "video i on rank i mod world (MuConEvaluator.evaluate(rank, world)); metric accumulators all-reduced as one vector, per-video records "
                                "gathered; ms_per_video = slowest rank's time for the pass / all videos"})
    out["config"] = (f"MuConEvaluator.evaluate(): {len(db)} videos, T={T}: eval forward + greedy decode ({max_words} words) + predict + "
                      f"Viterbi (fs=30) + MoF/IoD/IoU/edit/F1 for y-, s- and Viterbi segmentations; tapes resident in HBM; batched "
                      f"(chunks of {ev.chunk_videos} videos: pooled round trips, one Viterbi launch, device metrics)")
    return out


def viterbi_bench_sharded(rank, world, sync, allmax, decode_batch, make_videos, videos=256, rounds=3):
    """BASELINE.json's "Viterbi ms/video at 1/2/4/8 GPU": decoding shards over videos with nothing exchanged, so at N ranks every rank
    decodes its OWN `videos` videos per call (seeded by rank), every call bracketed by (synchronize + barrier) on both sides, and
    ms_per_video = the slowest rank's call time / (world x videos) -- the aggregate rate of the job, as `value` is for the dense path.
    `decode_batch(lps, trs, Ps)`: the binding's batched call (ops.viterbi_decode_batch); `make_videos(kind, rank, videos)` ->
    (lps, trs, Ps) for kind "config5" (T = 16,384, N = 64) / "T2000_N6"."""
    out = {"n_gpus": world, "videos_per_rank_and_call": videos}
    for kind, key in (("config5", "ms_per_video_batch256"), ("T2000_N6", "ms_per_video_T2000_N6_batch256")):
        lps, trs, Ps = make_videos(kind, rank, videos)
        decode_batch(lps, trs, Ps)            # warm-up (allocations, the library's staging buffers)
        per = []
        for _ in range(rounds):
            sync()
            t0 = time.perf_counter()
            decode_batch(lps, trs, Ps)
            sync()
            per.append(allmax(time.perf_counter() - t0))
        out[key.replace("256", str(videos)) if videos != 256 else key] = round(sorted(per)[len(per) // 2] / (world * videos) * 1e3, 5)
        del lps
    out["config"] = (f"every rank decodes {videos} videos of its own per call (config 5: T=16384 / N=64; Breakfast-typical: T=2000 / N=6), whole "
                     f"ops.viterbi_decode_batch calls, median of {rounds} barrier-bracketed calls, slowest rank's time / ({world} x {videos}) videos; no collective on the data path")
    return out


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` started bare: N child processes of this script, one rank each (the variables torch.distributed.run
    would set), rendezvous on 127.0.0.1.  Rank 0 inherits stdout (its one JSON line is the launcher's), the other ranks' stdout
    goes to stderr.  Returns the exit code: 0 only if every rank returned 0.  Nothing here touches the GPU."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's buffer exchange between the ranks needs it
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    codes = [0] * n
    pending = set(range(n))
    while pending:                         # a rank that dies takes the others down (they would wait in a collective for ever)
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                codes[r] = rc
                pending.discard(r)
                if rc != 0:
                    for q in pending:
                        procs[q].terminate()
        time.sleep(0.05)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py launcher: ranks exited non-zero: {bad}", file=sys.stderr)
        return 1
    return 0


def stub_main(args, rank, world):
    """MUCON_BENCH_STUB=1 -- the launcher's and the timing protocol's self-test on a box without GPUs (tests/test_bench_launcher.py):
    the same rendezvous, warm-up, (barrier + K steps + barrier) regions, max over ranks and ONE JSON line from rank 0, with a step that
    is one gloo all-reduce of a small host buffer.  No number in its line means anything and the line says so."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.full((1024,), float(rank + 1))

    def step(i):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.mul_(1.0 / world)

    for i in range(args.warmup):
        step(i)
    regions = []
    for r in range(args.repeats):
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        dist.barrier()
        regions.append(time.perf_counter() - t0)
    t = torch.tensor(regions, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    regions = t.tolist()
    elapsed = sorted(regions)[len(regions) // 2]

    # the sharded Viterbi / evaluation legs of the N > 1 line through the same functions, on stand-ins that need no GPU: the NumPy decode of
    # mucon_amd/cpu_plumbing.py on small videos, the evaluator on cfg.system.device = "cpu"
    def allmax(x):
        t_ = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    def make_videos(kind, rank_, videos):
        from mucon_amd.core.viterbi import PoissonModel
        Tv, Nv = (600, 8) if kind == "config5" else (300, 4)
        gc = torch.Generator().manual_seed(7000 + rank_)
        tr = torch.randint(0, 48, (Nv,), generator=gc).numpy().astype(np.int32)
        mu = np.ones(48)
        mu[np.unique(tr)] = Tv / Nv
        P = PoissonModel(mu).rows_for(tr, 30)
        return [torch.log_softmax(3 * torch.randn(Tv, 48, generator=gc), dim=1).numpy() for _ in range(videos)], [tr] * videos, [P] * videos

    def decode_batch(lps, trs, Ps):
        from mucon_amd import cpu_plumbing
        return [cpu_plumbing.viterbi_decode(a, b_, c, 30, None) for a, b_, c in zip(lps, trs, Ps)]

    vit = viterbi_bench_sharded(rank, world, dist.barrier, allmax, decode_batch, make_videos, videos=4, rounds=2)
    evl = eval_bench("cpu", 2, rank, world, dist.barrier, allmax, T=160, max_words=4)
    if rank == 0:
        B, T = args.batch, args.frames
        out = {"metric": "STUB: launcher / timing-protocol self-test, no GPU work (MUCON_BENCH_STUB=1)", "value": round(world * B * T * args.steps / elapsed, 1),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub",
               "config": {"workload": "stub", "global_batch": world * B, "frames_per_video": T, "parallelism": f"dp{world}"},
               "repeats": args.repeats, "ms_per_step_repeats": [round(r / args.steps * 1e3, 4) for r in regions],
               "rccl": {"world": world, "backend": dist.get_backend(), "bytes_per_step": int(buf.numel() * 4), "collectives_per_step": 1},
               "viterbi": vit, "evaluation": evl}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def live_traffic(args):
    """roofline.traffic measured ON THIS BOX (r5; the round-4 verdict: a constant read from profiles/ cannot notice a traffic regression): two short child runs of
    this script's hot-path leg under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters alone, as MI355X_MICROARCH.md prescribes --,
    started as CHILD processes before this process touches the GPU.  Per launch of the dominant kernel and of first_conv's forward: HBM bytes = 2 x FETCH_SIZE KiB
    (the gfx950 correction of the guide: the counter reports half the bytes) + WRITE_SIZE KiB.  None when rocprofv3 is missing, this process is itself being profiled, or
    a pass fails (the committed profile's constant is reported then, `traffic_measured: false` with the reason).  -> (result or None, reason or None).
    (r6: the passes run BEHIND the timed regions -- in front of them they heated the box the regions were then timed on.)"""
    import csv, glob, shutil, subprocess, tempfile
    if args.no_traffic:
        return None, "--no-traffic"
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH")) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself being profiled"
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None, "rocprofv3 not found"
    norm = lambda k: "".join(k.split())        # (kernel names compared with all whitespace stripped: rocprofv3's demangled spelling is not a contract)
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            d = tempfile.mkdtemp(prefix="mucon_pmc_", dir="/tmp")
        except OSError as e:
            return None, f"no scratch directory: {e}"
        try:
            cmd = [tool, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                   "--repeats", "1", "--prewarm-steps", "10", "--batch", str(args.batch), "--frames", str(args.frames), "--tapes", str(args.tapes),
                   "--no-cpu-baseline", "--no-viterbi", "--no-calibration", "--no-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=120)
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            rows = sorted(csv.DictReader(open(files[0])), key=lambda x: int(x["Dispatch_Id"]))
            fwd, wg = [], []
            for x in rows:
                if x["Counter_Name"] != counter:
                    continue
                k = norm(x["Kernel_Name"])
                if "nt_split16_kernel<true,false,false,false>" in k or "nt_split_kernel<true,false,false,false>" in k:
                    fwd.append(float(x["Counter_Value"]))
                if "ts_runs_kernel" in k or "ts_batched_kernel" in k or "tn_batched_kernel" in k:
                    wg.append(float(x["Counter_Value"]))
            if not fwd or not wg:
                return None, f"the {counter} pass holds no launch of first_conv forward / the weight-gradient kernel"
            sums[counter] = (sum(fwd) / len(fwd) * 1024.0, sum(wg) / len(wg) * 1024.0, len(wg))      # counter unit: KiB
        except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
            return None, f"{type(e).__name__}: {e}"[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"first_conv_fwd": 2.0 * sums["FETCH_SIZE"][0] + sums["WRITE_SIZE"][0], "weight_gradients": 2.0 * sums["FETCH_SIZE"][1] + sums["WRITE_SIZE"][1],
            "launches_sampled": min(sums["FETCH_SIZE"][2], sums["WRITE_SIZE"][2])}, None


def batch1_dense_legs(dev, spec, C, params, with_cpu=True):
    """BASELINE.md section 3's dense legs at batch 1 -- the hot path (encoder + y-head, fwd+bwd+SGD, training mode) on ONE video of T = 2,000
    (Breakfast-typical) and T = 16,384 (BASELINE config 5's dense half: 'HBM roofline point') --, tapes resident in HBM and rotated past the
    256 MiB Infinity Cache, each with its own roofline sub-record (16,768 algorithmic bytes per frame) and the CPU module graph at
    the same shape.  Outside the headline's timed regions; the weights are the random-init ones (restored by the caller afterwards)."""
    import types
    from mucon_amd import ops
    enc_params, wc, bc = params[:-2], params[-2], params[-1]
    wc2 = wc.reshape(wc.shape[0], wc.shape[1]) if wc.dim() == 3 else wc
    sgd = ops.FusedClipSGD([params], None, types.SimpleNamespace(param_groups=[{"lr": 0.01, "weight_decay": 0.005, "momentum": 0.0}]))
    out = {}
    for T, steps in ((2000, 60), (16384, 30)):
        n_tapes = max(2, -(-(320 << 20) // (T * spec.in_dim * 4)))          # > 256 MiB of distinct tape between two reads of the same one
        g = torch.Generator(device=dev).manual_seed(4000 + T)
        tapes = [torch.randn(1, T, spec.in_dim, device=dev, generator=g) for _ in range(n_tapes)]
        dlogp = torch.randn(1, T, C, device=dev, generator=g) / T

        def step(i):
            enc, c_enc = ops.run_forward(ops._EncoderFn, tapes[i % n_tapes], spec, True, int(i), *enc_params)
            (_, logp), c_head = ops.run_forward(ops._HeadFn, enc, wc2, bc, int(T), False, True)
            c_head.defer_reduce = os.environ.get("MUCON_BENCH_DEFER_HEAD", "1") == "1"
            c_head.reuse_grads = True
            d_enc, d_w, d_b = ops.run_backward(ops._HeadFn, c_head, None, dlogp)[:3]
            wc.grad, bc.grad = d_w.view_as(wc), d_b
            c_enc.reuse_grads = True      # (a one-video step is bound by the host as much as by the GPU: the gradient views are kept, as in MuCon.fused_train_step)
            for p_, g_ in zip(enc_params, ops.run_backward(ops._EncoderFn, c_enc, d_enc)[4:]):
                p_.grad = g_
            sgd.step()

        for i in range(8):
            step(i)
        per = []
        for r in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(8 + r * steps + i)
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / steps)
        ms = sorted(per)[1] * 1e3
        fps = T / (ms * 1e-3)
        leg = {"ms_per_step": round(ms, 4), "frames_per_s": round(fps, 1), "ms_per_step_repeats": [round(x * 1e3, 4) for x in per],
               "tapes_rotated": n_tapes, "tape_bytes_resident": n_tapes * T * spec.in_dim * 4,
               "roofline": {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS, "achieved": round(fps * BYTES_PER_FRAME_FWD_BWD / 1e9, 1),
                            "frac": round(fps * BYTES_PER_FRAME_FWD_BWD / 1e9 / PEAK_HBM_GBS, 5),
                            "note": "whole step against the HBM roofline (16,768 algorithmic bytes per frame): a batch-1 step is ~50 launches of 16 - 128 "
                                    "workgroups -- latency-bound, not bandwidth-bound; the B = 8 x T = 4096 line is the throughput figure"}}
        if with_cpu:
            cb = cpu_baseline(spec, C, T, B=1)
            leg["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind")}
            leg["cpu_baseline"]["sample"] = f"oracle/dense.py:module_graph fwd+bwd fp32, B=1 x T={T} x D={spec.in_dim}, best of 3 after one warm-up"
        out[f"B1_T{T}"] = leg
        del tapes, dlogp
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))        # the launcher: N children of this script, nothing on the GPU here
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("MUCON_BENCH_STUB") == "1":
        return stub_main(args, rank, world)
    # (child processes, before this one touches the GPU; the whole-line runs only: a run with a leg switched off -- the A/B tools' -- reports the profile's constant)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("MUCON_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path even with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist

        # single node, xGMI only (north_star): keep RCCL off any InfiniBand / RoCE NIC the box may have.  Defaults only:
        # an operator's own NCCL_* settings win.
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mucon_amd import _lib, ops

    lib = _lib.load()
    spec = ops.EncoderSpec()
    C, B, T = 48, args.batch, args.frames
    names, params = make_params(spec, C, dev)
    enc_params, wc, bc = params[:-2], params[-2], params[-1]
    flat_params = params
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    # resident in HBM; several distinct batches, rotated, so that no tape read is served by the 256 MiB Infinity Cache
    tapes = [torch.randn(B, T, spec.in_dim, device=dev, generator=g) for _ in range(max(1, args.tapes))]
    dlogp = torch.randn(B, T, C, device=dev, generator=g) / (B * T)     # dL/dlogp handed to the backward
    lr, wd = 0.01, 0.005                                                # reference default.py:21-24
    if os.environ.get("MUCON_BENCH_LR") is not None:                   # (A/B hook: is the step's speed a function of the weights?)
        lr = float(os.environ["MUCON_BENCH_LR"])
        wd = 0.0 if lr == 0.0 else wd
    import types
    sgd = ops.FusedClipSGD([flat_params], None,                         # SGD(lr, weight_decay) in one launch (csrc/optim.hpp)
                           types.SimpleNamespace(param_groups=[{"lr": lr, "weight_decay": wd, "momentum": 0.0}]))
    if dist is not None:   # identical replicas whatever the seeds did
        for p_ in params:
            dist.broadcast(p_.data, 0)
    rank_key = (rank * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF          # every rank draws its own dropout masks

    wc2 = wc.reshape(wc.shape[0], wc.shape[1]) if wc.dim() == 3 else wc
    # N > 1 (or MUCON_BENCH_FORCE_DIST=1), MUCON_BENCH_OVERLAP=1: the gradient exchange in two parts, the larger one under first_conv's weight-gradient
    # launch.  Built, tested (tests/test_gpu_dense.py::test_data_parallel_overlap_hook_*) and measured at world size 1, where it can only lose: 0.783 ms per
    # step against 0.7345 with the one collective behind the backward and 0.711 without RCCL (profiles/r06_same_box_abs.txt section 4) -- two collectives'
    # stream hand-offs (+25 us), two weight-gradient + reduction launches instead of one (+10), 248 instead of 256 workgroups.  It pays once the 4 MB
    # all-reduce takes more than ~65 us; no box here has a second GPU to say whether it does, so the DEFAULT stays the one collective.  The N = 1 step is untouched.
    overlap = dist is not None and os.environ.get("MUCON_BENCH_OVERLAP", "0") == "1" and os.environ.get("MUCON_BENCH_COALESCE") != "1"
    if overlap:
        ov_event, ov_side = torch.cuda.Event(), torch.cuda.Stream(device=dev)
        ov_max_wg = max(8, torch.cuda.get_device_properties(dev).multi_processor_count - 8)

    defer_head = os.environ.get("MUCON_BENCH_DEFER_HEAD", "1") == "1"   # (A/B hook)

    def step(i):
        # forward and backward of the hot path as direct calls of the autograd Functions (the same C entry points in the
        # same order as logp.backward() would issue them; ops.run_forward / run_backward) -- no graph walk on the host, so
        # the step stays GPU-bound also on a busy host (the autograd route needs ~1.0 ms of host time per 1.2 ms step)
        tape = tapes[i % len(tapes)]
        enc, c_enc = ops.run_forward(ops._EncoderFn, tape, spec, True, int(i) ^ rank_key, *enc_params)
        (_, logp), c_head = ops.run_forward(ops._HeadFn, enc, wc2, bc, int(T), False, True)
        c_head.reuse_grads = True
        c_head.defer_reduce = defer_head   # (r6) the y-head's slab sums ride in the encoder backward's first launch (mucon_head_bwd_defer): one launch fewer
        d_enc, d_w, d_b = ops.run_backward(ops._HeadFn, c_head, None, dlogp)[:3]
        wc.grad, bc.grad = d_w.view_as(wc), d_b
        if dist is not None and os.environ.get("MUCON_BENCH_COALESCE") != "1":
            c_enc.flat_extra = d_w.numel() + d_b.numel()          # room for the y-head's gradients behind the encoder's
        if overlap:
            # (r6) the exchange in two parts: everything but first_conv's gradients is final BEFORE the tape's second pass (first_conv's weight
            # gradient, the last ~85 us of the backward) -- the library records ov_event there and runs its weight-gradient launches on
            # ncu - 8 workgroups, so RCCL's kernel has CUs while that part travels on the side stream
            c_enc.dp_overlap = (ov_event, ov_max_wg)
        c_enc.reuse_grads = True      # (r6) the flat gradient buffer and its views serve every step (sgd.step() below has consumed them before the next pass writes)
        g_enc = ops.run_backward(ops._EncoderFn, c_enc, d_enc)[4:]
        for p_, g_ in zip(enc_params, g_enc):
            p_.grad = g_
        if overlap:
            buf = ops.flat_grad_buffers(enc_params)[0]
            off = c_enc.flat_rest_off
            ov_side.wait_event(ov_event)
            with torch.cuda.stream(ov_side):
                dist.all_reduce(buf[off:], op=dist.ReduceOp.AVG)                      # 3 of the 4 MB, under first_conv's launch
            tail = c_enc.flat_tail
            torch.cat([d_w.reshape(-1), d_b.reshape(-1)], out=tail)
            wc.grad, bc.grad = tail[:d_w.numel()].view_as(wc), tail[d_w.numel():]
            dist.all_reduce(buf[:off], op=dist.ReduceOp.AVG)                          # first_conv's gradients + the y-head's: what only the pass's end has
            torch.cuda.current_stream().wait_stream(ov_side)
        elif dist is not None:
            # the one exchange step: ONE all-reduce per optimizer step.  The encoder's gradients are views of one flat buffer;
            # the two y-head tensors are appended to a copy of it (every RCCL call costs ~25 us of stream hand-offs even at
            # world size 1: three calls were +75 us per step, one packed call +37, one in-place call on the shared buffer less).  (Splitting the buffer so that all but first_conv's
            # part overlaps the last launch was tried: +35 us of extra launches at world size 1 -- left out.)
            bufs = ops.flat_grad_buffers(enc_params)
            if os.environ.get("MUCON_BENCH_COALESCE") == "1":   # one RCCL group call on the three tensors in place (A/B hook)
                with dist._coalescing_manager(device=dev):
                    for t_ in bufs + [d_w, d_b]:
                        dist.all_reduce(t_, op=dist.ReduceOp.AVG)
            else:
                # the encoder's gradients are views of one flat buffer with room behind them (flat_extra): the y-head's two
                # tensors are copied there (one small launch), their .grad become views of it, and the whole buffer is reduced
                # in place -- one collective, no 4 MB gather and scatter around it
                tail = c_enc.flat_tail
                torch.cat([d_w.reshape(-1), d_b.reshape(-1)], out=tail)
                wc.grad, bc.grad = tail[:d_w.numel()].view_as(wc), tail[d_w.numel():]
                assert len(bufs) == 1
                dist.all_reduce(bufs[0], op=dist.ReduceOp.AVG)
        sgd.step()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.prewarm_seconds > 0:             # bring the GPU to its steady state with work that is NOT the path's (a kernel trace of this command then
        a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)   # holds the path's kernels in their steady state only): bf16 GEMMs + a tape-sized copy
        b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        scratch = torch.empty_like(tapes[0])
        t_end = time.perf_counter() + args.prewarm_seconds
        while time.perf_counter() < t_end:
            for _ in range(8):
                torch.matmul(a, b)
            scratch.copy_(tapes[0])
            torch.cuda.synchronize()
        del a, b, scratch
    init_state = [p_.detach().clone() for p_ in params]      # every timed region measures the RANDOM-INIT network (the workload BASELINE.json names):

    def restore_init():                                         # SGD on synthetic noise drifts the weights, and the step's speed with them (--keep-drift)
        with torch.no_grad():
            for p_, q_ in zip(params, init_state):
                p_.copy_(q_)

    for i in range(args.prewarm_steps):
        step(i)
    sync()
    if not args.keep_drift:
        restore_init()
    for i in range(args.warmup):
        step(i)
    sync()
    _lib.check(lib.mucon_profile_stride(max(1, args.time_every)), "profile_stride")
    _lib.check(lib.mucon_profile_begin(args.steps * args.repeats), "profile_begin")
    regions = []
    for r in range(args.repeats):        # each region: exactly --steps steps between two (barrier + synchronize) brackets
        if r and not args.keep_drift:
            restore_init()                   # (untimed; region 0 continues from the warm-up steps as the contract describes)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + r * args.steps + i)
            if args.drain_every and (i + 1) % args.drain_every == 0:
                torch.cuda.current_stream().synchronize()
        sync()
        regions.append(time.perf_counter() - t0)
    tot_ms = (ctypes.c_float * 2)()
    cnt = (ctypes.c_int32 * 2)()
    _lib.check(lib.mucon_profile_end(tot_ms, cnt), "profile_end")
    weights_finite = bool(all(torch.isfinite(p_).all().item() for p_ in params))      # (a region that ran on NaN operands would have been faster: see --keep-drift)
    calib = box_calibration(lib, dev) if (rank == 0 and not args.no_calibration) else None   # behind the timed regions: the box in the state the regions saw
    if dist is not None:
        t = torch.tensor(regions, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions = t.tolist()
    elapsed = sorted(regions)[len(regions) // 2]       # the median region

    rccl = None
    if dist is not None:
        # the exchange step by itself, outside the timed regions: HIP events on the step's stream around the one collective (what the
        # step waits for: RCCL's own stream, the hand-offs to it and back, the wire), on the buffer the step reduces
        step(args.warmup + args.repeats * args.steps)
        buf = ops.flat_grad_buffers(enc_params)[0]
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(20)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(20)]
        sync()
        for a, b in zip(e0, e1):
            a.record()
            dist.all_reduce(buf, op=dist.ReduceOp.AVG)
            b.record()
        sync()
        ar = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
        t = torch.tensor([ar[len(ar) // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rccl = {"world": world, "backend": dist.get_backend(), "bytes_per_step": int(buf.numel() * 4),
                "allreduce_ms": round(float(t.item()), 4), "collectives_per_step": 2 if overlap else 1,
                "overlap": ("all but first_conv's gradients (3 of 4 MB) all-reduced on a side stream behind an event the backward records in front of first_conv's "
                            "weight-gradient launch, which runs on ncu - 8 workgroups; first_conv's + the y-head's part behind the pass (MUCON_BENCH_OVERLAP=0: one collective)") if overlap else None,
                "nccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()),
                "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))},
                "note": "median of 20 all-reduces of the step's flat gradient buffer (encoder + y-head), HIP events on the step's stream, "
                        "max over ranks; measured after the timed regions"}

    sharded = None
    if dist is not None and not args.no_viterbi:
        # BASELINE.json's metric names "Viterbi ms/video at 1/2/4/8 GPU": at N > 1 (and with MUCON_BENCH_FORCE_DIST=1) every rank runs the
        # decode and evaluation legs on its own shard of videos; no collective on the data path, max-over-ranks timing as for the step
        from mucon_amd.core.viterbi import poisson_params_for_many

        def allmax(x):
            t_ = torch.tensor([x], device=dev, dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())

        def make_videos(kind, rank_, videos):
            Tv, Nv = (16384, 64) if kind == "config5" else (2000, 6)
            gc = torch.Generator(device="cpu").manual_seed(7000 + rank_)
            gd = torch.Generator(device=dev).manual_seed(8000 + rank_)
            tr = torch.randint(0, C, (Nv,), generator=gc).numpy().astype(np.int32)
            mu = np.ones(C)
            mu[np.unique(tr)] = Tv / Nv
            P = poisson_params_for_many([mu], [tr], 30, 2000)[0].params      # [3, N]: the length scores are built on the device (ABI 7)
            distinct = min(videos, 64 if kind == "config5" else videos)      # config 5: 64 emission tensors (3.1 MB each), each used videos / 64 times
            base = [torch.log_softmax(3 * torch.randn(Tv, C, device=dev, generator=gd), dim=1) for _ in range(distinct)]
            return (base * (videos // distinct + 1))[:videos], [tr] * videos, [P] * videos

        log_fact = poisson_params_for_many([np.ones(C)], [[0]], 30, 2000)[0].log_fact
        sharded = {"viterbi": viterbi_bench_sharded(rank, world, sync, allmax, lambda a, b_, c: ops.viterbi_decode_batch(a, b_, c, 30, 2000, log_fact=log_fact), make_videos),
                   "evaluation": eval_bench(dev, 32, rank, world, sync, allmax)}

    if rank == 0:
        frames = world * B * T * args.steps
        value = frames / elapsed
        k_fwd_ms = tot_ms[0] / max(cnt[0], 1)
        k_wg_ms = tot_ms[1] / max(cnt[1], 1)
        flops_first = 2.0 * B * T * spec.in_dim * spec.hidden      # first_conv: forward, and again inside the weight-gradient launch
        # the ONE batched weight-gradient launch: every residual layer (dilated conv taps + conv_1x1), last_conv, first_conv
        rows, flops_wg, bytes_wg = T, flops_first, B * T * (spec.in_dim + spec.hidden) * 4.0
        for l, d in enumerate(spec.stages):
            taps = 1 if d >= rows else 3          # dilation past the sequence: centre tap only
            flops_wg += 2.0 * B * rows * spec.hidden * spec.hidden * (taps + 1)
            bytes_wg += 4.0 * B * rows * spec.hidden * 4   # two gradient and two activation operands, each read once
            if spec.pooling and l in spec.pooling_layers:
                rows //= 2
        flops_wg += 2.0 * B * rows * spec.hidden * spec.hidden
        bytes_wg += 2.0 * B * rows * spec.hidden * 4
        # dominant kernel = the longest launch of the step: the batched weight gradients.  They run on the bf16 MFMA with exactly
        # split fp32 operands (six bf16 MFMAs per fp32 product block: csrc/gemm_tn_split.hpp), so two fractions are given: the
        # ALGORITHMIC fp32 FLOP rate against the f32-MFMA peak (what an exact-f32 instruction stream is capped at; it can pass 1)
        # and the bf16 FLOPs actually issued against the dense bf16 peak -- the number that says how far the kernel is from ITS roof
        split_tn = os.environ.get("MUCON_TN_SPLIT", "1") != "0"
        mfma16 = int(lib.mucon_test_get_knob(b"MUCON_MFMA16"))     # which MFMA shape the two tape-streaming launches ran on
        runs = int(lib.mucon_test_get_knob(b"MUCON_TS_RUNS")) == 1
        dom = ((f"{'ts_runs_kernel' if runs else 'ts_batched_kernel'}: all weight gradients of the step in one launch"
                f"{' of one persistent workgroup per CU, each a contiguous share of the (column, video, tile) line with its accumulators kept in registers' if runs else ''}"
                f" (bf16 MFMA 32x32x16 on exactly split fp32 operands)"
                if split_tn else "tn_batched_kernel<2>: all weight gradients of the step in one launch (f32 MFMA)"), k_wg_ms)
        achieved = flops_wg / (dom[1] * 1e-3) / 1e12
        # the kernel's OWN ceiling: it issues 6 bf16 MFMA FLOP per algorithmic fp32 FLOP, so 2.5 PFLOP/s dense bf16 / 6 = 416.7 TFLOP/s
        # fp32-equivalent (the f32-input MFMA peak, 157.3, is what a plain fp32 kernel could reach -- kept as a secondary field)
        peak_own = PEAK_BF16_MFMA_TFLOPS / 6.0 if split_tn else PEAK_F32_MFMA_TFLOPS
        traffic = traffic_fwd = traffic_src = None
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
            with open(cands[-1]) as f:
                tj = json.load(f)
            traffic_src = "profiles/" + os.path.basename(cands[-1])
            traffic = tj.get("weight_gradients", {}).get("hbm_bytes_per_launch")
            traffic_fwd = tj.get("first_conv_fwd", {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError, IndexError):
            pass
        traffic_note = (f"committed profile, builder's box ({traffic_src}): a constant of the profile run, not measured in this run") if traffic_src else None
        # (child processes of the whole-line runs only: a run with a leg switched off -- the A/B tools' -- reports the profile's constant)
        live, live_why = (live_traffic(args) if world == 1 and not (args.no_viterbi or args.no_cpu_baseline or args.no_calibration)
                          else (None, "a leg of the line is switched off (A/B run)" if world == 1 else "N > 1"))
        if live is not None:
            traffic, traffic_fwd = live["weight_gradients"], live["first_conv_fwd"]
            traffic_note = (f"measured on this box behind the timed regions: two child runs of the hot-path leg under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
                            f"{live['launches_sampled']} launches; HBM bytes = 2 x FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB")
        bytes_fwd = B * T * (spec.in_dim + spec.hidden) * 4.0
        out = {
            "metric": "frames/sec fwd+bwd (Breakfast I3D Tx2048)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 storage / accumulate; GEMM operands split exactly into three bf16 (x = hi + mid + lo), 6 of the 9 partial "
                          "products on the bf16 MFMA (the dropped three are < 2^-25 of the product); departs from north_star's 'no MFMA' "
                          "because at hidden 128 the path is compute-bound (SURVEY.md 0.3); +-inf in a GEMM operand yields NaN (inf - inf in "
                          "the split), fp32 subnormals may flush (INTEGRATION.md)",
            "config": {"workload": f"synthetic I3D tapes, B={B} videos/GPU x T={T} frames x D={spec.in_dim}, {C} classes, "
                                   f"hidden {spec.hidden}, 11 dilated layers (BASELINE config 3 shape); training mode "
                                   f"(dropout on), fwd+bwd+SGD, tapes resident in HBM",
                       "global_batch": world * B, "frames_per_video": T, "parallelism": f"dp{world}"},
            "repeats": args.repeats, "ms_per_step_repeats": [round(r / args.steps * 1e3, 4) for r in regions],
            "prewarm_steps": args.prewarm_steps, "weights_finite_after_last_region": weights_finite,
            "weights": ("random init, restored (untimed) behind the pre-warm steps and in front of every timed region after the first" if not args.keep_drift
                        else "left to drift under SGD on synthetic noise (--keep-drift: experiment)"),
            "data_dependence_note": "the step's speed depends on the values: after ~500 SGD steps on synthetic noise the same launches run 9 % faster (0.77 -> 0.70 ms; "
                                    "MFMA kernels against the power cap); not with lr = 0, not after GEMMs of other data -- so every region starts from the random-init weights",
            "tape_batches_rotated": len(tapes), "tape_bytes_resident": len(tapes) * B * T * spec.in_dim * 4,
            "roofline": {"bound": "mfma", "kernel": dom[0], "achieved": round(achieved, 2), "peak": round(peak_own, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / peak_own, 4), "traffic": traffic,
                         "traffic_source": traffic_note, "traffic_measured": live is not None, "traffic_not_measured_because": live_why,
                         "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "algorithmic_bytes_per_launch": bytes_wg, "avg_launch_ms": round(dom[1], 4),
                         "flops_per_launch": flops_wg, "launches_timed": int(cnt[1]),
                         "timed_every_nth_step": max(1, args.time_every),
                         "issued_bf16_tflops": round((6 if split_tn else 1) * achieved, 1) if split_tn else None,
                         "frac_of_bf16_mfma_peak": round(6 * achieved / PEAK_BF16_MFMA_TFLOPS, 4) if split_tn else None,
                         "hbm_frac": round(bytes_wg / (dom[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                         "note": "achieved = algorithmic fp32 FLOP / launch time; peak = the kernel's own ceiling = 2.5 PFLOP/s dense bf16 MFMA / 6 "
                                 "(it issues 6 bf16 MFMA FLOP per algorithmic FLOP), so frac = frac_of_bf16_mfma_peak <= 1; "
                                 "frac_of_f32_mfma_peak prices the same FLOP against the 157.3 TFLOP/s f32-input MFMA peak a plain fp32 kernel is capped at",
                         "rocprof_summary": "profiles/r06_kernel_stats_hotpath.csv (hot-path leg alone; the default command's summary "
                                            "mixes in the 10x smaller launches of the end-to-end leg)"},
            # first_conv forward: the kernel that streams the tape.  bf16 MFMA on exactly split fp32 operands
            # (csrc/gemm_split.hpp): its roof is HBM, the f32-MFMA roof (0.109 ms) no longer applies
            "roofline_first_conv_fwd": {"bound": "hbm", "kernel": ("nt_split16_kernel<true, false, false, false> (v_mfma_f32_16x16x32_bf16)" if mfma16 & 1
                                                                   else "nt_split_kernel<true, false, false, false> (v_mfma_f32_32x32x16_bf16)"),
                                        "achieved": round(bytes_fwd / (k_fwd_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                        "unit": "GB/s", "frac": round(bytes_fwd / (k_fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                        "traffic": traffic_fwd, "algorithmic_bytes_per_launch": bytes_fwd,
                                        "avg_launch_ms": round(k_fwd_ms, 4), "flops_per_launch": flops_first,
                                        "fp32_equivalent_tflops": round(flops_first / (k_fwd_ms * 1e-3) / 1e12, 1),
                                        "bf16_mfma_frac": round(6 * flops_first / (k_fwd_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)},
            "roofline_hbm_whole_path": {"bound": "hbm", "achieved": round(value / world * BYTES_PER_FRAME_FWD_BWD / 1e9, 1),
                                        "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": round(value / world * BYTES_PER_FRAME_FWD_BWD / 1e9 / PEAK_HBM_GBS, 4),
                                        "bytes_per_frame": BYTES_PER_FRAME_FWD_BWD,
                                        "note": "per GPU; 2.52 MFLOP/frame: an exact-f32 instruction stream caps this at 13% of the HBM peak, "
                                                "the split-bf16 kernels at about 35%"},
            "fp32_fraction_whole_path": round(value / world * 2.517e6 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        }
        if calib is not None:
            # The probes are reported RAW.  Round 5 first fitted ms_per_step ~ (LDS-fed loop's rate)^-3 on five boxes (7.9 % spread -> 1.8 %), then met boxes the fit
            # fails on outright: the loop at 1,482 - 1,530 TFLOP/s (the lowest readings of the round) beside a step of 0.702 - 0.706 ms (among the fastest) --
            # DESIGN.md section 5 has the ten (probe, step) pairs.  No code-independent probe found predicts the step to better than the boxes' own spread, so no
            # normalised figure is emitted (the key stays, null: the round-4 verdict's contract); same-box A/Bs (tools/flag_ab.sh, tools/knob_sweep.sh) remain the method.
            out["box_calibration"] = calib
            out["ms_per_step_at_reference_box"] = None
            out["normalisation"] = ("none: over ten boxes of the pool none of the probes (register-only MFMA loops of both shapes, LDS-fed MFMA loop, 1 GiB copy, the "
                                    "frozen round-2 first_conv kernel) predicts ms_per_step -- DESIGN.md section 5; box_calibration holds them raw")
        if rccl is not None:
            out["rccl"] = rccl
        # the single-GPU legs (CPU baseline, Viterbi, end-to-end, evaluation) belong to the N = 1 line: at N > 1 the other ranks would sit in the
        # closing barrier for their two minutes
        if sharded is not None:
            out["viterbi"], out["evaluation"] = sharded["viterbi"], sharded["evaluation"]
            b5 = 16384 * C * 4 + 16384 * 4
            gbs = b5 / (out["viterbi"]["ms_per_video_batch256"] * world * 1e-3) / 1e9        # per GPU: a rank's own rate
            out["roofline_viterbi"] = {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS, "per_gpu": True,
                                       "config5_T16384_N64": {"algorithmic_bytes_per_video": b5,
                                                              "batch256": {"ms_per_video_per_gpu": round(out["viterbi"]["ms_per_video_batch256"] * world, 5),
                                                                           "achieved": round(gbs, 3), "frac": round(gbs / PEAK_HBM_GBS, 6)}}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(spec, C, T)
        if world == 1 and not args.no_viterbi:
            restore_init()
            out["dense_batch1"] = batch1_dense_legs(dev, spec, C, params, with_cpu=not args.no_cpu_baseline)
            restore_init()
        if world == 1 and not (args.no_viterbi or args.no_calibration):
            # box-independent: shader cycles counted inside the four kernel families, from the stamped library variant in a CHILD process (one extra
            # step outside every timed region; tools/kernel_cycles.py).  Boxes differ by 8 % in clock; a cycle count only moves with the code.
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_cycles.py")], capture_output=True, text=True, timeout=240)
                out["kernel_cycles"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]}
            except (OSError, subprocess.SubprocessError, ValueError, IndexError) as e:
                out["kernel_cycles"] = {"error": str(e)[:300]}
        if not args.no_viterbi and world == 1 and sharded is None:
            out["viterbi"] = viterbi_bench(dev, C)
            out["roofline_viterbi"] = viterbi_roofline(out["viterbi"])
            out["end_to_end"] = end_to_end_bench(dev)
            out["evaluation"] = eval_bench(dev)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)   # RCCL's banner sits in C stdio buffers: get it out before the result line
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
