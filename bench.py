#!/usr/bin/env python3
"""Benchmark of the MuCon temporal hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Metric (BASELINE.json): frames/sec of encoder + y-head forward+backward on Breakfast-I3D-shaped
tapes (T x 2048), plus Viterbi ms/video as extra fields.  One step = one pass of the hot path over
one batch of synthetic tapes that are already resident in HBM:
    encoder fwd (training mode, dropout on) -> y-head fwd (log-softmax) -> dL/dlogp given ->
    y-head bwd -> encoder bwd -> [N>1: RCCL all-reduce of the flat gradient] -> SGD update.
Workload: BASELINE config 3 shape per GPU (B=8 videos x T=4096 frames x D=2048, 48 classes) --
268 MB of tape per step, i.e. larger than the 256 MiB Infinity Cache.  Weak scaling: every rank
has its own batch; videos are independent, the only exchange is the gradient all-reduce.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel, timed per launch with HIP events inside the timed region
  cpu_baseline  the oracle's CPU path (torch CPU ops, fp32, all host cores) on a bounded sample
  viterbi       decode latency (single video, and amortised over a batch of videos) vs the C oracle
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
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="videos per GPU per step")
    ap.add_argument("--frames", type=int, default=4096, help="frames per video")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


def cpu_baseline(spec, C, T, budget_s=20.0):
    """The oracle's dense path (same torch ops the reference issues: conv/matmul, pooling, GroupNorm,
    nearest upsample, log_softmax) fp32 on the host cores, fwd+bwd, on a bounded sample of the workload."""
    from oracle import dense as od

    ocfg = od.EncoderConfig(in_dim=spec.in_dim, hidden=spec.hidden, num_classes=C, stages=list(spec.stages),
                            pooling=spec.pooling, pooling_type=spec.pooling_type, pooling_layers=list(spec.pooling_layers),
                            leaky_relu=spec.leaky_relu, last_gn=spec.last_gn, last_gn_num_groups=spec.last_gn_num_groups,
                            last_relu=spec.last_relu)
    params = od.to_torch(od.seeded_params(ocfg, 1), torch.float32, requires_grad=True)
    B = 2
    tape = torch.randn(B, T, spec.in_dim)
    w = torch.randn(B, T, C)
    frames, t_used = 0, 0.0
    for it in range(100):
        t0 = time.perf_counter()
        enc = od.encoder_forward(tape, params, ocfg)
        _, logp = od.head_forward(enc, params, ocfg, T)
        (w * logp).sum().backward()
        dt = time.perf_counter() - t0
        if it > 0:  # first pass warms the allocator
            frames += B * T
            t_used += dt
        if t_used > budget_s:
            break
    return {"value": round(frames / t_used, 1), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/dense.py fwd+bwd fp32, B={B} x T={T} x D={spec.in_dim}, {frames // (B * T)} passes in {t_used:.1f}s"}


def viterbi_bench(dev, C=48):
    """BASELINE config 5 shape: T=16384, 64-state transcript.  Single-stream latency and amortised
    latency with 64 videos per launch; the C oracle (1 core) on the same input beside it."""
    import oracle
    from mucon_amd import ops
    from mucon_amd.core.viterbi import PoissonModel

    fs, max_len = 30, 2000
    g = torch.Generator(device="cpu").manual_seed(7)
    out = {}
    # a Breakfast-typical video first (T ~ 2000 frames, 6 actions), single stream
    T, N = 2000, 6
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    P = PoissonModel(mu).rows_for(tr, fs)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for _ in range(3):
        ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.viterbi_decode_batch([lp], [tr], [P], fs, max_len)
    torch.cuda.synchronize()
    out["ms_per_video_T2000_N6_single"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    lps256 = [torch.log_softmax(3 * torch.randn(T, C, device=dev), dim=1) for _ in range(256)]   # 256 videos in flight
    ops.viterbi_decode_batch(lps256, [tr] * 256, [P] * 256, fs, max_len)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ops.viterbi_decode_batch(lps256, [tr] * 256, [P] * 256, fs, max_len)
    torch.cuda.synchronize()
    out["ms_per_video_T2000_N6_batch256"] = round((time.perf_counter() - t0) / 3 / 256 * 1e3, 5)
    del lps256
    lp_h = lp.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(5):
        oracle.viterbi_decode_table(lp_h, tr, P, fs, max_len)
    out["cpu_oracle_ms_per_video_T2000_N6"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    out["reference_python_ms_per_video_T2000_N6"] = 68.0   # measured in the build container, context only

    T, N = 16384, 64
    tr = torch.randint(0, C, (N,), generator=g).numpy().astype(np.int32)
    mu = np.ones(C)
    mu[np.unique(tr)] = T / N
    P = PoissonModel(mu).rows_for(tr, fs)
    lp = torch.log_softmax(3 * torch.randn(T, C, generator=g), dim=1).to(dev)
    for label, nv, reps in (("single", 1, 10), ("batch64", 64, 3), ("batch256", 256, 2)):
        if nv == 1:
            lps = [lp]
        elif nv == 64:
            lps = base64 = [torch.log_softmax(3 * torch.randn(T, C, device=dev), dim=1) for _ in range(64)]
        else:
            lps = base64 * (nv // 64)          # 256 videos in flight (the 64 emission tensors four times over)
        ops.viterbi_decode_batch(lps, [tr] * nv, [P] * nv, fs, max_len)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.viterbi_decode_batch(lps, [tr] * nv, [P] * nv, fs, max_len)
        torch.cuda.synchronize()
        out[f"ms_per_video_{label}"] = round((time.perf_counter() - t0) / reps / nv * 1e3, 4)
    lp_h = lp.cpu().numpy()
    t0 = time.perf_counter()
    oracle.viterbi_decode_table(lp_h, tr, P, fs, max_len)
    out["cpu_oracle_ms_per_video"] = round((time.perf_counter() - t0) * 1e3, 3)
    out["config"] = (f"ms_per_video_single/batch64/batch256 and cpu_oracle_ms_per_video: BASELINE config 5, T={T}, N={N}, C={C}, fs={fs} "
                     f"(K=546 columns, 64x66 hypotheses); every timing includes the upload of the job table and the result D2H")
    out["algorithmic_bytes_per_video"] = T * C * 4 + T * 4
    return out


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
    for i in range(5):
        trainer._train_1_batch(i, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer._train_1_batch(5 + i, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"videos_per_s": round(1.0 / dt, 1), "frames_per_s": round(T / dt, 1), "ms_per_video": round(dt * 1e3, 3),
            "config": f"full MuCon train step, batch 1, T={T}, N={N}: all-HIP, graph-free: encoder, s-head (persistent LSTM / decoder), y-head, fused losses, backward, fused clip+SGD as one straight line of launches (MuCon.fused_train_step)",
            "reference_readme_it_per_s": [14.67, 16.23]}


def eval_bench(dev, n_videos=16):
    """Evaluation scope: MuConEvaluator.evaluate() per test video = eval-mode forward (greedy s-head decode, at most 8
    words here), predict, Viterbi decode on the device-resident log-probs, and the reference's full metric set on the host
    (reference src/mucon/evaluators.py:121-257).  Random-init weights; the EOS logit is biased down so that the greedy
    decode emits a transcript (an untrained s-head emits EOS first, on which the reference's evaluator fails as well)."""
    from mucon_amd import synth
    from mucon_amd.config import get_cfg_defaults, update_config
    from mucon_amd.core.datasets import Batch
    from mucon_amd.mucon.evaluators import MuConEvaluator
    from mucon_amd.mucon.models import create_model

    T, N, C = 2000, 6, 48
    cfg = update_config(get_cfg_defaults(), [], [])
    torch.manual_seed(0)
    model = create_model(cfg, C, 8, 2048).to(dev)
    with torch.no_grad():
        model.fs_decoder_transcript[2].bias[C] = -20.0

    class Videos:
        background_class_ids = [0]

        def __init__(self):
            self.items = []
            for v in range(n_videos):
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
    ev.evaluate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.evaluate()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(db)
    return {"videos_per_s": round(1.0 / dt, 1), "ms_per_video": round(dt * 1e3, 3),
            "config": f"MuConEvaluator.evaluate(): {n_videos} videos, T={T}: eval forward + greedy decode (8 words) + predict + "
                      f"Viterbi (fs=30) + MoF/IoD/IoU/edit/F1 for y-, s- and Viterbi segmentations; tapes resident in HBM"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("MUCON_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path even with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist

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
    tape = torch.randn(B, T, spec.in_dim, device=dev, generator=g)      # resident in HBM
    dlogp = torch.randn(B, T, C, device=dev, generator=g) / (B * T)     # dL/dlogp handed to the backward
    lr, wd = 0.01, 0.005                                                # reference default.py:21-24
    import types
    sgd = ops.FusedClipSGD([flat_params], None,                         # SGD(lr, weight_decay) in one launch (csrc/optim.hpp)
                           types.SimpleNamespace(param_groups=[{"lr": lr, "weight_decay": wd, "momentum": 0.0}]))

    wc2 = wc.reshape(wc.shape[0], wc.shape[1]) if wc.dim() == 3 else wc

    def step(i):
        # forward and backward of the hot path as direct calls of the autograd Functions (the same C entry points in the
        # same order as logp.backward() would issue them; ops.run_forward / run_backward) -- no graph walk on the host, so
        # the step stays GPU-bound also on a busy host (the autograd route needs ~1.0 ms of host time per 1.2 ms step)
        enc, c_enc = ops.run_forward(ops._EncoderFn, tape, spec, True, int(i), *enc_params)
        (_, logp), c_head = ops.run_forward(ops._HeadFn, enc, wc2, bc, int(T), False, True)
        d_enc, d_w, d_b = ops.run_backward(ops._HeadFn, c_head, None, dlogp)[:3]
        wc.grad, bc.grad = d_w.view_as(wc), d_b
        g_enc = ops.run_backward(ops._EncoderFn, c_enc, d_enc)[4:]
        for p_, g_ in zip(enc_params, g_enc):
            p_.grad = g_
        if dist is not None:
            # the one exchange step: ONE all-reduce per optimizer step.  The encoder's gradients are views of one flat buffer;
            # the two y-head tensors are appended to a copy of it (every RCCL call costs ~25 us of stream hand-offs even at
            # world size 1: three calls were +75 us per step, one is +30).  (Splitting the buffer so that all but first_conv's
            # part overlaps the last launch was tried: +35 us of extra launches at world size 1 -- left out.)
            bufs = ops.flat_grad_buffers(enc_params)
            if os.environ.get("MUCON_BENCH_COALESCE") == "1":   # one RCCL group call on the three tensors in place (A/B hook)
                with dist._coalescing_manager(device=dev):
                    for t_ in bufs + [d_w, d_b]:
                        dist.all_reduce(t_, op=dist.ReduceOp.AVG)
            else:
                parts = [b_.reshape(-1) for b_ in bufs] + [d_w.reshape(-1), d_b.reshape(-1)]
                packed = torch.cat(parts)
                dist.all_reduce(packed, op=dist.ReduceOp.AVG)
                off = 0
                for t_ in parts:
                    t_.copy_(packed[off: off + t_.numel()])
                    off += t_.numel()
        sgd.step()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    _lib.check(lib.mucon_profile_begin(args.steps), "profile_begin")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    tot_ms = (ctypes.c_float * 2)()
    cnt = (ctypes.c_int32 * 2)()
    _lib.check(lib.mucon_profile_end(tot_ms, cnt), "profile_end")
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

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
        # dominant kernel = the longest launch of the step: the batched weight gradients (f32-input MFMA, exact fp32)
        dom = ("tn_batched_kernel<2>: all weight gradients of the step in one launch (f32 MFMA)", k_wg_ms)
        achieved = flops_wg / (dom[1] * 1e-3) / 1e12
        traffic = traffic_fwd = None   # HBM bytes per launch from the rocprofv3 --pmc passes of tools/profile_round.sh (profiles/)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tj = json.load(f)
            traffic = tj.get("weight_gradients", {}).get("hbm_bytes_per_launch")
            traffic_fwd = tj.get("first_conv_fwd", {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        bytes_fwd = B * T * (spec.in_dim + spec.hidden) * 4.0
        out = {
            "metric": "frames/sec fwd+bwd (Breakfast I3D Tx2048)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic I3D tapes, B={B} videos/GPU x T={T} frames x D={spec.in_dim}, {C} classes, "
                                   f"hidden {spec.hidden}, 11 dilated layers (BASELINE config 3 shape); training mode "
                                   f"(dropout on), fwd+bwd+SGD, tapes resident in HBM",
                       "global_batch": world * B, "frames_per_video": T, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": dom[0], "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_wg, "avg_launch_ms": round(dom[1], 4),
                         "flops_per_launch": flops_wg, "all_weight_gradients_launch_ms": round(k_wg_ms, 4),
                         "rocprof_summary": "profiles/r01_kernel_stats_hotpath.csv (hot-path leg alone; the default command's summary "
                                            "mixes in the 10x smaller launches of the end-to-end leg)"},
            # first_conv forward: the kernel that streams the tape.  bf16 MFMA on exactly split fp32 operands
            # (csrc/gemm_split.hpp): its roof is HBM, the f32-MFMA roof (0.109 ms) no longer applies
            "roofline_first_conv_fwd": {"bound": "hbm", "kernel": "nt_split_kernel<true, false, false, false>",
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
                                        "note": "per GPU; the path is f32-compute-bound (2.52 MFLOP/frame): ceiling 13% of HBM peak"},
            "fp32_fraction_whole_path": round(value / world * 2.517e6 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, C, T)
        if not args.no_viterbi:
            out["viterbi"] = viterbi_bench(dev, C)
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
