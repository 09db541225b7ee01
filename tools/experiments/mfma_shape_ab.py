"""ON THE GPU BOX: the two MFMA shapes of the split-bf16 kernels, alternated in ONE process on random data.

    python3 tools/mfma_shape_ab.py [first|tn|both] [rounds]

first: first_conv forward (gemm_split.hpp: nt_split_kernel on v_mfma_f32_32x32x16_bf16 vs nt_split16_kernel on 16x16x32) at the bench
       shape B = 8 x T = 4096 x 2048, `iters` back-to-back launches per arm and round (HIP events around the run).
tn:    the batched weight-gradient launch (gemm_tn_split.hpp) through the encoder backward of the bench step, timed by the library's
       profile slots (HIP events around the one launch, every step).
With a -DCLK_STAMP=1 build (MUCON_HIPCC_FLAGS) the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over
workgroups of the LAST launch) is printed per arm."""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

import torch

from mucon_amd import _lib

lib = _lib.load()
what = sys.argv[1] if len(sys.argv) > 1 else "both"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = _lib.current_stream_ptr()


def clock(slot):
    buf = (ctypes.c_longlong * (2 * 4096))()
    n = lib.mucon_test_read_clock(slot, buf, 2 * 4096)
    if n <= 0:
        return None
    ghz = sorted(buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(n) if buf[2 * i + 1] > 0)
    return ghz[len(ghz) // 2] if ghz else None


def first_conv():
    B, T, D = 8, 4096, 2048
    g = torch.Generator(device="cuda").manual_seed(1)
    tapes = [torch.randn(B, T, D, device="cuda", generator=g) for _ in range(2)]   # 2 x 268 MB: past the Infinity Cache
    W = torch.randn(128, D, device="cuda", generator=g) * 0.02
    b = torch.randn(128, device="cuda", generator=g)
    out = torch.empty(B, T, 128, device="cuda")
    planes = torch.empty(3 * 128 * D * 2, dtype=torch.uint8, device="cuda")
    ms = ctypes.c_float()
    res = {0: [], 1: []}
    clk = {0: None, 1: None}
    outs = {}
    for rnd in range(rounds):
        for shape in (0, 1):
            _lib.set_knob("MUCON_MFMA16", shape)
            tape = tapes[rnd & 1]
            _lib.check(lib.mucon_test_first_conv_split(_lib.ptr(tape), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), B, T, D, 1, _lib.ptr(planes),
                                                       planes.numel(), 400, ctypes.byref(ms), s), "split")
            res[shape].append(ms.value * 1e3)
            clk[shape] = clock(0)
            if rnd == 0:
                outs[shape] = out.clone()
    ref = torch.relu(tapes[0].double() @ W.double().T + b.double())
    for shape, name in ((0, "32x32x16"), (1, "16x16x32")):
        us = res[shape]
        err = (outs[shape].double() - ref).abs().max().item()
        gb = (B * T * D * 4 + B * T * 128 * 4) / 1e3
        c = f"  in-kernel clock {clk[shape]:.3f} GHz" if clk[shape] else ""
        print(f"first_conv fwd  {name}: median {statistics.median(us):7.2f} us  (min {min(us):.2f}, max {max(us):.2f}; {rounds} x 400 launches)  "
              f"{gb / statistics.median(us) / 1e3:.2f} TB/s  max|err vs f64| {err:.2e}{c}")
    print(f"first_conv fwd  ratio 16x16x32 / 32x32x16 = {statistics.median(res[1]) / statistics.median(res[0]):.4f}")


def weight_gradients():
    from mucon_amd import ops, synth
    from oracle import dense as od   # (seeded parameter shapes only)
    B, T = 8, 4096
    spec, ocfg = ops.EncoderSpec(), od.EncoderConfig()
    params_np = od.seeded_params(ocfg, 3)
    P = [torch.tensor(params_np[k], device="cuda", requires_grad=True) for k in ops.param_names(spec)]
    g = torch.Generator(device="cuda").manual_seed(2)
    tapes = [torch.randn(B, T, 2048, device="cuda", generator=g) for _ in range(2)]
    denc = torch.randn(B, spec.out_length(T), 128, device="cuda", generator=g)
    res = {0: [], 1: []}
    clk = {0: None, 1: None}
    grads = {}
    tot = (ctypes.c_float * 2)()
    cnt = (ctypes.c_int32 * 2)()
    for rnd in range(rounds):
        for shape in (0, 2):
            _lib.set_knob("MUCON_MFMA16", shape)
            for it in range(3 + 60):
                if it == 3:
                    torch.cuda.synchronize()
                    _lib.check(lib.mucon_profile_stride(1), "stride")
                    _lib.check(lib.mucon_profile_begin(64), "begin")
                for p in P:
                    p.grad = None
                enc = ops.encoder_forward(tapes[it & 1], P, spec, training=True, seed=7 + it)
                enc.backward(denc)
            _lib.check(lib.mucon_profile_end(tot, cnt), "end")
            res[shape >> 1].append(tot[1] / max(cnt[1], 1) * 1e3)
            clk[shape >> 1] = clock(1)
            if rnd == 0:
                grads[shape >> 1] = [p.grad.clone() for p in P]
    for k, name in ((0, "32x32x16"), (1, "16x16x32")):
        us = res[k]
        c = f"  in-kernel clock {clk[k]:.3f} GHz" if clk[k] else ""
        print(f"weight gradients {name}: median {statistics.median(us):7.2f} us per launch (min {min(us):.2f}, max {max(us):.2f}; {rounds} x 60 steps){c}")
    rel = max(((a - b).norm() / (a.norm() + 1e-30)).item() for a, b in zip(grads[0], grads[1]))
    print(f"weight gradients ratio 16x16x32 / 32x32x16 = {statistics.median(res[1]) / statistics.median(res[0]):.4f}; "
          f"largest relative L2 distance between the two shapes' gradients {rel:.2e}")


if what in ("first", "both"):
    first_conv()
if what in ("tn", "both"):
    weight_gradients()
_lib.set_knob("MUCON_MFMA16", _lib.mfma16_default())
