"""first_conv forward on the bf16 MFMA with exactly split operands (csrc/gemm_split.hpp) against float64, next to the
f32-MFMA kernel it replaces for chip-filling launches.

Claim under test: x = hi + mid + lo is exact, six of the nine partial products are kept, and the result is fp32-grade:
its distance to the float64 product is of the size of the f32-MFMA kernel's own (both are dominated by the fp32
accumulation over K = 2048), far inside the 1e-4 tolerance of the dense path (test_gpu_dense.py)."""
import ctypes

import numpy as np
import pytest
import torch

from mucon_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=[0, 1], ids=["mfma32x32x16", "mfma16x16x32"])
def mfma_shape(request):
    """Every test of this file runs on both instantiations of the kernel: v_mfma_f32_32x32x16_bf16 (nt_split_kernel) and
    v_mfma_f32_16x16x32_bf16 (nt_split16_kernel; MUCON_MFMA16 bit 0).  The shipped default is restored afterwards."""
    from mucon_amd import _lib
    default = _lib.mfma16_default()
    _lib.set_knob("MUCON_MFMA16", (default & ~1) | request.param)
    try:
        yield request.param
    finally:
        _lib.set_knob("MUCON_MFMA16", default)


def _run_split(tape, W, bias, relu):
    from mucon_amd import _lib
    lib = _lib.load()
    B, T, D = tape.shape
    out = torch.full((B, T, 128), float("nan"), device=DEV)
    planes = torch.empty(3 * 128 * D * 2, dtype=torch.uint8, device=DEV)
    _lib.check(lib.mucon_test_first_conv_split(_lib.ptr(tape), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), B, T, D, relu,
                                               _lib.ptr(planes), planes.numel(), 1, None, _lib.current_stream_ptr()),
               "first_conv_split")
    return out, planes


@pytest.mark.parametrize("B,T,D,kind", [(1, 128, 2048, "pm1"), (1, 257, 2048, "pm1"), (3, 1000, 2048, "pm1"), (2, 130, 128, "pm1"),
                                         (1, 77, 1024, "pm1"), (2, 515, 2048, "abs_normal")])
def test_split_first_conv_matches_float64(B, T, D, kind):
    """kind abs_normal: non-negative features, as real post-ReLU I3D tapes -- sums that do not cancel."""
    from mucon_amd import _lib
    lib = _lib.load()
    tape = torch.tensor(synth.uniform_pm1(11, (B, T, D)), device=DEV)
    if kind == "abs_normal":
        tape = torch.randn(B, T, D, generator=torch.Generator().manual_seed(14)).abs().to(DEV)
    W = torch.tensor(synth.uniform_pm1(12, (128, D)), device=DEV) * 0.05
    bias = torch.tensor(synth.uniform_pm1(13, (128,)), device=DEV)
    ref = tape.double() @ W.double().T + bias.double()
    out, _ = _run_split(tape, W, bias, 0)
    assert torch.isfinite(out).all()
    err_split = (out.double() - ref).abs().max().item()
    # the f32-MFMA kernel on the same operands
    out32 = torch.full((B * T, 128), float("nan"), device=DEV)
    _lib.check(lib.mucon_test_gemm_nt(_lib.ptr(tape), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out32), B * T, D, 1,
                                      _lib.current_stream_ptr()), "gemm_nt")
    err_f32 = (out32.double() - torch.relu(ref).reshape(B * T, 128)).abs().max().item()
    scale = ref.abs().max().item()
    print(f"B={B} T={T} D={D}: max|err| split {err_split:.3e}  f32-MFMA {err_f32:.3e}  (|out| up to {scale:.2f})")
    assert err_split <= 2e-6 * scale + 4 * err_f32   # fp32-grade: same size as the f32 kernel's own error
    out_relu, _ = _run_split(tape, W, bias, 1)
    torch.testing.assert_close(out_relu, torch.relu(out), rtol=0, atol=0)


def test_split_is_exact(mfma_shape):
    """hi + mid + lo == x bit for bit (the planes pack_weights writes), for values across the exponent range."""
    D = 256
    g = torch.Generator().manual_seed(5)
    w = torch.randn(128, D, generator=g) * torch.exp2(torch.randint(-40, 40, (128, D), generator=g).float())
    # exact for 0 and for 2^-100 < |x| < the bf16 maximum (3.39e38); below that the low parts are denormal
    w[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -23, 1.0e38, 1.0e-30, -2.5])
    W = w.to(DEV)
    tape = torch.zeros(1, 128, D, device=DEV)
    _, planes = _run_split(tape, W, torch.zeros(128, device=DEV), 0)
    if mfma_shape == 0:
        # fragment order: [k-tile D/64][k-step 4][plane 3][lane half 2][channel 128][slot 8], k16 = 4*half + slot (slot < 4) or 8 + 4*half + slot - 4
        img = planes.view(torch.bfloat16).reshape(D // 64, 4, 3, 2, 128, 2, 4).double()   # slot = 4*hi + lo4
        # k = 64*S + 16*s + 8*hi + 4*half + lo4  ->  order the axes as (plane, n, S, s, hi, half, lo4)
        p = img.permute(2, 4, 0, 1, 5, 3, 6).reshape(3, 128, D)
    else:
        # 16x16x32 order: [k-tile D/64][k-step 2][plane 3][h 4][channel 128][slot 8], k32 = 4*h + slot (slot < 4) or 16 + 4*h + slot - 4
        img = planes.view(torch.bfloat16).reshape(D // 64, 2, 3, 4, 128, 2, 4).double()   # slot = 4*hi + lo4
        # k = 64*S + 32*ks + 16*hi + 4*h + lo4  ->  (plane, n, S, ks, hi, h, lo4)
        p = img.permute(2, 4, 0, 1, 5, 3, 6).reshape(3, 128, D)
    np.testing.assert_array_equal((p[0] + p[1] + p[2]).cpu().numpy(), W.double().cpu().numpy())


def test_encoder_uses_split_kernel_and_agrees_with_f32_path():
    """Above the size threshold mucon_encoder_fwd takes the split kernel: x[0] agrees with float64 as above."""
    from mucon_amd import ops
    from oracle import dense as od
    B, T = 4, 4096
    spec, ocfg = ops.EncoderSpec(), od.EncoderConfig()
    params_np = od.seeded_params(ocfg, 31)
    P = [torch.tensor(params_np[k], device=DEV) for k in ops.param_names(spec)]
    tape = torch.tensor(synth.tape(32, B, T, 2048), device=DEV)
    with torch.no_grad():
        enc = ops.encoder_forward(tape, P, spec, training=False)
    x0 = torch.relu(tape.double() @ P[0][:, :, 0].double().T + P[1].double())
    out, _ = _run_split(tape, P[0][:, :, 0].contiguous(), P[1], 1)
    assert (out.double() - x0).abs().max().item() <= 2e-6 * x0.abs().max().item() + 1e-6
    assert torch.isfinite(enc).all()


def _f32_first_conv(tape, W, bias):
    from mucon_amd import _lib
    lib = _lib.load()
    B, T, D = tape.shape
    out32 = torch.full((B * T, 128), float("nan"), device=DEV)
    _lib.check(lib.mucon_test_gemm_nt(_lib.ptr(tape), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out32), B * T, D, 0,
                                      _lib.current_stream_ptr()), "gemm_nt")
    return out32.reshape(B, T, 128)


def test_split_first_conv_wide_dynamic_range():
    """Tape channels scaled by 2^-60 ... 2^60 (the weights by the inverse, so that every product matters): each output is
    judged against sum |a| |w|, next to the f32-MFMA kernel."""
    B, T, D = 1, 300, 512
    g = torch.Generator().manual_seed(41)
    e = torch.linspace(-60, 60, D).round()
    tape = ((torch.rand(B, T, D, generator=g) * 2 - 1) * torch.exp2(e)).to(DEV)
    W = ((torch.rand(128, D, generator=g) * 2 - 1) * torch.exp2(-e) * 0.05).to(DEV)
    bias = torch.zeros(128, device=DEV)
    ref = tape.double() @ W.double().T
    mag = tape.double().abs() @ W.double().abs().T
    out, _ = _run_split(tape, W, bias, 0)
    out32 = _f32_first_conv(tape, W, bias)
    assert torch.isfinite(out).all()
    rel_s = ((out.double() - ref).abs() / mag).max().item()
    rel_f = ((out32.double() - ref).abs() / mag).max().item()
    print(f"error / sum|a||w|: split {rel_s:.3e}  f32-MFMA {rel_f:.3e}")
    assert rel_s <= 2e-7 + 4 * rel_f


def test_split_first_conv_nan_inf_subnormal_in_the_tape():
    """A NaN stays a NaN and stays in its frame; an infinity makes its frame non-finite (NaN where an fp32 chain gives +-inf:
    inf - bf16(inf) is NaN -- never a finite number); subnormal features cost at most their own magnitude."""
    B, T, D = 1, 260, 256
    tape = torch.tensor(synth.uniform_pm1(42, (B, T, D)), device=DEV)
    W = torch.tensor(synth.uniform_pm1(43, (128, D)), device=DEV) * 0.05
    bias = torch.tensor(synth.uniform_pm1(44, (128,)), device=DEV)
    clean, _ = _run_split(tape, W, bias, 0)
    keep = torch.ones(T, dtype=torch.bool, device=DEV)
    keep[100] = False
    for bad in (float("nan"), float("inf"), float("-inf")):
        t = tape.clone()
        t[0, 100, 37] = bad
        out, _ = _run_split(t, W, bias, 0)
        assert not torch.isfinite(out[0, 100]).any(), bad
        if bad != bad:
            assert torch.isnan(out[0, 100]).all()
        torch.testing.assert_close(out[0, keep], clean[0, keep], rtol=0, atol=0)
    t = tape.clone()
    t[0, 100, :] = 1e-40        # a subnormal frame: the output is the bias up to ~D * 1e-40 * |w|
    t[0, 7, 5] = 2e-39
    out, _ = _run_split(t, W, bias, 0)
    assert torch.isfinite(out).all()
    assert (out[0, 100] - bias).abs().max().item() <= 1e-37
    ref7 = t[0, 7].double() @ W.double().T + bias.double()
    torch.testing.assert_close(out[0, 7], ref7.float(), rtol=1e-5, atol=1e-5)
