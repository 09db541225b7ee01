"""Weight gradients on the bf16 MFMA with exactly split operands (csrc/gemm_tn_split.hpp) against float64, next to the
f32-MFMA kernel they replace (csrc/gemm_tn.hpp, MUCON_TN_SPLIT=0), through the C ABI's test hook.

Claims under test
  * fp32-grade: the distance to the float64 product is of the size of the f32-MFMA kernel's own (both are dominated by the
    fp32 accumulation over time), column by column, also when columns differ by 2^+-100 in scale;
  * operands with special values: a NaN stays confined to its output column / row and stays a NaN; an infinity makes its
    column / row non-finite (the split gives NaN where an fp32 product chain gives +-inf: inf - bf16(inf) is NaN -- a non-finite
    input never comes out as a finite number); subnormal inputs cost at most their own magnitude;
  * every tile shape: odd chunk counts (a half workgroup without columns), partial last tiles, one-tile chunks.
The taps / conv_1x1 (dropout replay) / bias-sum / non-linearity paths of the kernel are driven by the encoder backward tests
(test_gpu_dense.py: reference gradients and the float64 oracle at 11 shapes, training mode included)."""
import numpy as np
import pytest
import torch

from mucon_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=[1, 0], ids=["static-runs", "workgroup-per-item"])
def launch_form(request):
    """Every test of this file runs on both launch forms of the split kernel: ts_runs_kernel (persistent workgroups, each one contiguous
    share of the (column, tile) line; the default) and ts_batched_kernel (one workgroup per (column, time chunk) item; MUCON_TS_RUNS=0)."""
    from mucon_amd import _lib
    _lib.set_knob("MUCON_TS_RUNS", request.param)
    try:
        yield request.param
    finally:
        _lib.set_knob("MUCON_TS_RUNS", 1)


def _tn(Y, X):
    from mucon_amd import _lib
    lib = _lib.load()
    M, K = X.shape
    out = torch.full((128, K), float("nan"), device=DEV)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    _lib.check(lib.mucon_test_gemm_tn(_lib.ptr(Y), _lib.ptr(X), _lib.ptr(out), M, K, _lib.ptr(ws), ws.numel(),
                                      _lib.current_stream_ptr()), "gemm_tn")
    torch.cuda.synchronize()
    return out


def _both(Y, X):
    """(split kernel, f32-MFMA kernel) on the same operands."""
    from mucon_amd import _lib
    _lib.set_knob("MUCON_TN_SPLIT", 1)
    a = _tn(Y, X)
    try:
        _lib.set_knob("MUCON_TN_SPLIT", 0)
        b = _tn(Y, X)
    finally:
        _lib.set_knob("MUCON_TN_SPLIT", 1)
    return a, b


@pytest.mark.parametrize("M,K", [(32, 128), (33, 256), (129, 384), (1000, 256), (1024, 128), (4097, 384), (700, 2048), (2600, 640)])
def test_tn_split_matches_float64(M, K):
    Y = torch.tensor(synth.uniform_pm1(4, (M, 128)), device=DEV)
    X = torch.tensor(synth.uniform_pm1(5, (M, K)), device=DEV)
    ref = Y.double().T @ X.double()
    split, f32 = _both(Y, X)
    assert torch.isfinite(split).all()
    err_s = (split.double() - ref).abs().max().item()
    err_f = (f32.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"M={M} K={K}: max|err| split {err_s:.3e}  f32-MFMA {err_f:.3e}  (|out| up to {scale:.1f})")
    assert err_s <= 2e-6 * scale + 4 * err_f
    torch.testing.assert_close(split, ref.float(), rtol=1e-5, atol=2e-4 * (M / 128) ** 0.5)


def test_tn_split_non_cancelling_sums():
    """Non-negative operands (post-ReLU activations, |N(0,1)| tapes: SURVEY.md 8d): sums that grow with M."""
    M, K = 3000, 512
    g = torch.Generator().manual_seed(21)
    Y = torch.randn(M, 128, generator=g).abs().to(DEV)
    X = torch.randn(M, K, generator=g).abs().to(DEV)
    ref = Y.double().T @ X.double()
    split, f32 = _both(Y, X)
    rel_s = ((split.double() - ref).abs() / ref).max().item()
    rel_f = ((f32.double() - ref).abs() / ref).max().item()
    print(f"relative error: split {rel_s:.3e}  f32-MFMA {rel_f:.3e}")
    assert rel_s <= 1e-6 + 4 * rel_f


def test_tn_split_wide_dynamic_range():
    """Columns of X scaled by 2^-100 ... 2^100, channels of Y by 2^-20 ... 2^20: every output element is judged at its own scale."""
    M, K = 640, 256
    g = torch.Generator().manual_seed(22)
    ex = torch.linspace(-100, 100, K).round()
    ey = torch.linspace(-20, 20, 128).round()
    X = (torch.rand(M, K, generator=g) * 2 - 1) * torch.exp2(ex)[None, :]
    Y = (torch.rand(M, 128, generator=g) * 2 - 1) * torch.exp2(ey)[None, :]
    X, Y = X.to(DEV), Y.to(DEV)
    ref = Y.double().T @ X.double()
    mag = Y.double().abs().T @ X.double().abs()          # sum |y| |x| per element: the scale of its rounding errors
    split, f32 = _both(Y, X)
    assert torch.isfinite(split).all()
    rel_s = ((split.double() - ref).abs() / mag).max().item()
    rel_f = ((f32.double() - ref).abs() / mag).max().item()
    print(f"error / sum|y||x|: split {rel_s:.3e}  f32-MFMA {rel_f:.3e}")
    assert rel_s <= 2e-7 + 4 * rel_f


def test_tn_split_nan_and_inf_stay_confined():
    M, K = 300, 256
    Y = torch.tensor(synth.uniform_pm1(6, (M, 128)), device=DEV)
    X = torch.tensor(synth.uniform_pm1(7, (M, K)), device=DEV)
    clean = _tn(Y, X)
    # a NaN / an infinity in X poisons exactly its output column
    for bad in (float("nan"), float("inf"), float("-inf")):
        Xb = X.clone()
        Xb[77, 130] = bad
        out = _tn(Y, Xb)
        assert not torch.isfinite(out[:, 130]).any(), bad
        if bad != bad:
            assert torch.isnan(out[:, 130]).all()
        keep = torch.ones(K, dtype=torch.bool, device=DEV)
        keep[130] = False
        torch.testing.assert_close(out[:, keep], clean[:, keep], rtol=0, atol=0)
    # ... and in Y exactly its output row
    for bad in (float("nan"), float("inf")):
        Yb = Y.clone()
        Yb[200, 5] = bad
        out = _tn(Yb, X)
        assert not torch.isfinite(out[5]).any(), bad
        keep = torch.ones(128, dtype=torch.bool, device=DEV)
        keep[5] = False
        torch.testing.assert_close(out[keep], clean[keep], rtol=0, atol=0)


def test_tn_split_subnormal_inputs():
    """Subnormal fp32 inputs may be flushed on the bf16 path: the damage is bounded by what they could contribute."""
    M, K = 256, 128
    Y = torch.tensor(synth.uniform_pm1(8, (M, 128)), device=DEV)
    X = torch.tensor(synth.uniform_pm1(9, (M, K)), device=DEV)
    X[:, 3] = 1e-40      # a whole subnormal column
    X[17, 50] = -3e-39   # one subnormal among normal values
    ref = Y.double().T @ X.double()
    out = _tn(Y, X)
    assert torch.isfinite(out).all()
    assert (out[:, 3].double() - ref[:, 3]).abs().max().item() <= 1e-40 * Y.abs().sum(0).max().item() * 1.01
    torch.testing.assert_close(out[:, 50], ref[:, 50].float(), rtol=1e-5, atol=1e-4)


def test_tn_split_is_deterministic():
    M, K = 2049, 512
    Y = torch.tensor(synth.uniform_pm1(10, (M, 128)), device=DEV)
    X = torch.tensor(synth.uniform_pm1(11, (M, K)), device=DEV)
    a = _tn(Y, X)
    b = _tn(Y, X)
    assert torch.equal(a, b)
