"""ON THE GPU BOX: first_conv forward, f32-MFMA kernel vs the split-bf16 kernel (ms per launch, HIP events)."""
import ctypes
import sys

import torch

from mucon_amd import _lib

lib = _lib.load()
B, T, D = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 4096, 2048)))
tape = torch.randn(B, T, D, device="cuda")
W = torch.randn(128, D, device="cuda") * 0.02
b = torch.randn(128, device="cuda")
out = torch.empty(B, T, 128, device="cuda")
planes = torch.empty(3 * 128 * D * 2, dtype=torch.uint8, device="cuda")
ms = ctypes.c_float()
s = _lib.current_stream_ptr()
for rep in range(3):
    _lib.check(lib.mucon_bench_first_conv(_lib.ptr(tape), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), B, T, D, 50, ctypes.byref(ms), s), "f32")
    f32 = ms.value
    _lib.check(lib.mucon_test_first_conv_split(_lib.ptr(tape), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), B, T, D, 1, _lib.ptr(planes),
                                               planes.numel(), 50, ctypes.byref(ms), s), "split")
    gb = (B * T * D * 4 + B * T * 128 * 4) / 1e9
    print(f"B={B} T={T} D={D}: f32-MFMA {f32*1e3:.1f} us   split-bf16 {ms.value*1e3:.1f} us  ({gb/ms.value:.0f} GB/s algorithmic, "
          f"{2*B*T*D*128/ms.value/1e9:.1f} TFLOP/s fp32-equivalent)")
