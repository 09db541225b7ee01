"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mucon_hip.h
declares.  No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("mucon_hip.h", "mucon_hip_test.h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mucon_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from mucon_amd import _lib

    assert declared_symbols() == sorted(_lib.SYMBOLS), "include/mucon_hip*.h and mucon_amd/_lib.py list different entry points"


def test_library_builds_and_exports_every_symbol():
    from mucon_amd import build

    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} not exported"
    lib.mucon_abi_version.restype = ctypes.c_int
    assert lib.mucon_abi_version() == 8


def test_host_only_queries():
    """Entry points that do not touch the device: shapes, workspace sizes, argument validation."""
    from mucon_amd import _lib
    from mucon_amd.ops import EncoderSpec

    lib = _lib.load()
    spec = EncoderSpec()
    for T in (130, 2000, 2097, 4096, 9741, 16384):
        cfg = spec.to_c(1, T, False, 0)
        assert lib.mucon_encoder_out_length(ctypes.byref(cfg)) == spec.out_length(T)
        assert lib.mucon_encoder_workspace_bytes(ctypes.byref(cfg)) > T * 128 * 4
    bad = spec.to_c(1, 8, False, 0)      # too short for 4 poolings
    assert lib.mucon_encoder_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"too short" in lib.mucon_last_error()
    bad = EncoderSpec(hidden=64).to_c(1, 100, False, 0)
    assert lib.mucon_encoder_out_length(ctypes.byref(bad)) == -1
    assert b"hidden size" in lib.mucon_last_error()
    assert lib.mucon_viterbi_job_workspace_bytes(3000, 48, 5, 30) >= 100 * 48 * 4 + 100 * 5
    assert ctypes.sizeof(_lib.ViterbiJob) == 64


def test_product_has_no_cpu_fallback():
    """CPU tensors must raise, not silently compute somewhere else."""
    import torch
    from mucon_amd import _lib, ops

    spec = ops.EncoderSpec()
    with pytest.raises(_lib.MuconHipError):
        ops.head_forward(torch.zeros(1, 8, 128), torch.zeros(48, 128, 1), torch.zeros(48), 130)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mucon_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
