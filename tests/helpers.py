"""Shared helpers for the parity tests: rebuild the seeded inputs of a golden Viterbi case."""
import json
import os

import numpy as np

from mucon_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C = 48


def load_viterbi_golden():
    z = np.load(os.path.join(GOLDEN, "viterbi_cases.npz"))
    with open(os.path.join(GOLDEN, "viterbi_cases.json")) as f:
        meta = json.load(f)
    return z, meta


def viterbi_case_inputs(z, cs):
    """Emissions [T x C] f32 for golden case `cs` (same recipe as tools/make_golden.py)."""
    nm, T, seed, kind = cs["name"], cs["T"], cs["seed"], cs["kind"]
    tr = z[f"{nm}__transcript"].astype(np.int64)
    if kind == "stored":
        return z[f"{nm}__lp"]
    if kind == "poisson":
        return synth.emissions(seed, T, C, labels=synth.segment_labels(seed + 11, T, tr))
    if kind in ("poisson_noise", "flat"):
        return synth.emissions(seed, T, C)
    if kind == "flat_const":
        return np.full((T, C), np.float32(-1.0), np.float32)
    raise ValueError(kind)


def f64_bits(x):
    return np.asarray(x, dtype=np.float64).view(np.uint64)


def dropout_keep_np(n, seed, site, p):
    """numpy replay of the kernels' counter-based dropout (mucon_amd/csrc/common.hpp: mix32 / make_drop):
    keep[e] for element index e of dropout site `site`."""
    M = 0xFFFFFFFF
    s0 = ((seed & M) ^ ((0x85EBCA6B * (site + 1)) & M)) & M
    s1 = ((seed >> 32) + 0xC2B2AE35 * (site + 1)) & M
    t = p * 4294967296.0
    thresh = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) ^ np.uint64(s0)) & np.uint64(M)
        x = (x * np.uint64(0x9E3779B1) + np.uint64(s1)) & np.uint64(M)
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7feb352d)) & np.uint64(M)
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846ca68b)) & np.uint64(M)
        x ^= x >> np.uint64(16)
    return x >= np.uint64(thresh)
