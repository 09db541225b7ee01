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
