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

    def mix(x):
        x &= M
        x ^= x >> 16
        x = (x * 0x7feb352d) & M
        x ^= x >> 15
        x = (x * 0x846ca68b) & M
        x ^= x >> 16
        return x

    lo, hi = seed & M, (seed >> 32) & M
    s0 = mix(hi ^ mix(lo + 0x9E3779B9 * (site + 1)))
    s1 = (mix(s0 ^ 0x85EBCA6B) + hi) & M
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


# ---------------------------------------------------------------------------- s-head goldens
def seeded_model_value(name, shape):
    """Parameter recipe of tools/make_golden_model.py / make_golden_shead.py (seed = crc32 of the name)."""
    import zlib
    from mucon_amd import synth
    u = synth.uniform_pm1(zlib.crc32(name.encode()), tuple(shape))
    if name == "ft_last_gn.weight":
        return np.float32(1.0) + np.float32(0.25) * u
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return u * np.float32(2.0 ** -int(round(np.log2(np.sqrt(max(fan_in, 1))))))
    return u * np.float32(0.125)


SHEAD_SHAPES = {
    "fs_decoder_attention_W1": (256, 128), "fs_decoder_attention_V": (128,),
    **{f"fs_encoder_lstm.{n}_l0{s}": sh for s in ("", "_reverse")
       for n, sh in (("weight_ih", (512, 128)), ("weight_hh", (512, 128)), ("bias_ih", (512,)), ("bias_hh", (512,)))},
    "fs_encoder_hidden_out.weight": (128, 256), "fs_encoder_hidden_out.bias": (128,),
    "fs_encoder_cn_out.weight": (128, 256), "fs_encoder_cn_out.bias": (128,),
    "fs_decoder_attention_l2.weight": (128, 128), "fs_decoder_attention_l2.bias": (128,),
    "fs_decoder_embedding.weight": (50, 128),
    "fs_decoder_attn_combine.weight": (128, 384), "fs_decoder_attn_combine.bias": (128,),
    "fs_decoder_lstm.weight_ih_l0": (512, 128), "fs_decoder_lstm.weight_hh_l0": (512, 128),
    "fs_decoder_lstm.bias_ih_l0": (512,), "fs_decoder_lstm.bias_hh_l0": (512,),
    "fs_decoder_transcript.0.weight": (128, 128), "fs_decoder_transcript.0.bias": (128,),
    "fs_decoder_transcript.2.weight": (49, 128), "fs_decoder_transcript.2.bias": (49,),
    "fs_decoder_length.0.weight": (64, 177), "fs_decoder_length.0.bias": (64,),
    "fs_decoder_length.2.weight": (1, 64), "fs_decoder_length.2.bias": (1,),
}


def shead_params(gold, case):
    """The s-head parameters of a tests/golden/shead_cases.npz case, as float32 torch tensors by state_dict name."""
    import torch
    scale = np.float32(gold[f"{case}__scale"])
    out = {}
    for name, shape in SHEAD_SHAPES.items():
        v = seeded_model_value(name, shape).astype(np.float32)
        if name.startswith("fs_decoder") and v.ndim >= 2:
            v = v * scale
        out[name] = torch.from_numpy(v)
    return out


def shead_case(gold, case):
    import torch
    from mucon_amd import synth
    Tz, N, seed, eos = [int(x) for x in gold[f"{case}__meta"]]
    tr = synth.transcript(seed + 1, N, 48, allow_repeats=True)
    return {
        "Tz": Tz, "N": N, "eos": eos,
        "enc": torch.from_numpy(synth.uniform_pm1(seed, (1, Tz, 128)).astype(np.float32))[0],
        "tf_in": torch.tensor([49] + tr.tolist()),
        "R1": torch.from_numpy(synth.uniform_pm1(seed + 2, (N + 1, 49)).astype(np.float32)),
        "r2": torch.from_numpy(synth.uniform_pm1(seed + 3, (N + 1,)).astype(np.float32)),
    }


# ---------------------------------------------------------------------------- loss goldens
def loss_inputs(T, N, seed):
    """Seeded head outputs of tools/make_golden_loss.py: (segmentation [T,48], transcript logits [N+1,49], length
    logits [N], transcript [N]) as numpy."""
    from mucon_amd import synth
    seg = synth.uniform_pm1(seed, (T, 48)).astype(np.float32) * np.float32(3.0)
    tl = synth.uniform_pm1(seed + 1, (N + 1, 49)).astype(np.float32) * np.float32(2.0)
    ln = synth.uniform_pm1(seed + 2, (N,)).astype(np.float32) * np.float32(3.0)
    tr = synth.transcript(seed + 3, N, 48, allow_repeats=True)
    if N >= 2:
        tr[0] = 0
    return seg, tl, ln, tr


def load_pruned_golden():
    """tests/golden/viterbi_pruned.{npz,json}: the reference's beam search (tools/make_golden_pruned.py)."""
    import json
    z = np.load(os.path.join(GOLDEN, "viterbi_pruned.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "viterbi_pruned.json")))
    return z, meta


def pruned_case_inputs(rec):
    """(lp [T x C] float32, transcript int32 [N], P [J x N] float64) of one record of viterbi_pruned.json -- the generator's inputs,
    restated (tools/make_golden_pruned.py:build_inputs)."""
    import oracle
    from mucon_amd import synth
    T, seed, tr = rec["T"], rec["seed"], np.asarray(rec["transcript"], dtype=np.int64)
    if rec["emissions"] == "const":
        lp = np.full((T, C), np.float32(-1.0), np.float32)
    elif rec["emissions"] == "noise":
        lp = synth.emissions(seed, T, C, labels=None)
    else:
        lp = synth.emissions(seed, T, C, labels=synth.segment_labels(seed + 11, T, tr))
    fs, max_len = rec["fs"], rec["max_len"]
    J = max_len // fs
    if rec["length_model"] == "flat":
        P = np.zeros((J, len(tr)), dtype=np.float64)
        P[(np.arange(1, J + 1) * fs) >= max_len, :] = -np.inf
    else:
        with np.errstate(all="ignore"):
            P = oracle.length_rows(oracle.poisson_table(np.asarray(rec["mu"]), max_len), tr, fs, max_len)
    return lp, tr.astype(np.int32), P
