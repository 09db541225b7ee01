"""CPU: oracle/dense.py (the restatement the HIP kernels are compared against) versus the outputs
of the reference's own modules (tests/golden/dense_cases.npz, made by tools/make_golden_dense.py).

Tolerance: the reference runs in float32 on torch CPU kernels; fp32 vs fp64
evaluation of the same graph differs by <= ~1e-6 abs on values of magnitude <= 5, so
atol = rtol = 1e-4 on encodings / log-probs and rtol = 1e-3 on gradients (vs the tensor norm)."""
import os

import numpy as np
import pytest
import torch

from mucon_amd import synth
from oracle import dense as od

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dense_cases.npz"))
CASES = [("t130", {}), ("t2000", {}), ("t2097", {}), ("b2_t777", {}), ("t4096", {}),
         ("sum_pool", {"pooling_type": "sum"}), ("leaky", {"leaky_relu": True}), ("no_gn", {"last_gn": False})]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("name,over", CASES, ids=[c[0] for c in CASES])
def test_dense_oracle_matches_reference(name, over, dtype):
    B, T, Tz, pseed, tseed = [int(x) for x in G[f"{name}__meta"]]
    cfg = od.EncoderConfig(**over)
    assert cfg.out_length(T) == Tz
    params = od.seeded_params(cfg, pseed)
    enc, logits, logp = od.hot_path(synth.tape(tseed, B, T, 2048), params, cfg, dtype)
    idx = G[f"{name}__idx"]
    np.testing.assert_array_equal(od.nearest_index(Tz, T), idx)
    np.testing.assert_allclose(enc, G[f"{name}__enc"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(logits, G[f"{name}__logits_z"][:, idx], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(logp, G[f"{name}__logp_z"][:, idx], rtol=1e-4, atol=1e-4)


def test_dense_oracle_grads_match_reference():
    B, T, Tz, pseed, tseed, wseed, vseed = [int(x) for x in G["grads__meta"]]
    cfg = od.EncoderConfig()
    params = od.seeded_params(cfg, pseed)
    w = synth.uniform_pm1(wseed, (B, T, 48))
    v = synth.uniform_pm1(vseed, (B, Tz, 128))
    grads, L = od.hot_path_grads(synth.tape(tseed, B, T, 2048), params, cfg, w, v, torch.float64)
    assert abs(L - G["grads__L"][0]) < 1e-3 * abs(G["grads__L"][0])
    for k in od.param_shapes(cfg):
        g = grads[k].reshape(-1)
        norm = float(G[f"grads__{k}__norm"][0])
        assert abs(np.linalg.norm(g) - norm) <= 1e-3 * norm, k
        sel = G[f"grads__{k}__idx"]
        np.testing.assert_allclose(g[sel], G[f"grads__{k}__val"], rtol=1e-3, atol=1e-3 * norm / np.sqrt(g.size), err_msg=k)


def test_nearest_index_rule_f32_vs_f64():
    """torch computes the nearest source index in float32; the float64 form agrees on Breakfast-like sizes."""
    for T in [130, 777, 2000, 2097, 4096, 9741, 16384]:
        Tz = od.EncoderConfig().out_length(T)
        f64 = np.minimum(np.floor(np.arange(T) * (Tz / T)).astype(np.int64), Tz - 1)
        np.testing.assert_array_equal(od.nearest_index(Tz, T), f64)
