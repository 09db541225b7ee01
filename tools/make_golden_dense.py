"""Dense-path goldens from the reference's own modules (called by tools/make_golden.py dense).

Builds the reference MuCon (src/mucon/models.py:134-317) with the default config
(src/configs/mucon/default.py), overwrites the hot-path parameters with the seeded values of
oracle.dense.seeded_params, switches to eval() (dropout off) and records
  enc    = MuCon.temporal_modeling_forward(tape)                 (models.py:746-773)
  logits = MuCon.frame_classifier_forward(enc^T, T) at z resolution + the nearest index vector
  logp   = F.log_softmax(logits, dim=1)                          (models.py:368)
and, for one small case, the reference's autograd gradients of a seeded linear functional.
"""
import os

import numpy as np
import torch

from mucon_amd import synth
from oracle import dense as od

CASES = [  # name, B, T, overrides
    ("t130", 1, 130, {}),
    ("t2000", 1, 2000, {}),
    ("t2097", 1, 2097, {}),       # odd lengths at several pooling levels
    ("b2_t777", 2, 777, {}),      # batch of 2: GroupNorm statistics are per sample
    ("t4096", 1, 4096, {}),
    ("sum_pool", 1, 1000, {"pooling_type": "sum"}),
    ("leaky", 1, 1000, {"leaky_relu": True}),
    ("no_gn", 1, 1000, {"last_gn": False}),
]


def build_reference_model(cfg_over):
    from configs.mucon.default import get_cfg_defaults
    from mucon.models import create_model

    cfg = get_cfg_defaults()
    for k, v in cfg_over.items():
        cfg.model.ft[k] = v
    torch.manual_seed(0)
    model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
    model.eval()
    return model, cfg


def load_seeded(model, ocfg, seed):
    params = od.seeded_params(ocfg, seed)
    sd = model.state_dict()
    for k, v in params.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)
    return params


def main(gold_dir):
    out = {}
    for ci, (name, B, T, over) in enumerate(CASES):
        ocfg = od.EncoderConfig(**over)
        model, _ = build_reference_model(over)
        load_seeded(model, ocfg, seed=11 + ci)
        tape = torch.from_numpy(synth.tape(900 + ci, B, T, 2048))
        with torch.no_grad():
            enc = model.temporal_modeling_forward(tape)                       # [B, Tz, H]
            seg = torch.stack([model.frame_classifier_forward(enc[b:b + 1].permute(0, 2, 1), T)[0].permute(1, 0)
                               for b in range(B)])                            # [B, T, C]
            logp = torch.log_softmax(seg, dim=2)
        Tz = enc.shape[1]
        idx = od.nearest_index(Tz, T)
        # the y-head is a per-frame gather of z-level rows: store the z-level rows + check the gather
        first = np.array([np.argmax(idx == z) for z in range(Tz)])
        assert np.array_equal(seg[:, first][:, idx].numpy(), seg.numpy())
        out[f"{name}__enc"] = enc.numpy()
        out[f"{name}__logits_z"] = seg[:, first].numpy()
        out[f"{name}__logp_z"] = logp[:, first].numpy()
        out[f"{name}__idx"] = idx.astype(np.int32)
        out[f"{name}__meta"] = np.asarray([B, T, Tz, 11 + ci, 900 + ci])
        print(f"  dense {name:10s} B={B} T={T} Tz={Tz} |enc|max={enc.abs().max():.3f} logp range=({logp.min():.2f},{logp.max():.2f})")

    # gradients of L = sum(w*logp) + sum(v*enc), default config, small T
    B, T = 1, 600
    ocfg = od.EncoderConfig()
    model, _ = build_reference_model({})
    load_seeded(model, ocfg, seed=31)
    tape = torch.from_numpy(synth.tape(931, B, T, 2048))
    enc = model.temporal_modeling_forward(tape)
    seg = model.frame_classifier_forward(enc.permute(0, 2, 1), T)[0].permute(1, 0)
    logp = torch.log_softmax(seg, dim=1)
    w = torch.from_numpy(synth.uniform_pm1(932, (T, 48)))
    v = torch.from_numpy(synth.uniform_pm1(933, tuple(enc.shape)))
    L = (w * logp).sum() + (v * enc).sum()
    L.backward()
    named = dict(model.named_parameters())
    out["grads__meta"] = np.asarray([B, T, enc.shape[1], 31, 931, 932, 933])
    out["grads__L"] = np.asarray([L.item()], dtype=np.float64)
    for i, k in enumerate(od.param_shapes(ocfg)):
        g = named[k].grad.detach().numpy().reshape(-1)
        sel = synth.integers(7000 + i, min(256, g.size), 0, g.size)
        out[f"grads__{k}__norm"] = np.asarray([np.linalg.norm(g.astype(np.float64))])
        out[f"grads__{k}__idx"] = sel.astype(np.int64)
        out[f"grads__{k}__val"] = g[sel].copy()
    print(f"  dense grads T={T} L={L.item():.4f}")
    np.savez_compressed(os.path.join(gold_dir, "dense_cases.npz"), **out)
