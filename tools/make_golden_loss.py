"""Loss goldens from the reference (build container only): MuCon.loss (reference src/mucon/models.py:376-565, with
src/mucon/masks.py) on seeded head outputs -- segmentation logits [T x 48], transcript log-probs [N+1 x 49], length
logits [N] -- over the loss configurations (flint / arithmetic, box / gaussian / trapezoid templates, overlap,
background class weights, smoothing on logits / log-probs, clamp active, transcript averaging, without teacher
forcing).  Stores the five loss values and the gradients of `main` w.r.t. the three inputs.
Pins oracle/losses.py (tests/test_oracle_losses.py) and, through it, the fused HIP loss kernels."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402

ref_harness.install()
from mucon_amd import synth  # noqa: E402

# name: (T, N, seed, teacher_forcing, [config overrides])
CASES = {
    "flint_box": (300, 4, 21, True, []),
    "flint_gauss_ov": (257, 5, 22, True, ["model.loss.mucon.template", "gaussian", "model.loss.mucon.overlap", 0.1]),
    "flint_trap_w": (190, 3, 23, True, ["model.loss.mucon.template", "trapezoid", "model.loss.mucon.overlap", 0.2,
                                        "model.loss.mucon_weight_background", True,
                                        "model.loss.transcript_weight_background", True,
                                        "model.loss.transcript_average", True]),
    "arith_box": (300, 4, 24, True, ["model.loss.mucon.type", "arithmetic"]),
    "arith_gauss_w": (222, 6, 25, True, ["model.loss.mucon.type", "arithmetic", "model.loss.mucon.template", "gaussian",
                                         "model.loss.mucon.overlap", 0.1, "model.loss.mucon_weight_background", True]),
    "smooth_logits_clamped": (150, 2, 26, True, ["model.loss.smoothing.log_softmax_before", False,
                                                 "model.loss.smoothing.clamp_max", 0.5]),
    "smooth_noclamp_mul": (150, 7, 27, True, ["model.loss.smoothing.clamp", False, "model.loss.mul_smoothing", 0.7,
                                              "model.loss.mul_length", 1.3, "model.loss.mul_mucon", 0.4,
                                              "model.loss.mul_transcript", 2.0, "model.loss.length_width", 0.3]),
    "no_tf": (300, 5, 28, False, []),
    "long": (2000, 12, 29, True, []),
    "one_segment": (64, 1, 30, True, []),
}


def inputs(T, N, seed):
    seg = synth.uniform_pm1(seed, (T, 48)).astype(np.float32) * np.float32(3.0)
    tl = synth.uniform_pm1(seed + 1, (N + 1, 49)).astype(np.float32) * np.float32(2.0)
    ln = synth.uniform_pm1(seed + 2, (N,)).astype(np.float32) * np.float32(3.0)
    tr = synth.transcript(seed + 3, N, 48, allow_repeats=True)
    if N >= 2:
        tr[0] = 0  # the background class: exercises the class weights
    return seg, tl, ln, tr


def main():
    from configs.mucon.default import get_cfg_defaults
    from core.datasets.general_dataset import Batch
    from mucon.models import MuConForwardOut, create_model

    out = {}
    meta = {}
    # every configuration under both grid conventions: "<case>" = align_corners False (this container's torch default, the
    # round-1 goldens), "<case>@ac" = True (the reference's pinned PyTorch 1.1; ref_harness.set_grid_convention)
    for case, (T, N, seed, tf, over), ac in [(c + ("@ac" if a else ""), v, a) for c, v in CASES.items() for a in (False, True)]:
        ref_harness.set_grid_convention(ac)
        meta[case] = {"T": T, "N": N, "seed": seed, "teacher_forcing": tf,
                      "overrides": list(over) + ["model.loss.mucon.align_corners", ac]}
        cfg = get_cfg_defaults()
        for key, val in zip(over[::2], over[1::2]):   # the harness's CfgNode stub has no merge_from_list
            node = cfg
            parts = key.split(".")
            for part in parts[:-1]:
                node = getattr(node, part)
            node[parts[-1]] = val
        model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
        model.eval()
        model.set_teacher_forcing(tf)
        seg, tl, ln, tr = inputs(T, N, seed)
        seg_t = torch.from_numpy(seg).requires_grad_(True)
        tl_raw = torch.from_numpy(tl).requires_grad_(True)
        tlogp = torch.log_softmax(tl_raw, dim=1)
        tlogp.retain_grad()
        ln_t = torch.from_numpy(ln).requires_grad_(True)
        batch = Batch(feats=torch.zeros(1, T, 1), gt_label=torch.zeros(T, dtype=torch.long), transcript=torch.from_numpy(tr),
                      transcript_tf_input=torch.tensor([49] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [48]),
                      video_name="synthetic")
        fo = MuConForwardOut(transcript=tlogp, lengths=ln_t, segmentation=seg_t)
        loss = model.loss(batch, fo)
        loss.main.backward()
        vals = np.asarray([loss.main.item(), loss.transcript_loss.item(), loss.length_loss.item(), loss.mucon_loss.item(),
                           loss.smoothing_loss.item()], dtype=np.float64)
        out[f"{case}__meta"] = np.asarray([T, N, seed, int(tf)])
        out[f"{case}__losses"] = vals
        sub = 1 if T <= 400 else 7
        out[f"{case}__d_seg"] = seg_t.grad.numpy()[::sub]
        out[f"{case}__d_seg_norm"] = np.asarray(np.linalg.norm(seg_t.grad.numpy().astype(np.float64)))
        out[f"{case}__d_tlogp"] = tlogp.grad.numpy()
        out[f"{case}__d_lengths"] = ln_t.grad.numpy()
        print(case, vals, "|d_seg|", float(out[f"{case}__d_seg_norm"]), "d_len", ln_t.grad.numpy()[:3])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss_cases.npz"), **out)
    import json
    with open(os.path.join(ROOT, "tests", "golden", "loss_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
