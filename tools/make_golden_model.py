"""Full-model goldens from the reference (build container only): MuCon.forward + predict + loss on one
tiny video in eval() mode with teacher forcing, every parameter set from the platform-independent
generator (seed = crc32 of the reference's parameter name).  Pins the model surface of
mucon_amd/mucon/models.py (reference src/mucon/models.py:319-396)."""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402

ref_harness.install()
from mucon_amd import synth  # noqa: E402


def seeded_value(name, shape):
    u = synth.uniform_pm1(zlib.crc32(name.encode()), tuple(shape))
    if name == "ft_last_gn.weight":
        return np.float32(1.0) + np.float32(0.25) * u
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return u * np.float32(2.0 ** -int(round(np.log2(np.sqrt(max(fan_in, 1))))))
    return u * np.float32(0.125)


def make_batch(T, N, C=48, seed=5):
    from core.datasets.general_dataset import Batch
    tr = synth.transcript(seed, N, C, allow_repeats=False)
    gt = synth.segment_labels(seed + 1, T, tr)
    feats = synth.uniform_pm1(seed + 2, (1, T, 2048))
    return Batch(feats=torch.from_numpy(feats), gt_label=torch.from_numpy(gt), transcript=torch.from_numpy(tr),
                 transcript_tf_input=torch.tensor([C + 1] + tr.tolist()), transcript_tf_target=torch.tensor(tr.tolist() + [C]),
                 video_name="synthetic")


def main():
    from configs.mucon.default import get_cfg_defaults
    from mucon.models import create_model

    out = {}
    for case, (T, N, over) in {"base": (400, 4, {}), "arith": (333, 3, {"type": "arithmetic"}),
                               "gauss": (450, 5, {"template": "gaussian", "overlap": 0.1})}.items():
        cfg = get_cfg_defaults()
        for k, v in over.items():
            cfg.model.loss.mucon[k] = v
        model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
        model.eval()
        model.set_teacher_forcing(True)
        batch = make_batch(T, N)
        fo = model.forward(batch)
        loss = model.loss(batch, fo)
        pred = model.predict(batch, fo)
        loss.main.backward()
        out[f"{case}__meta"] = np.asarray([T, N])
        out[f"{case}__transcript"] = fo.transcript.detach().numpy()
        out[f"{case}__lengths"] = fo.lengths.detach().numpy()
        out[f"{case}__segmentation_sub"] = fo.segmentation.detach().numpy()[::7]
        out[f"{case}__pred_lengths"] = pred.lengths.detach().numpy()
        out[f"{case}__pred_logp_sub"] = pred.segmentation_logits.detach().numpy()[::7]
        out[f"{case}__loss"] = np.asarray([loss.main.item(), loss.transcript_loss.item(), loss.mucon_loss.item(),
                                           loss.length_loss.item(), loss.smoothing_loss.item()])
        names = ["ft.first_conv.weight", "ft.l_3.dilated_conv.weight", "ft.l_10.conv_1x1.bias", "ft_last_gn.weight",
                 "conv_classifier.weight", "fs_encoder_lstm.weight_ih_l0", "fs_decoder_attention_W1",
                 "fs_decoder_length.2.weight"]
        named = dict(model.named_parameters())
        out[f"{case}__grad_norms"] = np.asarray([named[k].grad.norm().item() for k in names])
        print(case, "loss", out[f"{case}__loss"], "grad norms", out[f"{case}__grad_norms"][:4])
    out["grad_names"] = np.asarray(names)
    sd = model.state_dict()
    out["state_keys"] = np.asarray(list(sd.keys()))
    out["state_shapes"] = np.asarray([",".join(str(d) for d in v.shape) for v in sd.values()])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "model_cases.npz"), **out)


if __name__ == "__main__":
    main()
