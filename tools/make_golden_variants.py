"""Goldens for the non-default encoders (build container only): the reference's MuCon.temporal_modeling_forward with
cfg.model.ft.type = "noft" / "mstcnpp" (src/core/modules/temporal.py:56-74, :150-204; models.py:746-773) on a seeded tape,
eval mode, parameters from the crc32-seeded generator.  Pins mucon_amd.core.modules.temporal.{NoFt, MSTCNPPFirstStage}."""
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
from make_golden_model import seeded_value  # noqa: E402


def main():
    from configs.mucon.default import get_cfg_defaults
    from mucon.models import create_model

    out = {}
    for kind, T in (("noft", 77), ("mstcnpp", 300)):
        cfg = get_cfg_defaults()
        cfg.model.ft.type = kind
        model = create_model(cfg, num_classes=48, max_decoding_steps=31, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(seeded_value(name, p.shape).astype(np.float32)))
        model.eval()
        tape = torch.from_numpy(synth.uniform_pm1(55, (1, T, 2048)))
        with torch.no_grad():
            enc = model.temporal_modeling_forward(tape)
        out[f"{kind}__enc"] = enc.numpy()
        out[f"{kind}__meta"] = np.asarray([T, enc.shape[1]])
        out[f"{kind}__keys"] = np.asarray([k for k in model.state_dict() if k.startswith("ft.")])
        print(kind, tuple(enc.shape), float(enc.abs().max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "variant_cases.npz"), **out)


if __name__ == "__main__":
    main()
