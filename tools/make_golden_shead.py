"""s-head goldens from the reference (build container only): MuCon.sequence_generation_forward (reference
src/mucon/models.py:585-744) on a seeded temporal encoding [1 x Tz x 128], every parameter from the
platform-independent generator (seed = crc32 of the reference's parameter name, as make_golden_model.py).

  tf:     eval() + teacher forcing: the per-step log-probs and lengths, and the gradients of
          sum(transcript * R1) + sum(lengths * r2) w.r.t. the input and every s-head parameter (small tensors
          whole, large ones every 29th element).
  greedy: eval() without teacher forcing: arg-max feedback, stop at EOS or max_decoding_steps.
Pins oracle/shead.py (tests/test_oracle_shead.py) and, through it, the HIP LSTM / decoder kernels."""
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
from make_golden_model import seeded_value  # noqa: E402

SUB = 29


def main():
    from configs.mucon.default import get_cfg_defaults
    from mucon.models import create_model

    out = {}
    cfg = get_cfg_defaults()
    # (Tz, N, seed, scale on the decoder's matrices, EOS id): scale > 1 leaves the near-linear regime of the
    # small seeded weights (distinct words per step); case "e" re-labels EOS so that the greedy loop stops early
    cases = {"a": (25, 4, 11, 1.0, 48), "b": (125, 7, 12, 1.0, 48), "c": (3, 1, 13, 1.0, 48),
             "d": (60, 5, 14, 4.0, 48), "e": (125, 3, 12, 8.0, 5)}
    for case, (Tz, N, seed, scale, eos) in cases.items():
        model = create_model(cfg, num_classes=48, max_decoding_steps=12, input_feature_size=2048)
        with torch.no_grad():
            for name, p in model.named_parameters():
                v = seeded_value(name, p.shape).astype(np.float32)
                if name.startswith("fs_decoder") and v.ndim >= 2:
                    v = v * np.float32(scale)
                p.copy_(torch.from_numpy(v))
        model.eval()
        model.EOS_token_id = eos
        names = [n for n, _ in model.named_parameters() if n.startswith("fs_") and "attention_l3" not in n]
        named = dict(model.named_parameters())
        enc = torch.from_numpy(synth.uniform_pm1(seed, (1, Tz, 128)).astype(np.float32)).requires_grad_(True)
        tr = synth.transcript(seed + 1, N, 48, allow_repeats=True)
        tf_in, tf_tgt = torch.tensor([49] + tr.tolist()), torch.tensor(tr.tolist() + [48])
        model.set_teacher_forcing(True)
        model.zero_grad()
        transcripts, lengths = model.sequence_generation_forward(enc, N + 1, tf_in, tf_tgt)
        logp, lens = torch.cat(transcripts, 0), torch.stack(lengths)
        R1 = torch.from_numpy(synth.uniform_pm1(seed + 2, tuple(logp.shape)).astype(np.float32))
        r2 = torch.from_numpy(synth.uniform_pm1(seed + 3, tuple(lens.shape)).astype(np.float32))
        ((logp * R1).sum() + (lens * r2).sum()).backward()
        out[f"{case}__meta"] = np.asarray([Tz, N, seed, eos])
        out[f"{case}__scale"] = np.asarray(scale, dtype=np.float32)
        out[f"{case}__tf_logp"] = logp.detach().numpy()
        out[f"{case}__tf_lengths"] = lens.detach().numpy()
        out[f"{case}__tf_d_enc"] = enc.grad.numpy()[0]
        for n in names:
            g = named[n].grad.numpy().reshape(-1)
            out[f"{case}__tf_grad__{n}"] = g if g.size <= 4096 else g[::SUB]
            out[f"{case}__tf_gnorm__{n}"] = np.asarray(np.linalg.norm(g.astype(np.float64)))
        model.set_teacher_forcing(False)
        with torch.no_grad():
            transcripts, lengths = model.sequence_generation_forward(enc, N + 1, tf_in, tf_tgt)
        glp = torch.cat(transcripts, 0)
        out[f"{case}__greedy_logp"] = glp.numpy()
        out[f"{case}__greedy_lengths"] = torch.stack(lengths).numpy()
        top2 = glp.topk(2, dim=1).values
        print(case, "tf steps", logp.shape[0], "greedy steps", len(transcripts), "words",
              [int(t.argmax()) for t in transcripts], "min arg-max margin %.3g" % float((top2[:, 0] - top2[:, 1]).min()))
    out["param_names"] = np.asarray(names)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "shead_cases.npz"), **out)


if __name__ == "__main__":
    main()
