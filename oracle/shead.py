"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the s-head in explicit float64 torch tensor arithmetic -- no nn.LSTM,
no nn.Linear: every equation is written out, so the HIP kernels (csrc/lstm.hpp, csrc/decoder.hpp) are compared
against the formulas, not against another library kernel.  Gradients come from autograd over these formulas.

  lstm()      torch.nn.LSTM(128, 128, batch_first, bidirectional) as the reference constructs and calls it:
              reference src/mucon/models.py:195-201, :605-611 (batch 1, zero initial state, gate order i,f,g,o)
  decoder()   the decoding loop, reference src/mucon/models.py:612-728, and the additive attention, :730-744

Parameters are a dict keyed by the reference's state_dict names (fs_encoder_lstm.weight_ih_l0, ...,
fs_decoder_length.2.bias) in the reference's layouts.

Parity pin: tests/golden/shead_cases.npz -- outputs and gradients of the reference's own
MuCon.sequence_generation_forward (made by tools/make_golden_shead.py); checked in tests/test_oracle_shead.py.
"""
from typing import Dict, Optional

import torch

F64 = torch.float64


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse: bool):
    """x [T, I] -> (out [T, H], h_T [H], c_T [H]) for one direction."""
    T, H = x.shape[0], w_hh.shape[1]
    h = torch.zeros(H, dtype=x.dtype)
    c = torch.zeros(H, dtype=x.dtype)
    out = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = w_ih @ x[t] + b_ih + w_hh @ h + b_hh
        i, f = torch.sigmoid(g[:H]), torch.sigmoid(g[H:2 * H])
        gg, o = torch.tanh(g[2 * H:3 * H]), torch.sigmoid(g[3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        out[t] = h
    return torch.stack(out), h, c


def lstm(x, params: Dict[str, torch.Tensor], prefix: str = "fs_encoder_lstm", bidirectional: bool = True):
    """x [T, I] -> (out [T, ndir*H], h_n [ndir, H], c_n [ndir, H])."""
    outs, hs, cs = [], [], []
    for d, suffix in enumerate(["", "_reverse"][:2 if bidirectional else 1]):
        w = [params[f"{prefix}.{n}_l0{suffix}"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        o, h, c = lstm_direction(x, *w, reverse=(d == 1))
        outs.append(o)
        hs.append(h)
        cs.append(c)
    return torch.cat(outs, dim=1), torch.stack(hs), torch.stack(cs)


def decoder(memory, h_n, c_n, params: Dict[str, torch.Tensor], tf_input, steps: int, teacher_forcing: bool,
            stop_on_eos: bool, eos: int, dropmask: Optional[torch.Tensor] = None):
    """memory [Tz, 2E], h_n / c_n [ndir, E] -> (logp [n, M+1], lengths [n]).

    dropmask [steps, D]: the embedding-dropout keep mask already divided by (1 - p) (None = eval mode)."""
    P = params
    lin = lambda name, v: P[f"{name}.weight"] @ v + P[f"{name}.bias"]  # noqa: E731
    h = lin("fs_encoder_hidden_out", h_n.reshape(-1))                   # models.py:612-614
    c = lin("fs_encoder_cn_out", c_n.reshape(-1))                       # models.py:615-617
    mp = memory @ P["fs_decoder_attention_W1"]                          # models.py:619-622
    H = h.shape[0]
    tok = int(tf_input[0])
    logps, lengths = [], []
    for step in range(steps):
        if teacher_forcing:
            tok = int(tf_input[step])
        emb = torch.relu(P["fs_decoder_embedding.weight"][tok])         # models.py:655-659
        if dropmask is not None:
            emb = emb * dropmask[step]
        q = lin("fs_decoder_attention_l2", h)                           # models.py:730-744
        score = torch.tanh(mp + q) @ P["fs_decoder_attention_V"]
        attn = torch.softmax(score, dim=0)
        context = (attn.unsqueeze(1) * memory).sum(dim=0)               # models.py:668-672
        mixed = torch.relu(lin("fs_decoder_attn_combine", torch.cat((emb, context))))
        g = (P["fs_decoder_lstm.weight_ih_l0"] @ mixed + P["fs_decoder_lstm.bias_ih_l0"]
             + P["fs_decoder_lstm.weight_hh_l0"] @ h + P["fs_decoder_lstm.bias_hh_l0"])
        i, f = torch.sigmoid(g[:H]), torch.sigmoid(g[H:2 * H])
        gg, o = torch.tanh(g[2 * H:3 * H]), torch.sigmoid(g[3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        logits = lin("fs_decoder_transcript.2", torch.relu(lin("fs_decoder_transcript.0", h)))    # models.py:690-694
        length = lin("fs_decoder_length.2", torch.relu(lin("fs_decoder_length.0", torch.relu(torch.cat((mixed, logits))))))
        logp = torch.log_softmax(logits, dim=0)
        logps.append(logp)
        lengths.append(length.reshape(()))
        word = int(logp.argmax())
        if stop_on_eos and word == eos:                                 # models.py:718-722
            break
        if not teacher_forcing:
            tok = word
    return torch.stack(logps), torch.stack(lengths)


def shead(enc, params: Dict[str, torch.Tensor], tf_input, steps: int, teacher_forcing: bool, stop_on_eos: bool, eos: int,
          dropmask: Optional[torch.Tensor] = None):
    """sequence_generation_forward: temporal encoding [Tz, 128] -> (logp [n, M+1], lengths [n])."""
    memory, h_n, c_n = lstm(enc, params)
    return decoder(memory, h_n, c_n, params, tf_input, steps, teacher_forcing, stop_on_eos, eos, dropmask)
