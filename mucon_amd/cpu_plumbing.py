"""`cfg.system.device = "cpu"` (reference src/core/config.py:16; BASELINE config 1: inside.yaml, device=cpu, one epoch).

PLUMBING ONLY.  The reference's entry script can be pointed at the host (`--set system.device cpu`,
src/train_test_mucon.py:41,51) to check a dataset tree, a config and the checkpoint folders without a GPU.
This module is what the mirror classes run for CPU tensors so that the same command works here: the
library ops the reference itself calls (`F.conv1d`, `max_pool1d`, `group_norm`, `interpolate`,
`log_softmax`: temporal.py:43-53,128-147, models.py:746-773,567-582) on the modules' own parameters, and a
plain NumPy restatement of the Viterbi recursion (viterbi.py:49-158).

It is NOT a fallback of the HIP path and carries no parity or performance claim:
  * `mucon_amd.ops.*` (the C-ABI wrappers) still raise MuconHipError on CPU tensors, and a missing
    libmucon_hip.so still raises -- nothing here is reached from them;
  * every function below refuses device tensors (`_host_only`), so a CUDA tensor can never be
    computed here (tests/test_cpu_plumbing.py);
  * nothing under oracle/ is imported; bench.py and the GPU tests never touch this module.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


class DevicePlumbingError(RuntimeError):
    """A device tensor reached the host plumbing path: a bug in the caller's dispatch."""


def _host_only(*ts):
    for t in ts:
        if isinstance(t, Tensor) and t.is_cuda:
            raise DevicePlumbingError("cpu_plumbing got a device tensor: device tensors run in the HIP kernels (mucon_amd.ops)")


def _act(x: Tensor, leaky: bool) -> Tensor:
    return F.leaky_relu(x) if leaky else F.relu(x)


def wavenet_forward(block, tape: Tensor, gn_weight: Tensor, gn_bias: Tensor, spec) -> Tensor:
    """tape [B, T, Cin] -> [B, Tz, H]: WaveNetBlock.forward + the GroupNorm / ReLU / Dropout wrapper
    (reference temporal.py:128-147, models.py:759-768) as library ops on `block`'s nn.Conv1d parameters."""
    _host_only(tape, gn_weight, gn_bias)
    training = block.training
    x = _act(block.first_conv(tape.permute(0, 2, 1)), False)
    for i, layer in enumerate(block.layers):
        h = _act(layer.dilated_conv(x), layer.leaky)
        x = x + F.dropout(layer.conv_1x1(h), p=float(block.dropout_rate), training=training)
        if block.pooling and i in block.pooling_layers:
            if block.pooling_type == "max":
                x = F.max_pool1d(x, kernel_size=2)
            else:
                x = F.avg_pool1d(x, kernel_size=2) * 2
    x = block.last_conv(F.relu(x))
    if spec.last_gn:
        x = F.group_norm(x, spec.last_gn_num_groups, gn_weight, gn_bias, spec.gn_eps)
    if spec.last_relu:
        x = F.relu(x)
    if spec.last_dropout:
        x = F.dropout(x, p=float(spec.last_dropout_rate), training=training)
    return x.permute(0, 2, 1).contiguous()


def head_forward(enc: Tensor, weight: Tensor, bias: Tensor, Tf: int) -> Tuple[Tensor, Tensor]:
    """enc [B, Tz, H] -> (logits [B, Tf, C], log-probs [B, Tf, C]): nearest interpolate, 1x1 conv, log_softmax
    (reference models.py:567-582, :368)."""
    _host_only(enc, weight, bias)
    up = F.interpolate(enc.permute(0, 2, 1), size=int(Tf))
    logits = F.conv1d(up, weight.reshape(weight.shape[0], weight.shape[1], 1), bias).permute(0, 2, 1)
    return logits, F.log_softmax(logits, dim=2)


def viterbi_decode(lp: np.ndarray, transcript: np.ndarray, P: np.ndarray, fs: int,
                   force: Optional[Tuple[int, int]]) -> Tuple[np.float64, List[int], List[int], bool]:
    """The recursion of reference viterbi.py:49-158 for a SingleTranscriptGrammar and a length table P[J x N]
    (P[j, n] = length score of (j+1)*fs frames of label a_n), as array operations per column.

    -> (score, labels[T], segment lengths, alive) -- `alive` False when no hypothesis survives (the reference's
    AttributeError).  `force` = (n, j): finalize on that hypothesis with score -inf (the degenerate outcomes the caller
    resolved on the host, core/viterbi/viterbi.py)."""
    lp = np.ascontiguousarray(lp, dtype=np.float32)
    T, _ = lp.shape
    tr = np.asarray(transcript, dtype=np.int64)
    N, J = len(tr), P.shape[0]
    K = T // fs
    cs = np.cumsum(lp[:, tr], axis=0, dtype=np.float32)              # sequential f32 sums (viterbi.py:51)
    ends = (np.arange(K) + 1) * fs - 1
    f = cs[ends].copy()
    f[1:] -= cs[ends[1:] - fs]                                        # frame_score (viterbi.py:68-72)
    S = np.full((N, J), -np.inf)
    S[0, 0] = np.float32(0.0) + f[0, 0]
    back = np.zeros((K, N), dtype=np.int64)                           # length index of the predecessor each entry came from
    alive_cnt = np.zeros((N, J), dtype=bool)
    alive_cnt[0, 0] = True
    for k in range(1, K):
        grown = S + f[k][:, None]
        grown[0] = grown[0].astype(np.float32)                        # transcript state 0 is a float32 + float32 chain
        if N > 1:
            cand = (grown[:-1] + P.T[:-1]) + 0.0                      # advance with the OLD label's frame score (viterbi.py:113)
            cand = np.where(alive_cnt[:-1], cand, -np.inf)
            with np.errstate(invalid="ignore"):
                best = np.nanmax(np.where(np.isnan(cand), -np.inf, cand), axis=1)
            # `<=` in HypDict.update: the last candidate among equals wins = the largest length index
            arg = J - 1 - np.argmax(((cand == best[:, None]) & alive_cnt[:-1])[:, ::-1], axis=1)
            any_alive = alive_cnt[:-1].any(axis=1)
        nS = np.full((N, J), -np.inf)
        nA = np.zeros((N, J), dtype=bool)
        nS[:, 1:], nA[:, 1:] = grown[:, :-1], alive_cnt[:, :-1]
        if N > 1:
            nS[1:, 0] = np.where(any_alive, best, -np.inf)
            nA[1:, 0] = any_alive
            back[k, 1:] = arg
        S, alive_cnt = nS, nA
    if force is not None:
        n, j, score = int(force[0]), int(force[1]), np.float64(-np.inf)
    else:
        last = alive_cnt[N - 1]
        if not last.any():
            return np.float64(-np.inf), [], [], False
        fin = np.where(last, (S[N - 1] + P[:, N - 1]) + 0.0, -np.inf)
        fin = np.where(np.isnan(fin), -np.inf, fin)
        score = np.float64(fin.max())
        n, j = N - 1, int(J - 1 - np.argmax(((fin == score) & last)[::-1]))   # `>=`: the last maximum in iteration order
    seg_len = [0] * (n + 1)
    k = K - 1
    while n >= 0:
        seg_len[n] = (j + 1) * fs
        k -= j + 1
        if n == 0:
            break
        j = int(back[k + 1, n])
        n -= 1
    n_seg = len(seg_len)
    seg_len[-1] += T - K * fs                                         # leftover frames join the last segment (viterbi.py:154-157)
    labels: List[int] = []
    rest = T - K * fs
    labels += [int(tr[n_seg - 1])] * rest                             # ... and are emitted first, with its label
    for s in range(n_seg):
        labels += [int(tr[s])] * (seg_len[s] - (rest if s == n_seg - 1 else 0))
    return score, labels, seg_len, True
