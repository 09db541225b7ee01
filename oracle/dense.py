"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the dense half of the hot path, written with plain torch CPU tensor ops in
float64 or float32 (this is the "torch fp32 reference" the floating-point kernels are compared
against; tolerance is stated in the tests):

  encoder      reference src/core/modules/temporal.py:128-147 (WaveNetBlock.forward),
               :43-53 (WaveNetLayer.forward)
  wrapper      reference src/mucon/models.py:746-773 (temporal_modeling_forward: GroupNorm, ReLU,
               Dropout -- dropout is identity here: the oracle is the eval()-mode function)
  y-head       reference src/mucon/models.py:567-582 (nearest interpolate + 1x1 conv) and the
               log-softmax of :367-368 / :403-405

Parameters are a dict keyed by the reference's state_dict names (ft.first_conv.weight, ft.l_3.
dilated_conv.weight, ft.l_3.conv_1x1.bias, ft.last_conv.*, ft_last_gn.*, conv_classifier.*), in
the reference's layouts ([out, in, k] for convs).  Activations here are time-major [B, T, H] --
the layout the HIP kernels use -- instead of the reference's [B, H, T].

Parity pin: tests/golden/dense_*.npz (outputs of the reference MuCon modules on seeded inputs,
made by tools/make_golden_dense.py); checked in tests/test_oracle_dense.py.
"""
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch


@dataclass
class EncoderConfig:
    """The cfg.model.ft.* keys the path depends on (reference configs/mucon/default.py:81-96)."""
    in_dim: int = 2048
    hidden: int = 128
    num_classes: int = 48
    stages: List[int] = field(default_factory=lambda: [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024])
    pooling: bool = True
    pooling_type: str = "max"
    pooling_layers: List[int] = field(default_factory=lambda: [1, 2, 4, 8])
    leaky_relu: bool = False
    last_gn: bool = True
    last_gn_num_groups: int = 32
    last_relu: bool = True
    gn_eps: float = 1e-5

    def out_length(self, T: int) -> int:
        for i in range(len(self.stages)):
            if self.pooling and i in self.pooling_layers:
                T = T // 2
        return T


def param_shapes(cfg: EncoderConfig) -> Dict[str, tuple]:
    H, D, C = cfg.hidden, cfg.in_dim, cfg.num_classes
    s = {"ft.first_conv.weight": (H, D, 1), "ft.first_conv.bias": (H,)}
    for i in range(len(cfg.stages)):
        s[f"ft.l_{i}.dilated_conv.weight"] = (H, H, 3)
        s[f"ft.l_{i}.dilated_conv.bias"] = (H,)
        s[f"ft.l_{i}.conv_1x1.weight"] = (H, H, 1)
        s[f"ft.l_{i}.conv_1x1.bias"] = (H,)
    s["ft.last_conv.weight"] = (H, H, 1)
    s["ft.last_conv.bias"] = (H,)
    s["ft_last_gn.weight"] = (H,)
    s["ft_last_gn.bias"] = (H,)
    s["conv_classifier.weight"] = (C, H, 1)
    s["conv_classifier.bias"] = (C,)
    return s


def seeded_params(cfg: EncoderConfig, seed: int) -> Dict[str, np.ndarray]:
    """Deterministic, platform-independent parameters: uniform[-1,1) * 2^-k with 2^-k ~ 1/sqrt(fan_in)
    (GroupNorm weight is 1 + 0.25*u so it stays positive-ish).  Same recipe in make_golden_dense.py."""
    from mucon_amd import synth

    out = {}
    for i, (name, shape) in enumerate(param_shapes(cfg).items()):
        u = synth.uniform_pm1(seed * 1000 + i, shape)
        if name.endswith("weight") and len(shape) == 3:
            fan_in = shape[1] * shape[2]
            scale = np.float32(2.0 ** -int(round(np.log2(np.sqrt(fan_in)))))
            out[name] = (u * scale).astype(np.float32)
        elif name == "ft_last_gn.weight":
            out[name] = (np.float32(1.0) + np.float32(0.25) * u).astype(np.float32)
        else:
            out[name] = (u * np.float32(0.125)).astype(np.float32)
    return out


def _act(x, leaky, mask=None):
    """relu / leaky_relu(0.01); with `mask` (bool, True = positive branch) the branch is forced instead
    of read off x -- used to compare gradients on the activation pattern a kernel actually took."""
    if mask is None:
        return torch.nn.functional.leaky_relu(x) if leaky else torch.relu(x)  # default slope 0.01
    m = mask.to(x.dtype)
    return x * (m + (0.01 if leaky else 0.0) * (1 - m))


def _conv_time_major(x, w, b, dilation=1):
    """x [B,T,Cin], w [Cout,Cin,k] (k in {1,3}), zero padding = dilation for k=3 (temporal.py:23-29)."""
    k = w.shape[2]
    if k == 1:
        return x @ w[:, :, 0].T + b
    B, T, _ = x.shape
    y = x @ w[:, :, 1].T + b
    d = dilation
    if d < T:
        y[:, d:, :] += x[:, : T - d, :] @ w[:, :, 0].T   # tap 0 reads x[t-d]
        y[:, : T - d, :] += x[:, d:, :] @ w[:, :, 2].T   # tap 2 reads x[t+d]
    return y


def encoder_forward(tape, params: Dict[str, torch.Tensor], cfg: EncoderConfig, return_intermediates=False,
                    drop=None, force=None):
    """tape [B,T,D] -> enc [B,Tz,H].  Restates temporal.py:128-147 + models.py:759-764 (eval mode).

    `drop` (optional, for checking the training-mode kernels): {layer index: multiplier [B,T_l,H]}
    applied where WaveNetLayer.drop sits (temporal.py:51) and {"last": multiplier [B,Tz,H]} for
    ft_last_dropout (models.py:767-768); a multiplier is keep_mask / (1 - p)."""
    p = params
    drop = drop or {}
    force = force or {}  # {"first" | ("dil", i) | "last_in" | "final": bool mask, ("pool", i): bool take-second}
    inter = {"masks": {}}
    margin = [float("inf")]  # distance of the closest ReLU input / max-pool pair to its kink

    def note(t):
        if return_intermediates:
            margin[0] = min(margin[0], float(t.detach().abs().min()))
        return t

    pre = note(_conv_time_major(tape, p["ft.first_conv.weight"], p["ft.first_conv.bias"]))
    inter["masks"]["first"] = (pre.detach(), pre.detach() > 0)
    x = _act(pre, cfg.leaky_relu, force.get("first"))
    inter["x0"] = x
    for i, d in enumerate(cfg.stages):
        pre = note(_conv_time_major(x, p[f"ft.l_{i}.dilated_conv.weight"], p[f"ft.l_{i}.dilated_conv.bias"], d))
        inter["masks"][("dil", i)] = (pre.detach(), pre.detach() > 0)
        h = _act(pre, cfg.leaky_relu, force.get(("dil", i)))
        y = _conv_time_major(h, p[f"ft.l_{i}.conv_1x1.weight"], p[f"ft.l_{i}.conv_1x1.bias"])
        if i in drop:
            y = y * drop[i]
        y = y + x
        if cfg.pooling and i in cfg.pooling_layers:
            Tl = y.shape[1] // 2
            a, b = y[:, 0:2 * Tl:2, :], y[:, 1:2 * Tl:2, :]
            if cfg.pooling_type == "max":
                note(a - b)
                inter["masks"][("pool", i)] = ((b - a).detach(), (b - a).detach() > 0)
                if ("pool", i) in force:
                    y = torch.where(force[("pool", i)], b, a)
                else:
                    y = torch.maximum(a, b)
            else:
                y = (a + b) / 2 * 2  # avg_pool1d * 2
        x = y
        inter[f"x{i + 1}"] = x
    inter["masks"]["last_in"] = (x.detach(), x.detach() > 0)
    z = _conv_time_major(_act(note(x), cfg.leaky_relu, force.get("last_in")), p["ft.last_conv.weight"],
                         p["ft.last_conv.bias"])
    inter["z"] = z
    if cfg.last_gn:
        B, Tz, H = z.shape
        G = cfg.last_gn_num_groups
        zg = z.reshape(B, Tz, G, H // G)
        mean = zg.mean(dim=(1, 3), keepdim=True)
        var = zg.var(dim=(1, 3), unbiased=False, keepdim=True)
        z = ((zg - mean) / torch.sqrt(var + cfg.gn_eps)).reshape(B, Tz, H) * p["ft_last_gn.weight"] + p["ft_last_gn.bias"]
    if cfg.last_relu:
        inter["masks"]["final"] = (z.detach(), z.detach() > 0)
        z = _act(note(z), False, force.get("final"))
    inter["kink_margin"] = margin[0]
    if "last" in drop:
        z = z * drop["last"]
    return (z, inter) if return_intermediates else z


def nearest_index(Tz: int, Tf: int) -> np.ndarray:
    """Source index of F.interpolate(mode='nearest') (models.py:574): min(floor(i * (Tz/Tf)), Tz-1),
    the scale and the product taken in float32 as torch does."""
    scale = np.float32(Tz) / np.float32(Tf)
    idx = np.floor(np.arange(Tf, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(idx, Tz - 1)


def head_forward(enc, params, cfg: EncoderConfig, Tf: int):
    """enc [B,Tz,H] -> (logits [B,Tf,C], logp [B,Tf,C]).  models.py:574-580 then log_softmax (:368)."""
    idx = torch.from_numpy(nearest_index(enc.shape[1], Tf))
    up = enc[:, idx, :]
    logits = up @ params["conv_classifier.weight"][:, :, 0].T + params["conv_classifier.bias"]
    return logits, torch.log_softmax(logits, dim=-1)


def to_torch(params_np, dtype=torch.float64, requires_grad=False):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in params_np.items()}


def hot_path(tape_np, params_np, cfg: EncoderConfig, dtype=torch.float64):
    """Convenience: numpy in, numpy out (enc, logits, logp)."""
    p = to_torch(params_np, dtype)
    tape = torch.tensor(tape_np, dtype=dtype)
    enc = encoder_forward(tape, p, cfg)
    logits, logp = head_forward(enc, p, cfg, tape.shape[1])
    return enc.numpy(), logits.numpy(), logp.numpy()


def hot_path_grads(tape_np, params_np, cfg: EncoderConfig, w_logp, w_enc, dtype=torch.float64):
    """Gradients of L = sum(w_logp * logp) + sum(w_enc * enc) w.r.t. every parameter (autograd on the
    restatement; the restatement itself is pinned by the forward goldens and by the reference's
    gradients stored in tests/golden/dense_grads.npz)."""
    p = to_torch(params_np, dtype, requires_grad=True)
    tape = torch.tensor(tape_np, dtype=dtype)
    enc = encoder_forward(tape, p, cfg)
    _, logp = head_forward(enc, p, cfg, tape.shape[1])
    L = (torch.tensor(w_logp, dtype=dtype) * logp).sum() + (torch.tensor(w_enc, dtype=dtype) * enc).sum()
    L.backward()
    return {k: v.grad.numpy() for k, v in p.items()}, float(L.detach())


# ---------------------------------------------------------------------------------------- CPU baseline (bench.py only)
def module_graph(cfg: EncoderConfig, params_np=None):
    """The dense hot path as the torch MODULE graph the reference runs: nn.Conv1d on [B, C, T] tensors,
    F.max_pool1d / avg_pool1d, nn.GroupNorm, F.interpolate(mode='nearest'), F.log_softmax -- the same calls, in the same order,
    as WaveNetBlock.forward (temporal.py:128-147), WaveNetLayer.forward (:43-53), temporal_modeling_forward
    (models.py:746-773: permute, encoder, GroupNorm, ReLU) and frame_classifier_forward + predict's log-softmax (models.py:567-582,
    :368).  Dropout is left out (eval-mode function, as the rest of this file).  bench.py's cpu_baseline leg times it."""
    import torch.nn as nn
    import torch.nn.functional as F

    class Layer(nn.Module):
        def __init__(self, H, d):
            super().__init__()
            self.dilated_conv = nn.Conv1d(H, H, 3, dilation=d, padding=d)
            self.conv_1x1 = nn.Conv1d(H, H, 1)

        def forward(self, x):
            y = self.dilated_conv(x)
            y = F.leaky_relu(y) if cfg.leaky_relu else F.relu(y)
            y = self.conv_1x1(y)
            return y + x

    class HotPath(nn.Module):
        def __init__(self):
            super().__init__()
            H = cfg.hidden
            self.first_conv = nn.Conv1d(cfg.in_dim, H, 1)
            self.layers = nn.ModuleList([Layer(H, d) for d in cfg.stages])
            self.last_conv = nn.Conv1d(H, H, 1)
            self.gn = nn.GroupNorm(cfg.last_gn_num_groups, H, eps=cfg.gn_eps) if cfg.last_gn else None
            self.classifier = nn.Conv1d(H, cfg.num_classes, 1)

        def forward(self, feats):                      # [B, T, D]
            act = F.leaky_relu if cfg.leaky_relu else F.relu
            x = feats.permute(0, 2, 1)                 # models.py:753
            x = act(self.first_conv(x))
            for i, l in enumerate(self.layers):
                x = l(x)
                if cfg.pooling and i in cfg.pooling_layers:
                    x = F.max_pool1d(x, kernel_size=2) if cfg.pooling_type == "max" else F.avg_pool1d(x, kernel_size=2) * 2
            x = self.last_conv(act(x))
            if self.gn is not None:
                x = self.gn(x)
            if cfg.last_relu:
                x = F.relu(x)
            up = F.interpolate(x, size=feats.shape[1], mode="nearest")    # models.py:574
            return F.log_softmax(self.classifier(up), dim=1)              # [B, C, T]

    m = HotPath()
    if params_np is not None:
        with torch.no_grad():
            sd = {"first_conv": "ft.first_conv", "last_conv": "ft.last_conv", "gn": "ft_last_gn", "classifier": "conv_classifier"}
            for ours, theirs in sd.items():
                mod = getattr(m, ours)
                if mod is not None:
                    mod.weight.copy_(torch.from_numpy(params_np[theirs + ".weight"]))
                    mod.bias.copy_(torch.from_numpy(params_np[theirs + ".bias"]))
            for i, l in enumerate(m.layers):
                for part in ("dilated_conv", "conv_1x1"):
                    getattr(l, part).weight.copy_(torch.from_numpy(params_np[f"ft.l_{i}.{part}.weight"]))
                    getattr(l, part).bias.copy_(torch.from_numpy(params_np[f"ft.l_{i}.{part}.bias"]))
    return m
