"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of MuCon.loss in explicit float64 torch arithmetic; gradients by autograd over
these formulas.  The mask construction is written out (no affine_grid / grid_sample):

  lengths -> masks   reference src/mucon/masks.py:8-15 (project_lengths_softmax), :44-74 (create_masks: in-place
                     rescale of the lengths, affine grid, bilinear sampling of the 100-point template with zero
                     padding -- in both conventions of affine_grid / grid_sample: align_corners=True is what the
                     PyTorch 1.1 the reference pins computes, False what a torch >= 1.3 does with the same code),
                     :19-41 (templates)
  mucon loss         reference src/mucon/models.py:414-515 ("flint": mask-averaged logits / length -> log-softmax ->
                     nll; "arithmetic": mask-weighted per-frame cross-entropy / T)
  smoothing loss     reference src/mucon/models.py:398-412
  length loss        reference src/mucon/models.py:517-531
  transcript loss    reference src/mucon/models.py:533-565
  main               reference src/mucon/models.py:376-396

Parity pin: tests/golden/loss_cases.npz (+ .json: the configuration of every case) -- values and gradients of the
reference's own MuCon.loss, made by tools/make_golden_loss.py; checked in tests/test_oracle_losses.py.
"""
from dataclasses import dataclass

import numpy as np
import torch

TEMPLATE_WIDTH = 100


@dataclass
class LossConfig:
    """cfg.model.loss.* (reference src/configs/mucon/default.py:48-78)."""
    mul_mucon: float = 1.0
    mul_transcript: float = 1.0
    mul_smoothing: float = 0.1
    mul_length: float = 0.1
    length_width: float = 2.0
    transcript_average: bool = False
    mucon_weight_background: bool = False
    mucon_weight_background_value: float = 0.5
    mucon_weight_background_index: int = 0
    transcript_weight_background: bool = False
    transcript_weight_background_value: float = 0.5
    transcript_weight_background_index: int = 0
    smoothing_log_softmax_before: bool = True
    smoothing_clamp: bool = True
    smoothing_clamp_min: float = 0.0
    smoothing_clamp_max: float = 16.0
    mucon_type: str = "flint"
    mucon_template: str = "box"
    mucon_overlap: float = 0.0
    mucon_align_corners: bool = True   # affine_grid / grid_sample convention (True: the PyTorch 1.1 the reference pins; False: torch >= 1.3)

    @staticmethod
    def from_overrides(pairs):
        """["model.loss.mucon.type", "arithmetic", ...] -> LossConfig."""
        c = LossConfig()
        for key, val in zip(pairs[::2], pairs[1::2]):
            name = key.replace("model.loss.", "").replace(".", "_")
            assert hasattr(c, name), key
            setattr(c, name, val)
        return c


def template(kind: str) -> torch.Tensor:
    """float32 values, as the reference builds them (masks.py:19-41), returned as float64."""
    n = np.arange(TEMPLATE_WIDTH, dtype=np.float64)
    if kind == "box":
        t = np.ones(TEMPLATE_WIDTH)
    elif kind == "gaussian":     # scipy.signal.windows.gaussian(M=100, std=20)
        t = np.exp(-0.5 * ((n - (TEMPLATE_WIDTH - 1) / 2.0) / (TEMPLATE_WIDTH / 5)) ** 2)
    elif kind == "trapezoid":    # linear ramps 0.5 -> 1 over the first / last quarter
        t = np.ones(TEMPLATE_WIDTH)
        ramp = torch.arange(start=0.5, end=1, step=0.5 / 25).numpy().astype(np.float64)
        t[:25] = ramp
        t[-25:] = torch.arange(start=1, end=0.5, step=-0.5 / 25).numpy().astype(np.float64)
    else:
        raise NameError(kind)
    return torch.from_numpy(t.astype(np.float32).astype(np.float64))


def masks_and_lengths(lengths: torch.Tensor, T: int, overlap: float, kind: str, align_corners: bool = True):
    """raw length logits [N] -> (masks [N, T], rescaled absolute lengths [N])."""
    A = T * torch.softmax(lengths, dim=0)
    start = torch.cumsum(A, 0) - A
    L = A * (1.0 + 2 * overlap)
    start = start - L * (overlap / 2)
    scale = T / L
    shift = (start + L / 2 - T / 2) / (-(L / 2))
    t = torch.arange(T, dtype=lengths.dtype)
    if align_corners:      # base grid linspace(-1, 1, T); pixel centres of the template's ends at -1 and +1
        xb = 2 * t / (T - 1) - 1 if T > 1 else torch.zeros_like(t)
    else:                  # half-pixel convention
        xb = (2 * t + 1) / T - 1
    x = scale[:, None] * xb[None, :] + shift[:, None]
    ix = (x + 1) / 2 * (TEMPLATE_WIDTH - 1) if align_corners else ((x + 1) * TEMPLATE_WIDTH - 1) / 2
    i0 = torch.floor(ix.detach())
    fx = ix - i0
    tm = template(kind).to(lengths.dtype)
    padded = torch.cat([torch.zeros(2, dtype=tm.dtype), tm, torch.zeros(2, dtype=tm.dtype)])   # index i -> padded[i + 2]
    j0 = torch.clamp(i0, -2, TEMPLATE_WIDTH + 1).long() + 2
    j1 = torch.clamp(i0 + 1, -2, TEMPLATE_WIDTH + 1).long() + 2
    return padded[j0] * (1 - fx) + padded[j1] * fx, L


def class_weight(n, enabled, index, value, dtype):
    if not enabled:
        return None
    w = torch.ones(n, dtype=dtype)
    w[index] = value
    return w


def nll(logp, target, weight, mean: bool):
    w = weight[target] if weight is not None else torch.ones(target.shape[0], dtype=logp.dtype)
    picked = -w * logp[torch.arange(target.shape[0]), target]
    return picked.sum() / w.sum() if mean else picked.sum()


def loss(cfg: LossConfig, segmentation, transcript_logp, lengths, mucon_target, transcript_target):
    """-> (main, transcript, length, mucon, smoothing).  segmentation [T, M] logits, transcript_logp [N+1, M+1],
    lengths [N]; mucon_target [N], transcript_target [N+1] (long)."""
    T, M = segmentation.shape
    dt = segmentation.dtype
    # transcript
    tw = class_weight(transcript_logp.shape[1], cfg.transcript_weight_background, cfg.transcript_weight_background_index,
                      cfg.transcript_weight_background_value, dt)
    t_loss = nll(transcript_logp, transcript_target, tw, cfg.transcript_average)
    # length
    w = cfg.length_width
    l_loss = torch.relu(lengths - w).sum() + torch.relu(-w - lengths).sum()
    # mucon
    masks, L = masks_and_lengths(lengths, T, cfg.mucon_overlap, cfg.mucon_template, cfg.mucon_align_corners)
    mw = class_weight(M, cfg.mucon_weight_background, cfg.mucon_weight_background_index, cfg.mucon_weight_background_value, dt)
    if cfg.mucon_type == "flint":
        windows = (masks @ segmentation) / L[:, None]
        m_loss = nll(torch.log_softmax(windows, dim=1), mucon_target, mw, True)
    elif cfg.mucon_type == "arithmetic":
        lsm = torch.log_softmax(segmentation, dim=1)
        wn = mw[mucon_target] if mw is not None else torch.ones(mucon_target.shape[0], dtype=dt)
        ce = -wn[None, :] * lsm[:, mucon_target]          # [T, N]
        m_loss = (ce * masks.t()).sum() / T
    else:
        raise Exception(cfg.mucon_type)
    # smoothing
    x = torch.log_softmax(segmentation, dim=1) if cfg.smoothing_log_softmax_before else segmentation
    s_loss = ((x[1:] - x[:-1].detach()) ** 2).mean()
    if cfg.smoothing_clamp:
        s_loss = torch.clamp(s_loss, min=cfg.smoothing_clamp_min, max=cfg.smoothing_clamp_max)
    main = cfg.mul_transcript * t_loss + cfg.mul_length * l_loss + cfg.mul_mucon * m_loss + cfg.mul_smoothing * s_loss
    return main, t_loss, l_loss, m_loss, s_loss
