"""Differentiable segment masks for the MuCon loss (reference src/mucon/masks.py:8-74).
PyTorch-ROCm ops (tiny tensors: N segments x T frames); out of the hand-written hot path
(SURVEY.md 2, row 9).  Same construction as the reference: a 100-sample template per segment is
resampled onto the T frames through an affine grid (scale = T / length, shift = segment centre).

`align_corners` is explicit here.  The reference calls affine_grid / grid_sample without it (masks.py:72-73), so what it
computes depends on the PyTorch underneath: the 1.1 its Dockerfile pins behaves as align_corners=True (the upstream training
recipe; the default of cfg.model.loss.mucon.align_corners), PyTorch >= 1.3 defaults to False.  Goldens exist for both."""
import torch
from torch import Tensor
from torch.nn.functional import affine_grid, grid_sample, softmax

TEMPLATE_WIDTH = 100


def project_lengths_softmax(T: int, L: Tensor) -> Tensor:
    """Raw length logits [M] -> absolute lengths summing to T (reference masks.py:8-15)."""
    return T * softmax(L, dim=0)


def _template(kind: str, n: int, like: Tensor) -> Tensor:
    if kind == "box":
        return like.new_ones((n, 1, TEMPLATE_WIDTH))
    if kind == "gaussian":
        from scipy.signal.windows import gaussian
        t = torch.tensor(gaussian(M=TEMPLATE_WIDTH, std=TEMPLATE_WIDTH / 5)).float()
    elif kind == "trapezoid":
        w1, lo = TEMPLATE_WIDTH / 2, 0.5
        q = int(w1 / 2)
        t = torch.ones(TEMPLATE_WIDTH)
        t[:q] = torch.arange(start=lo, end=1, step=(1 - lo) / (w1 / 2))
        t[-q:] = torch.arange(start=1, end=lo, step=(lo - 1) / (w1 / 2))
    else:
        raise NameError(f"Invalid template name ({kind})")
    return t.repeat((n, 1)).view(n, 1, -1).float().to(like.device)


def create_masks(T: int, L: Tensor, overlap: float = 0.0, template: str = "box", align_corners: bool = True) -> Tensor:
    """[M] absolute lengths -> [M x T] soft masks.  NB: like the reference (masks.py:58-61) this
    rescales L in place by (1 + 2*overlap)."""
    n = L.size(0)
    tmpl = _template(template, n, L)
    start = torch.cumsum(L, 0) - L
    L *= 1.0 + 2 * overlap
    start = start - L * (overlap / 2)
    scale = T / L                                    # normalised size
    shift = (start + L / 2 - T / 2) / (-(L / 2))     # normalised location
    theta = L.new_zeros((n, 2, 3))
    theta[:, 0, 0] = scale
    theta[:, 0, 2] = shift
    theta[:, 1, 1] = scale
    grid = affine_grid(theta.float(), torch.Size((n, 1, 1, T)), align_corners=align_corners)
    return grid_sample(tmpl.view(n, 1, 1, TEMPLATE_WIDTH), grid, align_corners=align_corners).view(n, T)
