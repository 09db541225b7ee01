"""Import harness for the read-only reference tree at /root/reference (THIS container only).

The reference needs three third-party packages that are not installed here and cannot be
installed (no network): fandak 0.1.3.1, yacs, edit_distance (reference requirements.txt:11-13).
None of them contributes arithmetic to the hot path, so this module installs
minimal in-memory stand-ins into sys.modules, patches two API drifts of the newer NumPy/SciPy
(np.float, scipy.signal.gaussian) and puts /root/reference/src on sys.path.

Used ONLY by tools/make_golden.py to emit fixtures under tests/golden/.  Nothing here, and
nothing of the reference, is imported by the product, the tests, bench.py or smoke().
"""
import dataclasses
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


class _CfgNode(dict):
    """dict-backed stand-in for yacs.config.CfgNode (attribute access, clone/defrost/freeze)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = _CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, _CfgNode) else (list(v) if isinstance(v, list) else v)
        return out

    def defrost(self):
        pass

    def freeze(self):
        pass


def install():
    if "fandak" in sys.modules:
        return
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

    import numpy as np
    import scipy.signal
    import scipy.signal.windows
    import torch
    import torch.nn as nn

    if not hasattr(np, "float"):
        np.float = float  # reference isba_code.py:46,90 / mstcn_code.py:30
    if not hasattr(scipy.signal, "gaussian"):
        scipy.signal.gaussian = scipy.signal.windows.gaussian  # reference masks.py:2

    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = _CfgNode
    yacs.config = yacs_config
    sys.modules["yacs"] = yacs
    sys.modules["yacs.config"] = yacs_config

    fandak = types.ModuleType("fandak")

    @dataclasses.dataclass(repr=False)
    class GeneralLoss:
        main: object

    @dataclasses.dataclass(repr=False)
    class GeneralForwardOut:
        pass

    class Model(nn.Module):
        def __init__(self, cfg):
            super().__init__()
            self.cfg = cfg

        def get_params(self, original_lr):
            return [{"params": self.parameters(), "lr": original_lr}]

    class Dataset(torch.utils.data.Dataset):
        def __init__(self, cfg):
            self.cfg = cfg

    class Trainer:
        pass

    class Evaluator:
        def __init__(self, cfg, test_db, model, device):
            self.cfg, self.test_db, self.model, self.device = cfg, test_db, model, device

    fandak.GeneralLoss, fandak.GeneralForwardOut = GeneralLoss, GeneralForwardOut
    fandak.Model, fandak.Dataset, fandak.Trainer, fandak.Evaluator = Model, Dataset, Trainer, Evaluator

    f_core = types.ModuleType("fandak.core")
    f_ds = types.ModuleType("fandak.core.datasets")

    @dataclasses.dataclass(repr=False)
    class GeneralBatch:
        def to(self, device):
            for f in dataclasses.fields(self):
                v = getattr(self, f.name)
                if isinstance(v, torch.Tensor):
                    setattr(self, f.name, v.to(device))

    f_ds.GeneralBatch = GeneralBatch
    f_ev = types.ModuleType("fandak.core.evaluators")

    @dataclasses.dataclass(repr=False)
    class GeneralEvaluatorResult:
        pass

    f_ev.GeneralEvaluatorResult = GeneralEvaluatorResult
    f_tr = types.ModuleType("fandak.core.trainers")
    f_tr.Scheduler = object
    f_utils = types.ModuleType("fandak.utils")
    f_utils.common_config = lambda f: f
    f_utorch = types.ModuleType("fandak.utils.torch")
    f_utorch.tensor_to_numpy = lambda t: t.detach().cpu().numpy()
    f_ucfg = types.ModuleType("fandak.utils.config")
    f_ucfg.update_config = lambda cfg, files, sets: cfg
    fandak.core, fandak.utils = f_core, f_utils
    f_core.datasets, f_core.evaluators, f_core.trainers = f_ds, f_ev, f_tr
    f_utils.torch, f_utils.config = f_utorch, f_ucfg
    for name, mod in [
        ("fandak", fandak), ("fandak.core", f_core), ("fandak.core.datasets", f_ds),
        ("fandak.core.evaluators", f_ev), ("fandak.core.trainers", f_tr),
        ("fandak.utils", f_utils), ("fandak.utils.torch", f_utorch), ("fandak.utils.config", f_ucfg),
    ]:
        sys.modules[name] = mod

    ed = types.ModuleType("edit_distance")

    class SequenceMatcher:  # only s_mat_score depends on it; off the hot path
        def __init__(self, a=None, b=None):
            import difflib
            self._m = difflib.SequenceMatcher(None, a, b, autojunk=False)

        def ratio(self):
            return self._m.ratio()

    ed.SequenceMatcher = SequenceMatcher
    sys.modules["edit_distance"] = ed

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    set_grid_convention(True)


def set_grid_convention(align_corners):
    """The reference's create_masks calls affine_grid / grid_sample WITHOUT align_corners (masks.py:70-71), so what it
    computes depends on the PyTorch underneath: the 1.1 its Dockerfile pins (docker/pytorch1.1/Dockerfile:25) has no such
    argument and behaves as align_corners=True -- base grid linspace(-1, 1, T), ix = (x + 1) / 2 * (W - 1) -- while this
    container's torch 2.10 defaults to False.  True (the harness default) makes the reference compute what it computes in
    its own pinned environment; False leaves today's default; the loss goldens are captured under both."""
    import functools

    import torch.nn.functional as F

    import mucon.masks as m
    m.affine_grid = functools.partial(F.affine_grid, align_corners=bool(align_corners))
    m.grid_sample = functools.partial(F.grid_sample, align_corners=bool(align_corners))
