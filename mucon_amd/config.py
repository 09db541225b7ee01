"""Configuration tree with the reference's keys (src/configs/mucon/default.py:7-116,
src/core/config.py:5-18) so its YAML files (src/configs/docker/inside.yaml, slow.yaml) and
`--set KEY VALUE` overrides load unchanged.  yacs is not available here; CfgNode below is the
small subset of yacs.config.CfgNode the reference uses (attribute access, clone/defrost/freeze,
merge_from_file, merge_from_list)."""
import copy
import os
from ast import literal_eval

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError(f"attempted to set {k} on a frozen CfgNode")
        self[k] = v

    def clone(self):
        out = CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, CfgNode) else copy.deepcopy(v)
        return out

    def _set_frozen(self, flag):
        self.__dict__["_frozen"] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {path}{k}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{path}{k} is a section")
                self[k]._merge(v, f"{path}{k}.")
            else:
                self[k] = _coerce(v, self[k], f"{path}{k}")

    def merge_from_file(self, filename):
        with open(filename) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, kv):
        assert len(kv) % 2 == 0, "override list must be KEY VALUE pairs"
        for key, val in zip(kv[0::2], kv[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            if isinstance(val, str):
                try:
                    val = literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _coerce(val, node[parts[-1]], key)

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self))


def _coerce(new, old, key):
    if old is None or isinstance(new, type(old)):
        return new
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new)
    if isinstance(old, bool) or isinstance(new, bool):
        raise ValueError(f"type mismatch for {key}: {type(old).__name__} vs {type(new).__name__}")
    if isinstance(old, str):
        return str(new)
    raise ValueError(f"type mismatch for {key}: {type(old).__name__} vs {type(new).__name__}")


_DEFAULTS = {
    "experiment_name": "mucon_default",
    "system": {"device": "cuda", "num_workers": 2, "seed": 1},
    "dataset": {
        "root": os.path.expanduser("~/work/MuCon/datasets"), "name": "breakfast", "feat_name": "i3d",
        "mapping_file_name": "mapping.txt", "split": 1,
        "mixed": {"full_supervision_percentage": 50.0},
    },
    "trainer": {
        "root": os.path.expanduser("~/work/MuCon/root"), "num_epochs": 150,
        "clip_grad_norm": True, "clip_grad_norm_separate": True, "clip_grad_norm_every_param": False,
        "clip_grad_norm_value": 100.0,
        "optimizer": "SGD", "learning_rate": 0.01, "momentum": 0.0, "weight_decay": 0.005,
        "accumulate_grad_every": 1,
        "scheduler": {
            "name": "step",
            "plateau": {"mode": "max", "factor": 0.1, "verbose": True, "patience": 20},
            "step": {"milestones": [70], "gamma": 0.1},
        },
        "save_every": 5, "eval_every": 1,
    },
    "evaluator": {"viterbi": {"multi_length": False}},
    "model": {
        "teacher_forcing": True, "name": "mucon", "first_gru_hidden_size": 128,
        "loss": {
            "mul_mucon": 1.0, "mul_transcript": 1.0, "mul_smoothing": 0.1, "mul_length": 0.1,
            "length_width": 2.0, "transcript_average": False,
            "mucon_weight_background": False, "mucon_weight_background_value": 0.5, "mucon_weight_background_index": 0,
            "transcript_weight_background": False, "transcript_weight_background_value": 0.5,
            "transcript_weight_background_index": 0,
            "fully_supervised": {"mul_classification": 1.0, "mul_supervised_length": 1.0},
            "smoothing": {"log_softmax_before": True, "clamp": True, "clamp_min": 0, "clamp_max": 16},
            # align_corners is this build's one added key: the reference's masks.py calls affine_grid / grid_sample without it,
            # and the PyTorch 1.1 its Dockerfile pins behaves as True (upstream recipe); False = a current torch's default
            "mucon": {"type": "flint", "template": "box", "overlap": 0.0, "align_corners": True},
        },
        "ft": {
            "type": "wavenet", "stages": [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024],
            "pooling": True, "pooling_type": "max", "pooling_layers": [1, 2, 4, 8],
            "hidden_size": 128, "dropout_rate": 0.25, "leaky_relu": False,
            "last_gn": True, "last_gn_num_groups": 32, "last_relu": True,
            "last_dropout": True, "last_dropout_rate": 0.25,
        },
        "fs": {
            "jit_no_reverse": True,
            "encoder": {"hidden_size": 128, "bidirectional": True, "dropout": 0.0},
            "decoder": {"embedding_dim": 128, "embedding_dropout": 0.25, "hidden_size": 128, "num_layers": 1,
                        "dropout": 0.0},
        },
        "fc": {},
    },
}


def get_cfg_defaults() -> CfgNode:
    """Same tree as reference src/configs/mucon/default.py:119-120."""
    return CfgNode(copy.deepcopy(_DEFAULTS))


def update_config(default_config: CfgNode, file_configs=(), set_configs=()) -> CfgNode:
    """fandak.utils.config.update_config as the reference calls it (src/train_test_mucon.py:18-22):
    YAML overlays in order, then KEY VALUE overrides, then freeze."""
    cfg = default_config.clone()
    for f in file_configs or ():
        cfg.merge_from_file(f)
    flat = []
    for item in set_configs or ():
        flat.extend(item if isinstance(item, (list, tuple)) else [item])
    cfg.merge_from_list(flat)
    cfg.freeze()
    return cfg
