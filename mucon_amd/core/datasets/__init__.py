"""Datasets with the reference's on-disk layout and Batch type (src/core/datasets/general_dataset.py:17-33,
138-167; src/core/datasets/breakfast.py:19-55):

    <root>/breakfast_i3d/{features,labels,transcripts}/<NAME>.npy, split{1-4}.{train,test}, mapping.txt

plus a generator of a Breakfast-shaped synthetic tree (the real dataset is not redistributable and
is absent from the build and GPU boxes).  Host-side I/O only -- out of the hot path."""
import dataclasses
import os
from dataclasses import dataclass
from pathlib import Path
from typing import List

import numpy as np
import torch
from torch import Tensor

POSSIBLE_SPLITS = [1, 2, 3, 4]
MAX_TRANSCRIPT_LENGTH = 30
FEAT_DIM_MAPPING = {"i3d": 2048}


@dataclass(repr=False)
class Batch:
    """One video (the reference is batch-size-1 by construction): T frames, D feature dims, N actions."""
    feats: Tensor                 # [1 x T x D] float
    gt_label: Tensor              # [T] long
    transcript: Tensor            # [N] long
    transcript_tf_input: Tensor   # [N + 1] long: SOS + transcript
    transcript_tf_target: Tensor  # [N + 1] long: transcript + EOS
    video_name: str

    def to(self, device):
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if isinstance(v, Tensor):
                setattr(self, f.name, v.to(device, non_blocking=True))
        return self


@dataclass(repr=False)
class FullySupervisedBatch(Batch):
    """+ the ground-truth segment lengths (reference general_dataset.py:36-39)."""
    absolute_lengths: Tensor = None   # [N] float


@dataclass(repr=False)
class MixedSupervisionBatch(FullySupervisedBatch):
    fully_supervised: bool = False    # whether full supervision should be used for this video (general_dataset.py:42-43)


def create_tf_input(transcript, sos_i: int) -> np.ndarray:
    return np.array([sos_i] + list(transcript))


def create_tf_target(transcript, eos_i: int) -> np.ndarray:
    return np.array(list(transcript) + [eos_i])


class GeneralDataset(torch.utils.data.Dataset):
    def __init__(self, cfg, root: Path, relative_path_to_list, relative_path_to_mapping="mapping.txt", feat_dim=-1):
        self.cfg, self.root = cfg, Path(root)
        with open(self.root / relative_path_to_list) as f:
            self.file_names = [x.strip() for x in f if len(x.strip()) > 0]
        self.action_id_to_name, self.action_name_to_id = {}, {}
        with open(self.root / relative_path_to_mapping) as f:
            for line in f:
                if line.strip():
                    i, name = line.strip().split()
                    self.action_id_to_name[int(i)] = name
                    self.action_name_to_id[name] = int(i)
        self.num_actions = len(self.action_id_to_name)
        self.eos_token_id, self.sos_token_id = self.num_actions, self.num_actions + 1
        self.feat_dim = feat_dim
        self.end_class_id, self.mof_eval_ignore_classes, self.background_class_ids = 0, [], [0]
        self.convenient_name, self.split, self.max_transcript_length = None, -1, 100

    def get_num_classes(self) -> int:
        return self.num_actions

    def __len__(self) -> int:
        return len(self.file_names)

    def __getitem__(self, item: int) -> Batch:
        name = self.file_names[item]
        feats = torch.tensor(np.load(str(self.root / "features" / f"{name}.npy"))).float().unsqueeze(0)
        gt = torch.tensor(np.load(str(self.root / "labels" / f"{name}.npy"))).long()
        tr = torch.tensor(np.load(str(self.root / "transcripts" / f"{name}.npy"))).long()
        return Batch(feats=feats, gt_label=gt, transcript=tr,
                     transcript_tf_input=torch.tensor(create_tf_input(tr.tolist(), self.sos_token_id)).long(),
                     transcript_tf_target=torch.tensor(create_tf_target(tr.tolist(), self.eos_token_id)).long(),
                     video_name=name)

    def collate_fn(self, items: List[Batch]) -> Batch:
        assert len(items) == 1  # the reference assumes batch_size = 1 (general_dataset.py:169-173)
        return items[0]


class GeneralFullySupervisedDataset(GeneralDataset):
    """+ <root>/lengths/<NAME>.npy: the segment lengths of the ground truth (reference general_dataset.py:176-208)."""

    def __getitem__(self, item: int) -> FullySupervisedBatch:
        b = super().__getitem__(item)
        lengths = torch.tensor(np.load(str(self.root / "lengths" / f"{self.file_names[item]}.npy")), dtype=torch.float32)
        return FullySupervisedBatch(**{f.name: getattr(b, f.name) for f in dataclasses.fields(Batch)}, absolute_lengths=lengths)


class GeneralMixedSupervisionDataset(GeneralFullySupervisedDataset):
    """A fixed, seeded subset of the videos is marked fully supervised (reference general_dataset.py:211-263): the
    same `random.seed(f"{seed}-{n}")` + `random.shuffle` draw, so the subset is the reference's."""

    def __init__(self, cfg, root, full_supervision_percentage: float, relative_path_to_list="split1.train",
                 relative_path_to_mapping="mapping.txt", feat_dim: int = -1):
        super().__init__(cfg, root, relative_path_to_list, relative_path_to_mapping, feat_dim)
        assert 0.0 < full_supervision_percentage < 100.0
        self.full_supervision_percentage = full_supervision_percentage
        n = len(self.file_names)
        self.number_of_full_supervision_examples = min(n, max(1, int(round(n * full_supervision_percentage / 100.0))))
        flags = [i < self.number_of_full_supervision_examples for i in range(n)]
        import random
        random.seed(f"{self.cfg.system.seed}-{self.number_of_full_supervision_examples}")
        random.shuffle(flags)
        self.is_it_supervised = flags

    def __getitem__(self, item: int) -> MixedSupervisionBatch:
        b = super().__getitem__(item)
        return MixedSupervisionBatch(**{f.name: getattr(b, f.name) for f in dataclasses.fields(FullySupervisedBatch)},
                                     fully_supervised=self.is_it_supervised[item])


def _breakfast(cfg, train: bool, cls, name_fmt: str, **extra):
    split, feat_name = cfg.dataset.split, cfg.dataset.feat_name
    assert split in POSSIBLE_SPLITS
    db_path = Path(cfg.dataset.root) / f"breakfast_{feat_name}"
    set_name = "train" if train else "test"
    db = cls(cfg, db_path, relative_path_to_list=f"split{split}.{set_name}", relative_path_to_mapping=cfg.dataset.mapping_file_name,
             feat_dim=FEAT_DIM_MAPPING[feat_name], **extra)
    db.convenient_name = name_fmt.format(split=split, set_name=set_name)
    db.split, db.max_transcript_length = split, MAX_TRANSCRIPT_LENGTH
    return db


def create_fully_supervised_breakfast_dataset(cfg, train: bool = True) -> GeneralFullySupervisedDataset:
    return _breakfast(cfg, train, GeneralFullySupervisedDataset, "fully_supervised_breakfast_split{split}_{set_name}")


def create_mixed_supervision_breakfast_dataset(cfg, train: bool = True) -> GeneralMixedSupervisionDataset:
    pct = cfg.dataset.mixed.full_supervision_percentage
    return _breakfast(cfg, train, GeneralMixedSupervisionDataset,
                      "mixed_supervision_percentage_" + str(pct) + "_breakfast_split{split}_{set_name}",
                      full_supervision_percentage=pct)


def handel_fully_supervised_dataset(cfg, train: bool) -> GeneralFullySupervisedDataset:
    if cfg.dataset.name == "breakfast":
        return create_fully_supervised_breakfast_dataset(cfg, train)
    raise Exception("Invalid dataset name.")


def handel_mixed_supervision_dataset(cfg, train: bool) -> GeneralMixedSupervisionDataset:
    if cfg.dataset.name == "breakfast":
        return create_mixed_supervision_breakfast_dataset(cfg, train)
    raise Exception("Invalid dataset name.")


def create_breakfast_dataset(cfg, train: bool = True) -> GeneralDataset:
    split, feat_name = cfg.dataset.split, cfg.dataset.feat_name
    assert split in POSSIBLE_SPLITS
    db_path = Path(cfg.dataset.root) / f"breakfast_{feat_name}"
    set_name = "train" if train else "test"
    db = GeneralDataset(cfg, db_path, f"split{split}.{set_name}", cfg.dataset.mapping_file_name,
                        feat_dim=FEAT_DIM_MAPPING[feat_name])
    db.convenient_name = f"breakfast_split{split}_{set_name}"
    db.split, db.max_transcript_length = split, MAX_TRANSCRIPT_LENGTH
    return db


def handel_dataset(cfg, train: bool) -> GeneralDataset:  # (sic) the reference's spelling
    if cfg.dataset.name == "breakfast":
        return create_breakfast_dataset(cfg, train)
    raise Exception(f"Invalid dataset name. ({cfg.dataset.name})")


def make_resident(db: GeneralDataset, device, max_bytes=None, indices=None):
    """The same dataset with its tapes cached in HBM (core/datasets/resident.py)."""
    from .resident import ResidentDataset
    return ResidentDataset(db, device, max_bytes=max_bytes, indices=indices)


def write_synthetic_breakfast(root, n_train=8, n_test=4, num_classes=48, feat_dim=2048, t_range=(130, 1200),
                              n_range=(2, 8), seed=0, splits=(1,)):
    """Write a Breakfast-I3D-shaped tree under <root>/breakfast_i3d with random tapes whose class signal is
    planted in a few feature dims (so training can learn something)."""
    from ... import synth

    base = Path(root) / "breakfast_i3d"
    for d in ("features", "labels", "transcripts", "lengths"):
        os.makedirs(base / d, exist_ok=True)
    with open(base / "mapping.txt", "w") as f:
        for c in range(num_classes):
            f.write(f"{c} action_{c}\n")
    names = {"train": [], "test": []}
    k = 0
    for part, n in (("train", n_train), ("test", n_test)):
        for _ in range(n):
            sd = seed * 100003 + k * 17
            T = int(synth.integers(sd, 1, t_range[0], t_range[1] + 1)[0])
            N = int(synth.integers(sd + 1, 1, n_range[0], n_range[1] + 1)[0])
            tr = synth.transcript(sd + 2, N, num_classes, allow_repeats=False)
            labels = synth.segment_labels(sd + 3, T, tr)
            feats = synth.uniform_pm1(sd + 4, (T, feat_dim)) * np.float32(0.5)
            feats[np.arange(T), labels % feat_dim] += np.float32(2.0)
            name = f"P{k:03d}_synth"
            np.save(base / "features" / f"{name}.npy", feats.astype(np.float32))
            np.save(base / "labels" / f"{name}.npy", labels.astype(np.int64))
            np.save(base / "transcripts" / f"{name}.npy", tr.astype(np.int64))
            cuts = np.flatnonzero(np.diff(labels)) + 1
            np.save(base / "lengths" / f"{name}.npy", np.diff(np.concatenate(([0], cuts, [T]))).astype(np.int64))
            names[part].append(name)
            k += 1
    for s in splits:
        for part in ("train", "test"):
            with open(base / f"split{s}.{part}", "w") as f:
                f.write("\n".join(names[part]) + "\n")
    return str(base)
