"""HBM-resident dataset cache.

The reference re-reads three .npy files per video per epoch and ships the tape over PCIe every step
(src/core/datasets/general_dataset.py:138-167: np.load -> torch.tensor -> .float() -> Batch; trainers.py moves it to the
device).  Streaming tapes from the host caps the hot path at 63 GB/s / 8 KB = 7.7 M frames/s per GPU, a third of what
the kernels sustain, and Breakfast-I3D is 27 GB: it fits 10x over in one MI355X's 288 GB.  So the tapes live in HBM:

  * one feature arena [sum T, D] f32, one int64 arena for labels | transcripts | teacher-forcing input | target;
    a video is four offsets, a Batch is five zero-copy views -- __getitem__ launches nothing and copies nothing;
  * filled once: .npy files are memory-mapped (numpy parses the header, no intermediate copy), staged through two
    pinned buffers and copied on a side stream, so the disk read of video k+1 overlaps the H2D copy of video k;
  * a byte budget (default: 80 % of the free HBM): videos that do not fit stay on the host and are served exactly as
    the reference serves them (resident_fraction tells how many made it).

Same surface as GeneralDataset (len / getitem / collate_fn / get_num_classes / the id maps), so trainers and evaluators
take either.  Works on "cpu" too (arena in host memory) -- that is how the CPU tests cover the bookkeeping.
"""
from typing import List, Optional

import numpy as np
import torch

from . import Batch, GeneralDataset, create_tf_input, create_tf_target


class ResidentDataset(torch.utils.data.Dataset):
    def __init__(self, base: GeneralDataset, device, max_bytes: Optional[int] = None, indices: Optional[List[int]] = None):
        self.base, self.device = base, torch.device(device)
        for attr in ("cfg", "root", "file_names", "action_id_to_name", "action_name_to_id", "num_actions", "eos_token_id",
                     "sos_token_id", "feat_dim", "end_class_id", "mof_eval_ignore_classes", "background_class_ids",
                     "convenient_name", "split", "max_transcript_length"):
            setattr(self, attr, getattr(base, attr))
        want = list(range(len(base))) if indices is None else list(indices)
        # pass 1: headers only (mmap), decide what fits
        shapes = {}
        for i in want:
            name = base.file_names[i]
            f = np.load(str(base.root / "features" / f"{name}.npy"), mmap_mode="r")
            n = int(np.load(str(base.root / "transcripts" / f"{name}.npy"), mmap_mode="r").shape[0])
            if f.ndim != 2:
                raise ValueError(f"{name}: features must be [T x D], got {f.shape}")
            shapes[i] = (int(f.shape[0]), int(f.shape[1]), n)
        if max_bytes is None:
            if self.device.type == "cuda":
                free, _total = torch.cuda.mem_get_info(self.device)
                max_bytes = int(free * 0.8)
            else:
                max_bytes = 1 << 62
        self._slot, used, n_frames, n_ints, D = {}, 0, 0, 0, None
        for i in want:
            T, d, n = shapes[i]
            D = d if D is None else D
            if d != D:
                raise ValueError(f"{base.file_names[i]}: feature dim {d} != {D}")
            need = T * d * 4 + (T + 3 * n + 2) * 8
            if used + need > max_bytes:
                continue
            self._slot[i] = (n_frames, T, n_ints, n)
            used, n_frames, n_ints = used + need, n_frames + T, n_ints + T + 3 * n + 2
        self.resident_bytes = used
        self.resident_fraction = len(self._slot) / max(len(want), 1)
        self._feats = torch.empty((n_frames, D or 0), dtype=torch.float32, device=self.device)
        self._ints = torch.empty((n_ints,), dtype=torch.int64, device=self.device)
        self._fill()

    # ------------------------------------------------------------------ fill
    def _fill(self):
        cuda = self.device.type == "cuda"
        max_T = max((s[1] for s in self._slot.values()), default=0)
        D = self._feats.shape[1]
        stage = [torch.empty((max_T, D), dtype=torch.float32, pin_memory=cuda) for _ in range(2 if cuda else 1)]
        events = [None, None]
        side = torch.cuda.Stream(self.device) if cuda else None
        ints_host = torch.empty_like(self._ints, device="cpu", pin_memory=cuda)
        for k, (i, (f0, T, i0, n)) in enumerate(self._slot.items()):
            name = self.base.file_names[i]
            buf = stage[k % len(stage)]
            if cuda and events[k % 2] is not None:
                events[k % 2].synchronize()                       # the copy that last used this staging buffer
            src = np.load(str(self.base.root / "features" / f"{name}.npy"), mmap_mode="r")
            np.copyto(buf[:T].numpy(), src, casting="same_kind")  # disk -> pinned (the .float() of the reference)
            if cuda:
                with torch.cuda.stream(side):
                    self._feats[f0:f0 + T].copy_(buf[:T], non_blocking=True)
                    events[k % 2] = torch.cuda.Event()
                    events[k % 2].record(side)
            else:
                self._feats[f0:f0 + T].copy_(buf[:T])
            gt = np.load(str(self.base.root / "labels" / f"{name}.npy")).astype(np.int64)
            tr = np.load(str(self.base.root / "transcripts" / f"{name}.npy")).astype(np.int64)
            if gt.shape[0] != T:
                raise ValueError(f"{name}: {gt.shape[0]} labels for {T} frames")
            row = np.concatenate([gt, tr, create_tf_input(tr.tolist(), self.sos_token_id),
                                  create_tf_target(tr.tolist(), self.eos_token_id)])
            ints_host[i0:i0 + row.shape[0]] = torch.from_numpy(row)
        if cuda:
            with torch.cuda.stream(side):
                self._ints.copy_(ints_host, non_blocking=True)
            side.synchronize()
        else:
            self._ints.copy_(ints_host)

    # ------------------------------------------------------------------ dataset surface
    def get_num_classes(self) -> int:
        return self.num_actions

    def __len__(self) -> int:
        return len(self.base)

    def is_resident(self, item: int) -> bool:
        return item in self._slot

    def __getitem__(self, item: int) -> Batch:
        slot = self._slot.get(item)
        if slot is None:
            return self.base[item]          # over budget: served from disk like the reference
        f0, T, i0, n = slot
        ints = self._ints
        return Batch(feats=self._feats[f0:f0 + T].unsqueeze(0), gt_label=ints[i0:i0 + T], transcript=ints[i0 + T:i0 + T + n],
                     transcript_tf_input=ints[i0 + T + n:i0 + T + 2 * n + 1],
                     transcript_tf_target=ints[i0 + T + 2 * n + 1:i0 + T + 3 * n + 2], video_name=self.file_names[item])

    def collate_fn(self, items: List[Batch]) -> Batch:
        assert len(items) == 1
        return items[0]
