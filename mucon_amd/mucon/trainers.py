"""Training step with the reference's semantics (src/mucon/trainers.py:18-56, 108-155) plus the
data-parallel exchange that is new in this build: one process per GPU, videos sharded across
ranks, ONE all-reduce of the flat gradient per optimizer step (RCCL over xGMI; gloo on CPU tests),
then the reference's two separate clip_grad_norm_ calls, then the optimizer step.
fandak's Trainer (checkpoint folders, tensorboard, progress bars) is third-party and out of scope."""
from typing import Optional

import torch
from torch import optim
from torch.nn.utils import clip_grad_norm_
from torch.optim.lr_scheduler import MultiStepLR, ReduceLROnPlateau


def create_optimizer(cfg, parameters):
    t = cfg.trainer
    if t.optimizer == "SGD":
        return optim.SGD(params=parameters, lr=t.learning_rate, weight_decay=t.weight_decay, momentum=t.momentum)
    if t.optimizer == "Adam":
        return optim.Adam(params=parameters, lr=t.learning_rate, weight_decay=t.weight_decay, amsgrad=True)
    raise Exception("Invalid optimizer name (%s)" % t.optimizer)


def create_scheduler(cfg, optimizer):
    s = cfg.trainer.scheduler
    if s.name == "none":
        return None
    if s.name == "plateau":
        return ReduceLROnPlateau(optimizer, mode=s.plateau.mode, factor=s.plateau.factor, patience=s.plateau.patience)
    if s.name == "step":
        return MultiStepLR(optimizer, milestones=s.step.milestones, gamma=s.step.gamma)
    raise Exception("Invalid scheduler name (%s)" % s.name)


def all_reduce_gradients(model, world_size: int):
    """Average the gradients over ranks (1,643,298 floats = 6.57 MB for the default model) in two collectives:
    gradients that are views of one flat buffer (the HIP encoder's backward writes all of its 50 gradients into one,
    ops._EncoderFn.backward) are all-reduced in place; the remaining, individually allocated ones (s-head, y-head) are
    packed into one buffer, all-reduced, and scattered back with one multi-tensor copy.  Parameters without a gradient on
    this rank contribute zeros."""
    import torch.distributed as dist

    params = [p for p in model.parameters() if p.requires_grad]
    by_storage = {}
    for p in params:
        if p.grad is not None:
            by_storage.setdefault(p.grad.untyped_storage().data_ptr(), []).append(p)
    loose = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
            loose.append(p)
        elif len(by_storage[p.grad.untyped_storage().data_ptr()]) == 1:
            loose.append(p)
    for group in by_storage.values():
        if len(group) > 1:   # one flat buffer behind several gradients: reduce it where it lies
            g = group[0].grad
            flat = torch.empty(0, dtype=g.dtype, device=g.device).set_(g.untyped_storage())
            dist.all_reduce(flat)
            flat /= world_size
    if loose:
        flat = torch.cat([p.grad.reshape(-1) for p in loose])
        dist.all_reduce(flat)
        flat /= world_size
        torch._foreach_copy_([p.grad for p in loose], [c.view_as(p) for c, p in zip(flat.split([p.numel() for p in loose]), loose)])


class SimpleTrainer:
    def __init__(self, cfg, model, device, train_db=None, world_size: int = 1, rank: int = 0):
        self.cfg, self.model, self.device, self.train_db = cfg, model, device, train_db
        self.world_size, self.rank = world_size, rank
        self.optimizer = create_optimizer(cfg, model.get_params(cfg.trainer.learning_rate))
        self.scheduler = create_scheduler(cfg, self.optimizer)
        self.clip_grad_norm: Optional[float] = cfg.trainer.clip_grad_norm_value if cfg.trainer.clip_grad_norm else None
        self.iter_num = 0
        self.fused_step = self._make_fused_step()
        self.fuse_step = True   # False: always go through torch.autograd

    def _make_fused_step(self):
        """The two clip_grad_norm_ calls + SGD.step() as ops.FusedClipSGD (two launches) when the configuration is the
        one those kernels implement: SGD on the GPU, no gradient accumulation, group-wise or global clipping."""
        from .. import ops
        t = self.cfg.trainer
        if not isinstance(self.optimizer, optim.SGD) or (t.accumulate_grad_every or 1) != 1:
            return None
        if not str(self.device).startswith("cuda") or len(self.optimizer.param_groups) != 1:
            return None
        pg = self.optimizer.param_groups[0]
        if pg.get("nesterov") or pg.get("dampening") or pg.get("maximize"):
            return None
        if self.clip_grad_norm is None:
            groups, mx = [list(self.model.parameters())], None
        elif t.clip_grad_norm_separate:
            groups, mx = [self.model.encode_params, self.model.decode_params], self.clip_grad_norm
            covered = {id(p) for g in groups for p in g}
            if any(id(p) not in covered for p in self.model.parameters()):   # a parameter outside both lists: torch path
                return None
        elif t.clip_grad_norm_every_param:
            return None
        else:
            groups, mx = [list(self.model.parameters())], self.clip_grad_norm
        return ops.FusedClipSGD(groups, mx, self.optimizer)

    def on_start_epoch(self, epoch_num: int):
        self.model.set_teacher_forcing(self.cfg.model.teacher_forcing)

    def _train_1_batch(self, iter_num: int, batch):
        """One video per rank (reference trainers.py:108-155)."""
        acc = self.cfg.trainer.accumulate_grad_every or 1
        if iter_num % acc == 0:
            self.optimizer.zero_grad()
        batch.to(self.device)
        if acc == 1 and self.fuse_step and hasattr(self.model, "can_fuse_step") and self.model.can_fuse_step(batch):
            # forward + loss + backward as one straight line of launches, no autograd graph (MuCon.fused_train_step)
            loss, forward_out = self.model.fused_train_step(batch)
        else:
            forward_out = self.model.forward(batch)
            loss = self.model.loss(batch, forward_out)
            (loss.main / acc).backward()
        last_of_group = iter_num % acc == (acc - 1)
        if last_of_group and self.world_size > 1:
            all_reduce_gradients(self.model, self.world_size)
        if self.fused_step is not None:
            self.fused_step.step()
            return loss, forward_out
        if self.clip_grad_norm is not None:
            t = self.cfg.trainer
            if t.clip_grad_norm_separate:
                clip_grad_norm_(self.model.encode_params, max_norm=self.clip_grad_norm)
                clip_grad_norm_(self.model.decode_params, max_norm=self.clip_grad_norm)
            elif t.clip_grad_norm_every_param:
                for p in self.model.parameters():
                    clip_grad_norm_(p, self.clip_grad_norm)
            else:
                clip_grad_norm_(self.model.parameters(), max_norm=self.clip_grad_norm)
        if last_of_group:
            self.optimizer.step()
        return loss, forward_out

    def train_epoch(self, epoch_num: int, shuffle_seed: Optional[int] = None):
        """Shards the (shuffled) video list over ranks; every rank takes the same number of steps."""
        self.on_start_epoch(epoch_num)
        self.model.train()
        n = len(self.train_db)
        g = torch.Generator().manual_seed((shuffle_seed if shuffle_seed is not None else self.cfg.system.seed) + epoch_num)
        order = torch.randperm(n, generator=g).tolist()
        steps = n // self.world_size if self.world_size > 1 else n
        losses = []
        for s in range(steps):
            batch = self.train_db[order[s * self.world_size + self.rank]]
            loss, _ = self._train_1_batch(self.iter_num, batch)
            losses.append(loss.main.detach())      # stays on the device: no host sync per step (the step is ~2 ms of
            self.iter_num += 1                     # asynchronous launches; a .item() here would serialise host and GPU)
        if self.scheduler is not None and not isinstance(self.scheduler, ReduceLROnPlateau):
            self.scheduler.step()
        return torch.stack(losses).tolist() if losses else []
