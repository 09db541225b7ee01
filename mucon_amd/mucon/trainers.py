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


def dist_all_reduce(t: torch.Tensor):
    """all_reduce(sum) of a tensor in place.  RCCL ("nccl") reduces device tensors directly; under the gloo backend (CPU tests,
    and several ranks sharing ONE GPU, which RCCL refuses) a device tensor travels through a host copy."""
    import torch.distributed as dist

    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)


def dist_broadcast(t: torch.Tensor, src: int = 0):
    import torch.distributed as dist

    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)


class GradBucket:
    """ONE all-reduce per optimizer step whose size and layout are fixed by the parameter list alone -- never by which
    gradients happen to exist, share a storage or were produced by the fused or the autograd path on this rank (ranks that
    disagreed on any of that would issue mismatched collectives: a hang or silently mixed-up gradients).

    Layout: every trainable parameter has a fixed slot in one persistent flat fp32 buffer, followed by one flag per parameter
    ("this rank produced a gradient").  pack -> all_reduce(sum) -> scale by 1 / world_size -> unpack:
      * a parameter that received a gradient on at least one rank gets the average (ranks without one contributed zeros),
        its .grad becomes a view of the flat buffer (no copy back; the buffer is not touched again before the optimizer ran);
      * a parameter without a gradient on ANY rank keeps grad = None on every rank -- exactly what a single process does (the
        optimizer skips it: no weight decay, no momentum update), so N ranks and 1 rank treat unused parameters alike."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.n = sum(self.sizes)
        self.flat = None

    def _ensure(self):
        if self.flat is None:
            p0 = self.params[0]
            self.flat = torch.zeros(self.n + len(self.params), dtype=torch.float32, device=p0.device)
            self.views = [v.view_as(p) for v, p in zip(self.flat[:self.n].split(self.sizes), self.params)]

    def all_reduce(self, world_size: int):
        import torch.distributed as dist

        if not self.params:
            return
        self._ensure()
        have = [p.grad is not None for p in self.params]
        dst_copy, src_copy, dst_zero = [], [], []
        for p, v, h in zip(self.params, self.views, have):
            if not h:
                dst_zero.append(v)
            elif p.grad.data_ptr() != v.data_ptr():     # (a gradient accumulated in place into last step's view is already there)
                dst_copy.append(v)
                src_copy.append(p.grad)
        if dst_copy:
            torch._foreach_copy_(dst_copy, src_copy)
        if dst_zero:
            torch._foreach_zero_(dst_zero)
        self.flat[self.n:] = torch.tensor([1.0 if h else 0.0 for h in have], dtype=torch.float32).to(self.flat.device, non_blocking=True)
        dist_all_reduce(self.flat)
        self.flat[:self.n].mul_(1.0 / world_size)
        counts = self.flat[self.n:].tolist()        # one small read-back per step: who has a gradient anywhere
        for p, v, c in zip(self.params, self.views, counts):
            p.grad = v if c > 0 else None


def all_reduce_gradients(model, world_size: int, bucket: "GradBucket" = None):
    """Average the gradients over ranks with one collective (1,643,298 floats = 6.57 MB for the default model + one flag per
    parameter); see GradBucket.  `bucket` keeps the flat buffer between steps (SimpleTrainer owns one)."""
    (bucket or GradBucket(model.parameters())).all_reduce(world_size)


def broadcast_parameters(model, src: int = 0):
    """Every rank starts from rank `src`'s parameters and buffers (equal seeds are not trusted): one flat broadcast."""
    import torch.distributed as dist

    tensors = [p.data for p in model.parameters()] + [b for b in model.buffers() if b.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist_broadcast(flat, src)
    torch._foreach_copy_(tensors, [c.view_as(t) for c, t in zip(flat.split([t.numel() for t in tensors]), tensors)])


class SimpleTrainer:
    def __init__(self, cfg, model, device, train_db=None, world_size: int = 1, rank: int = 0):
        self.cfg, self.model, self.device, self.train_db = cfg, model, device, train_db
        self.world_size, self.rank = world_size, rank
        self.optimizer = create_optimizer(cfg, model.get_params(cfg.trainer.learning_rate))
        self.scheduler = create_scheduler(cfg, self.optimizer)
        self.clip_grad_norm: Optional[float] = cfg.trainer.clip_grad_norm_value if cfg.trainer.clip_grad_norm else None
        self.iter_num = 0
        self.bucket = None
        if world_size > 1:
            broadcast_parameters(model, 0)
            self.bucket = GradBucket(model.parameters())
        # every rank draws its own dropout masks (the kernels' counter-based dropout is keyed by model.dropout_rank)
        model.dropout_rank = int(rank)
        self.fused_step = self._make_fused_step()
        self.fuse_step = True   # False: always go through torch.autograd

    def _make_fused_step(self):
        """The two clip_grad_norm_ calls + optimizer.step() as ops.FusedClipSGD / ops.FusedClipAdam (two launches) when the
        configuration is one those kernels implement: SGD or Adam (the reference's two optimizers, trainers.py:31-36) on the GPU,
        group-wise or global clipping; with gradient accumulation the non-stepping iterations clip only (mucon_clip_grads)."""
        from .. import ops
        t = self.cfg.trainer
        adam = type(self.optimizer) is optim.Adam
        if not (isinstance(self.optimizer, optim.SGD) or adam):
            return None
        if not str(self.device).startswith("cuda") or len(self.optimizer.param_groups) != 1:
            return None
        pg = self.optimizer.param_groups[0]
        if pg.get("nesterov") or pg.get("dampening") or pg.get("maximize") or pg.get("capturable") or pg.get("differentiable"):
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
        return (ops.FusedClipAdam if adam else ops.FusedClipSGD)(groups, mx, self.optimizer)

    def _zero_grad(self):
        """optimizer.zero_grad() (set_to_none, torch's default) for the parameters of the model: `p.grad = None` in a plain loop.  torch's version
        takes ~35 us of host time for this model's 90 parameters (profiler hooks, foreach grouping) -- 5 % of a one-video training step, which is
        bound by host code as much as by the GPU."""
        cached = self.__dict__.get("_zero_params")
        n = sum(len(g["params"]) for g in self.optimizer.param_groups)
        if cached is None or len(cached) != n:
            cached = self._zero_params = [p for g in self.optimizer.param_groups for p in g["params"]]
        for p in cached:
            p.grad = None

    def on_start_epoch(self, epoch_num: int):
        self.model.set_teacher_forcing(self.cfg.model.teacher_forcing)

    def _train_1_batch(self, iter_num: int, batch):
        """One video per rank (reference trainers.py:108-155)."""
        acc = self.cfg.trainer.accumulate_grad_every or 1
        if iter_num % acc == 0:
            self._zero_grad()
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
            all_reduce_gradients(self.model, self.world_size, self.bucket)
        if self.fused_step is not None:
            if last_of_group:
                self.fused_step.step()
            else:
                self.fused_step.clip_only()      # the reference clips the accumulated gradient at every iteration (trainers.py:131-147)
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

    def check_health(self):
        """The device-side failure words of the asynchronous step (a decoder hand-over time-out, a non-finite gradient norm whose update the
        fused optimizer skipped), read where the host waits for the device anyway; raises MuconHipError (mucon_amd/ops.py:check_health)."""
        from .. import ops
        ops.check_health([self.fused_step] if self.fused_step is not None else [])

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
            losses.append(loss.main.detach())      # stays on the device: no host sync per step (the step is ~0.7 ms of
            self.iter_num += 1                     # asynchronous launches; a .item() here would serialise host and GPU)
            if (s & 31) == 31 and torch.cuda.is_available() and str(self.device).startswith("cuda"):
                # ... but the host must not run FAR ahead either: with ~700 launches queued (it enqueues a step in 0.6 ms, the GPU needs 0.72) the
                # HIP runtime stalls for 15 - 140 ms at a time (tools/e2e_queue_probe.py: 0.875 ms per step un-synchronised, 0.750 with a
                # synchronisation every 40 steps).  Draining every 32 steps costs one ~20 us pipeline restart per 32 steps.
                torch.cuda.current_stream().synchronize()
                self.check_health()
        if steps and torch.cuda.is_available() and str(self.device).startswith("cuda"):
            self.check_health()
        if self.scheduler is not None and not isinstance(self.scheduler, ReduceLROnPlateau):
            self.scheduler.step()
        return torch.stack(losses).tolist() if losses else []

    def step_scheduler_on_eval(self, eval_result) -> None:
        """The plateau scheduler is driven by the evaluation: the reference feeds it eval_results[0].s_mof_nbg
        (trainers.py:157-163, figure_scheduler_input).  Called by the training script after every evaluation."""
        if isinstance(self.scheduler, ReduceLROnPlateau):
            self.scheduler.step(float(eval_result["s_mof_nbg"]))


class TrainerForTFExperiments(SimpleTrainer):
    """SimpleTrainer that trains with teacher forcing up to `turnoff_tf_after_epoch` and without it from that epoch on
    (reference trainers.py:166-191)."""

    def __init__(self, cfg, model, device, train_db=None, world_size: int = 1, rank: int = 0, turnoff_tf_after_epoch: int = 1000):
        super().__init__(cfg, model, device, train_db, world_size=world_size, rank=rank)
        self.turnoff_tf_after_epoch = turnoff_tf_after_epoch

    def on_start_epoch(self, epoch_num: int):
        self.model.set_teacher_forcing(epoch_num < self.turnoff_tf_after_epoch)
