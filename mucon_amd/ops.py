"""torch.autograd wrappers over the C ABI (mucon_amd/_lib.py -> libmucon_hip.so).

PyTorch is plumbing here: it owns the HBM buffers, the current HIP stream and autograd's graph;
all arithmetic of the hot path runs in the hand-written gfx950 kernels.  Device tensors only --
a CPU tensor raises (there is no CPU fallback in the product).
"""
import collections.abc
import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib


@dataclass
class EncoderSpec:
    """cfg.model.ft.* (reference src/configs/mucon/default.py:81-96) as the kernels need it."""
    in_dim: int = 2048
    hidden: int = 128
    stages: List[int] = field(default_factory=lambda: [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024])
    pooling: bool = True
    pooling_type: str = "max"
    pooling_layers: List[int] = field(default_factory=lambda: [1, 2, 4, 8])
    leaky_relu: bool = False
    dropout_rate: float = 0.25
    last_gn: bool = True
    last_gn_num_groups: int = 32
    last_relu: bool = True
    last_dropout: bool = True
    last_dropout_rate: float = 0.25
    gn_eps: float = 1e-5

    def out_length(self, T: int) -> int:
        for i in range(len(self.stages)):
            if self.pooling and i in self.pooling_layers:
                T //= 2
        return T

    def to_c(self, B: int, T: int, training: bool, seed: int) -> _lib.EncoderCfg:
        if self.pooling_type not in ("max", "sum"):
            # the reference treats every non-"max" value as avg*2 (temporal.py:138-142)
            pool_type = 1
        else:
            pool_type = 0 if self.pooling_type == "max" else 1
        if len(self.stages) > _lib.MAX_LAYERS:
            raise ValueError(f"{len(self.stages)} stages > {_lib.MAX_LAYERS}")
        c = _lib.EncoderCfg()
        c.B, c.T, c.D, c.H = B, T, self.in_dim, self.hidden
        c.n_layers = len(self.stages)
        for i, d in enumerate(self.stages):
            c.dilation[i] = int(d)
            c.pool_after[i] = 1 if (self.pooling and i in self.pooling_layers) else 0
        c.pool_type = pool_type
        c.leaky = 1 if self.leaky_relu else 0
        c.last_gn, c.gn_groups, c.gn_eps = (1 if self.last_gn else 0), self.last_gn_num_groups, self.gn_eps
        c.last_relu = 1 if self.last_relu else 0
        c.training = 1 if training else 0
        c.p_drop_layer = float(self.dropout_rate)
        c.p_drop_last = float(self.last_dropout_rate) if self.last_dropout else 0.0
        c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        return c


def param_names(spec: EncoderSpec) -> List[str]:
    """Reference state_dict keys of the encoder parameters, in the order the ops take them."""
    names = ["ft.first_conv.weight", "ft.first_conv.bias"]
    for i in range(len(spec.stages)):
        names += [f"ft.l_{i}.dilated_conv.weight", f"ft.l_{i}.dilated_conv.bias",
                  f"ft.l_{i}.conv_1x1.weight", f"ft.l_{i}.conv_1x1.bias"]
    names += ["ft.last_conv.weight", "ft.last_conv.bias", "ft_last_gn.weight", "ft_last_gn.bias"]
    return names


def _pack_params(spec: EncoderSpec, tensors: Sequence[torch.Tensor]) -> _lib.EncoderParams:
    L = len(spec.stages)
    assert len(tensors) == 6 + 4 * L, (len(tensors), L)
    p = _lib.EncoderParams()
    p.first_w, p.first_b = tensors[0].data_ptr(), tensors[1].data_ptr()
    for i in range(L):
        p.dil_w[i], p.dil_b[i], p.pw_w[i], p.pw_b[i] = (t.data_ptr() for t in tensors[2 + 4 * i: 6 + 4 * i])
    p.last_w, p.last_b, p.gn_w, p.gn_b = (t.data_ptr() for t in tensors[2 + 4 * L: 6 + 4 * L])
    return p


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.MuconHipError("mucon_amd ops need device tensors: there is no CPU fallback")
        if t is not None and t.dtype != torch.float32:
            raise _lib.MuconHipError(f"float32 expected, got {t.dtype}")


_STRUCT_CACHE = {}


def _param_struct(kind, tensors, build):
    """The ctypes struct of device pointers `build(tensors)` for a list of PARAMETER tensors, kept while they live where they did: a struct is a
    function of the addresses alone; the key is (kind, (address, dtype, numel) per tensor).  On a miss the tensors are checked (device, float32, contiguous); on a hit
    nothing but 46 data_ptr() calls happens -- these lists were re-checked and re-packed three times per training step (0.1 ms of the host's
    0.7 ms per one-video step, which is bound by host code as much as by the GPU: tools/e2e_host_profile.py)."""
    # (dtype and element count ride in the key: an address alone can be re-used by another tensor once a parameter's storage is freed)
    key = (kind,) + tuple((t.data_ptr(), t.dtype, t.numel()) for t in tensors)
    hit = None if os.environ.get("MUCON_NO_STRUCT_CACHE") else _STRUCT_CACHE.get(key)
    if hit is None:
        _check_dev(*tensors)
        for t in tensors:
            if not t.is_contiguous():
                raise _lib.MuconHipError("tensor handed to the C ABI must be contiguous")
        if len(_STRUCT_CACHE) >= 64:
            _STRUCT_CACHE.clear()
        hit = _STRUCT_CACHE[key] = build(tensors)
    return hit


class _NoGradCtx:
    """Stands in for autograd's context when gradients are off (evaluation): `Fn.forward(_NoGradCtx(), ...)` runs the same code
    without torch.autograd.Function.apply's bookkeeping (~10 us per call of a forward that is ~35 launches of ~3.5 us each)."""
    needs_input_grad = (False,) * 64

    def save_for_backward(self, *tensors):
        pass

    def set_materialize_grads(self, flag):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


def _apply(fn, *args):
    """fn.apply(*args), or -- with gradients off -- fn.forward on a stand-in context."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoGradCtx(), *args)


_PARAM_GRADS = {}     # _param_grads under ctx.reuse_grads: (tag, parameter identities) -> (gradient tensors, their struct or None, the parameters); <= 8 entries


def _param_grads(ctx, tag, params, pack=None):
    """Gradient tensors for `params` (+ their ctypes struct when `pack` is given).  Fresh ones, as autograd expects -- unless ctx.reuse_grads (the fused step
    paths: the optimizer has consumed the previous step's gradients before this pass writes): then the tensors of the previous step serve again.  Besides the
    allocations this keeps the fused optimizer's table identical step after step (no upload, ops.FusedClipSGD skips the unchanged records)."""
    if not getattr(ctx, "reuse_grads", False):
        grads = [torch.empty_like(w) for w in params]
        return grads, (pack(grads) if pack else None)
    key = (tag, tuple(id(w) for w in params))
    hit = _PARAM_GRADS.get(key)
    if hit is not None and all(g.shape == w.shape and g.device == w.device for g, w in zip(hit[0], params)):
        return hit[0], hit[1]
    grads = [torch.empty_like(w) for w in params]
    st = pack(grads) if pack else None
    if len(_PARAM_GRADS) >= 8:
        _PARAM_GRADS.clear()
    _PARAM_GRADS[key] = (grads, st, list(params))   # (the parameters are held: their ids cannot be recycled while the entry lives)
    return grads, st


_GRAD_BUFFERS = {}   # _EncoderFn.backward under ctx.reuse_grads: (parameter identities, layout) -> (flat buffer, per-parameter views, their struct, tail offset); <= 4 entries


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tape, spec, training, seed, *params):
        lib = _lib.load()
        _check_dev(tape)
        tape = tape.contiguous()
        params = [p if p.is_contiguous() else p.contiguous() for p in params]
        B, T, D = tape.shape
        if D != spec.in_dim:
            raise ValueError(f"tape feature dim {D} != {spec.in_dim}")
        cfg = spec.to_c(B, T, training, seed)
        nbytes = lib.mucon_encoder_workspace_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            _lib.check(_lib.E_ARG, "mucon_encoder_workspace_bytes")
        Tz = lib.mucon_encoder_out_length(ctypes.byref(cfg))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=tape.device)
        enc = torch.empty((B, Tz, spec.hidden), dtype=torch.float32, device=tape.device)
        cp = _param_struct(("enc", len(spec.stages)), params, lambda ts: _pack_params(spec, ts))
        _lib.check(lib.mucon_encoder_fwd(ctypes.byref(cfg), ctypes.byref(cp), _lib.ptr(tape), _lib.ptr(enc),
                                         _lib.ptr(ws), nbytes, _lib.current_stream_ptr()), "mucon_encoder_fwd")
        ctx.spec, ctx.cfg, ctx.ws, ctx.nbytes = spec, cfg, ws, nbytes
        ctx.save_for_backward(tape, *params)
        return enc

    @staticmethod
    def backward(ctx, d_enc):
        lib = _lib.load()
        tape, *params = ctx.saved_tensors
        d_enc = d_enc.contiguous()
        # one flat gradient buffer, the per-parameter gradients are views into it: data-parallel training
        # all-reduces the flat buffer directly (flat_grad_buffers), no gather / scatter copies
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]   # 256-byte aligned slots
        # (ctx.flat_extra floats of room behind them for the caller's other gradients -- ctx.flat_tail: then ONE collective on
        # one buffer reduces everything, with no gather copy)
        extra = int(getattr(ctx, "flat_extra", 0))
        ov = getattr(ctx, "dp_overlap", None)
        # (r6) ctx.reuse_grads (the fused step paths, where the optimizer has consumed the previous step's gradients before this pass runs): the flat buffer,
        # its ~50 per-parameter views and their ctypes struct are kept between steps -- building them is 0.13 ms of the host's 0.72 ms per one-video
        # training step (tools/experiments/e2e_host_sections.py), which is bound by the host as much as by the GPU.  The pass writes every member.
        reuse_key = None
        if getattr(ctx, "reuse_grads", False):
            reuse_key = (tuple(id(p) for p in params), tuple(sizes), extra, ov is not None, tape.device)
            hit = _GRAD_BUFFERS.get(reuse_key)
            if hit is not None:
                flat, grads, cg, tail_at, _ = hit
                ctx.flat_tail = flat[tail_at: tail_at + extra] if extra else flat[:0]
                ctx.flat_rest_off = sizes[0] + sizes[1] + extra if ov is not None else 0
                cp = _param_struct(("enc", len(ctx.spec.stages)), params, lambda ts: _pack_params(ctx.spec, ts))
                return _EncoderFn._launch_backward(ctx, lib, tape, d_enc, cp, cg, ov, grads)
        flat = torch.empty(sum(sizes) + extra, dtype=torch.float32, device=tape.device)
        # layout: [every parameter][extra] -- or, for the overlapped data-parallel step, [first_conv.weight, .bias][extra][the rest]: what is
        # final only at the pass's end (first_conv's gradients, the caller's tail) is one contiguous range, what is final at the event the other
        head = sizes[0] + sizes[1]
        tail_at = head if ov is not None else sum(sizes)
        ctx.flat_tail = flat[tail_at: tail_at + extra]
        ctx.flat_rest_off = head + extra if ov is not None else 0
        grads, off = [], 0
        for k, (p, n) in enumerate(zip(params, sizes)):
            if ov is not None and k == 2:
                off += extra
            grads.append(flat[off: off + p.numel()].view(p.shape))
            off += n
        cp, cg = _param_struct(("enc", len(ctx.spec.stages)), params, lambda ts: _pack_params(ctx.spec, ts)), _pack_params(ctx.spec, grads)
        if reuse_key is not None:
            if len(_GRAD_BUFFERS) >= 4:
                _GRAD_BUFFERS.clear()
            _GRAD_BUFFERS[reuse_key] = (flat, grads, cg, tail_at, list(params))   # (the parameters are held: their ids cannot be recycled while the entry lives)
        return _EncoderFn._launch_backward(ctx, lib, tape, d_enc, cp, cg, ov, grads)

    @staticmethod
    def _launch_backward(ctx, lib, tape, d_enc, cp, cg, ov, grads):
        if ov is not None:
            # data-parallel step: (torch.cuda.Event, max workgroups) -- the event is recorded once every gradient but first_conv's is final, the
            # weight-gradient launches leave CUs free for RCCL (include/mucon_hip.h: mucon_encoder_bwd_overlap).  flat[ctx.flat_rest_off:] is what is
            # final at the event.
            ev, max_wg = ov
            ev.record()            # (materialises the event's handle; recorded again, in its place, by the library)
            _lib.check(lib.mucon_encoder_bwd_overlap(ev.cuda_event, int(max_wg)), "mucon_encoder_bwd_overlap")
        _lib.check(lib.mucon_encoder_bwd(ctypes.byref(ctx.cfg), ctypes.byref(cp), _lib.ptr(tape), _lib.ptr(d_enc),
                                         _lib.ptr(ctx.ws), ctx.nbytes, ctypes.byref(cg), _lib.current_stream_ptr()),
                   "mucon_encoder_bwd")
        return (None, None, None, None, *grads)


def flat_grad_buffers(params: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """The distinct underlying buffers of the parameters' .grad tensors (the encoder's gradients live in ONE
    flat buffer, see _EncoderFn.backward).  All-reducing these in place all-reduces every gradient."""
    seen, out = set(), []
    for p in params:
        g = p.grad
        if g is None:
            continue
        st = g.untyped_storage()
        key = st.data_ptr()
        if key not in seen:
            seen.add(key)
            out.append(torch.empty(0, dtype=g.dtype, device=g.device).set_(st))   # 1-D tensor over the whole storage
    return out


def encoder_saved(enc: torch.Tensor, kind: str, layer: int = 0) -> torch.Tensor:
    """Intermediate the forward left in its workspace, as a [B, rows, 128] view: kind in
    {"x", "h", "ypre", "z"} (see mucon_encoder_saved_view).  For tests / debugging."""
    fn = enc.grad_fn
    if fn is None or not hasattr(fn, "ws"):
        raise ValueError("enc is not the output of encoder_forward with autograd enabled")
    lib = _lib.load()
    off, rows = ctypes.c_size_t(0), ctypes.c_int32(0)
    _lib.check(lib.mucon_encoder_saved_view(ctypes.byref(fn.cfg), {"x": 0, "h": 1, "ypre": 2, "z": 3}[kind], int(layer),
                                            ctypes.byref(off), ctypes.byref(rows)), "mucon_encoder_saved_view")
    n = fn.cfg.B * rows.value * fn.cfg.H
    return fn.ws[off.value: off.value + 4 * n].view(torch.float32).view(fn.cfg.B, rows.value, fn.cfg.H)


def encoder_forward(tape: torch.Tensor, params: Sequence[torch.Tensor], spec: EncoderSpec, training: bool = False,
                    seed: int = 0) -> torch.Tensor:
    """tape [B,T,D] -> enc [B,Tz,H]: MuCon.temporal_modeling_forward (reference models.py:746-773).
    `params` in param_names(spec) order.  Differentiable w.r.t. params (the tape needs no grad)."""
    return _apply(_EncoderFn, tape, spec, bool(training), int(seed), *params)


class _LinearFn(torch.autograd.Function):
    """out [B,T,128] = tape [B,T,D] @ w[128,D]^T + b on the first_conv kernels (the encoder variant "noft",
    reference core/modules/temporal.py:56-74).  The tape gets no gradient (it is the dataset's features)."""

    @staticmethod
    def forward(ctx, tape, w, b):
        lib = _lib.load()
        _check_dev(tape, w, b)
        if getattr(ctx, "needs_input_grad", (False,))[0]:
            # first_conv's kernels produce no gradient for their input (the dataset's features need none): refuse loudly
            # rather than hand autograd a silent None where nn.Conv1d would propagate
            raise _lib.MuconHipError("linear_forward: the input tape requires a gradient, which the HIP path does not produce "
                                     "(detach it, or use torch ops for this input)")
        tape, w, b = tape.contiguous(), w.contiguous(), b.contiguous()
        B, T, D = tape.shape
        nbytes = lib.mucon_linear_workspace_bytes(B, T, D)
        if nbytes == 0:
            _lib.check(_lib.E_ARG, "mucon_linear_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=tape.device)
        out = torch.empty((B, T, 128), dtype=torch.float32, device=tape.device)
        _lib.check(lib.mucon_linear_fwd(B, T, D, _lib.ptr(tape), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), _lib.ptr(ws), nbytes,
                                        _lib.current_stream_ptr()), "mucon_linear_fwd")
        ctx.dims, ctx.ws, ctx.nbytes = (B, T, D), ws, nbytes
        ctx.save_for_backward(tape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        (tape,) = ctx.saved_tensors
        B, T, D = ctx.dims
        d_w = torch.empty((128, D), dtype=torch.float32, device=tape.device)
        d_b = torch.empty(128, dtype=torch.float32, device=tape.device)
        _lib.check(lib.mucon_linear_bwd(B, T, D, _lib.ptr(tape), _lib.ptr(d_out.contiguous()), _lib.ptr(d_w), _lib.ptr(d_b),
                                        _lib.ptr(ctx.ws), ctx.nbytes, _lib.current_stream_ptr()), "mucon_linear_bwd")
        return None, d_w, d_b


class _Conv128Fn(torch.autograd.Function):
    """y [B,T,128] = conv1d over time of x [B,T,128] with w [128,128,taps] (nn.Conv1d layout), padding = dilation -- the building
    block of the encoder variant "mstcnpp" (reference core/modules/temporal.py:150-204) on the encoder's f32-MFMA kernels
    (mucon_conv128_fwd / _dgrad / _wgrad).  The weight re-layouts are two small permutes."""

    @staticmethod
    def forward(ctx, x, w, b, dilation):
        lib = _lib.load()
        _check_dev(x, w)
        x, w = x.contiguous(), w.contiguous()
        B, T, Cc = x.shape
        taps = int(w.shape[2])
        assert Cc == 128 and tuple(w.shape[:2]) == (128, 128) and taps in (1, 3)
        w_fwd = w.permute(0, 2, 1).contiguous()                      # [o][tap][i]
        y = torch.empty_like(x)
        _lib.check(lib.mucon_conv128_fwd(B, T, taps, int(dilation), _lib.ptr(x), _lib.ptr(w_fwd),
                                         _lib.ptr(b.contiguous()) if b is not None else None, _lib.ptr(y),
                                         _lib.current_stream_ptr()), "mucon_conv128_fwd")
        ctx.dims = (B, T, taps, int(dilation), b is not None)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        B, T, taps, dilation, has_b = ctx.dims
        g = g.contiguous()
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            w_bwd = w.permute(1, 2, 0).contiguous()                  # [i][tap][o]
            d_x = torch.empty_like(x)
            _lib.check(lib.mucon_conv128_dgrad(B, T, taps, dilation, _lib.ptr(g), _lib.ptr(w_bwd), _lib.ptr(d_x),
                                               _lib.current_stream_ptr()), "mucon_conv128_dgrad")
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            nbytes = lib.mucon_conv128_workspace_bytes(B, T, taps)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            d_w = torch.empty_like(w)
            d_b = torch.empty(128, dtype=torch.float32, device=x.device) if has_b else None
            _lib.check(lib.mucon_conv128_wgrad(B, T, taps, dilation, _lib.ptr(g), _lib.ptr(x), _lib.ptr(d_w),
                                               _lib.ptr(d_b) if has_b else None, _lib.ptr(ws), nbytes,
                                               _lib.current_stream_ptr()), "mucon_conv128_wgrad")
        return d_x, d_w, d_b, None


class _MstcnFuseFn(torch.autograd.Function):
    """The tail of an MS-TCN++ layer (reference core/modules/temporal.py:196-201) as one launch:
    y = [max_pool1d(2)](f + dropout(relu(conv_fusion(cat(a, b))))) -- mucon_mstcn_fuse_fwd; its backward is one element-wise
    launch (mucon_mstcn_tail_bwd: un-pooling + the ReLU / dropout mask) and the convolution gradients of mucon_conv128_*."""

    @staticmethod
    def forward(ctx, a, b, f, w, bias, p_drop, seed, training, pool):
        lib = _lib.load()
        _check_dev(a, b, f, w, bias)
        a, b, f = a.contiguous(), b.contiguous(), f.contiguous()
        w2 = w.reshape(w.shape[0], w.shape[1]).contiguous()
        B, T, Cc = a.shape
        assert Cc == 128 and tuple(w2.shape) == (128, 256) and a.shape == b.shape == f.shape
        dev = a.device
        y = torch.empty((B, T // 2 if pool else T, 128), dtype=torch.float32, device=dev)
        y_pre = torch.empty_like(a) if pool else None
        need_bwd = any(ctx.needs_input_grad[:5]) if hasattr(ctx, "needs_input_grad") else False
        x_act = torch.empty_like(a) if need_bwd else None
        _lib.check(lib.mucon_mstcn_fuse_fwd(B, T, _lib.ptr(a), _lib.ptr(b), _lib.ptr(w2), _lib.ptr(bias.contiguous()) if bias is not None else None,
                                            _lib.ptr(f), float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF, int(bool(training)), int(bool(pool)),
                                            _lib.ptr(y), _lib.ptr(y_pre), _lib.ptr(x_act), _lib.current_stream_ptr()), "mucon_mstcn_fuse_fwd")
        ctx.dims = (B, T, bool(pool), (1.0 / (1.0 - p_drop)) if (training and p_drop > 0) else 1.0, bias is not None)
        ctx.save_for_backward(a, b, w2, y_pre, x_act)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b, w2, y_pre, x_act = ctx.saved_tensors
        B, T, pool, scale, has_b = ctx.dims
        g = g.contiguous()
        d_sum, d_u = torch.empty_like(a), torch.empty_like(a)
        st = _lib.current_stream_ptr()
        _lib.check(lib.mucon_mstcn_tail_bwd(B, T, int(pool), _lib.ptr(g), _lib.ptr(y_pre), _lib.ptr(x_act), float(scale), _lib.ptr(d_sum),
                                            _lib.ptr(d_u), st), "mucon_mstcn_tail_bwd")
        d_a = d_b = d_w = d_bias = None
        wa, wb = w2[:, :128], w2[:, 128:]
        if ctx.needs_input_grad[0]:
            d_a = torch.empty_like(a)      # d a = d_u . W_a: a 1x1 "data gradient" whose operand is W_a as [in][out]
            _lib.check(lib.mucon_conv128_dgrad(B, T, 1, 0, _lib.ptr(d_u), _lib.ptr(wa.t().contiguous()), _lib.ptr(d_a), st), "mucon_conv128_dgrad")
        if ctx.needs_input_grad[1]:
            d_b = torch.empty_like(b)
            _lib.check(lib.mucon_conv128_dgrad(B, T, 1, 0, _lib.ptr(d_u), _lib.ptr(wb.t().contiguous()), _lib.ptr(d_b), st), "mucon_conv128_dgrad")
        if ctx.needs_input_grad[3] or (has_b and ctx.needs_input_grad[4]):
            nbytes = lib.mucon_conv128_workspace_bytes(B, T, 1)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
            d_w = torch.empty((128, 256, 1), dtype=torch.float32, device=a.device)
            halves = torch.empty((2, 128, 128), dtype=torch.float32, device=a.device)
            d_bias = torch.empty(128, dtype=torch.float32, device=a.device) if has_b else None
            for k, src in enumerate((a, b)):
                _lib.check(lib.mucon_conv128_wgrad(B, T, 1, 0, _lib.ptr(d_u), _lib.ptr(src), _lib.ptr(halves[k]),
                                                   _lib.ptr(d_bias) if (has_b and k == 0) else None, _lib.ptr(ws), nbytes, st),
                           "mucon_conv128_wgrad")
            d_w[:, :128, 0], d_w[:, 128:, 0] = halves[0], halves[1]
        return d_a, d_b, (d_sum if ctx.needs_input_grad[2] else None), d_w, d_bias, None, None, None, None


def mstcn_fuse_forward(a: torch.Tensor, b: torch.Tensor, f: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                       p_drop: float, seed: int, training: bool, pool: bool) -> torch.Tensor:
    """a, b, f [B,T,128] time-major; weight = conv_fusion.weight [128,256,1] -> [B, T or T//2, 128]."""
    return _MstcnFuseFn.apply(a, b, f, weight, bias, float(p_drop), int(seed), bool(training), bool(pool))


def conv128_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], dilation: int = 1) -> torch.Tensor:
    """x [B,T,128] time-major -> [B,T,128]; weight [128,128,1|3] (nn.Conv1d layout), zero padding = dilation."""
    return _Conv128Fn.apply(x, weight, bias, dilation)


def linear_forward(tape: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """tape [B,T,D] -> [B,T,128]; weight [128,D] (or the Conv1d form [128,D,1])."""
    return _LinearFn.apply(tape, weight.reshape(weight.shape[0], weight.shape[1]), bias)


DEFER_NEXT_HEAD_FORWARD = False   # set by MuCon.fused_train_step in front of its y-head forward (a forward has no ctx to carry the wish)


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, w, b, Tf, want_logits, want_logp):
        lib = _lib.load()
        _check_dev(enc, w, b)
        enc, w, b = enc.contiguous(), w.contiguous(), b.contiguous()
        B, Tz, H = enc.shape
        C = w.shape[0]
        nbytes = lib.mucon_head_workspace_bytes(B, Tz, H, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=enc.device)
        logits = torch.empty((B, Tf, C), dtype=torch.float32, device=enc.device) if want_logits else None
        logp = torch.empty((B, Tf, C), dtype=torch.float32, device=enc.device) if want_logp else None
        global DEFER_NEXT_HEAD_FORWARD
        if DEFER_NEXT_HEAD_FORWARD:
            # (r6) MuCon.fused_train_step: the LSTM's forward follows on this stream before anyone reads the classifier's outputs -- the kernel rides in that
            # pass's recurrence launch (include/mucon_hip.h: mucon_head_fwd_defer).  One-shot; the outputs below are written by THAT call.
            DEFER_NEXT_HEAD_FORWARD = False
            _lib.check(lib.mucon_head_fwd_defer(1), "mucon_head_fwd_defer")
        _lib.check(lib.mucon_head_fwd(B, Tz, Tf, H, C, _lib.ptr(enc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(logits),
                                      _lib.ptr(logp), _lib.ptr(ws), nbytes, _lib.current_stream_ptr()),
                   "mucon_head_fwd")
        ctx.dims, ctx.ws, ctx.nbytes = (B, Tz, Tf, H, C), ws, nbytes
        ctx.save_for_backward(enc, w)
        ctx.set_materialize_grads(False)
        out_logits = logits if want_logits else enc.new_empty(0)
        out_logp = logp if want_logp else enc.new_empty(0)
        unused = [t for t, want in ((out_logits, want_logits), (out_logp, want_logp)) if not want]
        if unused:
            ctx.mark_non_differentiable(*unused)
        return out_logits, out_logp

    @staticmethod
    def backward(ctx, d_logits, d_logp):
        lib = _lib.load()
        enc, w = ctx.saved_tensors
        B, Tz, Tf, H, C = ctx.dims
        dl = d_logits.contiguous() if (d_logits is not None and d_logits.numel() == B * Tf * C) else None
        dp = d_logp.contiguous() if (d_logp is not None and d_logp.numel() == B * Tf * C) else None
        d_enc = torch.empty_like(enc)
        # (keyed by the weight's ADDRESS and shape: the callers hand in a fresh [C, H] view of conv_classifier.weight every step, whose id() says nothing)
        hkey = ("head", w.data_ptr(), tuple(w.shape), w.device)
        hit = _PARAM_GRADS.get(hkey) if getattr(ctx, "reuse_grads", False) else None
        if hit is not None and hit[0][0].shape == w.shape and hit[0][0].device == w.device:
            d_w, d_b = hit[0]                     # (the fused step paths: the previous step's pair, see _param_grads)
        else:
            d_w = torch.empty_like(w)
            d_b = torch.empty(C, dtype=torch.float32, device=enc.device)
            if getattr(ctx, "reuse_grads", False):
                if len(_PARAM_GRADS) >= 8:
                    _PARAM_GRADS.clear()
                _PARAM_GRADS[hkey] = ([d_w, d_b], None, [w])
        if getattr(ctx, "defer_reduce", False):
            # (r6) the fused step paths (bench.py, MuCon.fused_train_step), where the encoder's backward follows on this stream before anyone reads
            # d_w / d_b: their slab sums ride in that pass's first launch (include/mucon_hip.h: mucon_head_bwd_defer).  ctx keeps the workspace alive.
            # ctx.defer_kernel (MuCon.fused_train_step only): the z-level kernel itself waits for the decoder's backward, which follows on this stream there
            _lib.check(lib.mucon_head_bwd_defer(3 if getattr(ctx, "defer_kernel", False) else 1), "mucon_head_bwd_defer")
            ctx.deferred = (d_w, d_b)
        _lib.check(lib.mucon_head_bwd(B, Tz, Tf, H, C, _lib.ptr(enc), _lib.ptr(w), _lib.ptr(dl), _lib.ptr(dp),
                                      _lib.ptr(d_enc), _lib.ptr(d_w), _lib.ptr(d_b), _lib.ptr(ctx.ws), ctx.nbytes,
                                      _lib.current_stream_ptr()), "mucon_head_bwd")
        return d_enc, d_w, d_b, None, None, None


def head_forward(enc: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, Tf: int, want_logits: bool = True,
                 want_logp: bool = True):
    """enc [B,Tz,H] -> (logits [B,Tf,C], logp [B,Tf,C]): frame_classifier_forward + log_softmax
    (reference models.py:567-582, :368).  weight is conv_classifier.weight [C,H,1] (or [C,H])."""
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    logits, logp = _apply(_HeadFn, enc, w2, bias, int(Tf), bool(want_logits), bool(want_logp))
    return (logits if want_logits else None), (logp if want_logp else None)


# --------------------------------------------------------------------------------------- s-head LSTM
class _LstmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ndir, *weights):
        lib = _lib.load()
        _check_dev(x)
        x = x.contiguous()
        weights = [w if w.is_contiguous() else w.contiguous() for w in weights]
        T, I = x.shape
        H = weights[1].shape[1]
        nbytes = lib.mucon_lstm_workspace_bytes(T, ndir)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
        out = torch.empty((T, ndir * H), dtype=torch.float32, device=x.device)
        hn = torch.empty((ndir, H), dtype=torch.float32, device=x.device)
        cn = torch.empty((ndir, H), dtype=torch.float32, device=x.device)
        _lib.check(lib.mucon_lstm_fwd(T, I, H, ndir, _lib.ptr(x), ctypes.byref(_param_struct(("lstm", ndir), weights, lambda ts: _lstm_params(ts, ndir))), _lib.ptr(out),
                                      _lib.ptr(hn), _lib.ptr(cn), _lib.ptr(ws), nbytes, _lib.current_stream_ptr()),
                   "mucon_lstm_fwd")
        ctx.dims, ctx.ws, ctx.nbytes = (T, I, H, ndir), ws, nbytes
        ctx.save_for_backward(x, out, *weights)
        ctx.set_materialize_grads(False)
        return out, hn, cn

    @staticmethod
    def backward(ctx, d_out, d_hn, d_cn):
        lib = _lib.load()
        x, out, *weights = ctx.saved_tensors
        T, I, H, ndir = ctx.dims
        d_out = d_out.contiguous() if d_out is not None else None
        d_hn = d_hn.contiguous() if d_hn is not None else None
        d_cn = d_cn.contiguous() if d_cn is not None else None
        # graph-free execution may hand in the gradient another consumer of x already produced (ctx.dx_accumulate, contiguous,
        # x's shape): the kernel adds the LSTM's own into it instead of a separate element-wise launch
        acc = getattr(ctx, "dx_accumulate", None)
        d_x = acc if acc is not None else torch.empty_like(x)
        grads, cg = _param_grads(ctx, ("lstm", ndir), weights, lambda ts: _lstm_params(ts, ndir))
        _lib.check(lib.mucon_lstm_bwd(T, I, H, ndir, _lib.ptr(x), ctypes.byref(_param_struct(("lstm", ndir), weights, lambda ts: _lstm_params(ts, ndir))), _lib.ptr(out),
                                      _lib.ptr(d_out), _lib.ptr(d_hn), _lib.ptr(d_cn), _lib.ptr(d_x), _lib.ptr(acc),
                                      ctypes.byref(cg), _lib.ptr(ctx.ws), ctx.nbytes,
                                      _lib.current_stream_ptr()), "mucon_lstm_bwd")
        return (d_x, None, *grads)


def _lstm_params(weights, ndir):
    """weights in torch.nn.LSTM._flat_weights order: (w_ih, w_hh, b_ih, b_hh) per direction."""
    p = _lib.LstmParams()
    for d in range(ndir):
        w_ih, w_hh, b_ih, b_hh = weights[4 * d:4 * d + 4]
        p.w_ih[d], p.w_hh[d], p.b_ih[d], p.b_hh[d] = _lib.ptr(w_ih), _lib.ptr(w_hh), _lib.ptr(b_ih), _lib.ptr(b_hh)
    return p


def lstm_forward(x: torch.Tensor, weights: Sequence[torch.Tensor], bidirectional: bool = True):
    """x [T, 128] -> (out [T, ndir*128], h_n [ndir, 128], c_n [ndir, 128]): torch.nn.LSTM(128, 128,
    bidirectional) at batch 1 with zero initial state, as the s-head calls it (reference models.py:605-611).
    `weights` = the module's parameters in its own order (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0
    [, *_reverse])."""
    ndir = 2 if bidirectional else 1
    if len(weights) != 4 * ndir:
        raise ValueError(f"lstm_forward: expected {4 * ndir} weight tensors, got {len(weights)}")
    return _apply(_LstmFn, x, ndir, *weights)


# --------------------------------------------------------------------------------------- s-head decoder
# the reference's state_dict names of the 23 decoder tensors, in _lib.DECODER_PARAM_FIELDS order
DECODER_STATE_NAMES = (
    "fs_encoder_hidden_out.weight", "fs_encoder_hidden_out.bias", "fs_encoder_cn_out.weight", "fs_encoder_cn_out.bias",
    "fs_decoder_attention_W1", "fs_decoder_attention_l2.weight", "fs_decoder_attention_l2.bias", "fs_decoder_attention_V",
    "fs_decoder_embedding.weight", "fs_decoder_attn_combine.weight", "fs_decoder_attn_combine.bias",
    "fs_decoder_lstm.weight_ih_l0", "fs_decoder_lstm.weight_hh_l0", "fs_decoder_lstm.bias_ih_l0", "fs_decoder_lstm.bias_hh_l0",
    "fs_decoder_transcript.0.weight", "fs_decoder_transcript.0.bias", "fs_decoder_transcript.2.weight",
    "fs_decoder_transcript.2.bias", "fs_decoder_length.0.weight", "fs_decoder_length.0.bias",
    "fs_decoder_length.2.weight", "fs_decoder_length.2.bias")
LSTM_STATE_NAMES = tuple(f"fs_encoder_lstm.{n}_l0{s}" for s in ("", "_reverse")
                         for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))


def _decoder_params(tensors):
    p = _lib.DecoderParams()
    for name, t in zip(_lib.DECODER_PARAM_FIELDS, tensors):
        setattr(p, name, _lib.ptr(t))
    return p


DECODER_HANDOVER_FAILED = ("mucon_decoder_fwd: a hand-over between the eight workgroups of the step kernel did not complete (they were not "
                           "co-resident for about a second: the GPU is oversubscribed); n_steps = -1, the outputs of this call are invalid")


NONFINITE_GRADIENT_NORM = ("fused clip + optimizer step: clipping group {group} met a non-finite gradient norm in {skipped} step(s) since the last check "
                           "(norm of the latest step: {norm}) -- those updates were NOT applied "
                           "(csrc/optim.hpp).  A non-finite gradient is a diverged step or a kernel that reported a failure by poisoning its outputs "
                           "(the eight-workgroup decoder's hand-over time-out fills its outputs / dV with NaN)")
_PENDING_DECODER_STATUS = []      # n_steps words (device int32 [1]) of teacher-forced decoder calls nobody has read yet


def check_health(optimizers=()):
    """To be called where the host synchronises anyway (SimpleTrainer.train_epoch drains the stream every 32 steps): reads the status words the
    asynchronous training path left on the device and raises MuconHipError on
      * a decoder launch whose eight workgroups failed a hand-over (n_steps = -1; a teacher-forced forward does not read the word itself),
      * a fused optimizer step that met a non-finite gradient norm (`last_norms`; the device skipped that group's update, so the weights
        are still the last good ones).
    Synchronises.  Cheap: one small device-to-host copy per call."""
    pend = list(_PENDING_DECODER_STATUS)
    del _PENDING_DECODER_STATUS[:]
    if pend:
        if int(torch.stack([t.reshape(()) for t in pend]).min().item()) < 0:
            raise _lib.MuconHipError(DECODER_HANDOVER_FAILED)
    for opt in optimizers:
        buf = getattr(opt, "_norms_buf", None)
        if buf is None or getattr(opt, "max_norm", None) is None:
            continue
        host = buf.detach().cpu()
        n = host.numel() // 2
        skipped = host[n:]
        bad = ((~torch.isfinite(host[:n])) | (skipped > 0)).nonzero().flatten().tolist()
        if bad:
            buf[n:].zero_()     # the counts are sticky on the device: reported once, then cleared here
            raise _lib.MuconHipError(NONFINITE_GRADIENT_NORM.format(group=bad[0], norm=float(host[bad[0]]), skipped=int(skipped[bad[0]])))


def _norms_for(holder, n_groups, dev):
    """The [2 * n_groups] device buffer of a fused optimizer's launches (include/mucon_hip.h): norms, then the STICKY counts of skipped steps.
    `holder.last_norms` is a view of the first half."""
    buf = getattr(holder, "_norms_buf", None)
    if buf is None or buf.device != dev or buf.numel() != 2 * n_groups:
        buf = holder._norms_buf = torch.zeros(2 * n_groups, dtype=torch.float32, device=dev)
        holder.last_norms = buf[:n_groups]
    return buf


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, memory, hn, cn, tf_input, dropmask, opts, *params):
        lib = _lib.load()
        _check_dev(memory, hn, cn, dropmask)
        if not tf_input.is_cuda:
            raise _lib.MuconHipError("mucon_amd ops need device tensors: there is no CPU fallback")
        memory, hn, cn = memory.contiguous(), hn.contiguous(), cn.contiguous()
        params = [w if w.is_contiguous() else w.contiguous() for w in params]
        tf_input = tf_input.contiguous().to(torch.int64)
        if dropmask is not None:
            dropmask = dropmask.contiguous()
        max_steps, teacher_forcing, stop_on_eos, eos = opts
        Tz, ME = memory.shape
        cfg = _lib.DecoderCfg(Tz=Tz, ME=ME, D=params[7].shape[0], NC=params[17].shape[0], n_emb=params[8].shape[0],
                              max_steps=max_steps, teacher_forcing=int(teacher_forcing), stop_on_eos=int(stop_on_eos), eos=eos)
        need = max_steps if teacher_forcing else 1
        if tf_input.numel() < need:
            raise ValueError(f"decoder_forward: tf_input has {tf_input.numel()} tokens, {need} needed")
        nbytes = lib.mucon_decoder_workspace_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            _lib.check(_lib.E_ARG, "mucon_decoder_workspace_bytes")
        dev = memory.device
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        logp = torch.empty((max_steps, cfg.NC), dtype=torch.float32, device=dev)
        lengths = torch.empty(max_steps, dtype=torch.float32, device=dev)
        n_steps = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(lib.mucon_decoder_fwd(ctypes.byref(cfg), ctypes.byref(_param_struct(("dec",), params, _decoder_params)), _lib.ptr(memory), _lib.ptr(hn),
                                         _lib.ptr(cn), _lib.ptr(tf_input), _lib.ptr(dropmask), _lib.ptr(logp), _lib.ptr(lengths),
                                         _lib.ptr(n_steps), _lib.ptr(ws), nbytes, _lib.current_stream_ptr()),
                   "mucon_decoder_fwd")
        n = int(n_steps.item()) if stop_on_eos else max_steps
        if n < 0:
            raise _lib.MuconHipError(DECODER_HANDOVER_FAILED)
        if not stop_on_eos:
            # a teacher-forced call does not wait for the device: its status word is read at the caller's next synchronisation point
            # (check_health); a failed launch has filled logp / lengths with NaN meanwhile, which the fused optimizer step refuses to apply
            _PENDING_DECODER_STATUS.append(n_steps)
            if len(_PENDING_DECODER_STATUS) > 4096:      # nobody is calling check_health: do not grow without bound
                check_health()
        logp, lengths = logp[:n], lengths[:n]
        ctx.cfg, ctx.n, ctx.ws, ctx.nbytes, ctx.dropmask = cfg, n, ws, nbytes, dropmask
        ctx.save_for_backward(memory, hn, cn, logp, *params)
        ctx.set_materialize_grads(False)
        return logp, lengths

    @staticmethod
    def backward(ctx, d_logp, d_len):
        lib = _lib.load()
        memory, hn, cn, logp, *params = ctx.saved_tensors
        d_logp = d_logp.contiguous() if d_logp is not None else None
        d_len = d_len.contiguous() if d_len is not None else None
        d_memory, d_hn, d_cn = torch.empty_like(memory), torch.empty_like(hn), torch.empty_like(cn)
        grads, cg = _param_grads(ctx, ("dec",), params, _decoder_params)
        if getattr(ctx, "defer_outer", False):
            # (r6) the fused step path, where the LSTM's backward follows on this stream before anyone reads the decoder's weight gradients: their outer
            # products ride in that pass's recurrence launch (include/mucon_hip.h: mucon_decoder_bwd_defer).  ctx keeps the workspace alive.
            _lib.check(lib.mucon_decoder_bwd_defer(1), "mucon_decoder_bwd_defer")
        _lib.check(lib.mucon_decoder_bwd(ctypes.byref(ctx.cfg), ctx.n, ctypes.byref(_param_struct(("dec",), params, _decoder_params)), _lib.ptr(memory),
                                         _lib.ptr(hn), _lib.ptr(cn), _lib.ptr(logp), _lib.ptr(d_logp), _lib.ptr(d_len),
                                         _lib.ptr(ctx.dropmask), _lib.ptr(d_memory), _lib.ptr(d_hn), _lib.ptr(d_cn),
                                         ctypes.byref(cg), _lib.ptr(ctx.ws), ctx.nbytes,
                                         _lib.current_stream_ptr()), "mucon_decoder_bwd")
        return (d_memory, d_hn, d_cn, None, None, None, *grads)


def decoder_forward(memory: torch.Tensor, h_n: torch.Tensor, c_n: torch.Tensor, tf_input: torch.Tensor,
                    params: Sequence[torch.Tensor], max_steps: int, teacher_forcing: bool, stop_on_eos: bool, eos: int,
                    dropmask: Optional[torch.Tensor] = None):
    """The s-head's decoding loop (reference models.py:612-744) in one persistent kernel.

    memory [Tz, 2E] (encoder LSTM output), h_n / c_n [ndir, E]; tf_input int64 tokens; params = the 23 tensors
    named by _lib.DECODER_PARAM_FIELDS in that order; dropmask [max_steps, 128] = the embedding-dropout keep mask
    already divided by (1 - p), or None.  Returns (logp [n, NC], lengths [n]) with n = max_steps, or the number of
    steps up to and including the first EOS arg-max when stop_on_eos."""
    if len(params) != len(_lib.DECODER_PARAM_FIELDS):
        raise ValueError(f"decoder_forward: expected {len(_lib.DECODER_PARAM_FIELDS)} parameter tensors, got {len(params)}")
    return _DecoderFn.apply(memory, h_n.reshape(-1), c_n.reshape(-1), tf_input, dropmask,
                            (int(max_steps), bool(teacher_forcing), bool(stop_on_eos), int(eos)), *params)


@torch.no_grad()
def decoder_forward_deferred(memory: torch.Tensor, h_n: torch.Tensor, c_n: torch.Tensor, tf_input: torch.Tensor,
                             params: Sequence[torch.Tensor], max_steps: int, eos: int):
    """Greedy decoding with the EOS stop (evaluation: no teacher forcing, no dropout) WITHOUT the host round trip: the same
    launch as decoder_forward(..., teacher_forcing=False, stop_on_eos=True), but the step count stays on the device.
    Returns (logp [max_steps, NC], lengths [max_steps], n_steps int32 [1]); rows from n_steps on are unwritten.  A batched
    evaluation fetches the step counts of many videos in one copy (mucon_amd/mucon/evaluators.py)."""
    lib = _lib.load()
    if len(params) != len(_lib.DECODER_PARAM_FIELDS):
        raise ValueError(f"decoder_forward_deferred: expected {len(_lib.DECODER_PARAM_FIELDS)} parameter tensors, got {len(params)}")
    memory, hn, cn = memory.contiguous(), h_n.reshape(-1).contiguous(), c_n.reshape(-1).contiguous()
    params = [w.contiguous() for w in params]
    _check_dev(memory, hn, cn, *params)
    if not tf_input.is_cuda:
        raise _lib.MuconHipError("mucon_amd ops need device tensors: there is no CPU fallback")
    tf_input = tf_input.contiguous().to(torch.int64)
    if tf_input.numel() < 1:
        raise ValueError("decoder_forward_deferred: tf_input needs the start token")
    Tz, ME = memory.shape
    cfg = _lib.DecoderCfg(Tz=Tz, ME=ME, D=params[7].shape[0], NC=params[17].shape[0], n_emb=params[8].shape[0],
                          max_steps=int(max_steps), teacher_forcing=0, stop_on_eos=1, eos=int(eos))
    nbytes = lib.mucon_decoder_workspace_bytes(ctypes.byref(cfg))
    if nbytes == 0:
        _lib.check(_lib.E_ARG, "mucon_decoder_workspace_bytes")
    dev = memory.device
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    logp = torch.empty((int(max_steps), cfg.NC), dtype=torch.float32, device=dev)
    lengths = torch.empty(int(max_steps), dtype=torch.float32, device=dev)
    n_steps = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(lib.mucon_decoder_fwd(ctypes.byref(cfg), ctypes.byref(_decoder_params(params)), _lib.ptr(memory), _lib.ptr(hn),
                                     _lib.ptr(cn), _lib.ptr(tf_input), None, _lib.ptr(logp), _lib.ptr(lengths),
                                     _lib.ptr(n_steps), _lib.ptr(ws), nbytes, _lib.current_stream_ptr()), "mucon_decoder_fwd")
    return logp, lengths, n_steps


# --------------------------------------------------------------------------------------- losses
@dataclass
class LossSpec:
    """cfg.model.loss.* as the fused loss kernels take it (include/mucon_hip.h: mucon_loss_cfg)."""
    mucon_type: str = "flint"
    overlap: float = 0.0
    smoothing_clamp: bool = True
    clamp_min: float = 0.0
    clamp_max: float = 16.0
    length_width: float = 2.0
    transcript_average: bool = False
    mul_transcript: float = 1.0
    mul_length: float = 0.1
    mul_mucon: float = 1.0
    mul_smoothing: float = 0.1
    align_corners: bool = True   # cfg.model.loss.mucon.align_corners: the affine_grid / grid_sample convention of the masks


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, sx, tlogp, lengths, spec, mtarget, ttarget, tmpl, mweight, tweight):
        lib = _lib.load()
        _check_dev(seg, sx, tlogp, lengths, tmpl, mweight, tweight)
        for t in (mtarget, ttarget):
            if not t.is_cuda or t.dtype != torch.int64:
                raise _lib.MuconHipError("loss targets must be int64 device tensors")
        seg, sx, tlogp, lengths = seg.contiguous(), sx.contiguous(), tlogp.contiguous(), lengths.contiguous()
        mtarget, ttarget = mtarget.contiguous(), ttarget.contiguous()
        T, M = seg.shape
        S, NC = tlogp.shape
        N = lengths.shape[0]
        if mtarget.numel() != N or ttarget.numel() != S or sx.shape != seg.shape or tmpl.numel() != 100:
            raise ValueError("losses_forward: inconsistent shapes")
        types = {"flint": 0, "arithmetic": 1}
        if spec.mucon_type not in types:
            raise Exception(f"Invalid mucon type ({spec.mucon_type})")   # the reference's error (models.py:513-515)
        cfg = _lib.LossCfg(T=T, M=M, N=N, S=S, NC=NC, mucon_type=types[spec.mucon_type],
                           smoothing_clamp=int(spec.smoothing_clamp), transcript_average=int(spec.transcript_average),
                           overlap=spec.overlap, clamp_min=spec.clamp_min, clamp_max=spec.clamp_max,
                           length_width=spec.length_width, mul_transcript=spec.mul_transcript, mul_length=spec.mul_length,
                           mul_mucon=spec.mul_mucon, mul_smoothing=spec.mul_smoothing, align_corners=int(spec.align_corners))
        nbytes = lib.mucon_loss_workspace_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            _lib.check(_lib.E_ARG, "mucon_loss_workspace_bytes")
        dev = seg.device
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        d_seg, d_sx, d_tlogp = torch.empty_like(seg), torch.empty_like(sx), torch.empty_like(tlogp)
        d_len_full = torch.empty(N + 1, dtype=torch.float32, device=dev)   # entry N = 0: the gradient of the EOS step's length
        d_len = d_len_full[:N]
        ctx.d_len_full = d_len_full
        _lib.check(lib.mucon_loss_fwd_bwd(ctypes.byref(cfg), _lib.ptr(seg), _lib.ptr(sx), _lib.ptr(tlogp), _lib.ptr(lengths),
                                          _lib.ptr(mtarget), _lib.ptr(ttarget), _lib.ptr(tmpl.contiguous()),
                                          _lib.ptr(mweight), _lib.ptr(tweight), _lib.ptr(losses), _lib.ptr(d_seg), _lib.ptr(d_sx),
                                          _lib.ptr(d_tlogp), _lib.ptr(d_len_full), _lib.ptr(ws), nbytes, _lib.current_stream_ptr()),
                   "mucon_loss_fwd_bwd")
        ctx.save_for_backward(d_seg, d_sx, d_tlogp, d_len)
        parts = losses[1:]          # (a view: a clone is one more launch on a GPU-bound step)
        ctx.mark_non_differentiable(parts)
        return losses[0], parts

    @staticmethod
    def backward(ctx, g_main, _g_parts):
        d_seg, d_sx, d_tlogp, d_len = ctx.saved_tensors
        return (d_seg * g_main, d_sx * g_main, d_tlogp * g_main, d_len * g_main, None, None, None, None, None, None)


def losses_forward(segmentation: torch.Tensor, smoothing_input: torch.Tensor, transcript_logp: torch.Tensor,
                   lengths: torch.Tensor, spec: LossSpec, mucon_target: torch.Tensor, transcript_target: torch.Tensor,
                   mask_template: torch.Tensor, mucon_class_weight: Optional[torch.Tensor] = None,
                   transcript_class_weight: Optional[torch.Tensor] = None):
    """MuCon.loss for one video in three launches (reference models.py:376-565 + masks.py:8-74).

    segmentation [T, M] logits; smoothing_input [T, M] (log-probs or the logits, per smoothing.log_softmax_before);
    transcript_logp [S, M+1]; lengths [N]; targets int64.  Returns (main, parts) with parts = [transcript, length, mucon,
    smoothing] (detached); main.backward() delivers the gradients computed in the same launches."""
    return _LossFn.apply(segmentation, smoothing_input, transcript_logp, lengths, spec, mucon_target, transcript_target,
                         mask_template, mucon_class_weight, transcript_class_weight)


# --------------------------------------------------------------------------------------- clip + SGD
class FusedClipSGD:
    """clip_grad_norm_ per parameter group + torch.optim.SGD.step() in two launches (reference trainers.py:137-140).

    groups: lists of parameters, one list per clipping group (the reference's model.encode_params /
    model.decode_params); max_norm: one value for all groups, or None for no clipping.  lr / weight_decay / momentum are
    read from `optimizer.param_groups[0]` at every step, so schedulers keep working.  Parameters whose .grad is None are
    skipped, as torch does."""

    def __init__(self, groups, max_norm, optimizer):
        self.groups, self.max_norm, self.optimizer = [list(g) for g in groups], max_norm, optimizer
        if not 1 <= len(self.groups) <= 8:
            raise ValueError("FusedClipSGD: 1..8 clipping groups")
        self._mom = {}
        self._ws = None
        self.last_norms = None

    def clip_only(self):
        """An iteration of a gradient-accumulation group that does not step: the clipping alone."""
        fused_clip_only(self.groups, self.max_norm, self)

    def step(self, _retry=False):
        lib = _lib.load()
        pg = self.optimizer.param_groups[0]
        lr, wd, mom = float(pg["lr"]), float(pg["weight_decay"]), float(pg["momentum"])
        # The table of (parameter, gradient, momentum buffer) records is kept between steps: which parameters have a gradient, their sizes and
        # groups rarely change; per step only the addresses are refreshed (a training step of one video is bound by this host code as much as by
        # the GPU: tools/e2e_host_vs_gpu.py).  A parameter's dtype / device / layout is checked when the table is built; its gradient has the
        # parameter's dtype and device by torch's own rules, its contiguity is checked every step.
        flat = self.__dict__.get("_flat")
        if flat is None:
            flat = self._flat = [(p, gi) for gi, params in enumerate(self.groups) for p in params]
        grads = [p.grad for p, _ in flat]
        have = tuple(g is not None for g in grads)
        plan = self.__dict__.get("_plan")
        if plan is None or plan[0] != have or plan[1] != (mom != 0.0):
            idx = [i for i, h in enumerate(have) if h]
            if not idx:
                return
            tab = (_lib.SgdTensor * len(idx))()
            total, bufs = 0, []
            for k, i in enumerate(idx):
                p, gi = flat[i]
                _check_dev(p, grads[i])
                if not p.is_contiguous():
                    raise _lib.MuconHipError("FusedClipSGD needs contiguous parameters and gradients")
                buf = None
                if mom != 0.0:
                    buf = self._mom.get(id(p))
                    if buf is None:
                        buf = self._mom[id(p)] = torch.zeros_like(p)   # buf = 0 -> first step gives buf = grad, as torch
                bufs.append(buf)
                tab[k].n, tab[k].group = p.numel(), gi
                tab[k].momentum_buf = buf.data_ptr() if buf is not None else None
                total += p.numel()
            plan = self._plan = (have, mom != 0.0, idx, tab, total, bufs)
        _, _, idx, tab, total, bufs = plan
        if not idx:
            return
        last = self.__dict__.get("_last_grads")
        if last is None or len(last) != len(grads):
            last = [None] * len(grads)
        self._last_grads = grads          # (held until the next step: an identity below means the very tensor the table was refreshed for)
        for k, i in enumerate(idx):
            g, p_ = grads[i], flat[i][0]
            t = tab[k]
            if g is last[i] and t.param == p_.data_ptr():
                continue                  # (r6) the gradient tensor of the previous step again (the encoder's cached views) and the parameter where it was: the record stands
            if not g.is_contiguous() or not p_.is_contiguous():
                raise _lib.MuconHipError("FusedClipSGD needs contiguous parameters and gradients")
            if t.n != p_.numel() or g.numel() != t.n or g.dtype != torch.float32 or p_.dtype != torch.float32:
                # a parameter's storage was replaced by one of another size / type: the cached table is stale -- and so is every momentum buffer
                # that no longer matches its parameter (id(p) is unchanged: the table would point the kernel at the old, smaller buffer)
                if _retry:
                    raise _lib.MuconHipError("FusedClipSGD: parameter / gradient sizes do not match after rebuilding the table")
                self._plan = None
                for q, _ in flat:
                    b = self._mom.get(id(q))
                    if b is not None and (b.shape != q.shape or b.dtype != q.dtype or b.device != q.device):
                        del self._mom[id(q)]
                return self.step(_retry=True)
            t.param, t.grad = p_.data_ptr(), g.data_ptr()
        n = len(idx)
        nbytes = lib.mucon_sgd_workspace_bytes(n, total)
        dev = flat[idx[0]][0].device
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        norms = _norms_for(self, len(self.groups), dev)
        mx = (ctypes.c_float * len(self.groups))(*[float(self.max_norm) if self.max_norm is not None else 0.0] * len(self.groups))
        _lib.check(lib.mucon_sgd_clip_step(n, tab, len(self.groups), mx, lr, wd, mom, _lib.ptr(norms),
                                           _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream_ptr()),
                   "mucon_sgd_clip_step")
        # this call stands in for optimizer.step(): tell torch's schedulers so (they warn about the call order otherwise)
        if isinstance(self.optimizer, torch.optim.Optimizer):
            self.optimizer._opt_called = True


def fused_clip_only(groups, max_norm, holder):
    """clip_grad_norm_ per parameter group without an optimizer step (mucon_clip_grads): the non-stepping iterations of a
    gradient-accumulation group.  `holder` keeps the workspace / norms tensors (a FusedClipSGD or FusedClipAdam)."""
    if max_norm is None:
        return
    lib = _lib.load()
    entries = [(p, gi) for gi, params in enumerate(groups) for p in params if p.grad is not None]
    if not entries:
        return
    n = len(entries)
    tab = (_lib.SgdTensor * n)()
    total = 0
    for i, (p, gi) in enumerate(entries):
        _check_dev(p, p.grad)
        if not p.grad.is_contiguous():
            raise _lib.MuconHipError("fused clipping needs contiguous gradients")
        tab[i].param, tab[i].grad, tab[i].momentum_buf = p.data_ptr(), p.grad.data_ptr(), None
        tab[i].n, tab[i].group = p.numel(), gi
        total += p.numel()
    nbytes = lib.mucon_sgd_workspace_bytes(n, total)
    dev = entries[0][0].device
    if getattr(holder, "_clip_ws", None) is None or holder._clip_ws.numel() < nbytes or holder._clip_ws.device != dev:
        holder._clip_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    norms = _norms_for(holder, len(groups), dev)
    mx = (ctypes.c_float * len(groups))(*[float(max_norm)] * len(groups))
    _lib.check(lib.mucon_clip_grads(n, tab, len(groups), mx, _lib.ptr(norms), _lib.ptr(holder._clip_ws),
                                    holder._clip_ws.numel(), _lib.current_stream_ptr()), "mucon_clip_grads")


class FusedClipAdam:
    """clip_grad_norm_ per parameter group + torch.optim.Adam.step() (optionally AMSGrad: the reference builds Adam with
    amsgrad=True, trainers.py:31-36) in two launches.  The moment buffers and the step counter are the torch optimizer's own
    `state` entries (created here on first use, with torch's keys), so `optimizer.state_dict()` checkpoints stay interchangeable;
    lr / betas / eps / weight_decay are read from `param_groups[0]` at every step."""

    def __init__(self, groups, max_norm, optimizer):
        self.groups, self.max_norm, self.optimizer = [list(g) for g in groups], max_norm, optimizer
        if not 1 <= len(self.groups) <= 8:
            raise ValueError("FusedClipAdam: 1..8 clipping groups")
        self._ws = None
        self.last_norms = None

    def clip_only(self):
        """An iteration of a gradient-accumulation group that does not step: the clipping alone."""
        fused_clip_only(self.groups, self.max_norm, self)

    def step(self):
        lib = _lib.load()
        pg = self.optimizer.param_groups[0]
        amsgrad = bool(pg.get("amsgrad", False))
        entries = []
        for gi, params in enumerate(self.groups):
            for p in params:
                if p.grad is None:
                    continue
                _check_dev(p, p.grad)
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.MuconHipError("FusedClipAdam needs contiguous parameters and gradients")
                entries.append((p, self.optimizer.state[p], gi))
        if not entries:
            return
        # validate BEFORE touching any state: torch keeps one step counter per tensor, this kernel takes one per call.  A
        # parameter whose first gradient arrives later than the others' would make the counters differ -- refuse with the
        # optimizer state untouched, so the caller can take torch's own optimizer.step() for this iteration
        counts = {int(st["step"]) if len(st) else 0 for _, st, _ in entries}
        if len(counts) != 1:
            raise _lib.MuconHipError("FusedClipAdam: parameters at different step counts (torch keeps one per tensor; this kernel one "
                                     "per call); the optimizer state was not modified")
        for p, st, _ in entries:
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if amsgrad:
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
        n = len(entries)
        tab = (_lib.AdamTensor * n)()
        total = 0
        for i, (p, st, gi) in enumerate(entries):
            tab[i].param, tab[i].grad = p.data_ptr(), p.grad.data_ptr()
            tab[i].exp_avg, tab[i].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            tab[i].max_exp_avg_sq = st["max_exp_avg_sq"].data_ptr() if amsgrad else None
            tab[i].n, tab[i].group = p.numel(), gi
            total += p.numel()
        steps = {counts.pop() + 1}
        nbytes = lib.mucon_adam_workspace_bytes(n, total)
        dev = entries[0][0].device
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        norms = _norms_for(self, len(self.groups), dev)
        mx = (ctypes.c_float * len(self.groups))(*[float(self.max_norm) if self.max_norm is not None else 0.0] * len(self.groups))
        b1, b2 = pg["betas"]
        _lib.check(lib.mucon_adam_clip_step(n, tab, len(self.groups), mx, float(pg["lr"]), float(b1), float(b2), float(pg["eps"]),
                                            float(pg["weight_decay"]), steps.pop(), _lib.ptr(norms), _lib.ptr(self._ws),
                                            self._ws.numel(), _lib.current_stream_ptr()), "mucon_adam_clip_step")
        if isinstance(self.optimizer, torch.optim.Optimizer):
            self.optimizer._opt_called = True


# --------------------------------------------------------------------------------------- graph-free execution
class PlainCtx:
    """Stand-in for torch.autograd's context object: lets the Functions above run their forward / backward as plain
    calls, without building an autograd graph (MuCon.fused_train_step: the training step as one straight line of
    launches -- the graph walk and its per-node Python round trips cost as much host time as the GPU needs for the step)."""
    saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def set_materialize_grads(self, flag):
        pass

    def mark_non_differentiable(self, *ts):
        pass


def run_forward(fn, *args):
    """(outputs, ctx) of an autograd Function's forward, called directly."""
    ctx = PlainCtx()
    with torch.no_grad():
        return fn.forward(ctx, *args), ctx


def run_backward(fn, ctx, *grads):
    with torch.no_grad():
        return fn.backward(ctx, *grads)


# --------------------------------------------------------------------------------------- viterbi
_VITERBI_VIDEO_DTYPE = np.dtype([("lp", np.uint64), ("transcript", np.uint64), ("table", np.uint64), ("T", np.int32), ("N", np.int32),
                                 ("force_n", np.int32), ("force_j", np.int32)])   # = _lib.ViterbiVideo = mucon_viterbi_video
_VITERBI_JOB_DTYPE = np.dtype([("lp", np.uint64), ("tr_off", np.int64), ("p_off", np.int64), ("label_off", np.int64), ("seg_off", np.int64),
                               ("ws_off", np.int64), ("T", np.int32), ("N", np.int32), ("force_n", np.int32), ("force_j", np.int32)])   # = mucon_viterbi_job
_VIT_LABEL_FORMATS = {"int32": _lib.VIT_LABELS_I32, "uint8": _lib.VIT_LABELS_U8, "lazy": _lib.VIT_LABELS_NONE}
_EMPTY_I32 = np.empty(0, np.int32)


def expand_labels(transcript: np.ndarray, seg_len: np.ndarray, n_seg: int, T: int, fs: int) -> np.ndarray:
    """Per-frame labels int32 [T] of a decoded segmentation -- the reference's traceback (src/core/viterbi/viterbi.py:140-158):
    the `T - K*fs` leftover frames carry the LAST segment's label and sit at the START of the video (their count is included in
    the last segment's length, :154-157), then every segment's label `length` times.  The same expansion the kernels perform
    for the int32 / uint8 label formats (csrc/viterbi.hip: vit_traceback_and_labels)."""
    if n_seg <= 0:
        return _EMPTY_I32
    missing = T - (T // fs) * fs
    lens = np.array(seg_len[:n_seg], dtype=np.int64)
    lens[-1] -= missing
    tr = np.asarray(transcript[:n_seg], dtype=np.int32)
    body = np.repeat(tr, lens)
    rest = T - missing - body.shape[0]
    if missing == 0 and rest == 0:
        return body
    last = tr[n_seg - 1]
    return np.concatenate((np.full(missing, last, np.int32), body, np.full(max(rest, 0), last, np.int32)))


class ViterbiResult:
    """One video's decode: score (np.float64), seg_len int32 [n_seg], n_seg, status (MUCON_VIT_*) and `labels` int32 [T].
    With the "lazy" label format the kernels write no per-frame labels at all; `labels` is then expanded from the segmentation
    on first access (expand_labels) and cached."""
    __slots__ = ("score", "seg_len", "n_seg", "status", "_labels", "_lazy")

    def __init__(self, score, labels, seg_len, n_seg, status, lazy=None):
        self.score, self.seg_len, self.n_seg, self.status = score, seg_len, n_seg, status
        self._labels, self._lazy = labels, lazy          # lazy = (transcript, T, fs) when the labels were not written

    @property
    def labels(self) -> np.ndarray:
        if self._labels is None:
            tr, T, fs = self._lazy
            self._labels = expand_labels(tr, self.seg_len, self.n_seg, T, fs)
        elif self._labels.dtype != np.int32:              # the uint8 format: widened on first access
            self._labels = self._labels.astype(np.int32)
        return self._labels

    @property
    def labels_raw(self) -> Optional[np.ndarray]:
        """The array the kernels wrote (int32 or uint8 [T]), None with the lazy format."""
        return self._labels

    def __repr__(self):
        return f"ViterbiResult(score={self.score!r}, n_seg={self.n_seg}, status={self.status})"


def viterbi_decode_beam(lps: Sequence[torch.Tensor], transcripts: Sequence, tables: Sequence[np.ndarray], fs: int, max_len: int,
                        max_hypotheses: int) -> List["ViterbiResult"]:
    """Viterbi.decode under the reference's beam (Viterbi(max_hypotheses=M), reference viterbi.py:34, :74-79) for M below the
    N * (max_len // fs) hypotheses a transcript can have alive: mucon_viterbi_decode_beam (csrc/viterbi_beam.hip), bit for bit.
    lps: device tensors [T x C] float32; transcripts: int sequences; tables: float64 [J x N].  -> a ViterbiResult per video (labels
    expanded from the segments on access).  A beam that lost every path into the last transcript state yields, as in the reference,
    score -inf and fewer segments than the transcript has entries (status VIT_TRUNCATED)."""
    lib = _lib.load()
    nv = len(lps)
    if nv == 0:
        return []
    J = int(max_len) // int(fs)
    vids = (_lib.ViterbiVideo * nv)()
    keep, trs = [], []
    C = int(lps[0].shape[1])
    for v, (lp, tr, P) in enumerate(zip(lps, transcripts, tables)):
        if not isinstance(lp, torch.Tensor) or not lp.is_cuda or lp.dtype != torch.float32:
            raise _lib.MuconHipError("mucon_amd ops need float32 device tensors: there is no CPU fallback")
        lp = _aligned_lp(lp)
        tr = np.ascontiguousarray(tr, dtype=np.int32)
        P = np.ascontiguousarray(P, dtype=np.float64)
        if int(lp.shape[1]) != C or P.shape != (J, len(tr)):
            raise ValueError(f"viterbi_decode_beam: video {v}: emissions [T x {C}] and a length table [{J} x {len(tr)}] expected")
        keep += [lp, tr, P]
        trs.append(tr)
        q = vids[v]
        q.lp, q.transcript, q.table, q.T, q.N, q.force_n, q.force_j = lp.data_ptr(), tr.ctypes.data, P.ctypes.data, int(lp.shape[0]), len(tr), -1, -1
    sum_n = sum(len(t) for t in trs)
    score, n_seg, status = np.empty(nv, np.float64), np.empty(nv, np.int32), np.empty(nv, np.int32)
    seg_len = np.zeros(max(sum_n, 1), np.int32)
    _lib.check(lib.mucon_viterbi_decode_beam(nv, vids, C, int(fs), int(max_len), int(max_hypotheses), score.ctypes.data, n_seg.ctypes.data,
                                             status.ctypes.data, seg_len.ctypes.data, _lib.current_stream_ptr()), "mucon_viterbi_decode_beam")
    out, off = [], 0
    for v in range(nv):
        n = len(trs[v])
        ns = int(n_seg[v])
        out.append(ViterbiResult(np.float64(score[v]), None, seg_len[off: off + ns].copy(), ns, int(status[v]), lazy=(trs[v], int(lps[v].shape[0]), int(fs))))
        off += n
    return out


def _aligned_lp(lp):
    if not lp.is_contiguous() or (lp.data_ptr() & 15):
        lp = lp.contiguous()
        if lp.data_ptr() & 15:                      # (a storage-offset view whose start is not 16-byte aligned)
            lp = lp.clone(memory_format=torch.contiguous_format)
    return lp


class ViterbiBatchResult(collections.abc.Sequence):
    """What viterbi_decode_batch returns: a sequence of ViterbiResult, one per video, over the call's flat result arrays
    (score float64 [nv], n_seg / status int32 [nv], seg_len int32 [sum N], labels [sum max(T, 1)] or None).  The per-video
    objects are views made on first access: a caller that reads the flat arrays (or a few videos) does not pay 256 object
    constructions for a call whose decode is shorter than they are."""
    __slots__ = ("score", "n_seg", "status", "seg_len", "labels_flat", "label_off", "seg_off", "_T", "_tr", "_fs", "_items", "_keep", "_lists")

    def __init__(self, score, n_seg, status, seg_len, labels_flat, label_off, seg_off, Ts, transcripts, fs, keep):
        self.score, self.n_seg, self.status, self.seg_len, self.labels_flat = score, n_seg, status, seg_len, labels_flat
        self.label_off, self.seg_off, self._T, self._tr, self._fs = label_off, seg_off, Ts, transcripts, fs
        self._items = [None] * len(Ts)
        self._keep = keep
        self._lists = None

    def __len__(self):
        return len(self._items)

    def __getitem__(self, v):
        if isinstance(v, slice):
            return [self[i] for i in range(*v.indices(len(self._items)))]
        r = self._items[v]
        if r is None:
            if self._lists is None:       # the index arrays as Python lists, once (256 x int(numpy scalar) cost more than the rest)
                self._lists = (self.status.tolist(), self.n_seg.tolist(), self.seg_off.tolist(),
                               self.label_off.tolist() if self.labels_flat is not None else None)
            sts, nss, sos, los = self._lists
            st, ns, so = sts[v], nss[v], sos[v]
            if st != _lib.VIT_OK and st != _lib.VIT_TRUNCATED:      # the error branches write status / n_seg / score only
                lab, lazy = _EMPTY_I32, None
            elif los is None:
                lab, lazy = None, (self._tr[v], self._T[v], self._fs)
            else:
                lab, lazy = self.labels_flat[los[v]: los[v] + self._T[v]], None
            r = self._items[v] = ViterbiResult(self.score[v], lab, self.seg_len[so: so + ns], ns, st, lazy)
        return r

    def __iter__(self):
        for v in range(len(self._items)):
            yield self[v]


def viterbi_decode_batch(lps: Sequence[torch.Tensor], transcripts: Sequence[np.ndarray],
                         tables: Sequence[np.ndarray], fs: int, max_len: int,
                         forces: Optional[Sequence[Optional[tuple]]] = None, labels: str = "lazy",
                         log_fact: Optional[np.ndarray] = None) -> ViterbiBatchResult:
    """Decode a batch of videos (one workgroup per video) through mucon_viterbi_decode_host.

    lps[v]: device float32 [T_v, C] log-probs (they stay where they are: every video is decoded in place, nothing is
    concatenated or copied); transcripts[v]: int [N_v]; tables[v]: float64 [J, N_v] with J = max_len // fs;
    forces[v]: None or (n, j) -- finalize on that hypothesis with score -inf (host-resolved degenerate outcomes of the
    reference).  The library reads the small inputs from and writes the results to its own pinned host buffers; a single
    short video is one launch whose completion the host sees through a flag (no copy calls, no stream synchronisation).

    log_fact (float64 [J]): the PoissonModel's length scores are BUILT ON THE DEVICE (mucon_viterbi_decode_host_poisson, ABI 7); tables[v] is
    then the video's float64 [3, N_v] parameter block (core/viterbi/length_model.py: PoissonParams.params, .log_fact) -- 3 doubles per
    transcript state over PCIe instead of J.

    labels: the form the per-frame labels leave the GPU in (include/mucon_hip.h, MUCON_VIT_LABELS_*): "int32" (the reference's
    ints, 4 T bytes per video over PCIe), "uint8" (T bytes) or "lazy" (default: none -- the kernels write the segmentation only and
    ViterbiResult.labels expands it on first access).  ViterbiResult.labels is an int32 array [T] whichever form was asked for.

    Returns a sequence of ViterbiResult (ViterbiBatchResult: per-video views over the flat result arrays, made on access).
    The per-video argument handling is a C loop (csrc/pyhost.c through ctypes.PyDLL): int32 / float64 C-contiguous NumPy arrays
    are pointed at where they lie, anything else (lists, other dtypes, strided views) is converted first."""
    lib = _lib.load()
    nv = len(lps)
    if nv == 0:
        return []
    fmt = _VIT_LABEL_FORMATS[labels]
    C = lps[0].shape[1]
    lps = list(lps)
    for v, lp in enumerate(lps):
        if not lp.is_cuda or lp.dtype != torch.float32:
            _check_dev(lp)
        if lp.shape[1] != C:
            raise ValueError(f"video {v}: {lp.shape[1]} classes (expected {C})")
        if not lp.is_contiguous():
            lps[v] = lp.contiguous()
    if (C & 3) == 0:
        # the pipelined frame-score kernel loads 16 bytes per lane: an emission tensor that does not start on a 16-byte boundary (a slice, a view) is
        # copied here, once, for the whole list -- the C loop reports ONE offending video per crossing (a batch of n such views cost n crossings)
        for v, lp in enumerate(lps):
            if lp.data_ptr() & 15:
                lps[v] = lp.clone(memory_format=torch.contiguous_format)
    ptrs = [lp.data_ptr() for lp in lps]
    Ts = [lp.shape[0] for lp in lps]
    trs, tabs = list(transcripts), list(tables)
    if forces is not None:
        forces = list(forces)
    lab_arr, lab_ptr = None, 0
    if fmt != _lib.VIT_LABELS_NONE and nv >= _lib.VIT_LATENCY_VIDEOS:
        # From MUCON_VIT_LATENCY_VIDEOS videos on the library writes a PINNED labels array in place (no staging copy): take it from
        # torch's caching host allocator; below that the results come through the library's own staging buffer anyway.
        lab_t = torch.empty(sum(T if T > 0 else 1 for T in Ts), dtype=torch.int32 if fmt == _lib.VIT_LABELS_I32 else torch.uint8, pin_memory=True)
        lab_arr, lab_ptr = lab_t.numpy(), lab_t.data_ptr()
    decode, J = _vit_entry(lib), max_len // fs
    fn_addr = _VIT_FN_ADDR[1 if log_fact is not None else 0]
    rows = 3 if log_fact is not None else J
    if log_fact is not None and not (isinstance(log_fact, np.ndarray) and log_fact.dtype == np.float64 and log_fact.shape == (J,) and log_fact.flags.c_contiguous):
        log_fact = np.ascontiguousarray(log_fact, dtype=np.float64).reshape(J)
    # Inputs that are not already what the C loop reads in place (1-d int32 transcripts, C-contiguous float64 [J x N] tables) are converted in ONE
    # pass here: the C loop reports one offending video per call, so a list of 256 int64 transcripts used to cost 256 extra crossings.
    for v in range(nv):
        t = trs[v]
        if not (isinstance(t, np.ndarray) and t.dtype == np.int32 and t.ndim == 1 and t.flags.c_contiguous):
            trs[v] = np.ascontiguousarray(t, dtype=np.int32).reshape(-1)
        t = tabs[v]
        if not (isinstance(t, np.ndarray) and t.dtype == np.float64 and t.ndim == 2 and t.flags.c_contiguous):
            tabs[v] = np.ascontiguousarray(t, dtype=np.float64)
        if tabs[v].shape != (rows, trs[v].shape[0]):
            raise ValueError(f"video {v}: length table {tabs[v].shape} (expected {(rows, trs[v].shape[0])})")
    start, last_bad = 0, -1
    while True:
        rc, bad, sum_T, sum_N, out = decode(ptrs, Ts, trs, tabs, forces, log_fact, C, fs, max_len, fmt, 1 if (C & 3) == 0 else 0, start, lab_ptr,
                                            fn_addr, _lib.current_stream_raw())
        if bad < 0:
            break
        # video `bad`: what is left after the pass above is an emission tensor whose start is not 16-byte aligned
        if bad == last_bad:      # the same video refused twice: nothing here can repair it -- never spin
            raise _lib.MuconHipError(f"viterbi_decode_batch: video {bad} is refused by the binding's C loop (transcript {trs[bad].dtype} "
                                     f"{trs[bad].shape}, table {tabs[bad].dtype} {tabs[bad].shape}, emissions at 0x{ptrs[bad]:x})")
        last_bad = bad
        if (C & 3) == 0 and (ptrs[bad] & 15):
            lps[bad] = lps[bad].clone(memory_format=torch.contiguous_format)
            ptrs[bad] = lps[bad].data_ptr()
        start = 0      # (every call rebuilds its records: the C side's scratch is not trusted across two crossings -- another Python thread may have used it)
    if rc != _lib.OK:
        _lib.check(rc, "mucon_viterbi_decode_host")
    # out: [score f64 nv][n_seg i32 nv][status i32 nv][seg_len i32 sum_N, padded][label offsets i64 nv + 1][segment offsets i64 nv + 1][labels]
    n_off = (16 * nv + 4 * sum_N + 7) // 8
    f64 = np.frombuffer(out, np.float64, n_off + 2 * nv + 2)
    i32 = f64.view(np.int32)
    i64 = f64.view(np.int64)
    if fmt != _lib.VIT_LABELS_NONE and lab_arr is None:
        lab_arr = np.frombuffer(out, np.int32 if fmt == _lib.VIT_LABELS_I32 else np.uint8, sum_T, 8 * (n_off + 2 * nv + 2))
    return ViterbiBatchResult(f64[:nv], i32[2 * nv: 3 * nv], i32[3 * nv: 4 * nv], i32[4 * nv: 4 * nv + sum_N], lab_arr,
                              i64[n_off: n_off + nv + 1], i64[n_off + nv + 1: n_off + 2 * nv + 2], Ts, trs, fs, (lps, tabs))


_VIT_FN_ADDR = [0, 0]


def _vit_entry(lib):
    """(pyhost's decode function; the addresses of mucon_viterbi_decode_host / _host_poisson are cached in _VIT_FN_ADDR)"""
    if not _VIT_FN_ADDR[0]:
        _VIT_FN_ADDR[0] = ctypes.cast(lib.mucon_viterbi_decode_host, ctypes.c_void_p).value
        _VIT_FN_ADDR[1] = ctypes.cast(lib.mucon_viterbi_decode_host_poisson, ctypes.c_void_p).value
    return _lib.pyhost().mucon_py_viterbi_decode


class ViterbiDeviceResult:
    """Results of viterbi_decode_batch_device: device tensors, valid once the stream has run the decode.
    labels: uint8 / int32 [sum max(T, 1)] (None with label_dtype=None), video v's at label_off[v]; seg_len int32 [sum N], video v's
    at seg_off[v]; n_seg int32 [nv]; score float64 [nv]; status int32 [nv]."""
    __slots__ = ("labels", "seg_len", "n_seg", "score", "status", "label_off", "seg_off", "T", "N", "_keep")


def viterbi_decode_batch_device(lps: Sequence[torch.Tensor], transcripts: Sequence[np.ndarray], tables: Sequence[np.ndarray],
                                fs: int, max_len: int, forces: Optional[Sequence[Optional[tuple]]] = None,
                                label_dtype: Optional[torch.dtype] = torch.uint8, log_fact: Optional[np.ndarray] = None) -> ViterbiDeviceResult:
    """The all-device decode, mucon_viterbi_decode_batch: nothing comes back to the host and the call does not synchronise --
    the job table, transcripts and length tables go up in ONE pinned copy on the current stream, the two launches (frame scores,
    DP) follow, the results stay in HBM for whatever consumes them there (device metrics, a later gather).  Arguments as
    viterbi_decode_batch (log_fact given: tables[v] is the [3, N] PoissonModel parameter block and the length scores are built on the device,
    mucon_viterbi_decode_batch_poisson); label_dtype: torch.uint8 (default), torch.int32 or None (segments only)."""
    lib = _lib.load()
    nv = len(lps)
    if nv == 0:
        raise ValueError("viterbi_decode_batch_device: no videos")
    _check_dev(*lps)
    dev = lps[0].device
    C, J = int(lps[0].shape[1]), max_len // fs
    fmt = {torch.uint8: _lib.VIT_LABELS_U8, torch.int32: _lib.VIT_LABELS_I32, None: _lib.VIT_LABELS_NONE}[label_dtype]
    lps = [_aligned_lp(lp) for lp in lps]
    trs = [np.ascontiguousarray(t, dtype=np.int32) for t in transcripts]
    tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in tables]
    Ts = np.array([lp.shape[0] for lp in lps], dtype=np.int64)
    Ns = np.array([t.shape[0] for t in trs], dtype=np.int64)
    R = 3 if log_fact is not None else J       # doubles per transcript state
    for v in range(nv):
        if tabs[v].shape != (R, Ns[v]) or lps[v].shape[1] != C:
            raise ValueError(f"video {v}: length table {tabs[v].shape} (expected {(R, int(Ns[v]))}) / {lps[v].shape[1]} classes (expected {C})")
    tr_off = np.concatenate(([0], np.cumsum(Ns)))
    lab_off = np.concatenate(([0], np.cumsum(np.maximum(Ts, 1))))
    ws_each = np.array([(lib.mucon_viterbi_job_workspace_bytes(int(T), C, int(N), fs) + 255) & ~255 for T, N in zip(Ts, Ns)], dtype=np.int64)
    ws_off = np.concatenate(([0], np.cumsum(ws_each)))
    jobs = np.zeros(nv, dtype=_VITERBI_JOB_DTYPE)
    jobs["lp"] = [lp.data_ptr() for lp in lps]
    jobs["tr_off"], jobs["p_off"], jobs["label_off"], jobs["seg_off"], jobs["ws_off"] = tr_off[:-1], tr_off[:-1] * R, lab_off[:-1], tr_off[:-1], ws_off[:-1]
    jobs["T"], jobs["N"] = Ts, Ns
    if forces is None:
        jobs["force_n"] = jobs["force_j"] = -1
    else:
        jobs["force_n"] = [int(f[0]) if f is not None else -1 for f in forces]
        jobs["force_j"] = [int(f[1]) if f is not None else -1 for f in forces]
    # one pinned staging tensor [jobs | length tables f64 | transcripts i32] -> one H2D copy
    sum_N = int(tr_off[-1])
    o_tab = (jobs.nbytes + 15) & ~15
    o_lf = o_tab + 8 * R * sum_N                   # (parameter form: the shared log-factorial row [J] behind the blocks)
    o_tr = o_lf + (8 * J if log_fact is not None else 0)
    total = (o_tr + 4 * sum_N + 15) & ~15
    stage = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    sn = stage.numpy()
    sn[:jobs.nbytes] = jobs.view(np.uint8)
    tab_v = sn[o_tab:o_lf].view(np.float64)
    tr_v = sn[o_tr:o_tr + 4 * sum_N].view(np.int32)
    if log_fact is not None:
        sn[o_lf:o_tr].view(np.float64)[:] = np.asarray(log_fact, dtype=np.float64).reshape(J)
    for v in range(nv):
        tab_v[tr_off[v] * R: tr_off[v + 1] * R] = tabs[v].reshape(-1)
        tr_v[tr_off[v]: tr_off[v + 1]] = trs[v]
    up = stage.to(dev, non_blocking=True)
    r = ViterbiDeviceResult()
    r.labels = torch.empty(int(lab_off[-1]), dtype=label_dtype, device=dev) if label_dtype is not None else None
    r.seg_len = torch.empty(sum_N, dtype=torch.int32, device=dev)
    r.n_seg = torch.empty(nv, dtype=torch.int32, device=dev)
    r.score = torch.empty(nv, dtype=torch.float64, device=dev)
    r.status = torch.empty(nv, dtype=torch.int32, device=dev)
    ws = torch.empty(int(ws_off[-1]) + 256, dtype=torch.uint8, device=dev)
    r.label_off, r.seg_off, r.T, r.N = lab_off, tr_off, Ts, Ns
    r._keep = (lps, up, ws, stage)
    base = up.data_ptr()
    if log_fact is not None:
        _lib.check(lib.mucon_viterbi_decode_batch_poisson(nv, base, C, fs, max_len, int(Ns.max()), base + o_tr, base + o_tab, base + o_lf,
                                                          _lib.ptr(r.labels), fmt, _lib.ptr(r.seg_len), _lib.ptr(r.n_seg), _lib.ptr(r.score),
                                                          _lib.ptr(r.status), _lib.ptr(ws), _lib.current_stream_ptr()),
                   "mucon_viterbi_decode_batch_poisson")
    else:
        _lib.check(lib.mucon_viterbi_decode_batch(nv, base, C, fs, max_len, int(Ns.max()), base + o_tr, base + o_tab,
                                                  _lib.ptr(r.labels), fmt, _lib.ptr(r.seg_len), _lib.ptr(r.n_seg), _lib.ptr(r.score),
                                                  _lib.ptr(r.status), _lib.ptr(ws), _lib.current_stream_ptr()),
                   "mucon_viterbi_decode_batch")
    return r
