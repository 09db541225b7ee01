"""Temporal encoder modules with the reference's names and state_dict keys
(src/core/modules/temporal.py:9-147): WaveNetBlock.{first_conv, l_0 .. l_{n-1}.{dilated_conv,
conv_1x1}, last_conv}.  The nn.Conv1d sub-modules only HOLD the parameters (same shapes, same
default initialisation as the reference, so reference checkpoints load); the arithmetic runs in
the gfx950 kernels behind mucon_amd.ops.encoder_forward.

The default encoder (cfg.model.ft.type == "wavenet") is the hand-written path.  The two non-default variants
(no shipped configuration selects them) keep the reference's names and state_dict keys: NoFt runs on
first_conv's kernels (mucon_linear_fwd / _bwd), MSTCNPPFirstStage on mucon_linear_fwd + the 128-channel convolution entry points
(mucon_conv128_fwd / _dgrad / _wgrad: the encoder's f32-MFMA kernels):
  NoFt               reference temporal.py:56-74    last_conv
  MSTCNPPFirstStage  reference temporal.py:150-204  conv_1x1_in, conv_dilated_1.{i}, conv_dilated_2.{i}, conv_fusion.{i}, conv_out"""
from typing import Iterable, List

import torch
import torch.nn as nn
from torch import Tensor

from ... import ops


class WaveNetLayer(nn.Module):
    """Parameter holder for one residual layer (reference temporal.py:9-53):
    y = x + Dropout(conv_1x1(act(dilated_conv(x)))), kernel 3, zero padding = dilation."""

    def __init__(self, num_channels: int, kernel_size: int, dilation: int, drop: float = 0.25, leaky: bool = False):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("the HIP encoder implements kernel_size=3 (the reference's only use)")
        self.num_channels, self.kernel_size, self.dilation, self.leaky = num_channels, kernel_size, dilation, leaky
        self.dilated_conv = nn.Conv1d(num_channels, num_channels, kernel_size, dilation=dilation, padding=dilation)
        self.conv_1x1 = nn.Conv1d(num_channels, num_channels, 1)
        self.drop = nn.Dropout(drop)

    def forward(self, x: Tensor) -> Tensor:
        raise RuntimeError("WaveNetLayer is fused into WaveNetBlock.forward on the HIP path; call the block")


class WaveNetBlock(nn.Module):
    def __init__(self, in_channels: int, stages: List[int] = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024),
                 out_dims: int = 64, kernel_size: int = 3, pooling=True, pooling_layers: Iterable[int] = (1, 2, 4, 8),
                 pooling_type: str = "max", dropout_rate=0.25, leaky=False):
        super().__init__()
        self.in_channels, self.stages, self.num_stages = in_channels, list(stages), len(stages)
        self.out_dims, self.kernel_size = out_dims, kernel_size
        self.pooling, self.pooling_type, self.pooling_layers = pooling, pooling_type, list(pooling_layers)
        self.dropout_rate, self.leaky = dropout_rate, leaky
        self.first_conv = nn.Conv1d(in_channels, out_dims, 1)
        self.last_conv = nn.Conv1d(out_dims, out_dims, 1)
        self.layers = []
        for i, stage in enumerate(self.stages):
            layer = WaveNetLayer(out_dims, kernel_size, stage, drop=dropout_rate, leaky=leaky)
            self.layers.append(layer)
            self.add_module("l_{}".format(i), layer)

    # ---------------------------------------------------------------------------------------
    def spec(self, last_gn=False, last_gn_num_groups=32, last_relu=False, last_dropout=False,
             last_dropout_rate=0.0) -> ops.EncoderSpec:
        return ops.EncoderSpec(in_dim=self.in_channels, hidden=self.out_dims, stages=list(self.stages),
                               pooling=bool(self.pooling), pooling_type=self.pooling_type,
                               pooling_layers=list(self.pooling_layers), leaky_relu=bool(self.leaky),
                               dropout_rate=float(self.dropout_rate), last_gn=last_gn,
                               last_gn_num_groups=last_gn_num_groups, last_relu=last_relu, last_dropout=last_dropout,
                               last_dropout_rate=last_dropout_rate)

    def ordered_parameters(self) -> List[Tensor]:
        """[first_w, first_b, (dil_w, dil_b, pw_w, pw_b) per layer, last_w, last_b] -- ops.param_names order."""
        # (cached: 50 attribute walks through nn.Module.__getattr__ per forward.  nn.Module keeps its Parameter OBJECTS across .to(),
        # load_state_dict() and optimizer steps; a caller that REPLACES one -- `layer.conv_1x1.weight = nn.Parameter(...)`, a pruning or
        # weight-norm re-parametrisation -- changes the module's _parameters dict, so every entry's identity is checked against the
        # dict it came from: 50 dict look-ups, still a fraction of the attribute walks)
        cache = self.__dict__.get("_ordered_params")
        if cache is not None:
            owners = self.__dict__["_ordered_param_owners"]
            if all(o[k] is t for (o, k), t in zip(owners, cache)):
                return cache
        mods = [self.first_conv] + [m for l in self.layers for m in (l.dilated_conv, l.conv_1x1)] + [self.last_conv]
        out, owners = [], []
        for m in mods:
            out += [m.weight, m.bias]
            owners += [(m._parameters, "weight"), (m._parameters, "bias")]
        self.__dict__["_ordered_params"] = out
        self.__dict__["_ordered_param_owners"] = owners
        return out

    def forward_time_major(self, tape: Tensor, gn_weight: Tensor, gn_bias: Tensor, spec: ops.EncoderSpec,
                           seed: int = 0) -> Tensor:
        """tape [B, T, Cin] (row-major, as the dataset delivers it) -> [B, Tz, out_dims].  Device tensors run in the HIP kernels;
        host tensors (cfg.system.device = "cpu", reference core/config.py:16) take the library-op plumbing path."""
        if not tape.is_cuda:
            from ... import cpu_plumbing
            return cpu_plumbing.wavenet_forward(self, tape, gn_weight, gn_bias, spec)
        if self.out_dims == 128:
            return ops.encoder_forward(tape, self.ordered_parameters() + [gn_weight, gn_bias], spec,
                                       training=self.training, seed=seed)
        if self.out_dims > 128:
            raise NotImplementedError("the HIP encoder is built for hidden sizes up to 128")
        return self._forward_padded(tape, gn_weight, gn_bias, spec, seed)

    def _forward_padded(self, tape: Tensor, gn_weight: Tensor, gn_bias: Tensor, spec: ops.EncoderSpec, seed: int) -> Tensor:
        """Hidden sizes below 128 (the constructor default of the reference's WaveNetBlock is 64, temporal.py:82) on the
        128-channel kernels: every parameter is zero-padded to 128 channels.  A padded channel is 0 after first_conv (zero weights,
        zero bias, ReLU) and stays 0 through every layer (its rows and columns of every weight are zero, dropout of 0 is 0, the
        residual adds 0), and the real channels only ever add exact zeros to their sums: the first out_dims channels of the
        result ARE the out_dims-channel network's, not an approximation of it; the gradients of the padding are sliced away by
        autograd.  GroupNorm / ReLU / dropout of the wrapper (models.py:759-768) run on the sliced rows as torch ops (the
        kernel's GroupNorm is laid out for 128 channels).  Costs the FLOPs of the 128-channel network."""
        import dataclasses

        import torch.nn.functional as F
        H, pad = self.out_dims, 128 - self.out_dims
        ps = self.ordered_parameters()
        padded = [F.pad(ps[0], (0, 0, 0, 0, 0, pad)), F.pad(ps[1], (0, pad))]              # first_conv [H, Cin, 1], [H]
        for w in ps[2:]:
            padded.append(F.pad(w, (0, 0, 0, pad, 0, pad)) if w.dim() == 3 else F.pad(w, (0, pad)))
        spec128 = dataclasses.replace(spec, hidden=128, last_gn=False, last_relu=False, last_dropout=False)
        ones = torch.ones(128, device=tape.device)
        z = ops.encoder_forward(tape, padded + [ones, torch.zeros_like(ones)], spec128, training=self.training, seed=seed)[..., :H]
        if spec.last_gn:
            z = F.group_norm(z.permute(0, 2, 1), spec.last_gn_num_groups, gn_weight, gn_bias, spec.gn_eps).permute(0, 2, 1)
        if spec.last_relu:
            z = torch.relu(z)
        if spec.last_dropout:
            z = F.dropout(z, p=spec.last_dropout_rate, training=self.training)
        return z.contiguous()

    def forward(self, x: Tensor) -> Tensor:
        """Reference signature (temporal.py:128-147): x [B, Cin, T] -> [B, out_dims, Tz].  A permuted view of a
        row-major [B, T, Cin] tensor (what MuCon.temporal_modeling_forward passes) is consumed without a copy."""
        tape = x.permute(0, 2, 1)
        ones = torch.ones(self.out_dims, device=x.device)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.training else 0
        z = self.forward_time_major(tape, ones, torch.zeros_like(ones), self.spec(), seed)
        return z.permute(0, 2, 1)



class NoFt(nn.Module):
    """No temporal modelling: one position-wise linear map in_channels -> out_dims (a kernel-size-1 convolution)."""

    def __init__(self, in_chnnels: int, out_dims: int, kernel_size: int = 1):   # (sic) the reference's argument name
        super().__init__()
        if kernel_size != 1:
            raise NotImplementedError("NoFt is used with kernel_size=1 only (reference models.py:181-185)")
        self.in_chnnels, self.out_dims, self.kernel_size = in_chnnels, out_dims, kernel_size
        self.last_conv = nn.Conv1d(in_chnnels, out_dims, kernel_size)

    def forward_time_major(self, tape: Tensor) -> Tensor:
        """[B, T, Cin] row-major -> [B, T, out_dims] on the tape as it lies in memory: first_conv's kernels without the
        non-linearity (ops.linear_forward: f32 MFMA, split-bf16 from 8,192 frames; weight gradient = first_conv's job).
        Device tensors of a shape those kernels are not built for (out_dims != 128, Cin not a multiple of 128) RAISE; host
        tensors (cfg.system.device = "cpu": plumbing, and the CPU test of the module surface) take one torch GEMM."""
        if tape.is_cuda:
            if not (self.out_dims == 128 and self.in_chnnels % 128 == 0 and tape.dtype == torch.float32):
                # no silent library-op path for device tensors: the kernels are what this package is
                raise NotImplementedError(f"NoFt on the HIP path needs out_dims = 128, in_channels a multiple of 128 and float32 "
                                          f"(got {self.in_chnnels} -> {self.out_dims}, {tape.dtype})")
            return ops.linear_forward(tape, self.last_conv.weight, self.last_conv.bias)
        return torch.matmul(tape, self.last_conv.weight[:, :, 0].t()) + self.last_conv.bias      # host tensors: cfg.system.device = "cpu" plumbing

    def forward(self, x: Tensor) -> Tensor:
        """Reference signature: [B, Cin, T] -> [B, out_dims, T]."""
        return self.forward_time_major(x.permute(0, 2, 1)).permute(0, 2, 1)


class MSTCNPPFirstStage(nn.Module):
    """First stage of MS-TCN++: per layer two dilated k=3 convolutions with mirrored dilations (2^(L-1-i) and 2^i), fused
    by a 1x1 convolution over their concatenation, ReLU, Dropout(0.5), residual; x2 max-pooling after `pooling_layers`."""

    def __init__(self, num_layers, num_f_maps, input_dim, output_dim, pooling_layers=(1, 2, 4, 8)):
        super().__init__()
        self.num_layers, self.pooling_layers = num_layers, pooling_layers
        self.conv_1x1_in = nn.Conv1d(input_dim, num_f_maps, 1)
        far = [2 ** (num_layers - 1 - i) for i in range(num_layers)]
        near = [2 ** i for i in range(num_layers)]
        self.conv_dilated_1 = nn.ModuleList(nn.Conv1d(num_f_maps, num_f_maps, 3, padding=d, dilation=d) for d in far)
        self.conv_dilated_2 = nn.ModuleList(nn.Conv1d(num_f_maps, num_f_maps, 3, padding=d, dilation=d) for d in near)
        self.conv_fusion = nn.ModuleList(nn.Conv1d(2 * num_f_maps, num_f_maps, 1) for _ in range(num_layers))
        self.dropout = nn.Dropout()
        self.conv_out = nn.Conv1d(num_f_maps, output_dim, 1)

    def _hip_ok(self, x: Tensor) -> bool:
        return (x.is_cuda and x.dtype == torch.float32 and self.conv_1x1_in.out_channels == 128
                and self.conv_1x1_in.in_channels % 128 == 0 and self.conv_out.out_channels == 128)

    def forward_time_major(self, tape: Tensor) -> Tensor:
        """[B, T, Cin] row-major -> [B, Tz, output_dim] on the encoder's kernels through the C ABI: conv_1x1_in = first_conv without
        its non-linearity (mucon_linear_fwd), the two dilated convolutions = mucon_conv128_fwd / _dgrad / _wgrad (f32 MFMA), and the
        layer's tail -- conv_fusion over the concatenation (a two-source 1x1 convolution), ReLU, dropout, residual, max-pooling --
        ONE launch (mucon_mstcn_fuse_fwd): three launches per layer, no torch glue in the forward."""
        f = ops.linear_forward(tape, self.conv_1x1_in.weight, self.conv_1x1_in.bias)
        for i, (far, near, fuse) in enumerate(zip(self.conv_dilated_1, self.conv_dilated_2, self.conv_fusion)):
            a = ops.conv128_forward(f, far.weight, far.bias, far.dilation[0])
            b = ops.conv128_forward(f, near.weight, near.bias, near.dilation[0])
            # conv_fusion over cat(a, b), ReLU, Dropout, the residual and the x2 max-pooling: ONE launch (mucon_mstcn_fuse_fwd).  The
            # dropout is the kernels' counter-based one, keyed by a seed drawn from torch's generator per layer call.
            drop_on = self.training and self.dropout.p > 0
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_on else 0
            f = ops.mstcn_fuse_forward(a, b, f, fuse.weight, fuse.bias, float(self.dropout.p), seed, drop_on, i in self.pooling_layers)
        return ops.conv128_forward(f, self.conv_out.weight, self.conv_out.bias, 1)

    def forward(self, x: Tensor) -> Tensor:
        """[B, Cin, T] -> [B, output_dim, Tz].  On the GPU with the reference's sizes (128 feature maps, Cin a multiple of 128): the
        hand-written kernels (a permuted view of a row-major [B, T, Cin] tensor is consumed without a copy); device tensors of other
        sizes RAISE; host tensors (cfg.system.device = "cpu" plumbing, the CPU test of the module surface): library ops."""
        if x.is_cuda:
            if not self._hip_ok(x):
                # no silent library-op path for device tensors: the kernels are what this package is
                raise NotImplementedError(f"MSTCNPPFirstStage on the HIP path needs 128 feature maps, an input width that is a multiple of "
                                          f"128 and float32 (got {self.conv_1x1_in.in_channels} -> {self.conv_1x1_in.out_channels} -> "
                                          f"{self.conv_out.out_channels}, {x.dtype})")
            return self.forward_time_major(x.permute(0, 2, 1)).permute(0, 2, 1)
        f = self.conv_1x1_in(x)                          # host tensors: cfg.system.device = "cpu" plumbing
        for i, (far, near, fuse) in enumerate(zip(self.conv_dilated_1, self.conv_dilated_2, self.conv_fusion)):
            both = torch.cat((far(f), near(f)), dim=1)
            f = f + self.dropout(torch.relu(fuse(both)))
            if i in self.pooling_layers:
                f = torch.nn.functional.max_pool1d(f, kernel_size=2)
        return self.conv_out(f)
