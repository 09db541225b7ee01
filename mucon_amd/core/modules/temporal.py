"""Temporal encoder modules with the reference's names and state_dict keys
(src/core/modules/temporal.py:9-147): WaveNetBlock.{first_conv, l_0 .. l_{n-1}.{dilated_conv,
conv_1x1}, last_conv}.  The nn.Conv1d sub-modules only HOLD the parameters (same shapes, same
default initialisation as the reference, so reference checkpoints load); the arithmetic runs in
the gfx950 kernels behind mucon_amd.ops.encoder_forward.

Only the default encoder (cfg.model.ft.type == "wavenet") is implemented; MSTCNPPFirstStage and
NoFt are non-default variants outside this round's scope."""
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
        out = [self.first_conv.weight, self.first_conv.bias]
        for l in self.layers:
            out += [l.dilated_conv.weight, l.dilated_conv.bias, l.conv_1x1.weight, l.conv_1x1.bias]
        return out + [self.last_conv.weight, self.last_conv.bias]

    def forward_time_major(self, tape: Tensor, gn_weight: Tensor, gn_bias: Tensor, spec: ops.EncoderSpec,
                           seed: int = 0) -> Tensor:
        """tape [B, T, Cin] (row-major, as the dataset delivers it) -> [B, Tz, out_dims]."""
        return ops.encoder_forward(tape, self.ordered_parameters() + [gn_weight, gn_bias], spec,
                                   training=self.training, seed=seed)

    def forward(self, x: Tensor) -> Tensor:
        """Reference signature (temporal.py:128-147): x [B, Cin, T] -> [B, out_dims, Tz].  A permuted view of a
        row-major [B, T, Cin] tensor (what MuCon.temporal_modeling_forward passes) is consumed without a copy."""
        tape = x.permute(0, 2, 1)
        ones = torch.ones(self.out_dims, device=x.device)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.training else 0
        z = self.forward_time_major(tape, ones, torch.zeros_like(ones), self.spec(), seed)
        return z.permute(0, 2, 1)
