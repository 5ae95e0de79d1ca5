"""Resnet1D / ResConv1DBlock with the reference's parameter tree (jukebox/vqvae/resnet.py:27-75).
The nn.Conv1d members only hold the parameters (so state_dict keys and shapes match released
checkpoints); the forward runs channels-last on the HIP tap-GEMM: a dilated k=3 convolution is
three row-shifted GEMMs accumulated in one launch, fused with the input ReLU, bias, output ReLU,
and the 1x1 convolution is fused with bias, res_scale and the residual add."""
import math

import torch.nn as nn

from .. import _lib as L
from .. import hip_ops as H
from .rows import PackedConvCache


class ResConv1DBlock(PackedConvCache, nn.Module):
    def __init__(self, n_in, n_state, dilation=1, zero_out=False, res_scale=1.0):
        super().__init__()
        self.model = nn.Sequential(nn.ReLU(), nn.Conv1d(n_in, n_state, 3, 1, dilation, dilation), nn.ReLU(),
                                   nn.Conv1d(n_state, n_in, 1, 1, 0))
        if zero_out:
            nn.init.zeros_(self.model[-1].weight)
            nn.init.zeros_(self.model[-1].bias)
        self.res_scale, self.dilation = res_scale, dilation

    def forward_rows(self, x, N, T):
        """x: (N*T, n_in) fp32 rows -> same shape: x + res_scale * conv1x1(relu(conv3_dil(relu(x))))."""
        d = self.dilation
        c3, c1 = self.model[1], self.model[3]
        h = H.gemm(x, self.packed(c3), bias=c3.bias.detach(), n_seq=N, t_in=T, shifts=(-d, 0, d), pre_relu=True,
                   act=L.ACT_RELU)
        return H.gemm(h, self.packed(c1), bias=c1.bias.detach(), res=x, res_scale=self.res_scale, n_seq=N, t_in=T)


class Resnet1D(nn.Module):
    def __init__(self, n_in, n_depth, m_conv=1.0, dilation_growth_rate=1, dilation_cycle=None, zero_out=False,
                 res_scale=False, reverse_dilation=False, checkpoint_res=False):
        super().__init__()
        depth_of = (lambda d: d) if dilation_cycle is None else (lambda d: d % dilation_cycle)
        blocks = [ResConv1DBlock(n_in, int(m_conv * n_in), dilation=dilation_growth_rate ** depth_of(d), zero_out=zero_out,
                                 res_scale=1.0 if not res_scale else 1.0 / math.sqrt(n_depth)) for d in range(n_depth)]
        if reverse_dilation:
            blocks = blocks[::-1]
        self.checkpoint_res = checkpoint_res
        if checkpoint_res == 1:          # resnet.py:62-67: parameter names become `blocks.*`
            self.blocks = nn.ModuleList(blocks)
        else:
            self.model = nn.Sequential(*blocks)

    def forward_rows(self, x, N, T):
        for blk in (self.blocks if self.checkpoint_res == 1 else self.model):
            x = blk.forward_rows(x, N, T)
        return x
