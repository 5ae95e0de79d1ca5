"""Encoder / Decoder conv stacks with the reference's parameter tree (jukebox/vqvae/encdec.py), executed
channels-last on the HIP tap-GEMM."""
import torch.nn as nn

from .resnet import Resnet1D
from .rows import PackedConvCache, conv_down, conv_k3, conv_transpose_up


class EncoderConvBlock(PackedConvCache, nn.Module):
    """encdec.py:6-26."""

    def __init__(self, input_emb_width, output_emb_width, down_t, stride_t, width, depth, m_conv,
                 dilation_growth_rate=1, dilation_cycle=None, zero_out=False, res_scale=False):
        super().__init__()
        blocks = []
        filter_t, pad_t = stride_t * 2, stride_t // 2
        if down_t > 0:
            for i in range(down_t):
                blocks.append(nn.Sequential(
                    nn.Conv1d(input_emb_width if i == 0 else width, width, filter_t, stride_t, pad_t),
                    Resnet1D(width, depth, m_conv, dilation_growth_rate, dilation_cycle, zero_out, res_scale)))
            blocks.append(nn.Conv1d(width, output_emb_width, 3, 1, 1))
        self.model = nn.Sequential(*blocks)
        self.down_t, self.stride_t = down_t, stride_t

    def forward_rows(self, x, N, T):
        for i in range(self.down_t):
            conv, res = self.model[i][0], self.model[i][1]
            x = conv_down(self, conv, x, N, T, self.stride_t)
            T //= self.stride_t
            x = res.forward_rows(x, N, T)
        if self.down_t > 0:
            x = conv_k3(self, self.model[self.down_t], x, N, T)
        return x, T


class DecoderConvBock(PackedConvCache, nn.Module):
    """encdec.py:28-46 (the reference's spelling)."""

    def __init__(self, input_emb_width, output_emb_width, down_t, stride_t, width, depth, m_conv,
                 dilation_growth_rate=1, dilation_cycle=None, zero_out=False, res_scale=False,
                 reverse_decoder_dilation=False, checkpoint_res=False):
        super().__init__()
        blocks = []
        if down_t > 0:
            filter_t, pad_t = stride_t * 2, stride_t // 2
            blocks.append(nn.Conv1d(output_emb_width, width, 3, 1, 1))
            for i in range(down_t):
                blocks.append(nn.Sequential(
                    Resnet1D(width, depth, m_conv, dilation_growth_rate, dilation_cycle, zero_out=zero_out,
                             res_scale=res_scale, reverse_dilation=reverse_decoder_dilation, checkpoint_res=checkpoint_res),
                    nn.ConvTranspose1d(width, input_emb_width if i == (down_t - 1) else width, filter_t, stride_t, pad_t)))
        self.model = nn.Sequential(*blocks)
        self.down_t, self.stride_t = down_t, stride_t

    def forward_rows(self, x, N, T):
        if self.down_t == 0:
            return x, T
        x = conv_k3(self, self.model[0], x, N, T)
        for i in range(self.down_t):
            res, up = self.model[1 + i][0], self.model[1 + i][1]
            x = res.forward_rows(x, N, T)
            x = conv_transpose_up(self, up, x, N, T)
            T *= 2
        return x, T


class Encoder(nn.Module):
    """encdec.py:48-85."""

    def __init__(self, input_emb_width, output_emb_width, levels, downs_t, strides_t, **block_kwargs):
        super().__init__()
        self.input_emb_width, self.output_emb_width, self.levels = input_emb_width, output_emb_width, levels
        self.downs_t, self.strides_t = downs_t, strides_t
        kw = dict(block_kwargs)
        kw.pop("reverse_decoder_dilation", None)
        self.level_blocks = nn.ModuleList(
            EncoderConvBlock(input_emb_width if level == 0 else output_emb_width, output_emb_width, down_t, stride_t, **kw)
            for level, down_t, stride_t in zip(range(levels), downs_t, strides_t))

    def forward_rows(self, x, N, T):
        """x (N*T, in) -> list of (rows, T_l) per level (each level feeds the next)."""
        xs = []
        for blk in self.level_blocks:
            x, T = blk.forward_rows(x, N, T)
            xs.append((x, T))
        return xs


class Decoder(PackedConvCache, nn.Module):
    """encdec.py:87-131 (all_levels=False path: only the deepest latent is decoded)."""

    def __init__(self, input_emb_width, output_emb_width, levels, downs_t, strides_t, **block_kwargs):
        super().__init__()
        self.input_emb_width, self.output_emb_width, self.levels = input_emb_width, output_emb_width, levels
        self.downs_t, self.strides_t = downs_t, strides_t
        self.level_blocks = nn.ModuleList(
            DecoderConvBock(output_emb_width, output_emb_width, down_t, stride_t, **block_kwargs)
            for down_t, stride_t in zip(downs_t, strides_t))
        self.out = nn.Conv1d(output_emb_width, input_emb_width, 3, 1, 1)

    def forward_rows(self, x, N, T):
        for blk in reversed(self.level_blocks):
            x, T = blk.forward_rows(x, N, T)
        return conv_k3(self, self.out, x, N, T), T
