"""VQVAE with the reference's module tree and encode/decode API (jukebox/vqvae/vqvae.py:42-144).  The conv
stacks and the bottleneck run on the HIP kernels; the training forward (losses) is out of scope."""
import numpy as np
import torch as t
import torch.nn as nn

from .bottleneck import Bottleneck
from .encdec import Decoder, Encoder


def calculate_strides(strides, downs):
    return [stride ** down for stride, down in zip(strides, downs)]


class VQVAE(nn.Module):
    def __init__(self, input_shape, levels, downs_t, strides_t, emb_width, l_bins, mu, commit, spectral, multispectral,
                 multipliers=None, use_bottleneck=True, **block_kwargs):
        super().__init__()
        assert use_bottleneck
        self.sample_length = input_shape[0]
        x_shape, x_channels = input_shape[:-1], input_shape[-1]
        self.x_shape = x_shape
        self.downsamples = calculate_strides(strides_t, downs_t)
        self.hop_lengths = np.cumprod(self.downsamples)
        self.z_shapes = [(x_shape[0] // self.hop_lengths[level],) for level in range(levels)]
        self.levels = levels
        self.multipliers = [1] * levels if multipliers is None else multipliers
        assert len(self.multipliers) == levels, "Invalid number of multipliers"

        def kw(level):
            k = dict(block_kwargs)
            k["width"] *= self.multipliers[level]
            k["depth"] *= self.multipliers[level]
            return k
        self.encoders = nn.ModuleList(Encoder(x_channels, emb_width, level + 1, downs_t[:level + 1], strides_t[:level + 1],
                                              **kw(level)) for level in range(levels))
        self.decoders = nn.ModuleList(Decoder(x_channels, emb_width, level + 1, downs_t[:level + 1], strides_t[:level + 1],
                                              **kw(level)) for level in range(levels))
        self.bottleneck = Bottleneck(l_bins, emb_width, mu, levels)
        self.downs_t, self.strides_t, self.l_bins = downs_t, strides_t, l_bins
        self.commit, self.spectral, self.multispectral = commit, spectral, multispectral

    def _decode(self, zs, start_level=0, end_level=None):
        """vqvae.py:101-113: only zs[0] (level start_level) is decoded (all_levels=False)."""
        if end_level is None:
            end_level = self.levels
        assert len(zs) == end_level - start_level
        z = zs[0]
        N, T = z.shape
        x = self.bottleneck.level_blocks[start_level].decode_rows(z)
        x, T = self.decoders[start_level].forward_rows(x, N, T)
        return x.view(N, T, -1)                     # already NTC: postprocess (:96-99) is a no-op here

    def decode(self, zs, start_level=0, end_level=None, bs_chunks=1):
        """vqvae.py:115-122.  bs_chunks is honoured (bounds activation memory) but the kernels batch fine."""
        z_chunks = [t.chunk(z, bs_chunks, dim=0) for z in zs]
        outs = []
        with t.no_grad():
            for i in range(len(z_chunks[0])):
                outs.append(self._decode([zc[i] for zc in z_chunks], start_level=start_level, end_level=end_level))
        return t.cat(outs, dim=0)

    def _encode(self, x, start_level=0, end_level=None):
        """vqvae.py:124-135: every level re-encodes the raw audio with its own encoder."""
        if end_level is None:
            end_level = self.levels
        N, T, Cx = x.shape
        rows = x.float().reshape(N * T, Cx).contiguous()
        zs = []
        for level in range(self.levels):
            h, Tl = self.encoders[level].forward_rows(rows, N, T)[-1]
            zs.append(self.bottleneck.level_blocks[level].encode_rows(h).view(N, Tl))
        return zs[start_level:end_level]

    def encode(self, x, start_level=0, end_level=None, bs_chunks=1):
        """vqvae.py:137-144."""
        zs_list = []
        with t.no_grad():
            for x_i in t.chunk(x, bs_chunks, dim=0):
                zs_list.append(self._encode(x_i, start_level=start_level, end_level=end_level))
        return [t.cat(level_list, dim=0) for level_list in zip(*zs_list)]

    def sample(self, n_samples):
        zs = [t.randint(0, self.l_bins, size=(n_samples, *z_shape), device=self.bottleneck.level_blocks[0].k.device)
              for z_shape in self.z_shapes]
        return self.decode(zs)
