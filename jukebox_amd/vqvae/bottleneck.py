"""Bottleneck with the reference's buffer layout (jukebox/vqvae/bottleneck.py): `level_blocks.{l}.k` is a
registered (k_bins, emb_width) buffer.  Only encode / decode (inference) are implemented; the EMA
codebook update is training-only."""
import torch as t
import torch.nn as nn

from .. import hip_ops as H


class BottleneckBlock(nn.Module):
    def __init__(self, k_bins, emb_width, mu):
        super().__init__()
        self.k_bins, self.emb_width, self.mu = k_bins, emb_width, mu
        self.register_buffer("k", t.zeros(k_bins, emb_width))      # no device required at construction
        self._kt = None

    def _apply(self, fn, *a, **k):
        self._kt = None
        return super()._apply(fn, *a, **k)

    def encode_rows(self, x):
        """x: (M, emb) fp32 rows -> (M,) int64 codes: argmin ||x||^2 - 2 x.k + ||k||^2 (:112-119)."""
        key = (self.k.data_ptr(), self.k._version)
        if self._kt is None or self._kt[0] != key:
            self._kt = (key, H.pack_linear_w(self.k.detach().float().contiguous(), t.float32))
        xk = H.gemm(x, self._kt[1])
        return H.vq_argmin(x, xk, self.k.detach().float().contiguous())

    def decode_rows(self, codes):
        """codes (N, T) int64 -> (N*T, emb) fp32 rows (dequantise :121-123, kept channels-last)."""
        return H.vq_gather(codes.contiguous(), self.k.detach().float().contiguous()).view(-1, self.emb_width)


class Bottleneck(nn.Module):
    def __init__(self, l_bins, emb_width, mu, levels):
        super().__init__()
        self.levels = levels
        self.level_blocks = nn.ModuleList(BottleneckBlock(l_bins, emb_width, mu) for _ in range(levels))
