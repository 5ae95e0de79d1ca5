"""Conditioners with the reference's parameter tree (jukebox/prior/conditioners.py).

`Conditioner` (upsampler x-conditioning) is the heavy one -- code embedding, a DecoderConvBock of dilated
residual convolutions upsampling x4, LayerNorm -- and runs on the HIP tap-GEMM / LayerNorm kernels,
channels-last end to end (the reference's two permutes disappear).  The label embeddings are a handful
of table lookups per window and stay as torch indexing (host glue)."""
import torch as t
import torch.nn as nn

from .. import hip_ops as H
from ..transformer.ops import LayerNorm
from ..vqvae.encdec import DecoderConvBock


class Conditioner(nn.Module):
    """conditioners.py:8-48."""

    def __init__(self, input_shape, bins, down_t, stride_t, out_width, init_scale, zero_out, res_scale, **block_kwargs):
        super().__init__()
        self.x_shape = input_shape
        self.width = out_width
        self.x_emb = nn.Embedding(bins, out_width)
        nn.init.normal_(self.x_emb.weight, std=0.02 * init_scale)
        self.cond = DecoderConvBock(self.width, self.width, down_t, stride_t, **block_kwargs, zero_out=zero_out,
                                    res_scale=res_scale)
        self.ln = LayerNorm(self.width)

    def forward(self, x, x_cond=None):
        N = x.shape[0]
        assert tuple(x.shape) == (N, *self.x_shape), f"Expected {(N, *self.x_shape)} got {tuple(x.shape)}"
        T = x.shape[1]
        rows = H.vq_gather(x.long().contiguous(), self.x_emb.weight.detach().float().contiguous()).view(N * T, self.width)
        if x_cond is not None:
            rows = rows + x_cond.reshape(N * T, self.width)
        rows, T = self.cond.forward_rows(rows.contiguous(), N, T)
        return self.ln(rows.view(N, T, self.width))


def _label_table(bins, out_width, init_scale):
    """An nn.Embedding initialised as the reference's label tables are (conditioners.py:62-63,76-77)."""
    table = nn.Embedding(bins, out_width)
    nn.init.normal_(table.weight, std=0.01 * init_scale)
    return table


def _require_2d(x, what="shape"):
    assert x.dim() == 2, f"Expected {what} with 2 dims, got {x.shape}"


class SimpleEmbedding(nn.Module):
    """A lookup table over ids 0 .. bins-1 (artist, genre words); state: `emb.weight` (conditioners.py:57-64)."""

    def __init__(self, bins, out_width, init_scale):
        super().__init__()
        self.bins = bins
        self.emb = _label_table(bins, out_width, init_scale)

    def forward(self, y):
        _require_2d(y)
        assert y.dtype == t.long, f"Expected dtype long, got {y.dtype}"
        in_range = (y >= 0) & (y < self.bins)
        assert bool(in_range.all()), f"Bins {self.bins}, got label {y}"
        return self.emb(y)


class RangeEmbedding(nn.Module):
    """A scalar in [range[0], range[1]) -- or a span [start, end) sampled at n_time equidistant points -- quantised into
    `bins` cells and embedded (timing signals of the label conditioner, conditioners.py:66-110).  The arithmetic keeps the
    reference's order of operations: the cell index is a floor and must not move by an ulp."""

    def __init__(self, n_time, bins, range, out_width, init_scale, clamp=False):
        super().__init__()
        self.n_time, self.bins, self.clamp = n_time, bins, clamp
        self.pos_min, self.pos_max = range
        self.emb = _label_table(bins, out_width, init_scale)

    def _check(self, pos, closed):
        hi_ok = (pos <= self.pos_max) if closed else (pos < self.pos_max)
        assert bool(((pos >= self.pos_min) & hi_ok).all()), f"Range is [{self.pos_min},{self.pos_max}), got {pos}"

    def forward(self, pos_start, pos_end=None):
        _require_2d(pos_start)
        self._check(pos_start, closed=False)
        position = pos_start.float()
        if pos_end is not None:
            pos_end = pos_end.clamp(self.pos_min, self.pos_max) if self.clamp else pos_end
            self._check(pos_end, closed=True)
        if self.n_time > 1:
            assert pos_end is not None
            frac = t.arange(0, self.n_time, dtype=t.float, device=position.device).view(1, self.n_time) / self.n_time
            position = position + (pos_end.float() - position) * frac
        cell = (self.bins * ((position - self.pos_min) / (self.pos_max - self.pos_min))).floor().long()
        return self.emb(cell.detach())


class LabelConditioner(nn.Module):
    """y = (total_length, offset, length, artist, genre words...) -> (start embedding (N, 1, W), timing signal (N, n_time, W) or
    None): the sum of the artist row and the bag of genre rows, and -- for models with a time signal -- total length +
    absolute position + position relative to the track (conditioners.py:112-157)."""

    def __init__(self, y_bins, t_bins, sr, min_duration, max_duration, n_time, out_width, init_scale,
                 max_bow_genre_size, include_time_signal):
        super().__init__()
        assert len(y_bins) == 2, f"Expecting (genre, artist) bins, got {y_bins}"
        self.n_time, self.out_width, self.max_bow_genre_size = n_time, out_width, max_bow_genre_size
        self.include_time_signal = include_time_signal
        self.bow_genre_emb = SimpleEmbedding(y_bins[0], out_width, init_scale)
        self.artist_emb = SimpleEmbedding(y_bins[1], out_width, init_scale)
        if include_time_signal:
            longest = max_duration * sr
            self.total_length_emb = RangeEmbedding(1, t_bins, (min_duration * sr, longest), out_width, init_scale)
            self.absolute_pos_emb = RangeEmbedding(n_time, t_bins, (0.0, longest), out_width, init_scale)
            self.relative_pos_emb = RangeEmbedding(n_time, t_bins, (0.0, 1.0), out_width, init_scale, clamp=True)

    def _who(self, artist, genre):
        words = self.bow_genre_emb(genre.clamp(min=0))                       # empty genre slots hold -1: masked out below
        present = (genre >= 0).unsqueeze(-1).to(words.dtype)
        return (words * present).sum(dim=1, keepdim=True) + self.artist_emb(artist)

    def _when(self, total_length, offset, length):
        total, begin = total_length.float(), offset.float()
        end = (offset + length).float()
        return self.total_length_emb(total) + self.absolute_pos_emb(begin, end) + self.relative_pos_emb(begin / total, end / total)

    def forward(self, y):
        _require_2d(y)
        n_cols = 4 + self.max_bow_genre_size
        assert y.shape[-1] == n_cols, f"Expected shape (N,{n_cols}), got {y.shape}"
        assert y.dtype == t.long
        N = y.shape[0]
        start_emb = self._who(y[:, 3:4], y[:, 4:])
        assert tuple(start_emb.shape) == (N, 1, self.out_width)
        pos_emb = None
        if self.include_time_signal:
            pos_emb = self._when(y[:, 0:1], y[:, 1:2], y[:, 2:3])
            assert tuple(pos_emb.shape) == (N, self.n_time, self.out_width)
        return start_emb, pos_emb
