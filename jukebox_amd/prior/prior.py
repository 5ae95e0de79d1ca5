"""SimplePrior with the reference's module tree and sampling API (jukebox/prior/prior.py:27-283): conditioning
assembly (labels, upper-level codes, lyric prefix with merged vocabulary) around the HIP decode engine.
Training methods (z_forward / forward) are out of scope."""
import numpy as np
import torch as t
import torch.nn as nn

from ..data.labels import EmptyLabeller, Labeller
from ..utils import dist_adapter as dist
from ..utils.dist_utils import print_once
from ..vqvae.vqvae import calculate_strides
from .autoregressive import ConditionalAutoregressive2D
from .conditioners import Conditioner, LabelConditioner


class SimplePrior(nn.Module):
    def __init__(self, z_shapes, l_bins, encoder, decoder, level, downs_t, strides_t, labels, prior_kwargs,
                 x_cond_kwargs, y_cond_kwargs, prime_kwargs, copy_input, labels_v3=False, merged_decoder=False,
                 single_enc_dec=False):
        super().__init__()
        self.use_tokens = prime_kwargs.pop("use_tokens")
        self.n_tokens = prime_kwargs.pop("n_tokens")
        self.prime_loss_fraction = prime_kwargs.pop("prime_loss_fraction")
        self.copy_input = copy_input
        if self.copy_input:
            prime_kwargs["bins"] = l_bins
        self.z_shapes, self.levels = z_shapes, len(z_shapes)
        self.z_shape = self.z_shapes[level]
        self.level = level
        assert level < self.levels, f"Total levels {self.levels}, got level {level}"
        self.l_bins = l_bins
        self.encoder, self.decoder = encoder, decoder          # bound methods of the VQ-VAE (no parameters captured)
        self.x_cond = level != (self.levels - 1)
        self.cond_level = level + 1
        self.y_cond = labels
        self.single_enc_dec = single_enc_dec
        if self.x_cond:
            self.conditioner_blocks = nn.ModuleList()
            self.conditioner_blocks.append(Conditioner(input_shape=z_shapes[self.cond_level], bins=l_bins,
                                                       down_t=downs_t[self.cond_level], stride_t=strides_t[self.cond_level],
                                                       **x_cond_kwargs))
        if self.y_cond:
            self.n_time = self.z_shape[0]
            self.y_emb = LabelConditioner(n_time=self.n_time, include_time_signal=not self.x_cond, **y_cond_kwargs)
        if single_enc_dec:
            # merged lyric + VQ vocabulary, one longer sequence (prior.py:90-103)
            self.prior_shapes = [(self.n_tokens,), prior_kwargs.pop("input_shape")]
            self.prior_bins = [prime_kwargs["bins"], prior_kwargs.pop("bins")]
            self.prior_dims = [int(np.prod(shape)) for shape in self.prior_shapes]
            self.prior_bins_shift = np.cumsum([0, *self.prior_bins])[:-1]
            self.prior_width = prior_kwargs["width"]
            self.prime_loss_dims, self.gen_loss_dims = self.prior_dims[0], self.prior_dims[1]
            self.total_loss_dims = self.prime_loss_dims + self.gen_loss_dims
            self.prior = ConditionalAutoregressive2D(input_shape=(sum(self.prior_dims),), bins=sum(self.prior_bins),
                                                     x_cond=(self.x_cond or self.y_cond), y_cond=True,
                                                     prime_len=self.prime_loss_dims, **prior_kwargs)
        else:
            if self.n_tokens != 0 and self.use_tokens:
                # separate lyric encoder whose last-layer states are cross-attended by the decoder (prior.py:104-117)
                from ..transformer.ops import Conv1D, LayerNorm
                prime_input_shape = (self.n_tokens,)
                self.prime_loss_dims = int(np.prod(prime_input_shape))
                self.prime_acts_width, self.prime_state_width = prime_kwargs["width"], prior_kwargs["width"]
                self.prime_prior = ConditionalAutoregressive2D(input_shape=prime_input_shape, x_cond=False, y_cond=False,
                                                               only_encode=True, **prime_kwargs)
                self.prime_state_proj = Conv1D(self.prime_acts_width, self.prime_state_width,
                                               init_scale=prime_kwargs["init_scale"])
                self.prime_state_ln = LayerNorm(self.prime_state_width)
                self.prime_bins = prime_kwargs["bins"]
                self.prime_x_out = nn.Linear(self.prime_state_width, self.prime_bins, bias=False)
                nn.init.normal_(self.prime_x_out.weight, std=0.02 * prior_kwargs["init_scale"])
            else:
                self.prime_loss_dims = 0
            self.gen_loss_dims = int(np.prod(self.z_shape))
            self.total_loss_dims = self.prime_loss_dims + self.gen_loss_dims
            self.prior = ConditionalAutoregressive2D(x_cond=(self.x_cond or self.y_cond), y_cond=self.y_cond,
                                                     encoder_dims=self.prime_loss_dims, merged_decoder=merged_decoder,
                                                     **prior_kwargs)
        self.n_ctx = self.gen_loss_dims
        self.downsamples = calculate_strides(strides_t, downs_t)
        self.cond_downsample = self.downsamples[level + 1] if level != self.levels - 1 else None
        self.raw_to_tokens = int(np.prod(self.downsamples[:level + 1]))
        self.sample_length = self.n_ctx * self.raw_to_tokens
        if labels:
            self.labels_v3 = labels_v3
            self.labeller = Labeller(self.y_emb.max_bow_genre_size, self.n_tokens, self.sample_length, v3=self.labels_v3)
        else:
            self.labeller = EmptyLabeller()

    def get_y(self, labels, start, get_indices=False):
        """The label matrix of the window that starts at token `start`: this level's window length, the offset moved to the
        window, and the lyric tokens re-windowed around it (prior.py:140-156).  None for unlabelled models."""
        if isinstance(self.labeller, EmptyLabeller):
            return None
        y = labels["y"].clone()
        y[:, 1] += int(start * self.raw_to_tokens)
        y[:, 2] = int(self.sample_length)
        indices = self.labeller.set_y_lyric_tokens(y, labels)
        return (y, indices) if get_indices else y

    def get_z_conds(self, zs, start, end):
        """The upper level's codes under the window [start, end): one tensor in a list, None at the top (prior.py:158-166)."""
        if self.level == self.levels - 1:
            return None
        cd = self.cond_downsample
        assert start % cd == 0 and end % cd == 0
        z_cond = zs[self.level + 1][:, start // cd:end // cd]
        assert z_cond.shape[1] == self.n_ctx // cd
        return [z_cond]

    def prior_preprocess(self, xs, conds):
        """single_enc_dec models: lyric tokens and music codes become ONE sequence over the merged vocabulary (each part
        shifted by the sizes of the parts before it), their conditioning one tensor, zeros where a part has none
        (prior.py:168-185)."""
        N = xs[0].shape[0]
        parts = []
        for x, bins, shift in zip(xs, self.prior_bins, self.prior_bins_shift):
            assert bool(((x >= 0) & (x < int(bins))).all())
            parts.append(x.reshape(N, -1) + int(shift))
        cond_parts = []
        for cond, dims in zip(conds, self.prior_dims):
            want = (N, dims, self.prior_width)
            if cond is None:
                cond = t.zeros(want, dtype=t.float, device=xs[0].device)
            assert tuple(cond.shape) == want
            cond_parts.append(cond)
        return t.cat(parts, dim=1), t.cat(cond_parts, dim=1)

    def prior_postprocess(self, z):
        """Inverse of prior_preprocess for the music part: the lyric prefix is dropped, the vocabulary shift undone, and
        an id sampled from the lyric range (possible: one softmax covers both) becomes code 0 (prior.py:187-203)."""
        n_lyric = self.prior_dims[0]
        music = z[:, n_lyric:] - int(self.prior_bins_shift[-1])
        music = music.clamp(min=0).reshape(z.shape[0], -1, *self.prior_shapes[-1][1:])
        assert bool((music < int(self.prior_bins[-1])).all())
        lyric = z[:, :n_lyric] - int(self.prior_bins_shift[0])
        assert bool((lyric.clamp(min=0) < int(self.prior_bins[0])).all())
        return music

    def x_emb(self, z_conds):
        """Upper-level codes -> (N, n_ctx, width) conditioning through this level's conditioner block(s), coarsest first."""
        n_blocks = self.cond_level - self.level
        z_conds = z_conds[:n_blocks]
        assert len(z_conds) == len(self.conditioner_blocks) == n_blocks
        x_cond = None
        for k in range(n_blocks - 1, -1, -1):
            x_cond = self.conditioner_blocks[k](z_conds[k], x_cond)
        return x_cond

    def _levels(self, start_level, end_level):
        return (self.level if start_level is None else start_level), (self.levels if end_level is None else end_level)

    def encode(self, x, start_level=None, end_level=None, bs_chunks=1):
        lo, hi = self._levels(start_level, end_level)
        with t.no_grad():
            return self.encoder(x, start_level=lo, end_level=hi, bs_chunks=bs_chunks)

    def decode(self, zs, start_level=None, end_level=None, bs_chunks=1):
        lo, hi = self._levels(start_level, end_level)
        assert len(zs) == hi - lo
        with t.no_grad():
            return self.decoder(zs, start_level=lo, end_level=hi, bs_chunks=bs_chunks)

    def get_cond(self, z_conds, y):
        """(x_cond, y_cond, lyric tokens) of one window: the label columns of y go through the label conditioner, its lyric
        columns are returned as they are, the upper-level codes through the conditioner; models without an upper level
        are conditioned on the labels' timing signal instead (prior.py:234-243)."""
        prime = None
        if y is not None:
            n_labels = 4 + self.y_emb.max_bow_genre_size
            assert y.shape[1] == n_labels + self.n_tokens
            y, prime = y[:, :n_labels], y[:, n_labels:]
        y_cond, y_pos = self.y_emb(y) if self.y_cond else (None, None)
        return (self.x_emb(z_conds) if self.x_cond else y_pos), y_cond, prime

    def get_encoder_kv(self, prime, fp16=False, sample=False):
        """prior.py:285-301: lyric tokens -> encoder activations -> projection -> LayerNorm (fp32), half when fp16."""
        if self.n_tokens != 0 and self.use_tokens:
            N = prime.shape[0]
            prime_acts = self.prime_prior(prime, None, None, None, fp16=fp16)
            assert tuple(prime_acts.shape) == (N, self.prime_loss_dims, self.prime_acts_width)
            assert prime_acts.dtype == t.float
            encoder_kv = self.prime_state_ln(self.prime_state_proj(prime_acts))
            assert encoder_kv.dtype == t.float
            if sample and fp16:
                encoder_kv = encoder_kv.half()
            return encoder_kv
        return None

    def sample(self, n_samples, z=None, z_conds=None, y=None, fp16=False, temp=1.0, top_k=0, top_p=0.0, chunk_size=None,
               sample_tokens=None, seed=0, sample_base=0, pos_base=0):
        """prior.py:245-283.  seed / sample_base / pos_base (extensions): the draw for global sample index
        sample_base + n at window position t is a pure function of (seed, level, index, pos_base + t) -- pos_base is the
        window's start at this level -- so sharded runs reproduce and no two windows or levels share a draw."""
        N = n_samples
        if z is not None:
            assert z.shape[0] == N, f"Expected shape ({N},**), got shape {z.shape}"
        if y is not None:
            assert y.shape[0] == N, f"Expected shape ({N},**), got shape {y.shape}"
        if z_conds is not None:
            for z_cond in z_conds:
                assert z_cond.shape[0] == N, f"Expected shape ({N},**), got shape {z_cond.shape}"
        no_past_context = z is None or z.shape[1] == 0
        if dist.get_rank() == 0:
            name = {True: "Ancestral", False: "Primed"}[no_past_context]
            print_once(f"{name} sampling {n_samples} samples with temp={temp}, top_k={top_k}, top_p={top_p}")
        kw = dict(fp16=fp16, temp=temp, top_k=top_k, top_p=top_p, seed=seed, sample_base=sample_base, pos_base=pos_base,
                  stream_id=int(self.level))
        self.prior.decode_tap = self._decode_tap()
        with t.no_grad():
            x_cond, y_cond, prime = self.get_cond(z_conds, y)
            if self.single_enc_dec:
                if no_past_context:
                    z, x_cond = self.prior_preprocess([prime], [None, x_cond])
                else:
                    z, x_cond = self.prior_preprocess([prime, z], [None, x_cond])
                if sample_tokens is not None:
                    sample_tokens += self.n_tokens
                z = self.prior.primed_sample(n_samples, z, x_cond, y_cond, chunk_size=chunk_size,
                                             sample_tokens=sample_tokens, **kw)
                z = self.prior_postprocess(z)
            else:
                encoder_kv = self.get_encoder_kv(prime, fp16=fp16, sample=True)
                if no_past_context:
                    z = self.prior.sample(n_samples, x_cond, y_cond, encoder_kv, sample_tokens=sample_tokens, **kw)
                else:
                    z = self.prior.primed_sample(n_samples, z, x_cond, y_cond, encoder_kv, chunk_size=chunk_size,
                                                 sample_tokens=sample_tokens, **kw)
            if sample_tokens is None:
                assert tuple(z.shape) == (N, *self.z_shape)
        self.prior.decode_tap = None
        return z

    def get_prime_loss(self, encoder_kv, prime_t):
        """prior.py:303-310: bits per lyric token of the separate lyric encoder's own prediction head."""
        if self.n_tokens != 0 and self.use_tokens:
            logits = self.prime_x_out(encoder_kv.float())
            return nn.functional.cross_entropy(logits.reshape(-1, self.prime_bins), prime_t.reshape(-1)) / np.log(2.0)
        return t.tensor(0.0, device=prime_t.device if prime_t is not None else "cpu")

    def z_forward(self, z, z_conds=[], y=None, fp16=False, get_preds=False, get_attn_weights=False):
        """prior.py:312-347, inference only: (loss, metrics) of a given code sequence -- bits per token of the music part
        (`bpd` / `gen_loss`), of the lyric part (`prime_loss`) and their weighted sum; metrics["preds"] with get_preds.
        Attention maps are recorded by jukebox_amd.align.get_alignment (engine prefill with recording), not here."""
        assert not get_attn_weights, "use jukebox_amd.align.get_alignment for attention weights"
        with t.no_grad():
            x_cond, y_cond, prime = self.get_cond(z_conds, y)
            if self.copy_input:
                prime = z[:, :self.n_tokens]
            if self.single_enc_dec:
                z, x_cond = self.prior_preprocess([prime, z], [None, x_cond])
                (prime_loss, gen_loss), preds = self.prior(z, x_cond, y_cond, fp16=fp16, get_sep_loss=True, get_preds=get_preds)
            else:
                encoder_kv = self.get_encoder_kv(prime, fp16=fp16)
                prime_loss = self.get_prime_loss(encoder_kv, prime) if encoder_kv is not None else t.tensor(0.0, device=z.device)
                if encoder_kv is not None and fp16:
                    encoder_kv = encoder_kv.half()
                gen_loss, preds = self.prior(z, x_cond, y_cond, encoder_kv, fp16=fp16, get_preds=get_preds)
            loss = (self.prime_loss_fraction * prime_loss * self.prime_loss_dims / self.total_loss_dims) + \
                (gen_loss * self.gen_loss_dims / self.total_loss_dims)
            metrics = dict(bpd=gen_loss.clone().detach(), prime_loss=prime_loss.clone().detach(),
                           gen_loss=gen_loss.clone().detach())
            if get_preds:
                metrics["preds"] = preds.clone().detach()
        return loss, metrics

    def forward(self, x, y=None, fp16=False, decode=False, get_preds=False):
        """prior.py:349-358: encode raw audio, evaluate it under this prior, optionally decode it back."""
        bs = x.shape[0]
        z, *z_conds = self.encode(x, bs_chunks=bs)
        loss, metrics = self.z_forward(z=z, z_conds=z_conds, y=y, fp16=fp16, get_preds=get_preds)
        x_out = self.decode([z, *z_conds]) if decode else None
        return x_out, loss, metrics

    def _decode_tap(self):
        """window_tap = (every, cb): cb(lo, hi, tokens) receives the window's music tokens [lo, hi) (window-relative,
        already through prior_postprocess) as soon as their decode steps are enqueued -- an extension used by the level
        pipeline; None (the default) samples the window in one piece."""
        wt = getattr(self, "window_tap", None)
        if wt is None:
            return None
        every, cb = wt
        off = self.n_tokens if self.single_enc_dec else 0
        shift = int(self.prior_bins_shift[-1]) if self.single_enc_dec else 0

        def fn(tokens, lo, hi):
            tok = tokens[:, lo:hi]
            if self.single_enc_dec:
                tok = t.clamp(tok - shift, min=0)                  # prior_postprocess on the music part
            cb(lo - off, hi - off, tok)
        return every, fn
