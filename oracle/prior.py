"""Oracle restatement of jukebox/prior/{conditioners,prior}.py sampling methods (numpy).
Test infrastructure only."""
import numpy as np

from .autoregressive import ConditionalAutoregressive2D
from .ops import F32, layer_norm
from .vqvae import DecoderConvBock


class Conditioner:
    """conditioners.py:8-48 -- code embedding -> DecoderConvBock (x4 upsample) -> LayerNorm."""

    def __init__(self, sd, prefix, down_t, stride_t, width, depth, m_conv, dilation_growth_rate,
                 dilation_cycle, res_scale, checkpoint_res):
        self.sd, self.prefix = sd, prefix
        self.cond = DecoderConvBock(sd, prefix + "cond.", down_t, stride_t, width, depth, m_conv,
                                    dilation_growth_rate, dilation_cycle, res_scale=res_scale,
                                    checkpoint_res=checkpoint_res)

    def __call__(self, z, x_cond=None):
        sd, p = self.sd, self.prefix
        x = sd[p + "x_emb.weight"][np.asarray(z)]                  # (N, T, W)
        if x_cond is not None:
            x = x + x_cond
        x = self.cond(np.transpose(x, (0, 2, 1)).astype(F32))
        x = np.transpose(x, (0, 2, 1))
        return layer_norm(x, sd[p + "ln.weight"], sd[p + "ln.bias"]).astype(F32)


def range_embedding(emb, n_time, bins, rng, pos_start, pos_end=None, clamp=False):
    """conditioners.py:66-110 -- interpolate [start, end) over n_time steps, bin, embed."""
    pos_min, pos_max = rng
    pos_start = np.asarray(pos_start, F32)
    if pos_end is not None:
        pos_end = np.asarray(pos_end, F32)
        if clamp:
            pos_end = np.clip(pos_end, F32(pos_min), F32(pos_max))
    if n_time != 1:
        interp = (np.arange(n_time, dtype=F32) / F32(n_time)).reshape(1, n_time)
        position = pos_start + (pos_end - pos_start) * interp
    else:
        position = pos_start
    norm = (position - F32(pos_min)) / F32(pos_max - pos_min)
    idx = np.floor(F32(bins) * norm.astype(F32)).astype(np.int64)
    return emb[idx]


class LabelConditioner:
    """conditioners.py:112-157."""

    def __init__(self, sd, prefix, y_bins, t_bins, sr, min_duration, max_duration, n_time, max_bow_genre_size,
                 include_time_signal):
        self.sd, self.p = sd, prefix
        self.t_bins, self.n_time, self.max_bow = t_bins, n_time, max_bow_genre_size
        self.include_time_signal = include_time_signal
        self.ranges = ((min_duration * sr, max_duration * sr), (0.0, max_duration * sr), (0.0, 1.0))

    def __call__(self, y):
        sd, p = self.sd, self.p
        y = np.asarray(y)
        total_length, offset, length, artist, genre = y[:, 0:1], y[:, 1:2], y[:, 2:3], y[:, 3:4], y[:, 4:]
        artist_emb = sd[p + "artist_emb.emb.weight"][artist]                         # (N,1,W)
        mask = (genre >= 0).astype(F32)[..., None]
        genre_emb = (sd[p + "bow_genre_emb.emb.weight"][np.clip(genre, 0, None)] * mask).sum(axis=1, keepdims=True)
        start_emb = (genre_emb + artist_emb).astype(F32)
        if not self.include_time_signal:
            return start_emb, None
        tl, start, end = total_length.astype(F32), offset.astype(F32), (offset + length).astype(F32)
        pos = (range_embedding(sd[p + "total_length_emb.emb.weight"], 1, self.t_bins, self.ranges[0], tl)
               + range_embedding(sd[p + "absolute_pos_emb.emb.weight"], self.n_time, self.t_bins, self.ranges[1], start, end)
               + range_embedding(sd[p + "relative_pos_emb.emb.weight"], self.n_time, self.t_bins, self.ranges[2],
                                 start / tl, end / tl, clamp=True))
        return start_emb, pos.astype(F32)


class SimplePrior:
    """prior.py:27-283, sampling methods only.  `hps` is the dict produced by the
    reference's setup_hparams for this prior; `z_shapes` as make_models.py:150-152."""

    def __init__(self, sd, hps, z_shapes, l_bins, downs_t, strides_t):
        self.sd = sd = {k: np.asarray(v) for k, v in sd.items()}
        self.hps = hps
        level = self.level = hps["level"]
        self.levels = len(z_shapes)
        self.z_shape = z_shapes[level]
        self.n_ctx = int(np.prod(self.z_shape))
        self.x_cond = level != self.levels - 1
        self.y_cond = bool(hps["labels"])
        self.single_enc_dec = bool(hps["single_enc_dec"])
        self.n_tokens = hps["n_tokens"]
        W = hps["prior_width"]
        downs = [s ** d for s, d in zip(strides_t, downs_t)]
        self.cond_downsample = downs[level + 1] if self.x_cond else None
        if self.x_cond:
            self.conditioner = Conditioner(sd, "conditioner_blocks.0.", downs_t[level + 1], strides_t[level + 1],
                                           hps["cond_width"], hps["cond_depth"], hps["cond_m_conv"],
                                           hps["cond_dilation_growth_rate"], hps["cond_dilation_cycle"],
                                           hps["cond_res_scale"], hps["cond_c_res"])
        if self.y_cond:
            self.y_emb = LabelConditioner(sd, "y_emb.", hps["y_bins"], hps["t_bins"], hps["sr"], hps["min_duration"],
                                          hps["max_duration"], self.n_ctx, hps["max_bow_genre_size"],
                                          include_time_signal=not self.x_cond)
        common = dict(width=W, depth=hps["prior_depth"], heads=hps["heads"], attn_order=hps["attn_order"],
                      blocks=hps["blocks"], m_attn=hps["m_attn"], m_mlp=hps["m_mlp"], res_scale=hps["res_scale"])
        if self.single_enc_dec:
            self.prior_dims = [self.n_tokens, self.n_ctx]
            self.prior_bins = [hps["n_vocab"], l_bins]
            self.prior_bins_shift = [0, hps["n_vocab"]]
            self.prior = ConditionalAutoregressive2D(sd, "prior.", (sum(self.prior_dims),), sum(self.prior_bins),
                                                     x_cond=(self.x_cond or self.y_cond), y_cond=True,
                                                     prime_len=self.n_tokens, **common)
        else:
            self.use_tokens = self.n_tokens != 0 and bool(hps["use_tokens"])
            if self.use_tokens:                     # prior.py:104-117
                self.prime_prior = ConditionalAutoregressive2D(
                    sd, "prime_prior.", (self.n_tokens,), hps["n_vocab"], width=hps["prime_width"], depth=hps["prime_depth"],
                    heads=hps["prime_heads"], attn_order=hps["prime_attn_order"], blocks=hps["prime_blocks"],
                    m_attn=hps["prime_m_attn"], m_mlp=hps["prime_m_mlp"], res_scale=hps["prime_res_scale"],
                    x_cond=False, y_cond=False, only_encode=True)
            self.prior = ConditionalAutoregressive2D(sd, "prior.", (self.n_ctx,), l_bins,
                                                     x_cond=(self.x_cond or self.y_cond), y_cond=self.y_cond,
                                                     encoder_dims=self.n_tokens if self.use_tokens else 0,
                                                     merged_decoder=hps["merged_decoder"], **common)

    def get_cond(self, z_conds, y):
        """prior.py:234-243."""
        if y is not None:
            n_labels = y.shape[1] - self.n_tokens
            y, prime = y[:, :n_labels], y[:, n_labels:]
        else:
            prime = None
        y_cond, y_pos = self.y_emb(y) if self.y_cond else (None, None)
        x_cond = self.conditioner(z_conds[0]) if self.x_cond else y_pos
        return x_cond, y_cond, prime

    def sample(self, n_samples, z=None, z_conds=None, y=None, fp16=False, temp=1.0, top_k=0, top_p=0.0,
               chunk_size=None, sample_tokens=None, seed=0):
        """prior.py:245-283."""
        no_past = z is None or z.shape[1] == 0
        x_cond, y_cond, prime = self.get_cond(z_conds, y)
        kw = dict(fp16=fp16, temp=temp, top_k=top_k, top_p=top_p, seed=seed)
        if self.single_enc_dec:
            # prior_preprocess :168-185 -- shift vocab, concat lyric prefix, zero-pad the conditioning
            parts = [np.asarray(prime) + self.prior_bins_shift[0]]
            if not no_past:
                parts.append(np.asarray(z) + self.prior_bins_shift[1])
            zz = np.concatenate(parts, axis=1)
            xc = np.concatenate([np.zeros((n_samples, self.n_tokens, self.hps["prior_width"]), F32), x_cond], axis=1)
            st = None if sample_tokens is None else sample_tokens + self.n_tokens
            out = self.prior.primed_sample(n_samples, zz, xc, y_cond, chunk_size=chunk_size, sample_tokens=st, **kw)
            # prior_postprocess :187-203 -- drop lyric part, un-shift, clamp at 0
            out = out[:, self.n_tokens:] - self.prior_bins_shift[1]
            return np.clip(out, 0, None)
        encoder_kv = self.get_encoder_kv(prime, fp16)
        if no_past:
            return self.prior.sample(n_samples, x_cond, y_cond, encoder_kv, sample_tokens=sample_tokens, **kw)
        return self.prior.primed_sample(n_samples, z, x_cond, y_cond, encoder_kv, chunk_size=chunk_size,
                                        sample_tokens=sample_tokens, **kw)

    def get_encoder_kv(self, prime, fp16=False):
        """prior.py:285-301."""
        if not getattr(self, "use_tokens", False):
            return None
        from .ops import conv1d, r16
        acts = self.prime_prior.forward_logits(np.asarray(prime), None, None, fp16=fp16)
        sd = self.sd
        ekv = layer_norm(conv1d(acts, sd["prime_state_proj.w"], sd["prime_state_proj.b"]),
                         sd["prime_state_ln.weight"], sd["prime_state_ln.bias"]).astype(F32)
        return r16(ekv, fp16)

    def z_forward(self, z, z_conds=(), y=None, fp16=False, get_preds=False):
        """prior.py:312-347 -- (loss, metrics) of a given code sequence: bits per token of the generated part (bpd), of
        the lyric part (prime_loss) and their weighted sum."""
        x_cond, y_cond, prime = self.get_cond(list(z_conds), y)
        hp = self.hps
        if self.single_enc_dec:
            zz = np.concatenate([np.asarray(prime) + self.prior_bins_shift[0], np.asarray(z) + self.prior_bins_shift[1]], axis=1)
            xc = np.concatenate([np.zeros((zz.shape[0], self.n_tokens, hp["prior_width"]), F32), x_cond], axis=1)
            (prime_loss, gen_loss), preds = self.prior.forward(zz, xc, y_cond, fp16=fp16, get_sep_loss=True, get_preds=True)
            prime_dims, gen_dims = self.prior_dims
        else:
            encoder_kv = self.get_encoder_kv(prime, fp16)
            if getattr(self, "use_tokens", False):
                from .ops import r16
                lg = np.matmul(np.asarray(encoder_kv, F32), self.sd["prime_x_out.weight"].T).astype(np.float64)
                lg = lg.reshape(-1, lg.shape[-1])
                m = lg.max(-1, keepdims=True)
                lse = m[:, 0] + np.log(np.exp(lg - m).sum(-1))
                prime_loss = F32((lse - lg[np.arange(lg.shape[0]), np.asarray(prime).reshape(-1)]).mean() / np.log(2.0))
                prime_dims = self.n_tokens
            else:
                prime_loss, prime_dims = F32(0.0), 0
            gen_loss, preds = self.prior.forward(z, x_cond, y_cond, encoder_kv, fp16=fp16, get_preds=True)
            gen_dims = self.n_ctx
        total = prime_dims + gen_dims
        loss = F32(hp["prime_loss_fraction"]) * prime_loss * F32(prime_dims) / F32(total) + gen_loss * F32(gen_dims) / F32(total)
        metrics = dict(bpd=gen_loss, prime_loss=prime_loss, gen_loss=gen_loss)
        if get_preds:
            metrics["preds"] = preds
        return F32(loss), metrics
