"""Oracle restatement of jukebox/prior/autoregressive.py (numpy).  Test infrastructure only."""
import numpy as np

from .ops import F32, filter_logits, softmax
from .transformer import Transformer


def split_chunks(length, chunk_size):
    """autoregressive.py:19-23."""
    n_passes = (length + chunk_size - 1) // chunk_size
    sizes = [*[chunk_size] * (n_passes - 1), (length - 1) % chunk_size + 1]
    assert sum(sizes) == length
    return sizes


class ConditionalAutoregressive2D:
    """autoregressive.py:48-359 restated over a reference-named state dict (prefix e.g. 'prior.')."""

    def __init__(self, sd, prefix, input_shape, bins, width, depth, heads, attn_order=0, blocks=None,
                 x_cond=False, y_cond=False, m_attn=0.25, m_mlp=1.0, encoder_dims=0, only_encode=False,
                 merged_decoder=False, prime_len=None, res_scale=False):
        self.input_dims = int(np.prod(input_shape))
        self.bins, self.width = bins, width
        self.x_cond, self.y_cond = x_cond, y_cond
        self.only_encode = only_encode
        self.prime_len = prime_len
        g = lambda n: np.asarray(sd[prefix + n], dtype=F32)
        self.x_emb = g("x_emb.weight")
        self.pos_emb = g("pos_emb.pos_emb")
        self.start_token = None if y_cond else g("start_token")
        self.transformer = Transformer(sd, prefix + "transformer.", n_in=width, n_ctx=self.input_dims, n_head=heads,
                                       n_depth=depth, attn_order=attn_order, blocks=blocks, m_attn=m_attn,
                                       m_mlp=m_mlp, encoder_dims=encoder_dims, prime_len=prime_len,
                                       res_scale=res_scale)
        self.add_cond_after_transformer = not merged_decoder      # :87-93
        if not only_encode:
            self.x_out = g("x_out.weight")                        # tied to x_emb unless merged_decoder (:95-98)

    def get_emb(self, sample_t, n_samples, x, x_cond, y_cond):
        """:177-197 -- start token / token embedding + pos_emb[t] + x_cond[:, t]."""
        if sample_t == 0:
            if self.y_cond:
                e = np.asarray(y_cond, F32).reshape(n_samples, 1, self.width).copy()
            else:
                e = np.broadcast_to(self.start_token.reshape(1, 1, self.width), (n_samples, 1, self.width)).copy()
        else:
            e = self.x_emb[np.asarray(x).reshape(n_samples)][:, None, :]
        if x_cond.shape[1] == self.input_dims:
            cond = x_cond[:, sample_t:sample_t + 1, :]
        else:
            cond = x_cond
        e = (e + self.pos_emb[sample_t:sample_t + 1][None]).astype(F32) + cond
        return e.astype(F32), cond

    def _draw(self, logits, temp, top_k, top_p, rng):
        """:233-235 -- temperature, filter_logits, Categorical(logits).sample().
        Greedy (top_k == 1) is deterministic: the surviving set is {argmax} (ties
        broken towards the lowest index here)."""
        x = (logits / F32(temp)).astype(F32)
        x = filter_logits(x, top_k=top_k, top_p=top_p)
        if top_k == 1:
            return x.argmax(axis=-1)
        p = softmax(x, axis=-1).astype(np.float64)
        p /= p.sum(axis=-1, keepdims=True)
        flat = p.reshape(-1, p.shape[-1])
        out = np.array([rng.choice(flat.shape[1], p=row) for row in flat])
        return out.reshape(p.shape[:-1])

    def _prep(self, n_samples, x_cond, y_cond):
        if not self.x_cond:
            x_cond = np.zeros((n_samples, 1, self.width), F32)            # :217-219
        return np.asarray(x_cond, F32), (None if y_cond is None else np.asarray(y_cond, F32))

    def sample(self, n_samples, x_cond=None, y_cond=None, encoder_kv=None, fp16=False, temp=1.0, top_k=0,
               top_p=0.0, get_preds=False, sample_tokens=None, seed=0):
        """:199-249."""
        if sample_tokens is None:
            sample_tokens = self.input_dims
        x_cond, y_cond = self._prep(n_samples, x_cond, y_cond)
        rng = np.random.default_rng(seed)
        tr = self.transformer
        tr.del_cache()
        xs, preds, x = [], [], None
        for t in range(sample_tokens):
            e, cond = self.get_emb(t, n_samples, x, x_cond, y_cond)
            h = tr.forward(e, encoder_kv=encoder_kv, fp16=fp16)
            if self.add_cond_after_transformer:
                h = h + cond
            logits = np.matmul(h, self.x_out.T).astype(F32)          # :229
            if get_preds:
                preds.append(logits.copy())
            x = self._draw(logits, temp, top_k, top_p, rng)           # (N,1)
            xs.append(x.copy())
        tr.del_cache()
        z = np.concatenate(xs, axis=1).astype(np.int64)
        return (z, np.concatenate(preds, axis=1)) if get_preds else z

    def primed_sample(self, n_samples, x, x_cond=None, y_cond=None, encoder_kv=None, fp16=False, temp=1.0,
                      top_k=0, top_p=0.0, get_preds=False, chunk_size=None, sample_tokens=None, seed=0):
        """:251-359 -- chunked prefill of the given tokens (fills the k/v caches), then the
        per-token loop from t = len(prime)."""
        if sample_tokens is None:
            sample_tokens = self.input_dims
        x = np.asarray(x).reshape(n_samples, -1).astype(np.int64)
        n_prime = x.shape[1]
        assert n_prime < sample_tokens
        x_cond, y_cond = self._prep(n_samples, x_cond, y_cond)
        rng = np.random.default_rng(seed)
        tr = self.transformer
        tr.del_cache()
        xs = [x[:, i:i + 1] for i in range(n_prime)]
        preds = []
        if chunk_size is None:
            chunk_size = n_prime
        start, prev, h_primes = 0, None, []
        for cur in split_chunks(n_prime, chunk_size):
            es, conds = [], []
            for t in range(start, start + cur):
                e, cond = self.get_emb(t, n_samples, prev, x_cond, y_cond)
                prev = xs[t]
                es.append(e)
                conds.append(np.broadcast_to(cond, e.shape))
            start += cur
            h = tr.forward(np.concatenate(es, axis=1), encoder_kv=encoder_kv, fp16=fp16)
            if get_preds:
                if self.add_cond_after_transformer:
                    h = h + np.concatenate(conds, axis=1)
                h_primes.append(h)
        if get_preds:
            preds.append(np.matmul(np.concatenate(h_primes, axis=1), self.x_out.T).astype(F32))
        tok = xs[-1]
        for t in range(n_prime, sample_tokens):
            e, cond = self.get_emb(t, n_samples, tok, x_cond, y_cond)
            h = tr.forward(e, encoder_kv=encoder_kv, fp16=fp16)
            if self.add_cond_after_transformer:
                h = h + cond
            logits = np.matmul(h, self.x_out.T).astype(F32)
            if get_preds:
                preds.append(logits.copy())
            tok = self._draw(logits, temp, top_k, top_p, rng)
            xs.append(tok.copy())
        tr.del_cache()
        z = np.concatenate(xs, axis=1).astype(np.int64)
        return (z, np.concatenate(preds, axis=1)) if get_preds else z

    def forward_logits(self, x, x_cond=None, y_cond=None, encoder_kv=None, fp16=False):
        """Teacher-forced logits for a full sequence (the `get_preds` output of forward,
        :114-175): position t sees tokens < t.  Runs as one prefill chunk."""
        x = np.asarray(x).reshape(x.shape[0], -1).astype(np.int64)
        N, D = x.shape
        x_cond, y_cond = self._prep(N, x_cond, y_cond)
        tr = self.transformer
        tr.del_cache()
        es, conds, prev = [], [], None
        for t in range(D):
            e, cond = self.get_emb(t, N, prev, x_cond, y_cond)
            prev = x[:, t:t + 1]
            es.append(e)
            conds.append(np.broadcast_to(cond, e.shape))
        h = tr.forward(np.concatenate(es, axis=1), encoder_kv=encoder_kv, fp16=fp16)
        tr.del_cache()
        if self.add_cond_after_transformer:
            h = h + np.concatenate(conds, axis=1)
        if self.only_encode:
            return h
        return np.matmul(h, self.x_out.T).astype(F32)

    def forward(self, x, x_cond=None, y_cond=None, encoder_kv=None, fp16=False, get_preds=False, get_sep_loss=False):
        """autoregressive.py:114-175 -- teacher-forced loss in bits per token (cross entropy / ln 2); with get_sep_loss the
        (prime, generated) parts separately, split at prime_len."""
        x = np.asarray(x).reshape(np.asarray(x).shape[0], -1).astype(np.int64)
        logits = self.forward_logits(x, x_cond, y_cond, encoder_kv, fp16)
        if self.only_encode:
            return logits

        def ce(lg, tg):
            lg = lg.reshape(-1, lg.shape[-1]).astype(np.float64)
            m = lg.max(-1, keepdims=True)
            lse = (m[:, 0] + np.log(np.exp(lg - m).sum(-1)))
            return F32((lse - lg[np.arange(lg.shape[0]), tg.reshape(-1)]).mean() / np.log(2.0))
        if get_sep_loss:
            pl = self.prime_len
            loss = (ce(logits[:, :pl], x[:, :pl]), ce(logits[:, pl:], x[:, pl:]))
        else:
            loss = ce(logits, x)
        return (loss, logits) if get_preds else (loss, None)
