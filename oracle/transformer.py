"""Oracle restatement of jukebox/transformer/{factored_attention,transformer}.py (numpy).

Test infrastructure only.  The sampling-mode forward (`sample=True`) is restated as
"attend over a closed-form key set per pattern with a full-length k/v cache"
(SURVEY.md Appendix B), which is what the reference's cache trimming + padding +
masking computes (factored_attention.py:135-193,230-271,310-353); the golden
fixtures generated from the reference pin that equivalence.
"""
import math

import numpy as np

from .ops import F32, conv1d, layer_norm, quick_gelu, r16, softmax

# transformer.py:110-126 -- attn_order -> per-layer attn_func
_ORDERS = {
    0: lambda d: 0,
    1: lambda d: [1, 2][d % 2],
    2: lambda d: [1, 2, 3][d % 3],
    3: lambda d: [1, 4][d % 2],
    4: lambda d: [1, 5][d % 2],
    5: lambda d: [1, 4, 1, 1][d % 4],
    6: lambda d: [1, 2, 3, 6][d % 4],
    7: lambda d: [*[1, 2, 3] * 5, 6][d % 16],
    8: lambda d: [1, 2, 3, 1, 2, 3, 1, 2, 3, 6][d % 10],
    9: lambda d: [1, 2, 3, 0][d % 4],
    10: lambda d: [*[1, 2, 3, 1, 2, 3, 1, 2, 3], *[1, 2, 3, 1, 2, 3, 1, 2, 3, 6] * 7][d % 79],
    11: lambda d: [6, 6, 0][d % 3] if d % 16 == 15 else [1, 2, 3][d % 3],
    12: lambda d: [7, 7, 0][d % 3] if d % 16 == 15 else [1, 2, 3][d % 3],
}


def attn_func_of_layer(attn_order, d):
    return _ORDERS[attn_order](d)


def rounded_prime_len(prime_len, blocks):
    """factored_attention.py:303-308 -- (prime_len // blocks + 1) * blocks."""
    return (prime_len // blocks + 1) * blocks


def allowed_keys(attn_func, p, j, bc, prime_len_r):
    """Boolean (len(p), len(j)): may query position p read key position j?
    dense :123-133 / block :135-150 / transpose :152-165 / prev :167-193 / prime :220-224."""
    p = np.asarray(p)[:, None]
    j = np.asarray(j)[None, :]
    if attn_func == 0:
        return j <= p
    if attn_func == 1:
        return (j <= p) & (j // bc == p // bc)
    if attn_func == 2:
        return (j <= p) & (j % bc == p % bc)
    if attn_func == 3:
        return j // bc == p // bc - 1
    if attn_func == 7:
        return (j <= p) & (j < prime_len_r)
    raise NotImplementedError(attn_func)


def decode_key_index(attn_func, p, bc, prime_len_r):
    """Key positions read by the single query at position p (q_l == 1, no mask,
    factored_attention.py:17-18).  Returns None for prev-block attention inside
    block 0: softmax over `block_ctx` all-zero rows times zero values == 0 (:178-181)."""
    if attn_func == 0:
        return np.arange(0, p + 1)
    if attn_func == 1:
        return np.arange((p // bc) * bc, p + 1)
    if attn_func == 2:
        return np.arange(p % bc, p + 1, bc)
    if attn_func == 3:
        blk = p // bc
        if blk == 0:
            return None
        return np.arange((blk - 1) * bc, blk * bc)
    if attn_func == 7:
        return np.arange(0, min(p + 1, prime_len_r))
    raise NotImplementedError(attn_func)


class Transformer:
    """transformer.py:88-200 restated over a reference-named state dict.

    cfg keys: n_in, n_ctx, n_head, n_depth, attn_order, blocks, m_attn, m_mlp,
    encoder_dims, prime_len, res_scale (bool, as passed to the reference ctor).
    """

    def __init__(self, sd, prefix, n_in, n_ctx, n_head, n_depth, attn_order=0, blocks=None,
                 m_attn=0.25, m_mlp=1.0, encoder_dims=None, prime_len=None, res_scale=False, funcs=None):
        self.n_in, self.n_ctx, self.n_head, self.n_depth = n_in, n_ctx, n_head, n_depth
        self.n_state = int(m_attn * n_in)
        self.blocks = blocks
        self.block_ctx = n_ctx // blocks if blocks is not None else None
        self.prime_len_r = rounded_prime_len(prime_len, blocks) if prime_len else None
        self.encoder_dims = encoder_dims
        self.res_scale = 1.0 / n_depth if res_scale else 1.0
        self.funcs = list(funcs) if funcs is not None else [attn_func_of_layer(attn_order, d) for d in range(n_depth)]
        g = lambda name: np.asarray(sd[prefix + name], dtype=F32)
        self.record, self.ws = set(), {}
        self.prime_len = prime_len
        self.layers = []
        for d in range(n_depth):
            p = f"_attn_mods.{d}."
            lay = dict(
                c_attn_w=g(p + "attn.c_attn.w"), c_attn_b=g(p + "attn.c_attn.b"),
                c_proj_w=g(p + "attn.c_proj.w"), c_proj_b=g(p + "attn.c_proj.b"),
                ln0_w=g(p + "ln_0.weight"), ln0_b=g(p + "ln_0.bias"),
                fc_w=g(p + "mlp.c_fc.w"), fc_b=g(p + "mlp.c_fc.b"),
                proj_w=g(p + "mlp.c_proj.w"), proj_b=g(p + "mlp.c_proj.b"),
                ln1_w=g(p + "ln_1.weight"), ln1_b=g(p + "ln_1.bias"),
            )
            if self.funcs[d] == 6:
                lay["c_enc_kv_w"] = g(p + "attn.c_enc_kv.w")
                lay["c_enc_kv_b"] = g(p + "attn.c_enc_kv.b")
            self.layers.append(lay)
        self.del_cache()

    def set_record_attn(self, layers):
        """transformer.py:141-167 -- layers whose softmax output is kept (None / empty set: off).  For a prime layer
        only music queries x lyric keys are kept (factored_attention.py:101-105)."""
        self.record = set(layers) if layers else set()
        self.ws = {}

    def del_cache(self):
        """factored_attention.py:375-381."""
        self.sample_t = 0
        self.k = [None] * self.n_depth
        self.v = [None] * self.n_depth

    # -- attention core (factored_attention.py:82-108) for one head-batch --
    def _attend(self, q, k, v, mask, fp16):
        # q (N,H,ql,d)  k (N,H,kl,d)  v (N,H,kl,d); mask (ql,kl) bool or None
        d = q.shape[-1]
        scale = 1.0 / math.sqrt(math.sqrt(d))
        w = r16(np.matmul(q, np.swapaxes(k, -1, -2)), fp16)
        w = r16(w * F32(scale * scale), fp16)
        if mask is not None:
            w = np.where(mask[None, None], w, F32(-1e9))
        pr = r16(softmax(w, axis=-1), fp16)
        if mask is not None:
            # rows with no admissible key (prev-block pattern inside block 0) attend
            # to all-zero values in the reference -> exact zeros (:178-181,:185-186)
            empty = ~mask.any(axis=1)
            if empty.any():
                pr = pr.copy()
                pr[:, :, empty, :] = 0.0
        self._last_pr = pr
        return r16(np.matmul(pr, v), fp16)

    def _split(self, x):
        N, l, S = x.shape
        return x.reshape(N, l, self.n_head, S // self.n_head).transpose(0, 2, 1, 3)

    def _merge(self, a):
        N, H, l, d = a.shape
        return a.transpose(0, 2, 1, 3).reshape(N, l, H * d)

    def _attention(self, d, h, t0, fp16, encoder_kv):
        """FactoredAttention.forward(sample=True) (factored_attention.py:289-301) for the
        q_l = h.shape[1] positions t0 .. t0+q_l-1."""
        lay, func = self.layers[d], self.funcs[d]
        N, ql, _ = h.shape
        S = self.n_state
        x = conv1d(h, lay["c_attn_w"], lay["c_attn_b"], fp16)
        if func == 6:
            # decode_qkv :273-287 -- k/v = c_enc_kv(encoder_kv), computed once at t == 0
            if self.k[d] is None:
                ekv = conv1d(r16(encoder_kv, fp16), lay["c_enc_kv_w"], lay["c_enc_kv_b"], fp16)
                self.k[d], self.v[d] = ekv[..., :S], ekv[..., S:]
            a = self._attend(self._split(x), self._split(self.k[d]), self._split(self.v[d]), None, fp16)
            a = self._merge(a)
        else:
            q, k, v = x[..., :S], x[..., S:2 * S], x[..., 2 * S:]
            cap = self.prime_len_r if func == 7 else None
            if self.k[d] is None:
                self.k[d] = np.zeros((N, 0, S), F32)
                self.v[d] = np.zeros((N, 0, S), F32)
            if cap is None or self.k[d].shape[1] < cap:
                # _append_cache :359-373 (prime layers stop at _prime_len, :260-263)
                self.k[d] = np.concatenate([self.k[d], k], axis=1)[:, :cap]
                self.v[d] = np.concatenate([self.v[d], v], axis=1)[:, :cap]
            K, V = self.k[d], self.v[d]
            bc = self.block_ctx
            if ql == 1:
                idx = decode_key_index(func, t0, bc, self.prime_len_r)
                if idx is None:
                    a = np.zeros((N, 1, S), F32)
                else:
                    a = self._merge(self._attend(self._split(q), self._split(K[:, idx]),
                                                 self._split(V[:, idx]), None, fp16))
            else:
                p = np.arange(t0, t0 + ql)
                kl = K.shape[1]
                j = np.arange(kl)
                mask = allowed_keys(func, p, j, bc, self.prime_len_r)
                cols = np.nonzero(mask.any(axis=0))[0]
                if len(cols) == 0:
                    a = np.zeros((N, ql, S), F32)
                else:
                    lo, hi = cols[0], cols[-1] + 1
                    a = self._merge(self._attend(self._split(q), self._split(K[:, lo:hi]),
                                                 self._split(V[:, lo:hi]), mask[:, lo:hi], fp16))
                    if d in self.record:
                        w = np.zeros((*self._last_pr.shape[:3], kl), F32)
                        w[..., lo:hi] = self._last_pr
                        if func == 7:
                            assert t0 == 0, "attention recording expects one full-length chunk"
                            w = w[:, :, self.prime_len:, :self.prime_len]
                        self.ws[d] = w
        return conv1d(a, lay["c_proj_w"], lay["c_proj_b"], fp16)

    def forward(self, x, encoder_kv=None, fp16=False, t0=None):
        """Transformer.forward(sample=True) (transformer.py:169-192) + ResAttnBlock sample
        branch (:62-66,82-86).  x: (N, q_l, n_in) float32 at positions sample_t.."""
        if t0 is None:
            t0 = self.sample_t
        assert t0 == self.sample_t, "chunks must be fed in order"
        x = r16(x, fp16)
        for d, lay in enumerate(self.layers):
            a = self._attention(d, r16(layer_norm(x, lay["ln0_w"], lay["ln0_b"]), fp16), t0, fp16, encoder_kv)
            xa = r16(x + a, fp16)
            h1 = r16(layer_norm(xa, lay["ln1_w"], lay["ln1_b"]), fp16)
            m = quick_gelu(conv1d(h1, lay["fc_w"], lay["fc_b"], fp16), fp16)
            m = conv1d(m, lay["proj_w"], lay["proj_b"], fp16)
            if self.res_scale == 1.0:
                x = r16(xa + m, fp16)          # h = x + a + m   (:82-83)
            else:
                x = r16(x + r16(F32(self.res_scale) * r16(a + m, fp16), fp16), fp16)
        self.sample_t = t0 + x.shape[1]
        return x.astype(F32)
