"""FactoredAttention with the reference's constructor, parameter names and sampling-mode semantics
(jukebox/transformer/factored_attention.py), computed by the HIP kernels.

The k/v cache is a static (N, cap, n_state) array per layer instead of a growing `t.cat`
(:359-373); `sample_t` counts cached positions exactly as in the reference (:75,235,265).
Only `sample=True` (inference) is implemented -- the training forward is out of scope."""
import torch as t
import torch.nn as nn

from .. import hip_ops as H
from .ops import Conv1D


class FactoredAttention(nn.Module):
    def __init__(self, n_in, n_ctx, n_state, n_head, attn_dropout=0.0, resid_dropout=0.0, scale=True, mask=False,
                 zero_out=False, init_scale=1.0, checkpoint_attn=0, attn_func=0, blocks=None, spread=None,
                 encoder_dims=None, prime_len=None):
        super().__init__()
        assert n_state % n_head == 0
        assert attn_func in (0, 1, 2, 3, 6, 7), f"attn_func {attn_func} has no HIP sampling path"
        self.n_in, self.n_ctx, self.n_state, self.n_head = n_in, n_ctx, n_state, n_head
        if attn_func == 6:           # cross attention to the lyric encoder (factored_attention.py:46-48)
            self.c_attn = Conv1D(n_in, n_state, init_scale=init_scale)
            self.c_enc_kv = Conv1D(n_in, n_state * 2, init_scale=init_scale)
        else:
            self.c_attn = Conv1D(n_in, n_state * 3, init_scale=init_scale)
        self.c_proj = Conv1D(n_state, n_in, zero_out, init_scale=init_scale)
        self.attn_func = attn_func
        self.blocks = blocks
        if blocks is not None:
            assert n_ctx % blocks == 0
            self.block_ctx = n_ctx // blocks
        else:
            self.block_ctx = 0
        self.prime_len = prime_len
        self.encoder_dims = encoder_dims
        self.sample_t = 0
        self.cache = {}

    @property
    def _prime_len(self):
        """:303-308."""
        assert self.prime_len is not None
        return (self.prime_len // self.blocks + 1) * self.blocks

    def _cap(self):
        if self.attn_func == 6:
            return self.encoder_dims
        return self._prime_len if self.attn_func == 7 else self.n_ctx

    def _suff_cache_len(self):
        """:328-353 -- how many positions the reference would keep; here: how many are valid."""
        return min(self.sample_t, self._cap())

    def forward(self, x, encoder_kv=None, sample=False):
        assert sample, "only the sampling-mode forward is implemented on the HIP path"
        N, ql, _ = x.shape
        S = self.n_state
        if self.attn_func == 6:
            # decode_qkv :273-287 -- k/v = c_enc_kv(encoder_kv) once at sample_t == 0, query from c_attn, no mask
            assert encoder_kv is not None and encoder_kv.shape[1] == self.encoder_dims
            if self.sample_t == 0:
                kv = self.c_enc_kv(encoder_kv.to(x.dtype).contiguous())
                self.cache = {"key": kv[..., :S].contiguous(), "value": kv[..., S:].contiguous()}
            q = self.c_attn(x.contiguous())
            a = H.attn_prefill(6, q.contiguous(), self.cache["key"], self.cache["value"], self.n_head, 0, self.sample_t)
            self.sample_t += ql
            return self.c_proj(a)
        assert encoder_kv is None
        if "key" not in self.cache or self.cache["key"].shape[0] != N or self.cache["key"].dtype != x.dtype:
            self.cache = {"key": t.zeros((N, self._cap(), S), dtype=x.dtype, device=x.device),
                          "value": t.zeros((N, self._cap(), S), dtype=x.dtype, device=x.device)}
            assert self.sample_t == 0
        t0 = self.sample_t
        assert t0 + ql <= self.n_ctx, "sampling past n_ctx"
        q = H.gemm_qkv(x.reshape(N * ql, -1).contiguous(), self.c_attn.packed(x.dtype), self.c_attn.b.detach().float(),
                       N, ql, S, self.cache["key"], self.cache["value"], t0)
        a = H.attn_prefill(self.attn_func, q.view(N, ql, S), self.cache["key"], self.cache["value"], self.n_head,
                           self.block_ctx, t0)
        self.sample_t += ql
        return self.c_proj(a)

    def del_cache(self):
        """:375-381."""
        self.sample_t = 0
        self.cache = {}

    def check_cache(self, n_samples, sample_t, fp16):
        """:412-422 adapted to the static cache: position count and dtype."""
        assert self.sample_t == sample_t, f"{self.sample_t} != {sample_t}"
        if sample_t == 0:
            assert self.cache == {} or self.cache["key"].shape[0] == n_samples
        else:
            dtype = {True: t.float16, False: t.float32}[fp16]
            assert self.cache["key"].shape == (n_samples, self._cap(), self.n_state)
            assert self.cache["key"].dtype == dtype and self.cache["value"].dtype == dtype
