"""Transformer / ResAttnBlock / MLP with the reference's module tree (jukebox/transformer/transformer.py) so
released checkpoints load with strict=True; forward(sample=True) runs on the HIP kernels.  The fused
decode loop lives in jukebox_amd.engine.PriorEngine; this module-level forward serves chunked calls
(the reference's check_sample/check_chunks style use) through the same kernels."""
import torch.nn as nn

from .. import _lib as L
from ..engine import attn_funcs
from .factored_attention import FactoredAttention
from .ops import Conv1D, LayerNorm


class MLP(nn.Module):
    def __init__(self, n_in, n_state, resid_dropout=0.0, afn="quick_gelu", zero_out=False, init_scale=1.0):
        super().__init__()
        assert afn == "quick_gelu"
        self.c_fc = Conv1D(n_in, n_state, init_scale=init_scale)
        self.c_proj = Conv1D(n_state, n_in, zero_out, init_scale=init_scale)

    def forward(self, x, res=None):
        return self.c_proj(self.c_fc(x, act=L.ACT_QUICK_GELU), res=res)


class ResAttnBlock(nn.Module):
    def __init__(self, n_in, n_ctx, n_head, attn_dropout=0.0, resid_dropout=0.0, afn="quick_gelu", scale=True,
                 mask=False, zero_out=False, init_scale=1.0, res_scale=1.0, m_attn=0.25, m_mlp=1.0,
                 checkpoint_attn=0, checkpoint_mlp=0, attn_func=0, blocks=None, spread=None, encoder_dims=None,
                 prime_len=None):
        super().__init__()
        self.attn = FactoredAttention(n_in=n_in, n_ctx=n_ctx, n_state=int(m_attn * n_in), n_head=n_head,
                                      zero_out=zero_out, init_scale=init_scale, attn_func=attn_func, blocks=blocks,
                                      spread=spread, encoder_dims=encoder_dims, prime_len=prime_len)
        self.ln_0 = LayerNorm(n_in)
        self.mlp = MLP(n_in=n_in, n_state=int(m_mlp * n_in), afn=afn, zero_out=zero_out, init_scale=init_scale)
        self.ln_1 = LayerNorm(n_in)
        self.res_scale = res_scale
        self.n_in, self.attn_func = n_in, attn_func

    def forward(self, x, encoder_kv, sample=False):
        """transformer.py:62-66,82-86 (sample branch): h = x + a + m."""
        assert sample and self.res_scale == 1.0
        a = self.attn(self.ln_0(x), encoder_kv, sample)
        xa = x + a
        return self.mlp(self.ln_1(xa), res=xa)


class Transformer(nn.Module):
    def __init__(self, n_in, n_ctx, n_head, n_depth, attn_dropout=0.0, resid_dropout=0.0, afn="quick_gelu",
                 scale=True, mask=False, zero_out=False, init_scale=1.0, res_scale=False, m_attn=0.25, m_mlp=1.0,
                 checkpoint_attn=0, checkpoint_mlp=0, checkpoint_res=0, attn_order=0, blocks=None, spread=None,
                 encoder_dims=None, prime_len=None):
        super().__init__()
        self.n_in, self.n_ctx, self.n_head, self.n_depth = n_in, n_ctx, n_head, n_depth
        self.encoder_dims, self.blocks, self.prime_len = encoder_dims, blocks, prime_len
        if blocks is not None:
            assert n_ctx % blocks == 0
            self.block_ctx = n_ctx // blocks
        self.attn_order, self.m_attn, self.m_mlp = attn_order, m_attn, m_mlp
        rs = 1.0 / n_depth if res_scale else 1.0
        funcs = attn_funcs(attn_order, n_depth)
        self._attn_mods = nn.ModuleList(
            ResAttnBlock(n_in=n_in, n_ctx=n_ctx, n_head=n_head, afn=afn, zero_out=zero_out if funcs[d] != 6 else True,
                         init_scale=init_scale, res_scale=rs, m_attn=m_attn, m_mlp=m_mlp, attn_func=funcs[d],
                         blocks=blocks, spread=spread, encoder_dims=encoder_dims, prime_len=prime_len)
            for d in range(n_depth))

    def forward(self, x, encoder_kv=None, sample=False, fp16=False, fp16_out=False):
        """transformer.py:169-192."""
        assert sample, "only the sampling-mode forward is implemented on the HIP path"
        x = x.half() if fp16 else x.float()
        for l in self._attn_mods:
            x = l(x, encoder_kv=encoder_kv if l.attn_func == 6 else None, sample=True)
        return x if fp16_out else x.float()

    def check_cache(self, n_samples, sample_t, fp16):
        for l in self._attn_mods:
            l.attn.check_cache(n_samples, sample_t, fp16)

    def del_cache(self):
        for l in self._attn_mods:
            l.attn.del_cache()
