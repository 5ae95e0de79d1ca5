"""Modules with the reference's names and parameter layout (jukebox/transformer/ops.py) whose forward
passes run on the HIP kernels.  There is no eager fallback: inputs must be GPU tensors."""
import numpy as np
import torch as t
import torch.nn as nn

from .. import _lib as L
from .. import hip_ops as H


class LayerNorm(nn.LayerNorm):
    """ops.py:14-24 -- fp32 statistics, result cast back to the input dtype.  Parameter names `weight`/`bias`
    as in torch.nn.LayerNorm / apex FusedLayerNorm so checkpoints load unchanged."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__(normalized_shape, eps=eps, elementwise_affine=elementwise_affine)
        self.width = int(np.prod(normalized_shape))

    def forward(self, input):
        x = input.contiguous()
        return H.layernorm(x, self.weight.float(), self.bias.float(), self.eps, out_dtype=input.dtype)


class Conv1D(nn.Module):
    """ops.py:83-101 -- y = x @ w + b with w stored (n_in, n_out).  The MFMA-ordered copy of `w` is built
    lazily per activation dtype (replacing the per-call `w.type_as(x)` cast) and dropped when the module moves."""

    def __init__(self, n_in, n_out, zero_out=False, init_scale=1.0):
        super().__init__()
        self.n_in, self.n_out = n_in, n_out
        if zero_out:
            w = t.zeros(n_in, n_out)
        else:
            w = t.empty(n_in, n_out)
            nn.init.normal_(w, std=0.02 * init_scale)
        self.w = nn.Parameter(w)
        self.b = nn.Parameter(t.zeros(n_out))
        self._packed = {}

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def packed(self, dtype):
        key = (dtype, self.w.data_ptr(), self.w._version)
        if key not in self._packed:
            self._packed = {key: H.pack_conv1d_w(self.w.detach().contiguous(), dtype)}
        return self._packed[key]

    def forward(self, x, act=L.ACT_NONE, res=None):
        size_out = (*x.size()[:-1], self.n_out)
        x2 = x.reshape(-1, x.size(-1)).contiguous()
        r2 = None if res is None else res.reshape(-1, self.n_out).contiguous()
        y = H.gemm(x2, self.packed(x.dtype), bias=self.b.detach().float(), res=r2, act=act)
        return y.view(*size_out)


def _convert_conv_weights_to_fp16(l):
    if isinstance(l, Conv1D):
        l.w.data = l.w.data.half()


def _convert_conv_weights_to_fp32(l):
    if isinstance(l, Conv1D):
        l.w.data = l.w.data.float()
