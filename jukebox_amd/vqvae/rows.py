"""Channels-last helpers shared by the conv stacks: every tensor is a row matrix (N*T, C) (the reference's
NTC flattened), which is what the HIP tap-GEMM consumes; NCT never materialises."""
import torch

from .. import hip_ops as H

SPLIT_F16 = True        # (tests switch it off to compare with the exact-fp32 kernel)


class PackedConvCache:
    """Mixin: lazily packed (MFMA-ordered, fp32) copies of nn.Conv1d / nn.ConvTranspose1d weights, dropped
    whenever the module is moved or cast (`_apply`) or a weight is updated in place.  Layers with a multiple of 32 input
    channels are packed as hi / lo pairs of f16 images: their fp32 convolution runs on the f16 matrix cores at fp32
    accuracy (jb_gemm_args.w_split); the others (the 1-channel audio input, tiny test models) keep the exact-fp32 kernel."""

    def packed(self, conv, transposed=False):
        cache = self.__dict__.setdefault("_packed_cache", {})
        key = (id(conv), conv.weight.data_ptr(), conv.weight._version)
        if key not in cache:
            cache[key] = H.pack_conv_taps(conv.weight.detach().float(), torch.float32, transposed=transposed,
                                          split=SPLIT_F16 and conv.in_channels % 32 == 0)
        return cache[key]

    def _apply(self, fn, *a, **k):
        self.__dict__["_packed_cache"] = {}
        return super()._apply(fn, *a, **k)


def conv_k3(owner, conv, x, N, T):
    """nn.Conv1d(k=3, s=1, p=1) on rows."""
    return H.gemm(x, owner.packed(conv), bias=conv.bias.detach(), n_seq=N, t_in=T, shifts=(-1, 0, 1))


def conv_down(owner, conv, x, N, T, stride):
    """nn.Conv1d(k=2*stride, s=stride, p=stride//2) on rows -> T // stride positions."""
    k, pad = conv.kernel_size[0], conv.padding[0]
    assert k <= 4, "tap GEMM takes up to 4 taps"
    return H.gemm(x, owner.packed(conv), bias=conv.bias.detach(), n_seq=N, t_in=T, t_out=T // stride, in_stride=stride,
                  shifts=tuple(i - pad for i in range(k)))


def conv_transpose_up(owner, conv, x, N, T):
    """nn.ConvTranspose1d(k=4, s=2, p=1) on rows -> 2T positions: even outputs use taps (1, 3) of inputs
    (t, t-1), odd outputs taps (0, 2) of inputs (t+1, t)."""
    assert conv.kernel_size[0] == 4 and conv.stride[0] == 2 and conv.padding[0] == 1
    pw = owner.packed(conv, transposed=True)
    out = torch.empty((N * 2 * T, conv.out_channels), dtype=x.dtype, device=x.device)
    H.gemm(x, H.tap_view(pw, [1, 3]), bias=conv.bias.detach(), out=out, n_seq=N, t_in=T, t_out=T, shifts=(0, -1),
           out_stride=2, out_offset=0, out_rows_per_seq=2 * T)
    H.gemm(x, H.tap_view(pw, [0, 2]), bias=conv.bias.detach(), out=out, n_seq=N, t_in=T, t_out=T, shifts=(1, 0),
           out_stride=2, out_offset=1, out_rows_per_seq=2 * T)
    return out
