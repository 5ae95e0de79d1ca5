"""Oracle restatement of jukebox/vqvae/{resnet,encdec,bottleneck,vqvae}.py (numpy, float32, NCT
layout like the reference).  Test infrastructure only."""
import math

import numpy as np

from .ops import F32


def conv1d_nct(x, w, b, stride=1, padding=0, dilation=1):
    """torch.nn.Conv1d forward (used at encdec.py:17,21,35,108; resnet.py:33,35).
    x (N,Ci,T), w (Co,Ci,k), b (Co,)."""
    N, Ci, T = x.shape
    Co, _, k = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding))).astype(F32)
    To = (T + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    out = np.zeros((N, Co, To), F32)
    for tap in range(k):
        xs = xp[:, :, tap * dilation: tap * dilation + (To - 1) * stride + 1: stride]
        out += np.matmul(w[:, :, tap].astype(F32), xs)
    return out + b.astype(F32)[None, :, None]


def conv_transpose1d_nct(x, w, b, stride=2, padding=1):
    """torch.nn.ConvTranspose1d forward (encdec.py:40).  x (N,Ci,T), w (Ci,Co,k)."""
    N, Ci, T = x.shape
    _, Co, k = w.shape
    full = np.zeros((N, Co, (T - 1) * stride + k), F32)
    for tap in range(k):
        full[:, :, tap: tap + (T - 1) * stride + 1: stride] += np.matmul(w[:, :, tap].T.astype(F32), x)
    To = (T - 1) * stride - 2 * padding + k
    return full[:, :, padding: padding + To] + b.astype(F32)[None, :, None]


def relu(x):
    return np.maximum(x, F32(0))


class Resnet1D:
    """resnet.py:27-75 -- n_depth x [x + res_scale * conv1x1(relu(conv3_dil(relu(x))))].
    Param names: `<prefix>model.{d}.model.{1,3}.*` or `<prefix>blocks.{d}.model.{1,3}.*`
    when checkpoint_res == 1 (resnet.py:62-67)."""

    def __init__(self, sd, prefix, n_in, n_depth, m_conv=1.0, dilation_growth_rate=1, dilation_cycle=None,
                 res_scale=False, reverse_dilation=False, checkpoint_res=False):
        container = "blocks" if checkpoint_res == 1 else "model"
        dils = [dilation_growth_rate ** (d if dilation_cycle is None else d % dilation_cycle) for d in range(n_depth)]
        if reverse_dilation:
            dils = dils[::-1]
        self.res_scale = 1.0 if not res_scale else 1.0 / math.sqrt(n_depth)
        self.blocks = []
        for d in range(n_depth):
            p = f"{prefix}{container}.{d}.model."
            self.blocks.append((dils[d], sd[p + "1.weight"], sd[p + "1.bias"], sd[p + "3.weight"], sd[p + "3.bias"]))

    def __call__(self, x):
        for dil, w1, b1, w3, b3 in self.blocks:
            h = conv1d_nct(relu(x), w1, b1, 1, dil, dil)
            h = conv1d_nct(relu(h), w3, b3, 1, 0, 1)
            x = x + F32(self.res_scale) * h
        return x


class DecoderConvBock:
    """encdec.py:28-46."""

    def __init__(self, sd, prefix, down_t, stride_t, width, depth, m_conv, dilation_growth_rate=1,
                 dilation_cycle=None, res_scale=False, reverse_decoder_dilation=False, checkpoint_res=False, **_):
        self.sd, self.prefix, self.down_t, self.stride_t = sd, prefix, down_t, stride_t
        self.res = [Resnet1D(sd, f"{prefix}model.{1 + i}.0.", width, depth, m_conv, dilation_growth_rate,
                             dilation_cycle, res_scale, reverse_decoder_dilation, checkpoint_res)
                    for i in range(down_t)]

    def __call__(self, x):
        sd, p = self.sd, self.prefix
        if self.down_t == 0:
            return x
        x = conv1d_nct(x, sd[p + "model.0.weight"], sd[p + "model.0.bias"], 1, 1, 1)
        for i in range(self.down_t):
            x = self.res[i](x)
            x = conv_transpose1d_nct(x, sd[f"{p}model.{1 + i}.1.weight"], sd[f"{p}model.{1 + i}.1.bias"],
                                     self.stride_t, self.stride_t // 2)
        return x


class EncoderConvBlock:
    """encdec.py:6-26."""

    def __init__(self, sd, prefix, down_t, stride_t, width, depth, m_conv, dilation_growth_rate=1,
                 dilation_cycle=None, res_scale=False, **_):
        self.sd, self.prefix, self.down_t, self.stride_t = sd, prefix, down_t, stride_t
        self.res = [Resnet1D(sd, f"{prefix}model.{i}.1.", width, depth, m_conv, dilation_growth_rate,
                             dilation_cycle, res_scale) for i in range(down_t)]

    def __call__(self, x):
        sd, p = self.sd, self.prefix
        for i in range(self.down_t):
            x = conv1d_nct(x, sd[f"{p}model.{i}.0.weight"], sd[f"{p}model.{i}.0.bias"],
                           self.stride_t, self.stride_t // 2, 1)
            x = self.res[i](x)
        return conv1d_nct(x, sd[f"{p}model.{self.down_t}.weight"], sd[f"{p}model.{self.down_t}.bias"], 1, 1, 1)


class VQVAE:
    """vqvae.py:42-144 (encode / decode only) over a reference-named state dict.
    hps keys: levels, downs_t, strides_t, emb_width, l_bins, width, depth, m_conv,
    dilation_growth_rate, dilation_cycle, vqvae_reverse_decoder_dilation, hvqvae_multipliers."""

    def __init__(self, sd, hps):
        self.sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
        sd = self.sd
        self.levels = hps["levels"]
        self.downs_t, self.strides_t = tuple(hps["downs_t"]), tuple(hps["strides_t"])
        mult = hps.get("hvqvae_multipliers") or [1] * self.levels
        self.encoders, self.decoders = [], []
        for level in range(self.levels):
            kw = dict(width=hps["width"] * mult[level], depth=hps["depth"] * mult[level], m_conv=hps["m_conv"],
                      dilation_growth_rate=hps["dilation_growth_rate"], dilation_cycle=hps.get("dilation_cycle"))
            enc = [EncoderConvBlock(sd, f"encoders.{level}.level_blocks.{i}.", self.downs_t[i], self.strides_t[i], **kw)
                   for i in range(level + 1)]
            dec = [DecoderConvBock(sd, f"decoders.{level}.level_blocks.{i}.", self.downs_t[i], self.strides_t[i],
                                   reverse_decoder_dilation=hps.get("vqvae_reverse_decoder_dilation", True), **kw)
                   for i in range(level + 1)]
            self.encoders.append(enc)
            self.decoders.append(dec)

    def codebook(self, level):
        return self.sd[f"bottleneck.level_blocks.{level}.k"]

    def quantise(self, x, level):
        """bottleneck.py:112-119 -- argmin_j ||x||^2 - 2 x.k_j + ||k_j||^2.  x (M, emb)."""
        k_w = self.codebook(level).T
        dist = (x ** 2).sum(-1, keepdims=True) - 2 * np.matmul(x, k_w) + (k_w ** 2).sum(0, keepdims=True)
        return dist.argmin(-1), dist

    def encode(self, x, start_level=0, end_level=None):
        """vqvae.py:124-144 + Encoder.forward encdec.py:70-85 + BottleneckBlock.encode
        bottleneck.py:125-136.  x (N,T,1) -> [codes (N, T_l)]."""
        if end_level is None:
            end_level = self.levels
        x_in = np.transpose(np.asarray(x, F32), (0, 2, 1))
        zs = []
        for level in range(self.levels):
            h = x_in
            for blk in self.encoders[level]:
                h = blk(h)
            N, Wd, T = h.shape
            flat = np.transpose(h, (0, 2, 1)).reshape(-1, Wd)
            codes, _ = self.quantise(flat, level)
            zs.append(codes.reshape(N, T).astype(np.int64))
        return zs[start_level:end_level]

    def decode(self, zs, start_level=0, end_level=None):
        """vqvae.py:101-122 + Decoder.forward(all_levels=False) encdec.py:110-131 +
        BottleneckBlock.decode bottleneck.py:138-147.  Only zs[0] (level start_level) is used."""
        if end_level is None:
            end_level = self.levels
        assert len(zs) == end_level - start_level
        z = np.asarray(zs[0])
        x = np.transpose(self.codebook(start_level)[z], (0, 2, 1))       # (N, emb, T)
        for blk in reversed(self.decoders[start_level]):
            x = blk(x)
        p = f"decoders.{start_level}.out."
        x = conv1d_nct(x, self.sd[p + "weight"], self.sd[p + "bias"], 1, 1, 1)
        return np.transpose(x, (0, 2, 1))                                 # (N, T, 1)
