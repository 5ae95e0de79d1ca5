"""Thin tensor-level wrappers over the C ABI (include/jukebox_hip.h).

torch supplies device memory and the current HIP stream only; every computation below happens
in libjukebox_hip.so.  Tensors must live on the GPU and be contiguous; there is no CPU path.
"""
import ctypes as C
import math

import torch

from . import _lib as L


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise L.JukeboxHipError("expected contiguous GPU tensors")


class PackedWeight:
    """A K x J matrix in MFMA fragment order (see csrc/gemm.hip) plus its logical dims.
    `taps` > 1 holds the per-tap matrices of a (transposed) convolution back to back."""

    def __init__(self, data, K, J, dtype, taps=1, split=False, unscale=1.0):
        self.data, self.K, self.J, self.dtype, self.taps = data, K, J, dtype, taps
        self.tap_stride = data.numel() // taps
        self.split = split              # fp32 matrix as hi / lo f16 images (jb_gemm_args.w_split); `data` holds the raw words
        self.unscale = unscale          # split images hold s * W for a power of two s: 1 / s (jb_gemm_args.w_split_unscale)

    @property
    def ptr(self):
        return self.data.data_ptr()


def pack_weight(w, K, J, stride_k, stride_j, dtype, out=None, offset_elems=0):
    """Pack logical matrix M[k, j] = w.flatten()[offset + k*stride_k + j*stride_j] into `dtype`."""
    _chk_cuda(w)
    code = L.dtype_code(dtype)
    nbytes = L.lib().jb_packed_weight_bytes(K, J, code)
    if out is None:
        out = torch.empty(nbytes // (2 if code == L.F16 else 4), dtype=dtype, device=w.device)
    src = w.data_ptr() + offset_elems * w.element_size()
    L.check(L.lib().jb_pack_weight(src, L.dtype_code(w.dtype), stride_k, stride_j, K, J, out.data_ptr(), code, L.stream()))
    return out


def pack_conv1d_w(w, dtype):
    """Conv1D.w (n_in, n_out) (jukebox/transformer/ops.py:89-95)."""
    K, J = w.shape
    return PackedWeight(pack_weight(w, K, J, J, 1, dtype), K, J, dtype)


class FoldedLN:
    """LayerNorm folded into the Conv1D that follows it (jb_gemv_args.ln_fold_c1): LN(x)·W + b is evaluated as
    rstd·(x·W' - mean·c1) + b' with W' = diag(gamma)·W (packed, engine dtype), c1 = column sums of W' as stored,
    b' = beta·W + b.  `w` is Conv1D.w (n_in, n_out); W is first rounded to the engine dtype, as the reference's
    `w.type_as(x)` does (jukebox/transformer/ops.py:99)."""

    def __init__(self, w, b, gamma, beta, dtype):
        K, J = w.shape
        w_used = w.to(dtype).double()
        wf = (gamma.double()[:, None] * w_used).to(dtype).contiguous()
        self.wf = wf                                           # unpacked diag(gamma)·W
        self.pw = PackedWeight(pack_weight(wf, K, J, J, 1, dtype), K, J, dtype)
        self.c1 = wf.double().sum(0).float().contiguous()
        self.bias = (beta.double() @ w_used + b.double()).float().contiguous()


def ln_fold_supported(dtype, K, J, n_rows):
    return bool(L.lib().jb_gemv_ln_fold_supported(L.dtype_code(dtype), K, J, n_rows))


def pack_linear_w(w, dtype):
    """nn.Linear.weight (out, in): logical [k=in][j=out]."""
    J, K = w.shape
    return PackedWeight(pack_weight(w, K, J, 1, K, dtype), K, J, dtype)


def pack_conv_taps(w, dtype, transposed=False, split=False):
    """nn.Conv1d weight (Cout, Cin, k) or nn.ConvTranspose1d weight (Cin, Cout, k): one K=Cin x J=Cout
    matrix per tap.  split (fp32 only, Cin a multiple of 32): every tap as the hi / lo pair of f16 images that lets the
    fp32 conv stacks run on the f16 matrix cores at fp32 accuracy (jb_gemm_args.w_split) -- the same bytes per tap."""
    w = w.contiguous()
    unscale = 1.0
    if split:
        assert dtype == torch.float32 and w.shape[0 if transposed else 1] % 32 == 0
        # (once per layer, at pack time) the image holds s * w with the power of two s that puts max |w| into [128, 256): every
        # weight down to 2^-22 of the largest keeps a NORMAL hi half, i.e. 2^-22 relative precision after the split, whatever the
        # layer's scale -- unscaled, a trained layer with |w| < 6e-5 would have subnormal hi halves (exact: powers of two)
        amax = float(w.abs().max())
        assert math.isfinite(amax), "f16-split weight image: weights must be finite"
        if amax > 0.0:
            scale = 2.0 ** (8 - math.frexp(amax)[1])
            w, unscale = (w.float() * scale).contiguous(), 1.0 / scale
    if transposed:
        Cin, Cout, k = w.shape
        sk, sj = Cout * k, k
    else:
        Cout, Cin, k = w.shape
        sk, sj = k, Cin * k
    code = L.F16_SPLIT if split else L.dtype_code(dtype)
    per = L.lib().jb_packed_weight_bytes(Cin, Cout, code) // (2 if code == L.F16 else 4)
    out = torch.empty(per * k, dtype=dtype, device=w.device)
    for tap in range(k):
        dst = out[tap * per:(tap + 1) * per]
        if split:
            L.check(L.lib().jb_pack_weight(w.data_ptr() + tap * w.element_size(), L.dtype_code(w.dtype), sk, sj, Cin, Cout,
                                           dst.data_ptr(), code, L.stream()))
        else:
            pack_weight(w, Cin, Cout, sk, sj, dtype, out=dst, offset_elems=tap)
    return PackedWeight(out, Cin, Cout, dtype, taps=k, split=split, unscale=unscale)


def layernorm(x, gamma, beta, eps=1e-5, out_dtype=None):
    """ops.LayerNorm.forward (jukebox/transformer/ops.py:14-24)."""
    _chk_cuda(x, gamma, beta)
    W = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    L.check(L.lib().jb_layernorm_fwd(x.data_ptr(), L.dtype_code(x.dtype), y.data_ptr(), L.dtype_code(y.dtype),
                                     gamma.data_ptr(), beta.data_ptr(), x.numel() // W, W, eps, L.stream()))
    return y


def gemm(A, pw, bias=None, out=None, res=None, n_seq=1, t_in=None, t_out=None, shifts=(0,), in_stride=1,
         out_stride=1, out_offset=0, out_rows_per_seq=None, pre_relu=False, act=L.ACT_NONE, res_scale=1.0, in_seq_pitch=None):
    """Sequence/tap GEMM (see jb_gemm).  A: (n_seq*t_in, lda>=K); returns (n_seq*out_rows_per_seq, J).
    in_seq_pitch: rows between the starts of two sequences in A when they are not packed (t_in must then be given)."""
    _chk_cuda(A, bias, out, res)
    rows_in = A.shape[0]
    assert in_seq_pitch is None or t_in is not None
    t_in = t_in if t_in is not None else rows_in // n_seq
    t_out = t_out if t_out is not None else t_in
    orps = out_rows_per_seq if out_rows_per_seq is not None else t_out * out_stride
    if out is None:
        out = torch.empty((n_seq * orps, pw.J), dtype=A.dtype, device=A.device)
    a = L.GemmArgs()
    a.dtype = L.dtype_code(A.dtype)
    a.A, a.lda = A.data_ptr(), A.stride(0)
    a.W, a.tap_stride = pw.ptr, pw.tap_stride
    a.bias = L.ptr(bias)
    a.out, a.ldo = out.data_ptr(), out.stride(0)
    a.res, a.ldr = L.ptr(res), (res.stride(0) if res is not None else 0)
    a.n_seq, a.t_in, a.t_out = n_seq, t_in, t_out
    a.in_seq_stride, a.out_seq_stride = (t_in if in_seq_pitch is None else in_seq_pitch), orps
    a.K, a.J = pw.K, pw.J
    if len(shifts) != pw.taps:
        raise L.JukeboxHipError(f"{len(shifts)} shifts for a weight with {pw.taps} taps")
    a.n_taps, a.in_stride = len(shifts), in_stride
    for i, s in enumerate(shifts):
        a.shift[i] = s
    a.out_stride, a.out_offset = out_stride, out_offset
    a.pre_relu, a.act, a.res_scale = int(pre_relu), act, res_scale
    a.w_split = int(pw.split)
    a.w_split_unscale = float(pw.unscale)
    if pw.split:            # scratch for the activations' hi / lo images (same bytes as the fp32 rows); the caching allocator recycles it
        scratch = torch.empty(n_seq * t_in * pw.K, dtype=torch.float32, device=A.device)
        a.a_split, a.a_split_bytes = scratch.data_ptr(), scratch.numel() * 4
    L.check(L.lib().jb_gemm(C.byref(a), L.stream()))
    return out


def check_split_overflow(wait=True):
    """Raise if a conv-stack launch on the f16-split path (jb_gemm_args.w_split) has seen an activation outside the half range
    (or a NaN) since the last check: its output is not the convolution.  The flag is a host-coherent word, so reading it costs
    nothing and waits for nothing: wait=True synchronises the device first (end of a job, tests: the launches just enqueued
    count), wait=False looks at what has FINISHED so far -- the sampler's look after every window, which must not stall the
    other levels' streams (a window's conv stacks ran seconds before its last token)."""
    if wait:
        torch.cuda.synchronize()
    rc = L.lib().jb_gemm_split_overflow(1)
    if rc < 0:
        L.check(rc)
    if rc:
        raise L.JukeboxHipError("a convolution on the f16-split path was given activations outside the half range (|x| > 65504 or "
                                "NaN): its output is invalid (jukebox_amd.vqvae.rows.SPLIT_F16 = False selects the exact-fp32 kernel)")


def gemm_qkv(h, pw, bias, n_seq, n_q, S, kcache, vcache, t0):
    """c_attn projection of a chunk with the k/v parts written into the caches at positions t0.. (jb_gemm
    qkv_split).  h: (n_seq*n_q, K).  Returns q: (n_seq*n_q, S)."""
    _chk_cuda(h, bias, kcache, vcache)
    q = torch.empty((n_seq * n_q, S), dtype=h.dtype, device=h.device)
    a = L.GemmArgs()
    a.dtype = L.dtype_code(h.dtype)
    a.A, a.lda = h.data_ptr(), h.stride(0)
    a.W, a.tap_stride, a.bias = pw.ptr, pw.tap_stride, L.ptr(bias)
    a.out, a.ldo = q.data_ptr(), S
    a.n_seq, a.t_in, a.t_out = n_seq, n_q, n_q
    a.in_seq_stride, a.out_seq_stride = n_q, n_q
    a.K, a.J = pw.K, pw.J
    a.n_taps, a.in_stride, a.out_stride, a.res_scale = 1, 1, 1, 1.0
    a.qkv_split, a.S = 1, S
    a.kcache, a.vcache, a.cache_cap, a.cache_t0 = kcache.data_ptr(), vcache.data_ptr(), kcache.shape[1], t0
    L.check(L.lib().jb_gemm(C.byref(a), L.stream()))
    return q


def tap_view(pw, taps):
    """PackedWeight holding only the listed taps (must be equally spaced) -- for transposed-conv phases."""
    step = taps[1] - taps[0] if len(taps) > 1 else 1
    assert all(taps[i] == taps[0] + i * step for i in range(len(taps)))
    base = pw.data[taps[0] * pw.tap_stride:]
    v = PackedWeight(base, pw.K, pw.J, pw.dtype, taps=1, split=pw.split, unscale=pw.unscale)
    v.tap_stride = pw.tap_stride * step
    v.taps = len(taps)
    return v


def gemv(x, pw, bias=None, ln=None, res=None, act=L.ACT_NONE, out=None, eps=1e-5, ln_fold=None, parts=None, out2=None,
         add2=None, t_dev=None, qkv=None):
    """Decode-step skinny GEMM (see jb_gemv).  x: (n_rows<=64, K).  ln = (gamma, beta): normalise the rows in the
    kernel; ln_fold = FoldedLN: the folded form (pw / bias are taken from it); parts = (x_parts (N, P, K) f16,
    ml (N, H, P, 2) fp32): the operand is the merge of the key-split attention's partial states (x must be None);
    out2 (N, J) fp32 [+ add2 (N, T, J) fp32 read at row *t_dev]: the second, fp32 output of the plain projection;
    qkv = (S, kcache, vcache or None, vcache_wide or None): c_attn of the decode step -- columns q | k | v | v' with the
    cache rows written at *t_dev; returns q (N, S)."""
    if ln_fold is not None:
        assert ln is None and bias is None and pw is None
        pw, bias = ln_fold.pw, ln_fold.bias
    _chk_cuda(x, bias, res, out, out2, add2)
    src = x if x is not None else parts[0]
    if out is None:
        out = torch.empty((src.shape[0], qkv[0] if qkv is not None else pw.J), dtype=src.dtype, device=src.device)
    a = L.GemvArgs()
    if qkv is not None:
        S, kc, vc, vw = qkv
        _chk_cuda(kc, vc, vw, t_dev)
        a.qkv_split, a.S, a.kcache, a.vcache, a.cache_cap = 1, S, kc.data_ptr(), L.ptr(vc), kc.shape[1]
        if vw is not None:
            a.vcache_wide, a.wide = vw.data_ptr(), vw.shape[2]
        a.t_dev = L.ptr(t_dev)
    a.dtype = L.dtype_code(src.dtype)
    a.n_rows = src.shape[0]
    if x is not None:
        a.x, a.ldx = x.data_ptr(), x.stride(0)
    else:
        xp, ml = parts
        _chk_cuda(xp, ml)
        a.x_parts, a.x_ml, a.n_parts, a.n_head = xp.data_ptr(), ml.data_ptr(), xp.shape[1], ml.shape[1]
        a.d_head, a.ldx = xp.shape[2] // ml.shape[1], xp.shape[2]
    if out2 is not None:
        a.out2, a.ldo2 = out2.data_ptr(), out2.stride(0)
        if add2 is not None:
            a.add2, a.add2_n_stride, a.add2_t_stride = add2.data_ptr(), add2.stride(0), add2.stride(1)
        a.t_dev = L.ptr(t_dev)
    if ln is not None:
        a.ln_gamma, a.ln_beta, a.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), eps
    if ln_fold is not None:
        a.ln_fold_c1, a.ln_eps = ln_fold.c1.data_ptr(), eps
    a.W, a.bias, a.K, a.J = pw.ptr, L.ptr(bias), pw.K, pw.J
    a.out, a.ldo = out.data_ptr(), out.stride(0)
    a.res, a.ldr = L.ptr(res), (res.stride(0) if res is not None else 0)
    a.act = act
    L.check(L.lib().jb_gemv(C.byref(a), L.stream()))
    return out


def attn_decode(func, q, kcache, vcache, n_head, block_ctx, t_dev, max_len):
    """q (N, S); caches (N, cap, S); returns (N, S)."""
    _chk_cuda(q, kcache, vcache, t_dev)
    N, S = q.shape
    out = torch.empty_like(q)
    L.check(L.lib().jb_attn_decode(L.dtype_code(q.dtype), func, q.data_ptr(), q.stride(0), kcache.data_ptr(),
                                   vcache.data_ptr(), kcache.shape[1], out.data_ptr(), out.stride(0), N, n_head,
                                   S // n_head, block_ctx or 0, t_dev.data_ptr(), max_len, L.stream()))
    return out


def attn_decode_wide(func, q, kcache, vcache_w, res, bias, block_ctx, t_dev, max_len):
    """Wide-value decode attention of a single-head fp16 layer (jb_attn_decode_wide): q (N, S); kcache (N, cap, S);
    vcache_w (N, cap, W) rows v' = v·Wp; returns x_out (N, W) = res + (sum_k p_k v'_k + bias)."""
    _chk_cuda(q, kcache, vcache_w, res, bias, t_dev)
    N, S = q.shape
    W = vcache_w.shape[2]
    out = torch.empty_like(res)
    L.check(L.lib().jb_attn_decode_wide(func, q.data_ptr(), q.stride(0), kcache.data_ptr(), vcache_w.data_ptr(),
                                        kcache.shape[1], res.data_ptr(), res.stride(0), bias.data_ptr(), out.data_ptr(),
                                        out.stride(0), N, S, W, block_ctx or 0, t_dev.data_ptr(), max_len, L.stream()))
    return out


def attn_decode_split(func, q, kcache, vcache, n_head, block_ctx, t_dev, max_keys, n_parts):
    """Key-split decode attention (fp16): returns (parts (N, n_parts, S) f16, ml (N, n_head, n_parts, 2) fp32)."""
    _chk_cuda(q, kcache, vcache, t_dev)
    N, S = q.shape
    parts = torch.empty((N, n_parts, S), dtype=q.dtype, device=q.device)
    ml = torch.empty((N, n_head, n_parts, 2), dtype=torch.float32, device=q.device)
    L.check(L.lib().jb_attn_decode_split(func, q.data_ptr(), q.stride(0), kcache.data_ptr(), vcache.data_ptr(),
                                         kcache.shape[1], parts.data_ptr(), ml.data_ptr(), N, n_head, S // n_head,
                                         block_ctx or 0, t_dev.data_ptr(), max_keys, n_parts, L.stream()))
    return parts, ml


def attn_prefill(func, q, kcache, vcache, n_head, block_ctx, t0):
    """q (N, n_q, S) for positions t0..t0+n_q-1; caches (N, cap, S) already hold them."""
    _chk_cuda(q, kcache, vcache)
    N, nq, S = q.shape
    out = torch.empty_like(q)
    L.check(L.lib().jb_attn_prefill(L.dtype_code(q.dtype), func, q.data_ptr(), kcache.data_ptr(), vcache.data_ptr(),
                                    kcache.shape[1], out.data_ptr(), N, n_head, S // n_head, block_ctx or 0, t0, nq,
                                    L.stream()))
    return out


def vq_gather(codes, codebook):
    """(N, T) int64 -> (N, T, emb) fp32 (BottleneckBlock.decode, channels-last)."""
    _chk_cuda(codes, codebook)
    out = torch.empty((*codes.shape, codebook.shape[1]), dtype=torch.float32, device=codes.device)
    L.check(L.lib().jb_vq_gather(codes.data_ptr(), codebook.data_ptr(), out.data_ptr(), codes.numel(),
                                 codebook.shape[1], codebook.shape[0], L.stream()))
    return out


def vq_argmin(x, xk, codebook):
    _chk_cuda(x, xk, codebook)
    codes = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    L.check(L.lib().jb_vq_argmin(x.data_ptr(), xk.data_ptr(), codebook.data_ptr(), codes.data_ptr(), x.shape[0],
                                 x.shape[1], codebook.shape[0], L.stream()))
    return codes


def sample_logits(logits, params_dev, tokens, t_dev, preds=None):
    _chk_cuda(logits, tokens, t_dev, preds)
    L.check(L.lib().jb_sample_logits(logits.data_ptr(), logits.shape[0], logits.shape[1], params_dev.data_ptr(),
                                     tokens.data_ptr(), tokens.stride(0), t_dev.data_ptr(), L.ptr(preds),
                                     preds.stride(0) if preds is not None else 0, L.stream()))


def make_sample_params(temp=1.0, top_k=0, top_p=0.0, seed=0, sample_base=0, pos_base=0, stream_id=0, device="cuda"):
    """Device copy of jb_sample_params."""
    p = L.SampleParams(temp, top_k, top_p, sample_base, seed & 0xFFFFFFFFFFFFFFFF, pos_base, stream_id)
    raw = bytes(p)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
