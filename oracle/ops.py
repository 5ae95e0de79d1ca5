"""Oracle restatement of jukebox/transformer/ops.py (numpy, float32).  Test infrastructure only."""
import numpy as np

F32 = np.float32


def r16(x, fp16=True):
    """Round to IEEE half and back: models a tensor the reference materialises as half."""
    if not fp16:
        return np.asarray(x, dtype=F32)
    return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


def layer_norm(x, weight, bias, eps=1e-5):
    """ops.py:14-24 -- LayerNorm computed in fp32 over the last dim (biased variance),
    affine; the caller casts the result back to the input dtype."""
    x = np.asarray(x, dtype=F32)
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))).astype(F32) * weight.astype(F32) + bias.astype(F32)


def conv1d(x, w, b, fp16=False):
    """ops.py:83-101 -- Conv1D.forward = addmm(b, x2d, w) with w stored (n_in, n_out);
    in half mode w and b are cast to half (`type_as(x)`) and the result is half."""
    w = r16(w, fp16)
    b = r16(b, fp16)
    y = np.matmul(np.asarray(x, dtype=F32), w) + b
    return r16(y, fp16)


def sigmoid(x):
    x = np.asarray(x, dtype=F32)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out.astype(F32)


def quick_gelu(x, fp16=False):
    """ops.py:33-35 -- x * sigmoid(1.702 * x); in half mode each op yields a half tensor."""
    u = r16(F32(1.702) * x, fp16)
    s = r16(sigmoid(u), fp16)
    return r16(x * s, fp16)


def softmax(x, axis=-1):
    x = np.asarray(x, dtype=F32)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True, dtype=F32)).astype(F32)


def filter_logits(logits, top_k=0, top_p=0.0, filter_value=-np.inf):
    """ops.py:113-142 -- top-k keeps everything >= the k-th largest logit (ties kept);
    nucleus: sort descending, softmax-cumsum, drop entries whose cumulative
    probability exceeds top_p, shifted right by one so the first crossing entry stays.
    (The reference's uint8 scatter breaks under torch>=2; this follows its rule with a
    bool mask -- SURVEY.md Appendix E.)"""
    logits = np.array(logits, dtype=F32, copy=True)
    top_k = min(top_k, logits.shape[-1])
    assert top_k == 0 or top_p == 0.0
    if top_k > 0:
        kth = np.sort(logits, axis=-1)[..., -top_k][..., None]
        logits[logits < kth] = filter_value
    if top_p > 0.0:
        order = np.argsort(-logits, axis=-1, kind="stable")
        sorted_logits = np.take_along_axis(logits, order, axis=-1)
        cum = np.cumsum(softmax(sorted_logits, axis=-1), axis=-1, dtype=F32)
        remove_sorted = cum > top_p
        remove_sorted[..., 1:] = remove_sorted[..., :-1].copy()
        remove_sorted[..., 0] = False
        remove = np.zeros_like(remove_sorted)
        np.put_along_axis(remove, order, remove_sorted, axis=-1)
        logits[remove] = filter_value
    return logits
