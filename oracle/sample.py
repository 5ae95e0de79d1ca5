"""Oracle restatement of the window/level driver in jukebox/sample.py:17-121 and
jukebox/utils/sample_utils.py (numpy).  Test infrastructure only."""
import numpy as np


def get_starts(total_length, n_ctx, hop_length):
    """utils/sample_utils.py:15-22."""
    starts = []
    for start in range(0, total_length - n_ctx + hop_length, hop_length):
        if start + n_ctx >= total_length:
            start = total_length - n_ctx
        starts.append(start)
    return starts


def split_batch(obj, n_samples, split_size):
    """utils/sample_utils.py:3-12."""
    n_passes = (n_samples + split_size - 1) // split_size
    if isinstance(obj, np.ndarray):
        return [obj[i * split_size:(i + 1) * split_size] for i in range(n_passes)]
    if isinstance(obj, list):
        return list(zip(*[split_batch(item, n_samples, split_size) for item in obj]))
    if obj is None:
        return [None] * n_passes
    raise TypeError("Unknown input type")


def window_plan(total_length, n_ctx, hop_length):
    """For each window: (start, conditioning tokens already present, new tokens) --
    what sample_single_window (sample.py:31-78) computes from zs[level].shape."""
    plan, have = [], 0
    for start in get_starts(total_length, n_ctx, hop_length):
        end = start + n_ctx
        cond = max(0, min(have, end) - start)
        new = n_ctx - cond
        plan.append((start, cond, new))
        have += new
    return plan


def sample_single_window(zs, y_fn, prior, level, start, n_samples, max_batch_size, **sampling_kwargs):
    """sample.py:31-78.  `y_fn(start)` returns the per-window label matrix (prior.get_y)."""
    n_ctx = prior.n_ctx
    end = start + n_ctx
    z = zs[level][:, start:end]
    sample_tokens = sampling_kwargs.pop("sample_tokens", end - start)
    new_tokens = sample_tokens - z.shape[1]
    if new_tokens <= 0:
        return zs
    if prior.x_cond:
        cd = prior.cond_downsample
        z_conds = [zs[level + 1][:, start // cd:end // cd]]        # prior.py:158-166
    else:
        z_conds = None
    y = y_fn(start)
    outs = []
    for z_i, zc_i, y_i in zip(split_batch(z, n_samples, max_batch_size), split_batch(z_conds, n_samples, max_batch_size),
                              split_batch(y, n_samples, max_batch_size)):
        zc_i = None if zc_i is None else list(zc_i)
        outs.append(prior.sample(n_samples=z_i.shape[0], z=z_i, z_conds=zc_i, y=y_i,
                                 sample_tokens=sample_tokens if sample_tokens != n_ctx else None, **sampling_kwargs))
    z_new = np.concatenate(outs, axis=0)[:, -new_tokens:]
    zs[level] = np.concatenate([zs[level], z_new], axis=1)
    return zs


def sample_level(zs, y_fn, prior, level, total_length, hop_length, n_samples, max_batch_size, **sampling_kwargs):
    """sample.py:81-88 (+ sample_partial_window :17-29 for total_length < n_ctx)."""
    if total_length >= prior.n_ctx:
        for start in get_starts(total_length, prior.n_ctx, hop_length):
            zs = sample_single_window(zs, y_fn, prior, level, start, n_samples, max_batch_size, **sampling_kwargs)
    else:
        n_ctx, cur = prior.n_ctx, zs[level].shape[1]
        if cur < n_ctx - total_length:
            st, start = cur + total_length, 0
        else:
            st, start = n_ctx, cur - n_ctx + total_length
        zs = sample_single_window(zs, y_fn, prior, level, start, n_samples, max_batch_size, sample_tokens=st,
                                  **sampling_kwargs)
    return zs
