"""Window / batch helpers of the sampling driver (same contracts as jukebox/utils/sample_utils.py:3-22, own
formulation; pinned to the reference's tables by tests/test_sample_windows.py against tests/golden/misc.npz)."""
import torch as t


def split_batch(obj, n_samples, split_size):
    """Cut a batch into sub-batches of at most `split_size` rows: a tensor gives a tuple of row slices, a list of
    tensors gives one tuple of slices per sub-batch, None gives one None per sub-batch."""
    bounds = [(lo, min(lo + split_size, n_samples)) for lo in range(0, n_samples, split_size)]
    if obj is None:
        return [None] * len(bounds)
    if isinstance(obj, t.Tensor):
        return tuple(obj[lo:hi] for lo, hi in bounds)
    if isinstance(obj, list):
        return [tuple(item[lo:hi] for item in obj) for lo, hi in bounds]
    raise TypeError("Unknown input type")


def get_starts(total_length, n_ctx, hop_length):
    """Start of every n_ctx-long window that covers [0, total_length): a hop grid, with the windows that would overrun
    the end right-aligned to it instead."""
    last = total_length - n_ctx
    n_windows = max(0, -(-(last + hop_length) // hop_length))            # ceil((last + hop) / hop) grid points below last + hop
    return [min(i * hop_length, last) for i in range(n_windows)]
