"""Small model / allocator helpers with the names the reference's modules expect (jukebox/utils/torch_utils.py)."""
import gc

import torch as t


def freeze_model(model):
    """Inference only: eval mode, no parameter tracks gradients (make_models.py:103,184)."""
    model.eval().requires_grad_(False)


def empty_cache():
    """Drop Python garbage, then hand the caching allocator's free blocks back (a device-wide synchronisation on a GPU:
    the resident / pipelined sampling modes do not call it between windows)."""
    gc.collect()
    if t.cuda.is_available():
        t.cuda.empty_cache()


def assert_shape(x, exp_shape):
    got, want = tuple(x.shape), tuple(exp_shape)
    assert got == want, f"Expected {exp_shape} got {x.shape}"
