"""Import shim that lets the UNMODIFIED reference (`/root/reference/jukebox`) run on CPU.

Used only by `tests/golden/gen_golden.py` (and ad-hoc probes) in the build
container; `/root/reference` does not exist on the GPU box, so nothing under
`-m gpu`, `smoke()` or `bench.py` imports this file.

What it does:
  * stub modules for packages the reference imports but that are not installed
    (fire, soundfile, librosa, unidecode, mpi4py, av, numba);
  * `.cuda()` becomes the identity, `torch.cuda.LongTensor` aliases
    `torch.LongTensor`, `torch.cuda.empty_cache` is a no-op;
  * a TorchFunctionMode that rewrites `device='cuda'` keyword arguments to 'cpu';
  * a single-rank gloo process group (the reference's dist_adapter needs one).
"""
import os
import sys
import tempfile
import types

import torch
from torch.overrides import TorchFunctionMode

REFERENCE_ROOT = os.environ.get("JUKEBOX_REFERENCE", "/root/reference")


class _CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = kwargs or {}
        dev = kwargs.get("device", None)
        if dev is not None and "cuda" in str(dev):
            kwargs = dict(kwargs)
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


_mode = None


def install():
    """Idempotent. Returns the imported top-level `jukebox` reference package."""
    global _mode
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for name in ("fire", "soundfile", "librosa", "unidecode", "mpi4py", "av", "numba"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["fire"].Fire = lambda f: None
    # unidecode is not installed: the same ASCII stand-in the product uses (accented Latin letters -> base letters)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from jukebox_amd.data.labels import to_ascii
    sys.modules["unidecode"].unidecode = to_ascii
    if not hasattr(sys.modules["mpi4py"], "MPI"):
        sys.modules["mpi4py"].MPI = types.SimpleNamespace()

    def _sf_write(fname, data, samplerate, format="wav"):
        from scipy.io import wavfile
        wavfile.write(fname, samplerate, data)
    sys.modules["soundfile"].write = _sf_write

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    ident = lambda self, *a, **k: self
    torch.Tensor.cuda = ident
    torch.nn.Module.cuda = ident
    torch.cuda.LongTensor = torch.LongTensor
    torch.cuda.empty_cache = lambda: None

    if _mode is None:
        _mode = _CudaToCpu()
        _mode.__enter__()

    import torch.distributed as dist
    if not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(prefix="jb_gloo_", delete=False)
        f.close()
        os.unlink(f.name)
        dist.init_process_group("gloo", init_method=f"file://{f.name}", rank=0, world_size=1)

    import jukebox  # noqa: F401  (the reference package)
    return jukebox
