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

_X000 = ("", "!", "C/", "PS", "$?", "Y=", "|", "SS", '"', "(c)", "a", "<<", "!", "", "(r)", "-",          # U+00A0 .. U+00AF
         "deg", "+-", "2", "3", "'", "u", "P", "*", ",", "1", "o", ">>", "1/4", "1/2", "3/4", "?",        # U+00B0 .. U+00BF
         "A", "A", "A", "A", "A", "A", "AE", "C", "E", "E", "E", "E", "I", "I", "I", "I",                 # U+00C0 .. U+00CF
         "D", "N", "O", "O", "O", "O", "O", "x", "O", "U", "U", "U", "U", "Y", "Th", "ss",                # U+00D0 .. U+00DF
         "a", "a", "a", "a", "a", "a", "ae", "c", "e", "e", "e", "e", "i", "i", "i", "i",                 # U+00E0 .. U+00EF
         "d", "n", "o", "o", "o", "o", "o", "/", "o", "u", "u", "u", "u", "y", "th", "y")                 # U+00F0 .. U+00FF
_X001 = ("AaAaAaCcCcCcCcDdDdEeEeEeEeEeGgGgGgGgHhHhIiIiIiIiIi", ("IJ", "ij"), "JjKkk", "LlLlLlLlLl", "NnNnNn", ("'n", "NG", "ng"),
         "OoOoOo", ("OE", "oe"), "RrRrRrSsSsSsSsTtTtTtUuUuUuUuUuUuWwYyYZzZzZz", ("s",))                   # U+0100 .. U+017F
_X020 = {0x2010: "-", 0x2011: "-", 0x2012: "-", 0x2013: "-", 0x2014: "--", 0x2015: "--", 0x2018: "'", 0x2019: "'", 0x201a: ",",
         0x201b: "'", 0x201c: '"', 0x201d: '"', 0x201e: ",,", 0x201f: '"', 0x2022: "*", 0x2026: "...", 0x2032: "'", 0x2033: '"',
         0x2039: "<", 0x203a: ">", 0x00a0: " "}


def _x001():
    out = []
    for part in _X001:
        out.extend(part if isinstance(part, tuple) else list(part))
    assert len(out) == 128, len(out)
    return out


def unidecode_stand_in(text):
    """ASCII transliteration of `text` by table lookup (see install): ASCII unchanged; U+00A0..U+017F and the listed
    punctuation as unidecode maps them; anything else dropped (unidecode would romanise it: no golden contains such text)."""
    x001 = _x001()
    out = []
    for ch in text:
        c = ord(ch)
        if c < 0x80:
            out.append(ch)
        elif c in _X020:
            out.append(_X020[c])
        elif 0xa0 <= c <= 0xff:
            out.append(_X000[c - 0xa0])
        elif 0x100 <= c <= 0x17f:
            out.append(x001[c - 0x100])
    return "".join(out)


def install():
    """Idempotent. Returns the imported top-level `jukebox` reference package."""
    global _mode
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for name in ("fire", "soundfile", "librosa", "unidecode", "mpi4py", "av", "numba"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["fire"].Fire = lambda f: None
    # unidecode (data/text_processor.py:2,12; not pinned in requirements.txt) is not installed.  The stand-in below is
    # INDEPENDENT of the product's `to_ascii` (jukebox_amd/data/labels.py: NFKD decomposition + a punctuation dict): an explicit
    # per-code-point table restating unidecode's published tables for Latin-1 Supplement (x000), the letters of Latin Extended-A
    # (x001) and the General Punctuation marks lyric sheets contain (x020) -- so that the lyric-token goldens are not circular in
    # that function (tests/test_oracle_golden.py::test_ascii_stand_ins_agree compares the two on every character of the tables).
    sys.modules["unidecode"].unidecode = unidecode_stand_in
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
