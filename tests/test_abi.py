"""CPU suite: the C-ABI shared library loads and exports every symbol include/jukebox_hip.h declares
(no compute calls -- there is no GPU here), and the argument-validation paths report errors."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from jukebox_amd.csrc.build import build
    return build()


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "jukebox_hip.h")).read()
    declared = set(re.findall(r"\b(jb_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"jb_status"}
    lib = C.CDLL(built)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    from jukebox_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_binding_loads_and_validates_arguments(built):
    from jukebox_amd import _lib as L
    lib = L.lib()
    assert lib.jb_version() == 1
    assert lib.jb_packed_weight_bytes(2048, 1536, L.F16) == 2048 * 1536 * 2
    assert lib.jb_packed_weight_bytes(70, 50, L.F32) == 80 * 64 * 4       # K padded to 16, J to 16
    # null pointers are rejected before any launch, with a message
    rc = lib.jb_layernorm_fwd(None, 0, None, 0, None, None, 4, 8, 1e-5, None)
    assert rc == -1 and b"null" in lib.jb_last_error()
    # host-side shape query of the folded-LayerNorm projection (no launch)
    assert lib.jb_gemv_ln_fold_supported(L.F16, 1920, 1440, 16) == 1 and lib.jb_gemv_ln_fold_supported(L.F16, 2048, 2048, 32) == 1
    assert lib.jb_gemv_ln_fold_supported(L.F16, 4800, 4800, 3) == 1       # 5b_lyrics width: 150 k-tiles on the 16-wave kernel
    assert lib.jb_gemv_ln_fold_supported(L.F16, 4800, 4800, 17) == 0 and lib.jb_gemv_ln_fold_supported(L.F32, 4800, 4800, 3) == 0
    assert lib.jb_gemv_ln_fold_supported(L.F16, 100, 64, 16) == 0 and lib.jb_gemv_ln_fold_supported(L.F32, 256, 64, 33) == 0
    a = L.GemvArgs()
    assert lib.jb_gemv(C.byref(a), None) == -1
    # wide-value layers: host-side shape queries and the argument checks of the v' column group (no launch)
    assert lib.jb_gemv_ln_fold_supported(L.F16, 1920, 2 * 480 + 1920, 16) == 1
    assert lib.jb_attn_decode_wide_supported(1, 480, 1920, 64, 8192) == 1 and lib.jb_attn_decode_wide_supported(2, 256, 1024, 128, 8192) == 1
    assert lib.jb_attn_decode_wide_supported(6, 480, 1920, 64, 8192) == 0        # cross-attention
    assert lib.jb_attn_decode_wide_supported(1, 150, 4800, 64, 8192) == 0        # head size the MFMA kernel does not take
    assert lib.jb_attn_decode_wide_supported(1, 480, 2000, 64, 8192) == 0        # width not a multiple of the head size
    assert lib.jb_attn_decode_wide_supported(1, 480, 1920, 0, 8192) == 0         # block pattern without block_ctx
    dummy = C.create_string_buffer(64)
    ptr = C.addressof(dummy)
    a = L.GemvArgs()
    a.dtype, a.x, a.ldx, a.n_rows, a.W, a.K, a.out, a.ldo = L.F16, ptr, 1920, 16, ptr, 1920, ptr, 480
    a.qkv_split, a.S, a.kcache, a.cache_cap, a.t_dev = 1, 480, ptr, 8, ptr
    a.J, a.wide = 2 * 480 + 1920, 1920                                           # v' columns announced, no cache for them
    assert lib.jb_gemv(C.byref(a), None) == -1 and b"vcache_wide" in lib.jb_last_error()
    a.vcache_wide, a.J = ptr, 3 * 480 + 1920                                     # v columns counted, no v cache given
    assert lib.jb_gemv(C.byref(a), None) == -1 and b"qkv split" in lib.jb_last_error()
    assert lib.jb_attn_decode_wide(1, ptr, 480, ptr, ptr, 8, ptr, 1920, ptr, ptr, 1920, 16, 150, 4800, 64, ptr, 8192, None) == -1
    assert b"wide-value attention" in lib.jb_last_error()
    with pytest.raises(L.JukeboxHipError):
        L.check(lib.jb_engine_decode(None, 0, 1, 0, None))


def test_struct_layouts_match_header():
    """ctypes mirrors vs the layout the C compiler computes from include/jukebox_hip.h: the size of every struct and the
    offset of every field (guards against silent field drift / reordering)."""
    import subprocess, tempfile
    from jukebox_amd import _lib as L
    pairs = [("jb_gemm_args", L.GemmArgs), ("jb_gemv_args", L.GemvArgs), ("jb_sample_params", L.SampleParams),
             ("jb_layer", L.Layer), ("jb_engine_cfg", L.EngineCfg)]
    lines, expect = [], []
    for cname, ct in pairs:
        lines.append(f'printf("%zu\\n", sizeof({cname}));')
        expect.append(C.sizeof(ct))
        for fname, _ in ct._fields_:
            lines.append(f'printf("%zu\\n", offsetof({cname}, {fname}));')
            expect.append(getattr(ct, fname).offset)
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"jukebox_hip.h\"\nint main(void) {\n" + "\n".join(lines) + "\nreturn 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert got == expect
