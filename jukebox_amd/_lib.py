"""ctypes binding of libjukebox_hip.so (the C ABI declared in include/jukebox_hip.h).

The product path has no CPU fallback: if the shared object is missing or a symbol cannot be
resolved, `lib()` raises.  Build it with `python -m jukebox_amd.csrc.build` or
`__graft_entry__.build()`.
"""
import ctypes as C
import os

F32, F16, F16_SPLIT = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_QUICK_GELU = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libjukebox_hip.so")
# measurement build of the same sources (-DJB_PIPE_SEGMENTS: per-segment clock stamps of the pipelined launches, common.h), loaded
# instead of the product library ONLY when a measurement tool asks for it (tools/phase_segments.py sets JB_LIB_SEGMENTS=1)
if os.environ.get("JB_LIB_SEGMENTS") == "1":
    LIB_PATH = os.path.join(_HERE, "csrc", "libjukebox_hip_segments.so")
if os.environ.get("JB_LIB_PATH"):       # A/B measurements of two builds in one GPU call (tools/): an explicit path, never a fallback
    LIB_PATH = os.environ["JB_LIB_PATH"]

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("dtype", i32), ("A", vp), ("lda", i64), ("W", vp), ("tap_stride", i64), ("bias", vp),
                ("out", vp), ("ldo", i64), ("res", vp), ("ldr", i64),
                ("n_seq", i32), ("t_in", i32), ("t_out", i32), ("in_seq_stride", i64), ("out_seq_stride", i64),
                ("K", i32), ("J", i32), ("n_taps", i32), ("in_stride", i32), ("shift", i32 * 4),
                ("out_stride", i32), ("out_offset", i32), ("pre_relu", i32), ("act", i32), ("res_scale", f32),
                ("qkv_split", i32), ("S", i32), ("kcache", vp), ("vcache", vp), ("cache_cap", i32), ("cache_t0", i32),
                ("w_split", i32), ("w_split_unscale", f32), ("a_split", vp), ("a_split_bytes", i64)]


class GemvArgs(C.Structure):
    _fields_ = [("dtype", i32), ("x", vp), ("ldx", i64), ("n_rows", i32),
                ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32),
                ("W", vp), ("bias", vp), ("K", i32), ("J", i32), ("out", vp), ("ldo", i64),
                ("res", vp), ("ldr", i64), ("act", i32), ("qkv_split", i32), ("S", i32),
                ("kcache", vp), ("vcache", vp), ("cache_cap", i32), ("t_dev", vp),
                ("ln_fold_c1", vp),
                ("out2", vp), ("ldo2", i64), ("add2", vp), ("add2_n_stride", i64), ("add2_t_stride", i64),
                ("x_parts", vp), ("x_ml", vp), ("n_parts", i32), ("n_head", i32), ("d_head", i32),
                ("vcache_wide", vp), ("wide", i32)]


class SampleParams(C.Structure):
    _fields_ = [("temp", f32), ("top_k", i32), ("top_p", f32), ("sample_base", i32), ("seed", C.c_uint64),
                ("pos_base", i32), ("stream_id", i32)]


class Layer(C.Structure):
    _fields_ = [("attn_func", i32), ("w_attn", vp), ("w_proj", vp), ("w_fc", vp), ("w_proj2", vp),
                ("b_attn", vp), ("b_proj", vp), ("b_fc", vp), ("b_proj2", vp),
                ("ln0_g", vp), ("ln0_b", vp), ("ln1_g", vp), ("ln1_b", vp),
                ("kcache", vp), ("vcache", vp), ("cache_cap", i32), ("w_enc_k", vp), ("w_enc_v", vp), ("b_enc_kv", vp),
                ("w_attn_f", vp), ("w_fc_f", vp), ("b_attn_f", vp), ("b_fc_f", vp), ("c1_attn", vp), ("c1_fc", vp),
                ("w_attn_fw", vp), ("b_attn_fw", vp), ("c1_attn_w", vp), ("vcache_w", vp)]


class EngineCfg(C.Structure):
    _fields_ = [("dtype", i32), ("n_batch", i32), ("width", i32), ("n_state", i32), ("n_head", i32), ("n_mlp", i32),
                ("n_layers", i32), ("seq_len", i32), ("block_ctx", i32), ("bins", i32), ("ln_eps", f32),
                ("x_emb", vp), ("pos_emb", vp), ("x_out_packed", vp), ("start", vp), ("start_stride", i64),
                ("x_cond", vp), ("xc_n_stride", i64), ("xc_t_stride", i64), ("add_cond_after", i32), ("encoder_kv", vp), ("enc_len", i32), ("hidden_out", vp), ("hidden_n_stride", i64),
                ("x_a", vp), ("x_b", vp), ("q", vp), ("att", vp), ("mlp", vp), ("xf", vp), ("logits", vp),
                ("att_parts", vp), ("att_ml", vp), ("ticket", vp),
                ("chunk_cap", i32), ("c_xa", vp), ("c_xb", vp), ("c_h", vp), ("c_q", vp), ("c_att", vp),
                ("c_mlp", vp), ("c_xf", vp), ("tokens", vp), ("tok_stride", i64), ("t_dev", vp),
                ("preds", vp), ("preds_n_stride", i64), ("sample_params", vp),
                ("rec_layer", i32), ("rec_head", i32), ("rec_keys", i32), ("rec_out", vp), ("rec_n_stride", i64),
                ("att_ld", i32), ("pipe_words", vp), ("act_rows", i32)]


_SIGS = {
    "jb_last_error": (C.c_char_p, []),
    "jb_version": (i32, []),
    "jb_stream_priority_range": (i32, [C.POINTER(i32), C.POINTER(i32)]),
    "jb_stream_create": (i32, [i32, C.POINTER(vp)]),
    "jb_stream_destroy": (i32, [vp]),
    "jb_stream_create_cu_mask": (i32, [vp, i32, C.POINTER(vp)]),
    "jb_cu_census": (i32, [i32, vp, vp]),
    "jb_clock_probe": (i32, [vp, i32, vp]),
    "jb_packed_weight_bytes": (i64, [i32, i32, i32]),
    "jb_pack_weight": (i32, [vp, i32, i64, i64, i32, i32, vp, i32, vp]),
    "jb_layernorm_fwd": (i32, [vp, i32, vp, i32, vp, vp, i64, i32, f32, vp]),
    "jb_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "jb_gemm_split_overflow": (i32, [i32]),
    "jb_gemv": (i32, [C.POINTER(GemvArgs), vp]),
    "jb_gemv_ln_fold_supported": (i32, [i32, i32, i32, i32]),
    "jb_attn_decode": (i32, [i32, i32, vp, i64, vp, vp, i32, vp, i64, i32, i32, i32, i32, vp, i32, vp]),
    "jb_tune_attn_decode": (None, [i32, i32]),
    "jb_attn_decode_split": (i32, [i32, vp, i64, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp]),
    "jb_attn_decode_split_parts": (i32, [i32, i32, i32]),
    # attn_func, q, ldq, kcache, vcache_w, cache_cap, res, ldr, bias, x_out, ldo, n_batch, d_head, width, block_ctx, t_dev, max_len, stream
    "jb_attn_decode_wide": (i32, [i32, vp, i64, vp, vp, i32, vp, i64, vp, vp, i64, i32, i32, i32, i32, vp, i32, vp]),
    "jb_attn_decode_wide_supported": (i32, [i32, i32, i32, i32, i32]),
    "jb_tune_attn_decode_wide_lean": (None, [i32]),
    "jb_tune_pipeline": (None, [i32]),
    "jb_tune_attn_decode_split": (None, [i32, i32]),
    "jb_tune_attn_decode_split_min_keys": (None, [i32]),
    "jb_tune_gemm_lds": (None, [i32]),
    "jb_tune_gemm_glds": (None, [i32]),
    "jb_tune_gemm_8phase": (None, [i32]),
    "jb_tune_gemm_presplit": (None, [i32]),
    "jb_tune_gemv_long": (None, [i32]),
    "jb_tune_attn_prefill_v2": (None, [i32]),
    "jb_attn_prefill": (i32, [i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "jb_attn_probs": (i32, [i32, i32, vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "jb_embed": (i32, [i32, vp, vp, i64, vp, vp, vp, i64, vp, i64, i64, i32, i32, i32, vp, i32, vp]),
    "jb_final_add": (i32, [i32, vp, vp, i64, vp, i64, i64, i32, i32, i32, vp, i32, vp]),
    "jb_sample_logits": (i32, [vp, i32, i32, vp, vp, i64, vp, vp, i64, vp]),
    "jb_sample_step": (i32, [vp, i32, i32, vp, vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, i64, i64, i32, i32, vp, vp]),
    "jb_vq_gather": (i32, [vp, vp, vp, i64, i32, i32, vp]),
    "jb_vq_argmin": (i32, [vp, vp, vp, vp, i64, i32, i32, vp]),
    "jb_engine_create": (i32, [C.POINTER(EngineCfg), C.POINTER(Layer), C.POINTER(vp)]),
    "jb_engine_destroy": (i32, [vp]),
    "jb_engine_set_encoder_kv": (i32, [vp, vp]),
    "jb_engine_prefill": (i32, [vp, i32, i32, vp]),
    "jb_engine_decode": (i32, [vp, i32, i32, i32, vp]),
    "jb_engine_probe_projection": (i32, [vp, i32, i32, vp, C.POINTER(C.c_double)]),
    "jb_engine_launches_per_step": (i32, [vp]),
    "jb_engine_pipeline": (i32, [vp, i32]),
    "jb_engine_pipelined": (i32, [vp]),
    "jb_engine_pipeline_resident": (i32, [vp]),
    "jb_engine_step_bytes": (C.c_double, [vp, i32]),
}
EXPORTS = tuple(_SIGS)

_lib = None


class JukeboxHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library with argument types set.  Raises if it is absent."""
    global _lib
    if _lib is None:
        # torch-ROCm bundles its own HIP runtime: import it first so that this process has exactly one
        # libamdhip64 (loading the system copy first leaves the second runtime without a device).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise JukeboxHipError(f"{LIB_PATH} is missing: build the HIP extension first "
                                  "(python -m jukebox_amd.csrc.build); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        # experiments: the hand-off form of the pipelined launches (jb_tune_pipeline) from the environment
        if os.environ.get("JB_PIPE_FRAG"):
            l.jb_tune_pipeline(int(os.environ["JB_PIPE_FRAG"]))
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise JukeboxHipError(f"libjukebox_hip status {rc}: {lib().jb_last_error().decode()}")


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float16:
        return F16
    if torch_dtype == torch.float32:
        return F32
    raise JukeboxHipError(f"unsupported dtype {torch_dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def priority_streams(classes, device=None):
    """torch-visible raw HIP streams, one per entry of `classes` (HIP priority values, clamped to the device range;
    numerically lower = higher priority).  Returns (streams, raw_handles); destroy with destroy_streams."""
    import torch
    least, greatest = i32(), i32()
    check(lib().jb_stream_priority_range(C.byref(least), C.byref(greatest)))
    streams, raw = [], []
    for c in classes:
        h = vp()
        check(lib().jb_stream_create(max(greatest.value, min(least.value, c)), C.byref(h)))
        raw.append(h)
        streams.append(torch.cuda.ExternalStream(h.value, device=device))
    return streams, raw


def cu_mask_stream(mask_bits, device=None):
    """torch-visible raw HIP stream confined to the CUs whose index is in `mask_bits` (iterable of bit positions < 256).
    Returns (stream, raw_handle); destroy with destroy_streams([raw_handle])."""
    import torch
    words = (C.c_uint32 * 8)()
    for b in mask_bits:
        words[b >> 5] |= 1 << (b & 31)
    h = vp()
    check(lib().jb_stream_create_cu_mask(words, 8, C.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=device), h


def destroy_streams(raw):
    for h in raw:
        lib().jb_stream_destroy(h)
