"""Host-side binding of one autoregressive prior to the HIP decode engine (jb_engine_*).

Plumbing only: torch allocates the device buffers (weights in MFMA order, k/v caches sized for
288 GB HBM, work buffers) and passes raw pointers to the C ABI; every FLOP of the token loop runs
in libjukebox_hip.so.  Mirrors the state that the reference keeps inside
Transformer / FactoredAttention while sampling (jukebox/transformer/factored_attention.py:75-76,
355-381): `sample_t` is the device counter `t_dev`, the per-layer `cache` dict is the static
(N, seq_len, n_state) k/v arrays.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import hip_ops as H

# transformer.py:110-126 (reference) -- attn_order -> attn_func per layer
_ORDERS = {
    0: lambda d: 0, 1: lambda d: [1, 2][d % 2], 2: lambda d: [1, 2, 3][d % 3], 3: lambda d: [1, 4][d % 2],
    4: lambda d: [1, 5][d % 2], 5: lambda d: [1, 4, 1, 1][d % 4], 6: lambda d: [1, 2, 3, 6][d % 4],
    7: lambda d: [*[1, 2, 3] * 5, 6][d % 16], 8: lambda d: [1, 2, 3, 1, 2, 3, 1, 2, 3, 6][d % 10],
    9: lambda d: [1, 2, 3, 0][d % 4],
    10: lambda d: [*[1, 2, 3, 1, 2, 3, 1, 2, 3], *[1, 2, 3, 1, 2, 3, 1, 2, 3, 6] * 7][d % 79],
    11: lambda d: [6, 6, 0][d % 3] if d % 16 == 15 else [1, 2, 3][d % 3],
    12: lambda d: [7, 7, 0][d % 3] if d % 16 == 15 else [1, 2, 3][d % 3],
}


def attn_funcs(attn_order, depth):
    return [_ORDERS[attn_order](d) for d in range(depth)]


def wide_value_weights(w_attn, w_proj, b_attn, S, dtype):
    """c_attn of a wide-value layer: (W x (2S + W) weight, bias) producing q | k | v'.  With one head,
    attn.c_proj(sum_k p_k v_k) = sum_k p_k (v_k·Wp) + bp (factored_attention.py:104-108,118-121 are linear in v), so the
    cache may hold v' = v·Wp = LN(x)·(Wv·Wp) + bv·Wp and the attention output is already the projected one.  Wv and Wp are
    first rounded to the engine dtype, as the reference uses them (`w.type_as(x)`, ops.py:99); the product is rounded once.
    Prefill keeps its own c_attn / attention / c_proj and fills v' from the v rows it caches (jb_engine_prefill)."""
    w16, wp16 = w_attn.to(dtype), w_proj.to(dtype)
    vp = (w16[:, 2 * S:].double() @ wp16.double()).to(dtype)                           # (W, W)
    b_vp = (b_attn[2 * S:].double() @ wp16.double()).float()
    return torch.cat([w16[:, :2 * S], vp], 1).contiguous(), torch.cat([b_attn[:2 * S].float(), b_vp]).contiguous()


class PackedPrior:
    """The weights of one ConditionalAutoregressive2D, re-laid once for the MFMA kernels (jb_pack_weight) in one engine
    dtype: per layer the four projections in fragment order, biases / LayerNorm parameters in fp32, the folded-LayerNorm
    images of c_attn / c_fc (hip_ops.FoldedLN) where the decode kernels can use them, the c_enc_kv halves of
    cross-attention layers, plus the embedding tables and the fp32 logits head.  Shared by every PriorEngine of the same
    prior (different batch sizes, with / without logits recording), so a new batch size costs caches and work buffers
    only -- not another pass over 1-10 GB of weights.

    sd: mapping reference-name -> GPU tensor for the keys under `prefix`
        (x_emb.weight, pos_emb.pos_emb, [start_token], transformer._attn_mods.*, x_out.weight)."""

    def __init__(self, sd, prefix, *, seq_len, bins, width, depth, heads, attn_order, blocks=None, m_attn=0.25, m_mlp=1.0,
                 prime_len=None, y_cond=False, add_cond_after=True, fp16=True, encoder_dims=0, only_encode=False,
                 fold_ln=None, wide_v=None, device="cuda"):
        L.lib()
        self.device = torch.device(device)
        self.T, self.bins, self.W = seq_len, bins, width
        self.S, self.M, self.H, self.depth = int(m_attn * width), int(m_mlp * width), heads, depth
        self.fp16 = bool(fp16)
        self.dtype = torch.float16 if fp16 else torch.float32
        self.code = L.F16 if fp16 else L.F32
        self.block_ctx = seq_len // blocks if blocks else 0
        self.prime_cap = (prime_len // blocks + 1) * blocks if prime_len else 0
        self.funcs = attn_funcs(attn_order, depth)
        self.y_cond, self.add_cond_after, self.only_encode = y_cond, add_cond_after, only_encode
        self.enc_len = int(encoder_dims or 0)
        # Decode step: LayerNorm folded into c_attn / c_fc (hip_ops.FoldedLN).  Default (fold_ln=None): on for fp16 engines;
        # the fp32 engine (the parity mode) normalises rows explicitly, operation for operation as the reference.
        if fold_ln is None:
            fold_ln = fp16
        self.fold_ln = bool(fold_ln) and not only_encode
        # Wide-value layers (jb_layer.vcache_w): single-head fp16 models cache v' = v·Wp, the value already carried through
        # attn.c_proj, so the decode step's attention writes the residual stream and the attn.c_proj launch disappears
        # (4 launches per layer instead of 5).  Needs the folded c_attn.  wide_v=False keeps the five-launch form.
        if wide_v is None:
            wide_v = True
        self.wide_v = bool(wide_v) and self.fold_ln and fp16 and heads == 1
        dev, dt = self.device, self.dtype
        g = lambda name: sd[prefix + name].to(dev).contiguous()
        f32 = lambda name: g(name).float().contiguous()
        self.x_emb = f32("x_emb.weight")
        self.pos_emb = f32("pos_emb.pos_emb")
        self.start_token = None if y_cond else f32("start_token")
        self.x_out = None if only_encode else H.pack_linear_w(f32("x_out.weight"), torch.float32)
        W, S, M = self.W, self.S, self.M
        self.layers = []
        for d in range(depth):
            p = f"transformer._attn_mods.{d}."
            func = self.funcs[d]
            if func not in (0, 1, 2, 3, 6, 7):
                raise L.JukeboxHipError(f"attn_func {func} is not supported by the engine")
            lay = dict(func=func,
                       ws=[H.pack_conv1d_w(g(p + n), dt) for n in ("attn.c_attn.w", "attn.c_proj.w", "mlp.c_fc.w", "mlp.c_proj.w")],
                       bs=[f32(p + n) for n in ("attn.c_attn.b", "attn.c_proj.b", "mlp.c_fc.b", "mlp.c_proj.b")],
                       lns=[f32(p + n) for n in ("ln_0.weight", "ln_0.bias", "ln_1.weight", "ln_1.bias")],
                       cap=self.prime_cap if func == 7 else (self.enc_len if func == 6 else self.T),
                       f_attn=None, f_fc=None, enc=None, b_enc=None, wide=None)
            if func == 6:
                # c_enc_kv (n_in, 2*n_state): key half and value half as separate packed matrices
                wkv = g(p + "attn.c_enc_kv.w")
                lay["enc"] = [H.PackedWeight(H.pack_weight(wkv, W, S, 2 * S, 1, dt, offset_elems=part * S), W, S, dt) for part in (0, 1)]
                lay["b_enc"] = f32(p + "attn.c_enc_kv.b")
            if self.fold_ln:
                j_attn = S if func == 6 else 3 * S
                if H.ln_fold_supported(dt, W, j_attn, 1):
                    lay["f_attn"] = H.FoldedLN(g(p + "attn.c_attn.w"), lay["bs"][0], lay["lns"][0], lay["lns"][1], dt)
                if H.ln_fold_supported(dt, W, M, 1):
                    lay["f_fc"] = H.FoldedLN(g(p + "mlp.c_fc.w"), lay["bs"][2], lay["lns"][2], lay["lns"][3], dt)
                if self.wide_v and func != 6 and lay["f_attn"] is not None and H.ln_fold_supported(dt, W, 2 * S + W, 1) and \
                        L.lib().jb_attn_decode_wide_supported(func, S, W, self.block_ctx, seq_len):
                    lay["wide"] = self._wide_images(g(p + "attn.c_attn.w"), g(p + "attn.c_proj.w"), lay["bs"][0],
                                                    lay["lns"][0], lay["lns"][1])
                for f in (lay["f_attn"], lay["f_fc"], lay["wide"] and lay["wide"]["fold"]):
                    if f:
                        f.wf = None                  # the unpacked image is bind-time only
            self.layers.append(lay)

    def _wide_images(self, w_attn, w_proj, b_attn, ln_g, ln_b):
        """Folded decode image of a wide-value layer's c_attn (see wide_value_weights)."""
        w_dec, b_dec = wide_value_weights(w_attn, w_proj, b_attn, self.S, self.dtype)
        return dict(fold=H.FoldedLN(w_dec, b_dec, ln_g, ln_b, self.dtype))

    def weight_bytes(self):
        n = self.x_out.data.numel() * 4 if self.x_out is not None else 0
        for lay in self.layers:
            for w in lay["ws"] + (lay["enc"] or []):
                n += w.data.numel() * w.data.element_size()
        return n


class PriorEngine:
    """One ConditionalAutoregressive2D bound to device buffers for a fixed batch size: k/v caches, work buffers,
    conditioning, token buffer, and the native engine handle (jb_engine_*).  The packed weights come from a PackedPrior
    (`packed=`), or are packed here from `sd` when none is given.

    The conditioning inputs are COPIED into persistent buffers, so the handle and its captured hipGraph survive across
    windows (set_cond per window is a device copy, not a re-capture)."""

    PIPE_STAMPS = 16           # int64 clock stamps per launch slot behind the completion words (common.h: JB_PIPE_STAMPS)

    def __init__(self, sd=None, prefix="", *, n_batch, chunk_cap=256, want_preds=False, record=None, packed=None,
                 attn_split=None, **model):
        if packed is None:
            packed = PackedPrior(sd, prefix, **model)
        self.packed = pk = packed
        self.device = pk.device
        self.N, self.T, self.bins, self.W = n_batch, pk.T, pk.bins, pk.W
        self.S, self.M, self.H, self.depth = pk.S, pk.M, pk.H, pk.depth
        self.dtype, self.code = pk.dtype, pk.code
        self.block_ctx, self.prime_cap, self.funcs = pk.block_ctx, pk.prime_cap, pk.funcs
        self.y_cond, self.add_cond_after, self.only_encode = pk.y_cond, pk.add_cond_after, pk.only_encode
        self.enc_len, self.fold_ln = pk.enc_len, pk.fold_ln
        self.chunk_cap = min(chunk_cap, self.T)
        dev, dt = self.device, self.dtype
        N, T, S, W, M = self.N, self.T, self.S, self.W, self.M
        self.layers_c = (L.Layer * self.depth)()
        self.kcaches, self.vcaches, self.vcaches_w = [], [], []
        # key-split decode attention (fp16 engines whose head size the split kernel takes) for layers with long key sets;
        # attn_split=False: never (the reference-ordered engine of the fp16 gate, tests/test_hip_baseline_configs.py)
        split_off = attn_split is False
        bc = max(self.block_ctx, 1)
        max_keys = lambda lay: {0: T, 1: bc, 2: (T + bc - 1) // bc, 3: bc}.get(lay["func"], lay["cap"])
        splits = lambda lay: (not self.only_encode and not split_off and N <= 32 and
                              L.lib().jb_attn_decode_split_parts(self.code, S // self.H, max_keys(lay)) > 0)
        for d, lay in enumerate(pk.layers):
            kc = torch.zeros((N, lay["cap"], S), dtype=dt, device=dev)
            vc = torch.zeros((N, lay["cap"], S), dtype=dt, device=dev)
            self.kcaches.append(kc)
            self.vcaches.append(vc)
            lc = self.layers_c[d]
            lc.attn_func = lay["func"]
            lc.w_attn, lc.w_proj, lc.w_fc, lc.w_proj2 = (w.ptr for w in lay["ws"])
            lc.b_attn, lc.b_proj, lc.b_fc, lc.b_proj2 = (b.data_ptr() for b in lay["bs"])
            lc.ln0_g, lc.ln0_b, lc.ln1_g, lc.ln1_b = (t.data_ptr() for t in lay["lns"])
            lc.kcache, lc.vcache, lc.cache_cap = kc.data_ptr(), vc.data_ptr(), lay["cap"]
            if self.fold_ln:
                j_attn = S if lay["func"] == 6 else 3 * S
                f = lay["f_attn"]
                if f is not None and H.ln_fold_supported(dt, W, j_attn, N):
                    lc.w_attn_f, lc.b_attn_f, lc.c1_attn = f.pw.ptr, f.bias.data_ptr(), f.c1.data_ptr()
                f = lay["f_fc"]
                if f is not None and H.ln_fold_supported(dt, W, M, N):
                    lc.w_fc_f, lc.b_fc_f, lc.c1_fc = f.pw.ptr, f.bias.data_ptr(), f.c1.data_ptr()
            if lay["func"] == 6:
                lc.w_enc_k, lc.w_enc_v, lc.b_enc_kv = lay["enc"][0].ptr, lay["enc"][1].ptr, lay["b_enc"].data_ptr()
            wd = lay["wide"]
            if wd is not None and lc.w_attn_f and H.ln_fold_supported(dt, W, 2 * S + W, N) and not splits(lay):
                vw = torch.zeros((N, lay["cap"], W), dtype=dt, device=dev)
                self.vcaches_w.append(vw)
                lc.w_attn_fw, lc.b_attn_fw, lc.c1_attn_w = wd["fold"].pw.ptr, wd["fold"].bias.data_ptr(), wd["fold"].c1.data_ptr()
                lc.vcache_w = vw.data_ptr()
            else:
                self.vcaches_w.append(None)

        e = lambda *shape, dtype=dt: torch.zeros(shape, dtype=dtype, device=dev)
        Cc = self.chunk_cap
        # attention output rows padded to whole k-tiles (zeros): attn.c_proj's branch-free path for n_state = 1200 (5b_lyrics)
        kt = 32 if self.dtype == torch.float16 else 16
        self.att_ld = (S + kt - 1) // kt * kt
        # x_a / x_b / mlp hold 16 rows: pipelined launches hand them over in MFMA operand order, [k-tile][lane][8 channels] of
        # always 16 rows (jb_engine_cfg.act_rows)
        self.act_rows = max(N, 16)
        R = self.act_rows
        self.buf = dict(x_a=e(R, W), x_b=e(R, W), q=e(N, S), att=e(N, self.att_ld), mlp=e(R, M),
                        xf=e(N, W, dtype=torch.float32), logits=e(N, max(self.bins, 1), dtype=torch.float32),
                        c_xa=e(N * Cc, W), c_xb=e(N * Cc, W), c_h=e(N * Cc, W), c_q=e(N * Cc, S), c_att=e(N * Cc, S),
                        c_mlp=e(N * Cc, M))
        # partial softmax states of the key-split layers
        self.att_parts = self.att_ml = None
        if any(splits(lay) for lay in pk.layers):
            self.att_parts = e(N, 4, S)
            self.att_ml = e(N, self.H, 4, 2, dtype=torch.float32)
        # completion words of software-pipelined launches (jb_engine_pipeline): counts + tickets per launch slot, error word last
        # (+ room for the JB_PIPE_DEBUG stamps: 4 x int64 per slot behind the error group)
        # (an odd number of launches per step gets one pad slot: 5 * depth + 3 covers every engine)
        self.pipe_words = torch.zeros((18 * (5 * self.depth + 3) + 1) * 32 + (5 * self.depth + 3) * 2 * self.PIPE_STAMPS,
                                      dtype=torch.int32, device=dev)
        self.pipelined = False
        self.tokens = torch.zeros((N, T), dtype=torch.int64, device=dev)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.preds = e(N, T, self.bins, dtype=torch.float32) if want_preds else None
        if want_preds:
            self.buf["c_xf"] = e(N * Cc, W, dtype=torch.float32)
        self.sample_params = H.make_sample_params(device=dev)
        # record = (layer, head, n_keys): keep that head's attention probabilities during prefill (alignment)
        self.record = record
        self.rec_out = e(N, T, record[2], dtype=torch.float32) if record else None
        self.hidden = e(N, T, W, dtype=torch.float32) if self.only_encode else None
        self.encoder_kv = e(N, self.enc_len, W) if self.enc_len else None
        # persistent conditioning: start row(s) and x_cond in the shape class last seen ("full" (N, T, W) / "bcast" (N, 1, W) / None)
        self.start = e(N, W, dtype=torch.float32) if self.y_cond else pk.start_token.reshape(W).contiguous()
        self.start_stride = W if self.y_cond else 0
        self.x_cond = None
        self._cond_mode = "unset"
        self.handle = None

    # -- conditioning / sampler state ------------------------------------------------------------------
    def set_cond(self, x_cond, y_cond):
        """x_cond: (N, T, W) / (N, 1, W) fp32 or None; y_cond: (N, 1, W) when the model is y-conditioned.  Values are
        copied into the engine's own buffers (same device addresses every window)."""
        mode = None
        if x_cond is not None:
            assert x_cond.shape[0] == self.N and x_cond.shape[2] == self.W and x_cond.shape[1] in (1, self.T)
            mode = "full" if x_cond.shape[1] == self.T and self.T > 1 else "bcast"
        if self.y_cond:
            assert y_cond is not None and tuple(y_cond.shape) == (self.N, 1, self.W)
            self.start.copy_(y_cond.reshape(self.N, self.W))
        if mode != self._cond_mode:
            self.x_cond = None if mode is None else torch.empty((self.N, self.T if mode == "full" else 1, self.W),
                                                                dtype=torch.float32, device=self.device)
            self._cond_mode = mode
            self._create()
        if x_cond is not None:
            self.x_cond.copy_(x_cond)

    def set_sampling(self, temp=1.0, top_k=0, top_p=0.0, seed=0, sample_base=0, pos_base=0, stream_id=0):
        new = H.make_sample_params(temp, top_k, top_p, seed, sample_base, pos_base, stream_id, device=self.device)
        self.sample_params.copy_(new)          # same device address: captured graphs stay valid

    def _create(self):
        self.close()
        pk = self.packed
        c = L.EngineCfg()
        c.dtype, c.n_batch, c.width, c.n_state, c.n_head, c.n_mlp = self.code, self.N, self.W, self.S, self.H, self.M
        c.n_layers, c.seq_len, c.block_ctx, c.bins, c.ln_eps = self.depth, self.T, self.block_ctx, self.bins, 1e-5
        c.x_emb, c.pos_emb = pk.x_emb.data_ptr(), pk.pos_emb.data_ptr()
        c.x_out_packed = pk.x_out.ptr if pk.x_out is not None else None
        if self.encoder_kv is not None:
            c.encoder_kv, c.enc_len = self.encoder_kv.data_ptr(), self.enc_len
        if self.hidden is not None:
            c.hidden_out, c.hidden_n_stride = self.hidden.data_ptr(), self.hidden.stride(0)
        c.start, c.start_stride = self.start.data_ptr(), self.start_stride
        if self.x_cond is not None:
            c.x_cond = self.x_cond.data_ptr()
            c.xc_n_stride = self.x_cond.stride(0)
            c.xc_t_stride = self.x_cond.stride(1) if self.x_cond.shape[1] > 1 else 0
        c.add_cond_after = int(self.add_cond_after)
        b = self.buf
        for k in ("x_a", "x_b", "q", "att", "mlp", "xf", "logits", "c_xa", "c_xb", "c_h", "c_q", "c_att", "c_mlp"):
            setattr(c, k, b[k].data_ptr())
        if self.att_parts is not None:
            c.att_parts, c.att_ml = self.att_parts.data_ptr(), self.att_ml.data_ptr()
        c.ticket = self.ticket.data_ptr()
        c.att_ld = self.att_ld
        c.pipe_words = self.pipe_words.data_ptr()
        c.act_rows = self.act_rows
        c.chunk_cap = self.chunk_cap
        c.c_xf = b["c_xf"].data_ptr() if "c_xf" in b else None
        c.tokens, c.tok_stride, c.t_dev = self.tokens.data_ptr(), self.tokens.stride(0), self.t_dev.data_ptr()
        if self.preds is not None:
            c.preds, c.preds_n_stride = self.preds.data_ptr(), self.preds.stride(0)
        c.sample_params = self.sample_params.data_ptr()
        c.rec_layer = -1
        if self.record:
            c.rec_layer, c.rec_head, c.rec_keys = self.record
            c.rec_out, c.rec_n_stride = self.rec_out.data_ptr(), self.rec_out.stride(0)
        h = C.c_void_p()
        L.check(L.lib().jb_engine_create(C.byref(c), self.layers_c, C.byref(h)))
        self.handle = h
        # Software-pipelined launches (one engine per process at a time, jb_engine_pipeline): requested by the sampler for the
        # level that is the long pole (set_pipelined), or for every eligible engine as it is created with
        # JB_PIPELINE_LAUNCHES=1 (first come, first served); JB_PIPELINE_LAUNCHES=0 forbids them.
        want = self.pipelined or os.environ.get("JB_PIPELINE_LAUNCHES", "") == "1"
        self.pipelined = False
        if want and not self.only_encode and os.environ.get("JB_PIPELINE_LAUNCHES", "") != "0":
            self.pipelined = L.lib().jb_engine_pipeline(self.handle, 1) == 0

    def set_pipelined(self, on, fresh=False):
        """Switch software-pipelined launches of the decode step on / off; returns whether they are on (they stay off for
        shapes without pipelined kernels, while another engine has them, and under JB_PIPELINE_LAUNCHES=0).  Switching them
        off RELEASES the engine's pair of streams and its two graphs (an idle pair slows the plain chains of every other
        engine of the process); the next pipelined decode makes new ones.  `fresh`: the same release while staying on."""
        if on and os.environ.get("JB_PIPELINE_LAUNCHES", "") == "0":
            return False
        if self.handle is None:
            self.pipelined = bool(on)          # remembered for _create
            return self.pipelined
        rc = L.lib().jb_engine_pipeline(self.handle, (2 if fresh else 1) if on else 0)
        if on and rc != 0:
            self.pipelined = False
            return False
        L.check(rc)
        self.pipelined = bool(on)
        return self.pipelined

    @property
    def pipeline_resident(self):
        """Whether the engine holds a pair of streams for pipelined launches (first pipelined decode .. set_pipelined(False))."""
        return bool(self.handle) and L.lib().jb_engine_pipeline_resident(self.handle) == 1

    def pipe_stamps(self):
        """JB_PIPE_DEBUG=1: (n_slots, PIPE_STAMPS) int64 ticks of the 100 MHz clock of the last pipelined step, workgroup 0 of
        every launch: 0 poll entered, 1 producer seen, 2 completion published (the launch's last workgroup), 3 stores issued;
        the measurement build (JB_LIB_SEGMENTS=1) adds 4 barrier behind the poll, 5 operands landed, 6 arithmetic retired and
        partials in LDS, 7 LDS exchange barrier, 8 stores drained, 9 own ticket returned (common.h)."""
        n = self.pipe_slots
        base = (18 * n + 1) * 32
        return self.pipe_words[base:base + n * 2 * self.PIPE_STAMPS].view(torch.int64).reshape(n, self.PIPE_STAMPS).cpu().numpy()

    def pipe_error(self):
        """0, or slot + 1 of a pipelined launch whose wait for its producer timed out (sticky)."""
        return int(self.pipe_words[18 * self.pipe_slots * 32].item())

    def clear_pipe_error(self):
        """Forget a recorded timeout (the engine must be idle).  Until then every pipelined wait gives up at once."""
        self.pipe_words[18 * self.pipe_slots * 32] = 0

    def close(self):
        if getattr(self, "handle", None):
            L.lib().jb_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_encoder_kv(self, encoder_kv):
        """encoder_kv (N, enc_len, W): lyric-encoder states for the cross-attention layers; projects them through every
        c_enc_kv into the layer caches (decode_qkv at sample_t == 0).  Call after set_cond."""
        assert self.encoder_kv is not None and tuple(encoder_kv.shape) == tuple(self.encoder_kv.shape)
        self.encoder_kv.copy_(encoder_kv.to(self.dtype))
        L.check(L.lib().jb_engine_set_encoder_kv(self.handle, L.stream()))

    # -- the two hot loops ---------------------------------------------------------------------------------
    def prefill(self, t0, n_t):
        L.check(L.lib().jb_engine_prefill(self.handle, t0, n_t, L.stream()))

    def decode(self, t0, n_steps, use_graph=True, plain=False):
        """use_graph: True / 1 = graph replay (pipelined launches while they are on), False / 0 = the eager plain chain, 2 =
        the pipelined launches without the graph executor (diagnostics); plain: the plain chain's graph even while pipelined
        launches are on (the pair of streams stays)."""
        L.check(L.lib().jb_engine_decode(self.handle, t0, n_steps, 3 if plain else int(use_graph), L.stream()))
        if self.pipelined and not L.lib().jb_engine_pipelined(self.handle):
            # the library keeps the plain chain when the pair of streams cannot have hardware queues of its own: say why, once
            self.pipelined = False
            import sys
            print("jukebox_amd: pipelined launches fell back to the plain chain:", L.lib().jb_last_error().decode(errors="replace"),
                  file=sys.stderr, flush=True)

    def timed_decode(self, t0, n_steps, plain=False):
        """decode + wait: seconds per step as the host sees them (the in-situ comparison of the two launch forms)."""
        import time
        torch.cuda.current_stream(self.device).synchronize()
        t = time.perf_counter()
        self.decode(t0, n_steps, plain=plain)
        torch.cuda.current_stream(self.device).synchronize()
        return (time.perf_counter() - t) / max(n_steps, 1)

    def probe_projection(self, t0, n_steps):
        """(avg_us_per_launch, launches, avg_algorithmic_bytes) of the LN-fused projection kernel, timed in situ."""
        out = (C.c_double * 3)()
        L.check(L.lib().jb_engine_probe_projection(self.handle, t0, n_steps, L.stream(), out))
        return out[0], int(out[1]), out[2]

    @property
    def pipe_slots(self):
        """Completion slots of a pipelined step: its launches, plus one pad launch when their number is odd (engine.hip)."""
        n = self.launches_per_step
        return n + (n & 1)

    @property
    def launches_per_step(self):
        return L.lib().jb_engine_launches_per_step(self.handle)

    def step_bytes(self, t):
        """Algorithmic HBM bytes of one decode step at position t."""
        return float(L.lib().jb_engine_step_bytes(self.handle, int(t)))

    def cache_bytes(self):
        return sum(k.numel() * k.element_size() * 2 for k in self.kcaches) + \
            sum(v.numel() * v.element_size() for v in self.vcaches_w if v is not None)

    def weight_bytes(self):
        return self.packed.weight_bytes()
