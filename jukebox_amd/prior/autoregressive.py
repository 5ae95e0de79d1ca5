"""ConditionalAutoregressive2D with the reference's parameter tree and sampling API
(jukebox/prior/autoregressive.py:48-359).  `sample` / `primed_sample` run the whole token loop inside
the HIP decode engine (jukebox_amd.engine.PriorEngine): one hipGraph replay per token, chunked MFMA
prefill for the primed part; nothing is computed in torch."""
import math
import os

import numpy as np
import torch as t
import torch.nn as nn

from ..engine import PackedPrior, PriorEngine
from ..transformer.transformer import Transformer


def get_normal(*shape, std=0.01):
    w = t.empty(shape)
    nn.init.normal_(w, std=std)
    return w


def split_chunks(length, chunk_size):
    """autoregressive.py:19-23."""
    n_passes = (length + chunk_size - 1) // chunk_size
    chunk_sizes = [*[chunk_size] * (n_passes - 1), (length - 1) % chunk_size + 1]
    assert sum(chunk_sizes) == length
    return chunk_sizes


class PositionEmbedding(nn.Module):
    def __init__(self, input_shape, width, init_scale=1.0, pos_init=False):
        super().__init__()
        assert not pos_init, "pos_init is not used by any released model"
        self.input_shape = input_shape
        self.input_dims = int(np.prod(input_shape))
        self.pos_emb = nn.Parameter(get_normal(self.input_dims, width, std=0.01 * init_scale))

    def forward(self):
        return self.pos_emb


class ConditionalAutoregressive2D(nn.Module):
    def __init__(self, input_shape, bins, width=128, depth=2, heads=1, attn_dropout=0.0, resid_dropout=0.0,
                 emb_dropout=0.0, mask=True, zero_out=False, init_scale=1.0, res_scale=False, pos_init=False,
                 m_attn=0.25, m_mlp=1, checkpoint_res=0, checkpoint_attn=0, checkpoint_mlp=0, attn_order=0,
                 blocks=None, spread=None, x_cond=False, y_cond=False, encoder_dims=0, only_encode=False,
                 merged_decoder=False, prime_len=None):
        super().__init__()
        self.input_shape = input_shape
        self.input_dims = int(np.prod(input_shape))
        self.encoder_dims, self.bins, self.width, self.depth = encoder_dims, bins, width, depth
        self.heads, self.attn_order, self.blocks, self.m_attn, self.m_mlp = heads, attn_order, blocks, m_attn, m_mlp
        assert not res_scale, "res_scale is not used by any released model"
        self.x_emb = nn.Embedding(bins, width)
        nn.init.normal_(self.x_emb.weight, std=0.02 * init_scale)
        self.y_cond, self.x_cond = y_cond, x_cond
        if not y_cond:
            self.start_token = nn.Parameter(get_normal(1, width, std=0.01 * init_scale))
        self.pos_emb = PositionEmbedding(input_shape=input_shape, width=width, init_scale=init_scale, pos_init=pos_init)
        self.transformer = Transformer(n_in=width, n_ctx=self.input_dims, n_head=heads, n_depth=depth, mask=mask,
                                       zero_out=zero_out, init_scale=init_scale, res_scale=res_scale, m_attn=m_attn,
                                       m_mlp=m_mlp, attn_order=attn_order, blocks=blocks, spread=spread,
                                       encoder_dims=encoder_dims, prime_len=prime_len)
        self.only_encode, self.prime_len = only_encode, prime_len
        self.add_cond_after_transformer = not merged_decoder
        self.share_x_emb_x_out = not merged_decoder
        if not only_encode:
            self.x_out = nn.Linear(width, bins, bias=False)
            if self.share_x_emb_x_out:
                self.x_out.weight = self.x_emb.weight
        self._engines, self._packed, self._last_engine = {}, {}, None

    def _apply(self, fn, *a, **k):
        before = (self.x_emb.weight.device, self.x_emb.weight.dtype, self.x_emb.weight.data_ptr())
        out = super()._apply(fn, *a, **k)
        after = (self.x_emb.weight.device, self.x_emb.weight.dtype, self.x_emb.weight.data_ptr())
        if before != after:             # device copies are dropped when the module really moves (prior.cpu())
            self._engines, self._packed, self._last_engine = {}, {}, None
        return out

    def preprocess(self, x):
        return x.view(x.shape[0], -1).long()

    def postprocess(self, x, sample_tokens=None):
        N = x.shape[0]
        assert (0 <= x).all() and (x < self.bins).all()
        if sample_tokens is None or sample_tokens == self.input_dims:
            return x.view(N, *self.input_shape)
        return x.view(N, -1)

    # ---- engine binding --------------------------------------------------------------------------------
    def engine(self, n_samples, fp16, want_preds=False, chunk_cap=2048):
        """The engine bound to this prior for a batch size.  The packed weights (PackedPrior) are built once per dtype
        and shared; an engine per (batch size, dtype, logits recording) adds only its k/v caches and work buffers and
        stays bound, so alternating batch sizes (split_batch tails) or get_preds calls never re-pack weights.
        chunk_cap: positions per prefill chunk (measured, upsampler, N = 16, 4096 primed tokens: 449 ms at 512, 380 ms at
        2048 -- M = 32768-row GEMMs leave no tail on the 256 x 128 tiling)."""
        fp16 = bool(fp16)
        packed = self.packed(fp16)
        key = (n_samples, fp16, bool(want_preds))
        if key not in self._engines:
            make = lambda: PriorEngine(packed=packed, n_batch=n_samples, chunk_cap=chunk_cap, want_preds=want_preds)
            try:
                self._engines[key] = make()
            except t.cuda.OutOfMemoryError:
                # the budget is checked after the new engine exists (its size is not known before): when the old engines plus
                # the new one do not fit, the old ones go first and the construction is repeated once
                for k in list(self._engines):
                    self._engines.pop(k).close()
                self._last_engine = None
                t.cuda.synchronize(self.x_emb.weight.device)
                t.cuda.empty_cache()
                self._engines[key] = make()
        eng = self._engines.pop(key)
        self._engines[key] = eng                     # dicts keep insertion order: most recently used last
        self._evict_engines()
        self._last_engine = eng
        self._apply_pipeline(eng)
        return eng

    def engine_cache_budget(self):
        """Bytes of k/v caches this prior keeps bound across calls (`engine_cache_bytes` attribute, JB_ENGINE_CACHE_GB, or 30 %
        of the device's memory: one 16-sample upsampler engine is 54 GB, 36 GB of it the wide-value cache)."""
        b = getattr(self, "engine_cache_bytes", None)
        if b is None and os.environ.get("JB_ENGINE_CACHE_GB"):
            b = float(os.environ["JB_ENGINE_CACHE_GB"]) * 1e9
        if b is None:
            dev = self.x_emb.weight.device
            b = 0.3 * t.cuda.get_device_properties(dev).total_memory if dev.type == "cuda" else float("inf")
        return b

    def _evict_engines(self):
        """Least recently used engines go while the bound caches exceed the budget; the one just asked for always stays.
        The packed weights are shared and stay."""
        budget = self.engine_cache_budget()
        keys = list(self._engines)
        total = sum(self._engines[k].cache_bytes() for k in keys)
        for k in keys[:-1]:
            if total <= budget:
                break
            eng = self._engines.pop(k)
            total -= eng.cache_bytes()
            if eng is self._last_engine:
                self._last_engine = None
            if self.x_emb.weight.device.type == "cuda":
                t.cuda.synchronize(self.x_emb.weight.device)     # its graphs and buffers may still be in flight (rare path)
            eng.close()

    @property
    def pipeline_candidate(self):
        """Whether this model's engines can run software-pipelined launches, from the geometry alone (the library decides for
        an engine: jb_engine_pipeline's eligibility rule -- fp16, <= 16 samples, one 480- or 256-channel head on wide-value
        attention, width and MLP of 32..64 k-tiles, key sets of <= 128 keys: the 1b upsamplers, small_prior / small_upsampler)."""
        S, M = int(self.m_attn * self.width), int(self.m_mlp * self.width)
        bc = self.input_dims // self.blocks if self.blocks else 0
        return (self.heads == 1 and S in (480, 256) and not self.only_encode and self.attn_order == 2 and 0 < bc <= 128
                and self.blocks <= 128 and self.width % 32 == 0 and M % 32 == 0 and 32 <= self.width // 32 <= 64
                and 32 <= M // 32 <= 64)

    def _apply_pipeline(self, eng):
        """Software-pipelined launches as the sampler asks for them: `pipeline_launches` is None (leave the engine alone), a
        bool, or a callable that is asked again before every decode call (the level pipeline: "does this level have the GPU to
        itself now?").  A verdict of the in-situ comparison (`_decode`) against them stands for as long as the engine's pair of
        streams would (`release_pipeline` forgets it)."""
        want = getattr(self, "pipeline_launches", None)
        if callable(want):
            want = want()
        if want is None:
            return
        want = bool(want) and getattr(eng, "_pipe_verdict", None) is not False
        if eng.pipelined != want:
            eng.set_pipelined(want)

    def release_pipeline(self):
        """End of the phase in which this prior's engines may run pipelined launches (its level has finished, a job starts or
        ends): every engine goes back to the plain chain and RELEASES its pair of streams and graphs -- a pair that merely
        exists slows the other levels' plain chains (BENCH_r04: levels 2 / 1 ran 3.8x / 1.8x slower in every job after a
        process's first), and the reference's loop leaves nothing behind either (jukebox/sample.py:90-121).  The in-situ
        verdict belongs to the pair it was measured on and goes with it; an engine on which a wait once TIMED OUT keeps the
        plain chain for good."""
        for eng in list(self._engines.values()):
            eng.set_pipelined(False)
            if not getattr(eng, "_pipe_timed_out", False):
                eng._pipe_verdict = None

    # steps between two looks at `pipeline_launches` while a window is decoded on the plain chain (_decode_window)
    PIPE_RECHECK_STEPS = 512

    def _decode_window(self, eng, t0, n_steps):
        """Decode a window that nobody taps.  When `pipeline_launches` is a callable that says "not now" (the level pipeline's
        lowest level while upper levels still run), the window starts on the plain chain in chunks of PIPE_RECHECK_STEPS steps
        -- one host wait per chunk, about a second apart -- and the sampler is asked again between them: the upper levels
        usually finish INSIDE one of this level's windows, and the rest of that window (up to 4000 steps at 1.86 instead of
        1.56 ms in the 20-second job) need not wait for the window's end.  Same tokens in every form."""
        want = getattr(self, "pipeline_launches", None)
        pos, end = t0, t0 + n_steps
        if callable(want):
            while True:
                self._apply_pipeline(eng)
                # the rest in one call: once the launches are on, once the in-situ comparison has decided against them (nothing
                # can re-open that verdict before the pair is released), or when too little of the window is left
                if eng.pipelined or getattr(eng, "_pipe_verdict", None) is False or end - pos < 2 * self.PIPE_RECHECK_STEPS:
                    break
                eng.timed_decode(pos, self.PIPE_RECHECK_STEPS)
                pos += self.PIPE_RECHECK_STEPS
        self._decode(eng, pos, end - pos)

    def _decode(self, eng, t0, n_steps):
        """eng.decode, with the two launch forms compared IN SITU the first time an engine runs pipelined launches: the same
        process has measured them at 1.6 ms per step (the pair of streams made early) and at 3.0 ms + 0.18 s per call (made
        late, a waiting packet in a neighbouring hardware queue: HISTORY.md section 4.2), against 1.87 ms for the plain chain.
        Both forms produce the same tokens bit for bit, so the window's first steps are the measurement: 384 pipelined steps,
        128 plain ones (after 16 untimed ones: the plain graph may not exist yet, and its capture is not the chain's speed;
        112 / 64 after 8 in a call of 256..1023 steps), and the engine keeps pipelined launches only if they were >= 3 %
        faster -- once more on a fresh pair of streams, where the call is long enough, before giving them up.  The verdict holds for as long as the pair lives (`release_pipeline`; `pipeline_report` keeps
        the numbers)."""
        if not eng.pipelined or getattr(eng, "_pipe_verdict", None) is not None or n_steps < 256:
            eng.decode(t0, n_steps)
            return
        # steps: untimed warm-up, timed pipelined, untimed plain warm-up, timed plain -- the short form fits one published
        # chunk of a tapped window (256 steps: the level pipeline's upper levels)
        warm, n_pipe, n_plain = (16, 384, 128) if n_steps >= 1024 else (8, 112, 64)
        report = dict(pipelined_ms=[], plain_ms=None, kept=False)
        pos, end = t0, t0 + n_steps
        for attempt in range(2):
            if end - pos < warm + n_pipe + (0 if report["plain_ms"] is not None else warm + n_plain):
                break
            eng.decode(pos, warm)              # the pair of streams and its graphs are made here, outside the timed steps
            pos += warm
            if eng.pipe_error():
                return                         # a wait timed out: the caller decodes the window again on the plain chain
            if not eng.pipelined:
                break
            report["pipelined_ms"].append(round(eng.timed_decode(pos, n_pipe) * 1e3, 4))
            pos += n_pipe
            if eng.pipe_error():
                return
            if not eng.pipelined:
                break
            if report["plain_ms"] is None:
                # the plain chain's graph, with the pair kept (switching off would release the very pair that was measured);
                # untimed steps first: an engine that was pipelined from its creation captures that graph here
                eng.decode(pos, warm, plain=True)
                pos += warm
                report["plain_ms"] = round(eng.timed_decode(pos, n_plain, plain=True) * 1e3, 4)
                pos += n_plain
            if report["pipelined_ms"][-1] < 0.97 * report["plain_ms"]:
                report["kept"] = True
                break
            if attempt == 0 and not eng.set_pipelined(True, fresh=True):
                break
        eng._pipe_verdict = report["kept"]
        eng.set_pipelined(report["kept"])
        self.pipeline_report = report
        self.pipeline_reports = (getattr(self, "pipeline_reports", None) or [])[-15:] + [report]
        if end > pos:
            eng.decode(pos, end - pos)

    def _pipe_give_up(self, eng):
        """A pipelined wait timed out on this engine: forget the error, release the pair, plain chain for the engine's lifetime."""
        eng.clear_pipe_error()
        eng.set_pipelined(False)
        eng._pipe_verdict = False
        eng._pipe_timed_out = True
        self.pipeline_report = dict(getattr(self, "pipeline_report", None) or {}, kept=False, timed_out=True)

    def packed(self, fp16):
        """This prior's weights in MFMA order for one engine dtype (built on first use, dropped when the module moves)."""
        if self.x_emb.weight.device.type != "cuda":
            raise RuntimeError("move the prior to the GPU before sampling (prior.cuda()); there is no CPU path")
        fp16 = bool(fp16)
        if fp16 not in self._packed:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._packed[fp16] = PackedPrior(sd, "", seq_len=self.input_dims, bins=self.bins,
                                             encoder_dims=self.encoder_dims if 6 in self._funcs() else 0,
                                             only_encode=self.only_encode, width=self.width, depth=self.depth,
                                             heads=self.heads, attn_order=self.attn_order, blocks=self.blocks,
                                             m_attn=self.m_attn, m_mlp=self.m_mlp, prime_len=self.prime_len,
                                             y_cond=self.y_cond, add_cond_after=self.add_cond_after_transformer,
                                             fp16=fp16, device=self.x_emb.weight.device)
        return self._packed[fp16]

    def bound_engine(self):
        """The engine used last (bench.py probes it); None before the first sample / forward call."""
        return self._last_engine

    def _funcs(self):
        from ..engine import attn_funcs
        return attn_funcs(self.attn_order, self.depth)

    def forward(self, x, x_cond=None, y_cond=None, encoder_kv=None, fp16=False, loss_full=False, encode=False,
                get_preds=False, get_acts=False, get_sep_loss=False):
        """autoregressive.py:114-175, inference only: the teacher-forced pass over a full sequence as ONE engine prefill
        (position t sees tokens < t).  `only_encode` models return the final activations (N, T, width) fp32 (the lyric
        encoder of separated enc-dec priors, prior.py:285-292); the others return (loss, preds | None) with the loss
        in bits per token -- with get_sep_loss the (prime, generated) parts separately, prime first."""
        assert not get_acts, "get_acts is not supported (activations are only exposed by only_encode models)"
        with t.no_grad():
            x = self.preprocess(x)
            N, D = x.shape
            assert D == self.input_dims and (0 <= x).all() and (x < self.bins).all()
            if self.only_encode:
                assert not self.x_cond and not self.y_cond and x_cond is None and y_cond is None and encoder_kv is None
                eng = self.engine(N, fp16)
                eng.set_cond(None, None)
                eng.tokens[:, :D] = x
                eng.prefill(0, D)
                return eng.hidden[:, :D].clone()
            has_cross = 6 in self._funcs()
            assert (encoder_kv is not None) == has_cross, "encoder_kv is required exactly for cross-attention models"
            self._check_cond(N, x_cond, y_cond)
            eng = self.engine(N, fp16, want_preds=True)
            eng.set_cond(x_cond, y_cond)
            if has_cross:
                eng.set_encoder_kv(encoder_kv)
            eng.tokens[:, :D] = x
            eng.prefill(0, D)
            preds = eng.preds[:, :D].clone()                               # (N, D, bins) fp32 logits
            ce = lambda lg, tg: nn.functional.cross_entropy(lg.reshape(-1, self.bins), tg.reshape(-1)) / math.log(2.0)
            if get_sep_loss:
                assert self.prime_len is not None
                loss = (ce(preds[:, :self.prime_len], x[:, :self.prime_len]), ce(preds[:, self.prime_len:], x[:, self.prime_len:]))
            else:
                loss = ce(preds, x)
            return (loss, preds) if get_preds else (loss, None)

    def _check_cond(self, N, x_cond, y_cond):
        D = self.input_dims
        if self.y_cond:
            assert y_cond is not None and tuple(y_cond.shape) == (N, 1, self.width)
        else:
            assert y_cond is None
        if self.x_cond:
            assert x_cond is not None
            assert tuple(x_cond.shape) in ((N, D, self.width), (N, 1, self.width)), \
                f"Got {x_cond.shape}, expected ({N}, {D}/{1}, {self.width})"
        else:
            assert x_cond is None

    def _run(self, n_samples, x_prime, x_cond, y_cond, encoder_kv, fp16, temp, top_k, top_p, get_preds, sample_tokens,
             seed=0, sample_base=0, pos_base=0, stream_id=0):
        assert self.training is False
        assert top_k == 0 or top_p == 0.0
        has_cross = 6 in self._funcs()
        assert (encoder_kv is not None) == has_cross, "encoder_kv is required exactly for cross-attention models"
        if sample_tokens is None:
            sample_tokens = self.input_dims
        self._check_cond(n_samples, x_cond, y_cond)
        eng = self.engine(n_samples, fp16, want_preds=get_preds)
        eng.set_cond(x_cond, y_cond)
        eng.set_sampling(temp=temp, top_k=top_k, top_p=top_p, seed=seed, sample_base=sample_base, pos_base=pos_base,
                         stream_id=stream_id)
        if has_cross:
            assert tuple(encoder_kv.shape) == (n_samples, self.encoder_dims, self.width)
            eng.set_encoder_kv(encoder_kv)
        n_prime = 0
        if x_prime is not None:
            x_prime = self.preprocess(x_prime)
            assert (0 <= x_prime).all() and (x_prime < self.bins).all()
            assert x_prime.shape[0] == n_samples
            n_prime = x_prime.shape[1]
            assert n_prime < sample_tokens
            eng.tokens[:, :n_prime] = x_prime
            eng.prefill(0, n_prime)
        tap = getattr(self, "decode_tap", None)
        if tap is None:
            self._decode_window(eng, n_prime, sample_tokens - n_prime)
            if eng.pipe_error():
                # A pipelined launch gave up waiting for its producer (bounded polls: no hang), so what it computed is void.
                # Nobody has seen the window's tokens yet and the draw of a position is a pure function of (seed, position):
                # the window is decoded again on the plain chain -- same tokens as if nothing had happened -- and this engine
                # keeps the plain chain from here on.
                import sys
                print(f"jukebox_amd: pipelined decode: launch slot {eng.pipe_error() - 1} timed out waiting for its producer; "
                      "decoding the window again on the plain launch chain", file=sys.stderr, flush=True)
                self._pipe_give_up(eng)
                eng.decode(n_prime, sample_tokens - n_prime)
        else:
            # (every, fn): hand the token buffer to fn after every `every` enqueued decode steps, so that a consumer on
            # another stream can start on a partial window (jukebox_amd.sample._sample_levels_pipelined)
            every, fn = tap
            pos = n_prime
            while pos < sample_tokens:
                n = min(int(every), sample_tokens - pos)
                self._apply_pipeline(eng)
                self._decode(eng, pos, n)
                if eng.pipelined and eng.pipe_error():       # (a pipelined decode is host-synchronous: the read costs no wait)
                    # before anybody is handed the chunk: what the failed steps wrote is void, the same chunk is decoded
                    # again on the plain chain (same tokens as if nothing had happened)
                    self._pipe_give_up(eng)
                    eng.decode(pos, n)
                fn(eng.tokens, pos, pos + n)
                pos += n
        x = eng.tokens[:, :sample_tokens].clone()
        if eng.pipe_error():                   # (a tapped window: a consumer may already have read what the failed steps wrote)
            raise RuntimeError(f"pipelined decode: launch slot {eng.pipe_error() - 1} timed out waiting for its producer; the "
                               "tokens of this window are not valid (JB_PIPELINE_LAUNCHES=0 selects the plain launch chain)")
        x = self.postprocess(x, sample_tokens)
        if get_preds:
            return x, eng.preds[:, :sample_tokens].clone()
        return x

    def sample(self, n_samples, x_cond=None, y_cond=None, encoder_kv=None, fp16=False, temp=1.0, top_k=0, top_p=0.0,
               get_preds=False, sample_tokens=None, seed=0, sample_base=0, pos_base=0, stream_id=0):
        """autoregressive.py:199-249.  Extensions: the draw for sample n at window position t is the counter-based uniform
        keyed by (seed, stream_id, sample_base + n, pos_base + t) -- see jb_sample_params."""
        with t.no_grad():
            return self._run(n_samples, None, x_cond, y_cond, encoder_kv, fp16, temp, top_k, top_p, get_preds,
                             sample_tokens, seed, sample_base, pos_base, stream_id)

    def primed_sample(self, n_samples, x, x_cond=None, y_cond=None, encoder_kv=None, fp16=False, temp=1.0, top_k=0,
                      top_p=0.0, get_preds=False, chunk_size=None, sample_tokens=None, seed=0, sample_base=0, pos_base=0,
                      stream_id=0):
        """autoregressive.py:251-359.  `chunk_size` is accepted for API compatibility; the engine prefills in its
        own (larger) chunks -- results are chunk-invariant (the reference's check_chunks, factored_attention.py:457-488)."""
        with t.no_grad():
            return self._run(n_samples, x, x_cond, y_cond, encoder_kv, fp16, temp, top_k, top_p, get_preds,
                             sample_tokens, seed, sample_base, pos_base, stream_id)
