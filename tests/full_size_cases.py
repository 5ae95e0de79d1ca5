"""Full-size parity cases of tests/test_hip_baseline_configs.py: one table of model geometries + the seeded inputs of a case,
shared by the GPU tests and by tests/golden/gen_full_size.py, which runs the numpy oracle's prefill of sample 0 HERE (no GPU:
minutes of CPU per case) and commits a digest -- so that the GPU lease is not spent on a CPU oracle.

Inputs come from numpy's PCG64 streams (float32 `standard_normal`), a pure function of (case seed, what, sample index): the
generator and the test build bit-identical arrays on any machine of this image.  The generator needs sample 0 only."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    # BASELINE config 2: small_prior (hparams.py:210-220), block row 63
    "small_prior": dict(W=1024, depth=48, heads=1, attn_order=2, blocks=64, seq=8192, bins=1024, prime_len=None, y_cond=False,
                        t0=8064, n_steps=64, N=16, seed=3),
    # BASELINE config 3: prior_1b_lyrics (hparams.py:165-188) right after the lyric prefill
    "1b_lyrics_top": dict(W=2048, depth=72, heads=2, attn_order=12, blocks=64, seq=6528, bins=2127, prime_len=384, y_cond=True,
                          t0=384, n_steps=64, N=16, seed=3),
    # ... and in the second window of the 20-second job: 384 lyric + 5398 primed positions (sample.py:17-88, hop 768)
    "1b_lyrics_top_window2": dict(W=2048, depth=72, heads=2, attn_order=12, blocks=64, seq=6528, bins=2127, prime_len=384,
                                  y_cond=True, t0=384 + 5398, n_steps=64, N=16, seed=11),
    # BASELINE config 4's dominant model: upsampler_level_0 / _1 (hparams.py:68-101), last block rows
    "upsampler": dict(W=1920, depth=72, heads=1, attn_order=2, blocks=128, seq=8192, bins=2048, prime_len=None, y_cond=True,
                      t0=8064, n_steps=64, N=4, seed=3),
    # BASELINE config 5: prior_5b_lyrics geometry (hparams.py:127-156), self-attention patterns at depth 12
    "5b": dict(W=4800, depth=12, heads=8, attn_order=2, blocks=128, seq=8192, bins=2048, prime_len=None, y_cond=True,
               t0=1100, n_steps=64, N=3, seed=5),
    # ... and as it is built: attn_order 10 (cross-attention layers 18 and 28 inside depth 29), merged_decoder, 512 encoder states
    "5b_order10": dict(W=4800, depth=29, heads=8, attn_order=10, blocks=128, seq=8192, bins=2048, prime_len=None, y_cond=True,
                       t0=700, n_steps=48, N=3, seed=7, enc_len=512, merged_decoder=True),
    # ... and at its REAL depth: all 79 layers of prior_5b_lyrics (hparams.py:127-156), cross-attention layers 18, 28, ..., 78
    # (22 GB of fp32 weights: generated on a thread pool, see state_dict)
    "5b_order10_full": dict(W=4800, depth=79, heads=8, attn_order=10, blocks=128, seq=8192, bins=2048, prime_len=None, y_cond=True,
                            t0=700, n_steps=64, N=3, seed=9, enc_len=512, merged_decoder=True),
}

# Whole greedy windows from the UNMODIFIED reference (tests/golden/gen_whole_window.py: CPU hours here, zero GPU minutes): the
# two geometries that carry the 20-second job, N = 2, every token of the window
WHOLE = {
    # BASELINE config 4's dominant model, a whole first window: ConditionalAutoregressive2D.sample (autoregressive.py:199-249)
    "upsampler_whole": dict(W=1920, depth=72, heads=1, attn_order=2, blocks=128, seq=8192, bins=2048, prime_len=None, y_cond=True,
                            t0=0, N=2, seed=23),
    # BASELINE config 3: primed_sample (autoregressive.py:251-359) over 384 lyric tokens in chunks of 32, then 6144 music tokens
    "1b_lyrics_top_whole": dict(W=2048, depth=72, heads=2, attn_order=12, blocks=64, seq=6528, bins=2127, prime_len=384, y_cond=True,
                                t0=384, N=2, seed=21),
}

# The arithmetic the bench times, pinned to the reference's OWN fp16 path (sample.py:239-241 samples with fp16=True at every level;
# transformer.py:169-192 x.half(), ops.py:14-24,97-101, factored_attention.py:82-108): tests/golden/gen_fp16_window.py runs the
# UNMODIFIED reference's primed_sample(fp16=True) at the upsampler geometry -- 4096 primed tokens in chunks of 32, then 4096 greedy
# steps with get_preds (a whole level-0 window of the 20-second job) -- and the fp32 reference teacher-forced on that stream (N = 2; CPU here, zero GPU minutes)
FP16 = {
    "upsampler_fp16": dict(W=1920, depth=72, heads=1, attn_order=2, blocks=128, seq=8192, bins=2048, prime_len=None, y_cond=True,
                           t0=4096, n_steps=4096, N=2, seed=29),
}

_ORDERS = {
    2: lambda d: [1, 2, 3][d % 3],
    10: lambda d: [*[1, 2, 3, 1, 2, 3, 1, 2, 3], *[1, 2, 3, 1, 2, 3, 1, 2, 3, 6] * 7][d % 79],
    12: lambda d: [7, 7, 0][d % 3] if d % 16 == 15 else [1, 2, 3][d % 3],
}


def attn_funcs(case):
    return [_ORDERS[case["attn_order"]](d) for d in range(case["depth"])]


def _normal(seed, what, idx, shape, scale):
    rng = np.random.default_rng([int(seed), int(what), int(idx)])
    return rng.standard_normal(shape, dtype=np.float32) * np.float32(scale)


def state_dict(case, threads=None):
    """Reference-named state dict of the case's prior (float32 numpy).  Every tensor has a stream of its own (case seed, layer,
    index), so the layers are generated side by side on a thread pool (numpy releases the GIL while it draws): the 5.5 billion
    weights of the full-depth 5b case take seconds on the GPU box's host cores, and the result does not depend on the pool."""
    W, depth, bins, seq, seed = case["W"], case["depth"], case["bins"], case["seq"], case["seed"]
    S = W // 4
    funcs = attn_funcs(case)
    sd = {"x_emb.weight": _normal(seed, 0, 0, (bins, W), 0.05), "pos_emb.pos_emb": _normal(seed, 0, 1, (seq, W), 0.01)}
    sd["x_out.weight"] = _normal(seed, 0, 2, (bins, W), 0.05) if case.get("merged_decoder") else sd["x_emb.weight"]
    if not case["y_cond"]:
        sd["start_token"] = _normal(seed, 0, 3, (1, W), 0.01)

    def layer(d):
        out = {}
        p = f"transformer._attn_mods.{d}."
        cross = funcs[d] == 6
        j_attn = S if cross else 3 * S
        for i, (nm, shape, sc) in enumerate((("attn.c_attn.w", (W, j_attn), 0.02), ("attn.c_proj.w", (S, W), 0.02),
                                             ("mlp.c_fc.w", (W, W), 0.02), ("mlp.c_proj.w", (W, W), 0.02),
                                             ("attn.c_attn.b", (j_attn,), 0.01), ("attn.c_proj.b", (W,), 0.01),
                                             ("mlp.c_fc.b", (W,), 0.01), ("mlp.c_proj.b", (W,), 0.01))):
            out[p + nm] = _normal(seed, 1 + d, i, shape, sc)
        if cross:
            out[p + "attn.c_enc_kv.w"] = _normal(seed, 1 + d, 8, (W, 2 * S), 0.02)
            out[p + "attn.c_enc_kv.b"] = _normal(seed, 1 + d, 9, (2 * S,), 0.01)
        for i, ln in enumerate(("ln_0", "ln_1")):
            out[p + ln + ".weight"] = 1 + _normal(seed, 1 + d, 10 + i, (W,), 0.05)
            out[p + ln + ".bias"] = _normal(seed, 1 + d, 12 + i, (W,), 0.02)
        return out

    if threads is None:
        threads = min(32, os.cpu_count() or 1) if depth * W * W > 2 ** 30 else 1
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as pool:
            layers = list(pool.map(layer, range(depth)))
    else:
        layers = [layer(d) for d in range(depth)]
    for out in layers:                       # (layer order: the dict reads as the sequential form built it)
        sd.update(out)
    return sd


def sample_inputs(case, n):
    """Inputs of sample n: (prefix tokens (t0,), x_cond (seq, W) or None, y_cond (1, W) or None, encoder states (enc_len, W) or None)."""
    W, seq, seed, t0 = case["W"], case["seq"], case["seed"], case["t0"]
    rng = np.random.default_rng([int(seed), 1000, int(n)])
    tokens = rng.integers(0, case["bins"] if case["prime_len"] is None else 79, t0)
    x_cond = _normal(seed, 1001, n, (seq, W), 0.05) if case["y_cond"] else None
    y_cond = _normal(seed, 1002, n, (1, W), 0.05) if case["y_cond"] else None
    enc = _normal(seed, 1003, n, (case["enc_len"], W), 1.0) if case.get("enc_len") else None
    return tokens, x_cond, y_cond, enc


def digest_positions(case):
    """Positions whose k / v rows the digest keeps in full: pattern and chunk boundaries, the ends, and a spread."""
    t0, bc = case["t0"], case["seq"] // case["blocks"]
    cand = {0, 1, bc - 1, bc, 2 * bc - 1, 383, 384, 447, 448, 511, 512, 1023, 1024, t0 // 2, t0 - bc, t0 - 2, t0 - 1}
    cand |= set(np.linspace(0, t0 - 1, 7).astype(int).tolist())
    pos = sorted(p for p in cand if 0 <= p < t0)
    if len(pos) > 12:                                   # keep the ends and an even spread of the rest
        keep = sorted(set(np.linspace(0, len(pos) - 1, 12).round().astype(int).tolist()))
        pos = [pos[i] for i in keep]
    return np.asarray(pos, dtype=np.int64)


def fp16_full_positions(case, n_steps=None):
    """Positions of an FP16 case whose logit rows the golden keeps in full: the last 64 primed ones, the first and the last 128
    sampled ones (fewer when the generator was asked for a short stream)."""
    t0 = case["t0"]
    n_steps = case["n_steps"] if n_steps is None else n_steps
    k = min(128, n_steps)
    return np.unique(np.concatenate([np.arange(max(t0 - 64, 0), t0 + k), np.arange(t0 + n_steps - k, t0 + n_steps)])).astype(np.int64)


def golden_path(tag):
    return os.path.join(GOLDEN_DIR, f"full_size_{tag}.npz")


def embed(sd, tokens, t0, n_t, x_cond, start):
    """get_emb (autoregressive.py:177-197) for positions t0..t0+n_t-1: tokens (N, >= t0+n_t-1), x_cond (N, seq, W) or None,
    start (N or 1, W) -> (N, n_t, W) float32."""
    pos = np.arange(t0, t0 + n_t)
    prev = np.where(pos > 0, pos - 1, 0)
    x = sd["x_emb.weight"][tokens[:, prev]]
    if t0 == 0:
        x[:, 0] = start
    x = x + sd["pos_emb.pos_emb"][pos][None]
    if x_cond is not None:
        x = x + x_cond[:, pos]
    return x.astype(np.float32)


def oracle_prefill(case, sd=None, progress=None):
    """The numpy oracle's prefill of sample 0 over the case's prefix: per layer (k rows, v rows) float32 (oracle.transformer,
    pinned to the reference's goldens by tests/test_oracle_golden.py)."""
    from oracle.transformer import Transformer as OracleTransformer
    sd = sd if sd is not None else state_dict(case)
    W, t0 = case["W"], case["t0"]
    tokens, x_cond, y_cond, enc = sample_inputs(case, 0)
    tr_sd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    tr = OracleTransformer(tr_sd, "", W, case["seq"], case["heads"], case["depth"], attn_order=case["attn_order"],
                           blocks=case["blocks"], prime_len=case["prime_len"], encoder_dims=case.get("enc_len") or None)
    start = y_cond.reshape(1, W) if case["y_cond"] else sd["start_token"].reshape(1, W)
    for c0 in range(0, t0, 1024):
        n = min(1024, t0 - c0)
        tr.forward(embed(sd, tokens[None], c0, n, None if x_cond is None else x_cond[None], start),
                   encoder_kv=None if enc is None else enc[None], t0=c0)
        if progress:
            progress(c0 + n)
    return [(tr.k[d][0], tr.v[d][0]) for d in range(case["depth"])]


def digest(case, kv):
    """What is committed of an oracle prefill: per layer the rows at digest_positions (those the layer holds) and, for EVERY
    row the layer holds, |k|^2 + |v|^2 -- a wrong row anywhere moves its number."""
    pos = digest_positions(case)
    depth, S = case["depth"], case["W"] // 4
    n_rows = np.asarray([k.shape[0] for k, _ in kv], dtype=np.int64)
    k_rows = np.zeros((depth, len(pos), S), np.float32)
    v_rows = np.zeros((depth, len(pos), S), np.float32)
    norm2 = np.zeros((depth, int(n_rows.max())), np.float32)
    for d, (k, v) in enumerate(kv):
        ok = pos < n_rows[d]
        k_rows[d, ok], v_rows[d, ok] = k[pos[ok]], v[pos[ok]]
        norm2[d, :n_rows[d]] = (k.astype(np.float64) ** 2).sum(1) + (v.astype(np.float64) ** 2).sum(1)
    return dict(pos=pos, n_rows=n_rows, k_rows=k_rows, v_rows=v_rows, norm2=norm2)
