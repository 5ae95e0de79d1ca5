"""CPU suite: pin the numpy oracle (oracle/) to the golden vectors produced by the reference
itself (tests/golden/gen_golden.py).  Tolerances: float32 summation-order differences only."""
import numpy as np
import pytest

import os

from conftest import ROOT, load_golden, sub_state
from oracle import ops
from oracle.autoregressive import ConditionalAutoregressive2D, split_chunks
from oracle.prior import SimplePrior
from oracle.sample import get_starts
from oracle.transformer import Transformer, allowed_keys, decode_key_index
from oracle.vqvae import VQVAE


class _OneLayer(Transformer):
    """A FactoredAttention alone: reuse Transformer._attention with dummy ln/mlp params."""

    def __init__(self, sd, func, n_in, n_ctx, n_head, blocks, prime_len, encoder_dims):
        z = lambda *s: np.zeros(s, np.float32)
        full = {"_attn_mods.0.attn." + k: v for k, v in sd.items()}
        for nm, shp in (("ln_0.weight", (n_in,)), ("ln_0.bias", (n_in,)), ("ln_1.weight", (n_in,)),
                        ("ln_1.bias", (n_in,)), ("mlp.c_fc.w", (n_in, n_in)), ("mlp.c_fc.b", (n_in,)),
                        ("mlp.c_proj.w", (n_in, n_in)), ("mlp.c_proj.b", (n_in,))):
            full["_attn_mods.0." + nm] = z(*shp)
        super().__init__(full, "", n_in, n_ctx, n_head, 1, blocks=blocks, funcs=[func],
                         m_attn=sd["c_proj.w"].shape[0] / n_in, encoder_dims=encoder_dims, prime_len=prime_len)


@pytest.mark.parametrize("func", [0, 1, 2, 3, 6, 7])
def test_attention_patterns(func):
    g = load_golden("attention")
    sd = sub_state(g, f"f{func}.")
    x = sd.pop("x"); y_full = sd.pop("y_full"); y_chunks = sd.pop("y_chunks")
    ekv = sd.pop("encoder_kv", None)
    L = x.shape[1]
    att = _OneLayer(sd, func, 32, L, 2, 8, 24 if func == 7 else None, 16 if func == 6 else None)
    # one full-length chunk == the training-style forward
    y = att._attention(0, x, 0, False, ekv)
    assert np.abs(y - y_full).max() < 2e-6
    # the reference's ragged chunk schedule, incl. q_l == 1 steps
    att.del_cache()
    ys, pos = [], 0
    for c in list(g["chunks"]) + [11] * 100:
        if pos >= L:
            break
        c = int(min(c, L - pos))
        ys.append(att._attention(0, x[:, pos:pos + c], pos, False, ekv))
        pos += c
    assert np.abs(np.concatenate(ys, 1) - y_chunks).max() < 2e-6


def test_key_sets_consistent():
    """decode_key_index (q_l == 1) and allowed_keys (masked prefill) describe the same sets."""
    for func in (0, 1, 2, 3, 7):
        for p in range(0, 70):
            idx = decode_key_index(func, p, 8, 32)
            m = allowed_keys(func, [p], np.arange(p + 1), 8, 32)[0]
            want = np.nonzero(m)[0]
            if func == 7:
                want = want[want < 32]
            got = np.array([], int) if idx is None else idx
            assert np.array_equal(got, want), (func, p)


def _ar(g, pfx, **kw):
    return ConditionalAutoregressive2D(sub_state(g, pfx), "", **kw)


def test_autoregressive_sample_fp32():
    g = load_golden("autoregressive")
    p = _ar(g, "a.", input_shape=(64,), bins=128, width=64, depth=6, heads=2, attn_order=2, blocks=8,
            x_cond=True, y_cond=True)
    z, preds = p.sample(3, g["a.x_cond"], g["a.y_cond"], top_k=1, get_preds=True)
    assert np.abs(preds - g["a.preds"]).max() < 2e-5
    assert np.array_equal(z, g["a.z"])
    z20 = p.sample(3, g["a.x_cond"], g["a.y_cond"], top_k=1, sample_tokens=20)
    assert np.array_equal(z20, g["a.z20"])
    lf = p.forward_logits(g["a.z"], g["a.x_cond"], g["a.y_cond"])
    assert np.abs(lf - g["a.preds_forward"]).max() < 2e-5


def test_autoregressive_sample_fp16_emulation():
    """reference run with fp16=True on CPU vs the oracle's rounding emulation (logit tolerance,
    SURVEY.md section 7 'hard parts': fp16 moves logits by ~4e-3)."""
    g = load_golden("autoregressive")
    p = _ar(g, "a.", input_shape=(64,), bins=128, width=64, depth=6, heads=2, attn_order=2, blocks=8,
            x_cond=True, y_cond=True)
    z, preds = p.sample(3, g["a.x_cond"], g["a.y_cond"], top_k=1, get_preds=True, fp16=True)
    agree = (z == g["a.z16"]).mean()
    # compare logits while the streams agree
    first_div = np.argmax((z != g["a.z16"]).any(0)) if (z != g["a.z16"]).any() else z.shape[1]
    n = max(int(first_div), 1)
    assert np.abs(preds[:, :n] - g["a.preds16"][:, :n]).max() < 2e-2
    assert agree > 0.9


def test_autoregressive_primed_order12():
    g = load_golden("autoregressive")
    p = _ar(g, "b.", input_shape=(120,), bins=80, width=32, depth=48, heads=2, attn_order=12, blocks=8,
            x_cond=False, y_cond=False, prime_len=24)
    z, preds = p.primed_sample(2, g["b.x_prime"], top_k=1, get_preds=True, chunk_size=7)
    assert np.abs(preds - g["b.preds"]).max() < 5e-5
    assert np.array_equal(z, g["b.z"])
    z60, preds60 = p.primed_sample(2, g["b.x_prime"], top_k=1, get_preds=True, sample_tokens=60)
    assert np.abs(preds60 - g["b.preds60"]).max() < 5e-5
    assert np.array_equal(z60, g["b.z60"])


def test_autoregressive_dense_ycond():
    g = load_golden("autoregressive")
    p = _ar(g, "c.", input_shape=(40,), bins=96, width=48, depth=3, heads=3, attn_order=0, blocks=None,
            x_cond=False, y_cond=True)
    z, preds = p.sample(2, None, g["c.y_cond"], top_k=1, get_preds=True)
    assert np.abs(preds - g["c.preds"]).max() < 2e-5
    assert np.array_equal(z, g["c.z"])


def test_vqvae_encode_decode(tiny_hps):
    g = load_golden("vqvae")
    vq = VQVAE(sub_state(g, "sd."), tiny_hps["tiny_vqvae"])
    zs = vq.encode(g["x"])
    for l in range(3):
        assert (zs[l] == g[f"z{l}"]).mean() > 0.995          # argmin near-ties may flip a code
        xd = vq.decode([g[f"z{l}"]] + [None] * (2 - l), start_level=l)
        assert np.abs(xd - g[f"xd{l}"]).max() < 1e-5


def _priors(tiny_hps):
    g = load_golden("priors")
    vq = tiny_hps["tiny_vqvae"]
    top_len = vq["sample_length"]
    hops = np.cumprod([s ** d for s, d in zip(vq["strides_t"], vq["downs_t"])])
    vq_shapes = [top_len // h for h in hops]
    out = []
    for i, nm in enumerate(["tiny_up0", "tiny_up1", "tiny_top"]):
        hps = tiny_hps[nm]
        z_shapes = [(zs * hps["n_ctx"] // vq_shapes[hps["level"]],) for zs in vq_shapes]
        out.append(SimplePrior(sub_state(g, f"p{i}."), hps, z_shapes, vq["l_bins"], vq["downs_t"], vq["strides_t"]))
    return g, out


def test_prior_conditioning_and_sampling(tiny_hps):
    g, (up0, up1, top) = _priors(tiny_hps)
    x_cond, y_cond, prime = top.get_cond(None, g["top.y0"])
    assert np.abs(x_cond - g["top.x_cond"]).max() < 1e-6 and np.abs(y_cond - g["top.y_cond"]).max() < 1e-6
    z = top.sample(3, z=np.zeros((3, 0), np.int64), y=g["top.y0"], top_k=1, chunk_size=5)
    assert np.array_equal(z, g["top.z_ancestral"])
    zp = top.sample(3, z=g["top.z_ancestral"][:, 24:], y=g["top.y24"], top_k=1, chunk_size=5)
    assert np.array_equal(zp, g["top.z_primed"])
    zpt = top.sample(3, z=g["top.z_ancestral"][:, :10], y=g["top.y0"], top_k=1, chunk_size=5, sample_tokens=30)
    assert np.array_equal(zpt, g["top.z_partial30"])
    for nm, p in (("up1", up1), ("up0", up0)):
        x_cond, y_cond, _ = p.get_cond([g[f"{nm}.z_cond"]], g[f"{nm}.y"])
        assert np.abs(x_cond - g[f"{nm}.x_cond"]).max() < 2e-5
        assert np.abs(y_cond - g[f"{nm}.y_cond"]).max() < 1e-6
        z, preds = p.prior.sample(3, x_cond, y_cond, None, top_k=1, get_preds=True)
        assert np.abs(preds - g[f"{nm}.preds"]).max() < 1e-4
        assert np.array_equal(z, g[f"{nm}.z"])
        zp = p.sample(3, z=g[f"{nm}.z"][:, :64], z_conds=[g[f"{nm}.z_cond"]], y=g[f"{nm}.y"], top_k=1, chunk_size=32)
        assert np.array_equal(zp, g[f"{nm}.z_primed"])


def test_top_prior_raw_logits(tiny_hps):
    g, (_, _, top) = _priors(tiny_hps)
    x_cond, y_cond, prime = top.get_cond(None, g["top.y0"])
    xc = np.concatenate([np.zeros((3, top.n_tokens, 32), np.float32), x_cond], 1)
    z, preds = top.prior.primed_sample(3, prime, xc, y_cond, top_k=1, chunk_size=5, get_preds=True)
    assert np.abs(preds - g["top.raw_preds"]).max() < 5e-5
    assert np.array_equal(z, g["top.raw_z"])


def test_misc_tables():
    g = load_golden("misc")
    i = 0
    while f"starts{i}" in g.files:
        assert list(g[f"starts{i}"]) == get_starts(*[int(v) for v in g[f"starts{i}.args"]])
        i += 1
    i = 0
    while f"chunks{i}" in g.files:
        assert list(g[f"chunks{i}"]) == split_chunks(*[int(v) for v in g[f"chunks{i}.args"]])
        i += 1
    for k in (1, 5, 97, 200):
        assert np.array_equal(ops.filter_logits(g["filter.logits"], top_k=k), g[f"filter.top_k{k}"])


def test_alignment_matches_reference(tiny_hps):
    """align.py:15-83 on the tiny top prior: attention of the prime layer (music queries x lyric keys) stitched over hops."""
    from oracle.align import get_alignment
    from jukebox_amd.data.labels import get_relevant_lyric_tokens
    g, (_, _, top) = _priors(tiny_hps)
    e = load_golden("e2e")
    labels_y = g["p2.labels_y"]
    full_tokens = [list(map(int, g[f"p2.full_tokens{j}"])) for j in range(3)]
    raw_to_tokens = 64

    def get_y(start):
        y = labels_y.copy()
        y[:, 2] = top.n_ctx * raw_to_tokens
        y[:, 1] += start * raw_to_tokens
        idx = []
        for i in range(3):
            toks, ind = get_relevant_lyric_tokens(full_tokens[i], top.n_tokens, int(y[i, 0]), int(y[i, 1]), int(y[i, 2]), f32=True)
            y[i, -top.n_tokens:] = toks
            idx.append(ind)
        return y, idx

    zs = [e["z0"], e["z1"], e["z2"]]
    al = get_alignment(zs, labels_y, full_tokens, top, get_y, False, [0.5, 0.5, 0.125], 15, 1)
    for j in range(3):
        assert al[j].shape == e[f"alignment{j}"].shape
        assert np.abs(al[j] - e[f"alignment{j}"]).max() < 1e-6


def _sep_prior(tiny_hps):
    g = load_golden("prior_sep")
    vq = tiny_hps["tiny_vqvae"]
    hops = np.cumprod([s ** d for s, d in zip(vq["strides_t"], vq["downs_t"])])
    vq_shapes = [vq["sample_length"] // h for h in hops]
    hps = tiny_hps["tiny_sep"]
    z_shapes = [(zs * hps["n_ctx"] // vq_shapes[hps["level"]],) for zs in vq_shapes]
    return g, SimplePrior(sub_state(g, "sd."), hps, z_shapes, vq["l_bins"], vq["downs_t"], vq["strides_t"])


def test_separated_encoder_decoder_prior(tiny_hps):
    """prior_5b_lyrics structure: lyric encoder (only_encode) -> prime_state_proj + LN -> cross-attention layers (func 6),
    merged_decoder (no cond add after the transformer, untied x_out)."""
    g, sep = _sep_prior(tiny_hps)
    x_cond, y_cond, prime = sep.get_cond(None, g["y0"])
    ekv = sep.get_encoder_kv(prime)
    assert np.abs(ekv - g["encoder_kv"]).max() < 2e-5
    z, preds = sep.prior.sample(3, x_cond, y_cond, ekv, top_k=1, get_preds=True)
    assert np.abs(preds - g["preds"]).max() < 5e-5
    assert np.array_equal(z, g["z_raw"])
    assert np.array_equal(sep.sample(3, z=np.zeros((3, 0), np.int64), y=g["y0"], top_k=1), g["z_ancestral"])
    assert np.array_equal(sep.sample(3, z=g["z_ancestral"][:, :20], y=g["y0"], top_k=1, chunk_size=6), g["z_primed"])


def test_torch_port_matches_numpy_oracle():
    """oracle/torch_port.py (what bench.py times as the CPU baseline, and the checker of the full-size decode steps) against the
    golden-pinned numpy oracle: the same hidden states step by step, for the block / transpose / prev, the prime and the
    cross-attention patterns (attn_order 10 as in prior_5b_lyrics: layer 18 of 20 reads the encoder states)."""
    from oracle.torch_port import TorchDecodeStack
    from oracle.transformer import Transformer
    rng = np.random.default_rng(0)
    for cfg in (dict(n_in=32, n_ctx=48, n_head=2, n_depth=6, attn_order=2, blocks=8),
                dict(n_in=32, n_ctx=64, n_head=2, n_depth=8, attn_order=12, blocks=8, prime_len=12),
                dict(n_in=24, n_ctx=20, n_head=3, n_depth=2, attn_order=0, blocks=None),
                dict(n_in=32, n_ctx=24, n_head=2, n_depth=20, attn_order=10, blocks=4, encoder_dims=7)):
        from oracle.transformer import attn_func_of_layer
        W, D = cfg["n_in"], cfg["n_depth"]
        S = W // 4
        sd = {}
        enc = rng.standard_normal((3, cfg["encoder_dims"], W)).astype(np.float32) if cfg.get("encoder_dims") else None
        for d in range(D):
            p = f"_attn_mods.{d}."
            cross = attn_func_of_layer(cfg["attn_order"], d) == 6
            if cross:
                sd[p + "attn.c_enc_kv.w"] = (0.2 * rng.standard_normal((W, 2 * S))).astype(np.float32)
                sd[p + "attn.c_enc_kv.b"] = (0.1 * rng.standard_normal(2 * S)).astype(np.float32)
            for nm, shp in (("attn.c_attn.w", (W, S if cross else 3 * S)), ("attn.c_proj.w", (S, W)), ("mlp.c_fc.w", (W, W)),
                            ("mlp.c_proj.w", (W, W))):
                sd[p + nm] = (0.2 * rng.standard_normal(shp)).astype(np.float32)
                sd[p + nm[:-1] + "b"] = (0.1 * rng.standard_normal(shp[1])).astype(np.float32)
            for ln in ("ln_0", "ln_1"):
                sd[p + ln + ".weight"] = (1 + 0.1 * rng.standard_normal(W)).astype(np.float32)
                sd[p + ln + ".bias"] = (0.1 * rng.standard_normal(W)).astype(np.float32)
        ref = Transformer(sd, "", **cfg)
        port = TorchDecodeStack(sd, "", n_batch=3, encoder_kv=enc, **{k: v for k, v in cfg.items() if k != "encoder_dims"})
        for t in range(cfg["n_ctx"]):
            x = rng.standard_normal((3, 1, W)).astype(np.float32)
            want = ref.forward(x, encoder_kv=enc)
            got = port.forward(x).numpy()
            assert np.abs(got - want).max() < 2e-4 * max(1.0, np.abs(want).max()), (cfg["attn_order"], t)


def test_teacher_forced_losses(tiny_hps):
    """SimplePrior.z_forward / ConditionalAutoregressive2D.forward (prior.py:312-347, autoregressive.py:114-175): the
    evaluation pass over a given code sequence -- loss, bits per token of the lyric and music parts, logits -- against
    the reference on the merged (1b_lyrics-like) top prior, both upsamplers and the separated enc-dec prior."""
    f = load_golden("forward")
    g, (up0, up1, top) = _priors(tiny_hps)
    gs, sep = _sep_prior(tiny_hps)
    cases = (("top", top, g["top.z_ancestral"], [], g["top.y0"]),
             ("up0", up0, g["up0.z"], [g["up0.z_cond"]], g["up0.y"]),
             ("up1", up1, g["up1.z"], [g["up1.z_cond"]], g["up1.y"]),
             ("sep", sep, gs["z_ancestral"][:1], [], gs["y0"][:1]))
    for tag, prior, z, z_conds, y in cases:
        loss, m = prior.z_forward(z, z_conds, y, get_preds=True)
        assert np.abs(m["preds"] - f[f"{tag}.preds"]).max() < 1e-4, tag
        for k in ("bpd", "prime_loss", "gen_loss"):
            assert abs(float(m[k]) - float(f[f"{tag}.{k}"])) < 2e-5, (tag, k, m[k], f[f"{tag}.{k}"])
        assert abs(float(loss) - float(f[f"{tag}.loss"])) < 2e-5, (tag, loss, f[f"{tag}.loss"])


def test_cpu_baseline_window_plan_matches_the_sampling_driver():
    """oracle/time_reference.py integrates the reference's per-step CPU time over the workload's decode steps and primed
    tokens: its window plan must be the one the sampling driver walks (get_starts + sample_single_window bookkeeping)."""
    import importlib.util
    from jukebox_amd.utils.sample_utils import get_starts
    spec = importlib.util.spec_from_file_location("time_reference", os.path.join(ROOT, "oracle", "time_reference.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    for total, n_ctx, hop in ((110240, 8192, 4096), (27560, 8192, 4096), (6890, 6144, 768), (8192, 8192, 4096), (5000, 8192, 4096),
                              (33072, 8192, 4096), (100, 48, 6)):
        decode, primed, have = 0, 0, 0
        if total < n_ctx:
            decode = total
        else:
            for s in get_starts(total, n_ctx, hop):
                new = s + n_ctx - have
                decode, primed, have = decode + new, primed + (n_ctx - new if have else 0), s + n_ctx
            assert have == total
        assert tr.window_plan(total, n_ctx, hop) == (decode, primed)
        assert decode == total                    # every token of the level is decoded exactly once


def test_ascii_stand_ins_agree():
    """`unidecode` (the reference's data/text_processor.py:2,12; not installed in this image) has two stand-ins written
    independently of each other: the product's `to_ascii` (NFKD decomposition + a dict of exceptions, jukebox_amd/data/labels.py)
    and the golden generators' `unidecode_stand_in` (tests/golden/refshim.py: a per-code-point restatement of unidecode's tables
    x000 / x001 / x020, no unicodedata).  The lyric-token goldens were produced through the second; the product must agree with it
    on every character of those tables -- and on a lyric sheet's worth of text."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("refshim_tables", os.path.join(ROOT, "tests", "golden", "refshim.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)                       # (defines the tables; nothing is installed)
    from jukebox_amd.data.labels import to_ascii
    chars = [chr(c) for c in list(range(0x20, 0x180)) + sorted(shim._X020)]
    bad = [(hex(ord(ch)), shim.unidecode_stand_in(ch), to_ascii(ch)) for ch in chars if shim.unidecode_stand_in(ch) != to_ascii(ch)]
    assert not bad, bad
    text = "Café — déjà vu… “quoted” ’tis Ærø Œuvre straße naïve Łódź Ðþ ½"
    assert shim.unidecode_stand_in(text) == to_ascii(text) == 'Cafe -- deja vu... "quoted" \'tis AEro OEuvre strasse naive Lodz Dth 1/2'
