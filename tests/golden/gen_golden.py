#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference/jukebox) on CPU under refshim.py.

Run once in the build container:  python tests/golden/gen_golden.py
The fixtures hold seeded inputs, the reference-named state dicts and the reference's
outputs; `tests/test_oracle_golden.py` pins the numpy oracle to them and the `-m gpu`
tests pin the HIP path to them.  /root/reference is never read at test time.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

import torch as t  # noqa: E402
from jukebox.hparams import HPARAMS_REGISTRY, Hyperparams, setup_hparams  # noqa: E402
from jukebox.make_models import make_prior, make_vqvae  # noqa: E402
from jukebox.prior.autoregressive import ConditionalAutoregressive2D, split_chunks  # noqa: E402
from jukebox.transformer.factored_attention import FactoredAttention  # noqa: E402
from jukebox.transformer.ops import filter_logits  # noqa: E402
from jukebox.utils.sample_utils import get_starts  # noqa: E402
import jukebox.sample as ref_sample  # noqa: E402

t.set_grad_enabled(False)


def npd(sd, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays)")


# ----------------------------------------------------------------------------------------
def gen_attention():
    """FactoredAttention per attn_func: full (training-style) forward and chunked sampling
    forward (factored_attention.py:424-488 are the reference's own self-checks of these)."""
    out = {}
    n_in, n_state, n_head, blocks, l = 32, 64, 2, 8, 128
    for func in [0, 1, 2, 3, 6, 7]:
        t.manual_seed(100 + func)
        enc = 16 if func == 6 else 0
        prime = 24 if func == 7 else 0
        attn = FactoredAttention(n_in, l + prime if func == 7 else l, n_state, n_head, mask=True, attn_func=func,
                                 blocks=blocks, encoder_dims=enc, prime_len=prime if func == 7 else None)
        attn.eval()
        L = attn.n_ctx
        x = t.randn(2, L, n_in)
        ekv = t.randn(2, enc, n_in) if func == 6 else None
        y_full = attn(x, encoder_kv=ekv)
        attn.del_cache()
        ys, pos = [], 0
        for c in [1, 1, 5, 7, 1, 16, 3] + [11] * 100:
            if pos >= L:
                break
            c = min(c, L - pos)
            ys.append(attn(x[:, pos:pos + c].contiguous(), encoder_kv=ekv, sample=True))
            pos += c
        attn.del_cache()
        y_chunks = t.cat(ys, dim=1)
        assert (y_full - y_chunks).abs().max() < 1e-5
        out.update(npd(attn.state_dict(), f"f{func}."))
        out[f"f{func}.x"] = x.numpy()
        out[f"f{func}.y_full"] = y_full.numpy()
        out[f"f{func}.y_chunks"] = y_chunks.numpy()
        if ekv is not None:
            out[f"f{func}.encoder_kv"] = ekv.numpy()
    out["chunks"] = np.array([1, 1, 5, 7, 1, 16, 3], dtype=np.int64)
    save("attention", **out)


# ----------------------------------------------------------------------------------------
def gen_autoregressive():
    """ConditionalAutoregressive2D.sample / primed_sample, greedy, with per-step logits."""
    out = {}
    # (a) attn_order 2, x_cond + y_cond, fp32 and fp16 activations
    t.manual_seed(7)
    p = ConditionalAutoregressive2D((64,), 128, width=64, depth=6, heads=2, attn_order=2, blocks=8,
                                    x_cond=True, y_cond=True, init_scale=1.0)
    p.eval()
    xc, yc = t.randn(3, 64, 64), t.randn(3, 1, 64)
    z, preds = p.sample(3, xc, yc, top_k=1, get_preds=True)
    z16, preds16 = p.sample(3, xc, yc, top_k=1, get_preds=True, fp16=True)
    zk, predsk = p.sample(3, xc, yc, top_k=1, get_preds=True, sample_tokens=20)
    out.update(npd(p.state_dict(), "a."))
    out.update({"a.x_cond": xc.numpy(), "a.y_cond": yc.numpy(), "a.z": z.numpy(), "a.preds": preds.numpy(),
                "a.z16": z16.numpy(), "a.preds16": preds16.numpy(), "a.z20": zk.numpy()})
    # teacher-forced logits on the sampled tokens (forward(), autoregressive.py:114-175)
    _, preds_forw = p.forward(z, xc, yc, get_preds=True)
    out["a.preds_forward"] = preds_forw.numpy()

    # (b) attn_order 12 (block/transpose/prev + prime layers at 15,31 + dense at 47), no x_cond,
    #     start token, primed with 34 tokens in chunks of 7
    t.manual_seed(8)
    p = ConditionalAutoregressive2D((120,), 80, width=32, depth=48, heads=2, attn_order=12, blocks=8,
                                    x_cond=False, y_cond=False, prime_len=24, init_scale=1.0)
    p.eval()
    xp = t.randint(0, 80, (2, 34))
    z, preds = p.primed_sample(2, xp.clone(), top_k=1, get_preds=True, chunk_size=7)
    z1, preds1 = p.primed_sample(2, xp.clone(), top_k=1, get_preds=True, chunk_size=None, sample_tokens=60)
    out.update(npd(p.state_dict(), "b."))
    out.update({"b.x_prime": xp.numpy(), "b.z": z.numpy(), "b.preds": preds.numpy(),
                "b.z60": z1.numpy(), "b.preds60": preds1.numpy()})

    # (c) attn_order 0 (dense) + y_cond only, temperature/top-k filtered logits recorded
    t.manual_seed(9)
    p = ConditionalAutoregressive2D((40,), 96, width=48, depth=3, heads=3, attn_order=0, blocks=None,
                                    x_cond=False, y_cond=True, init_scale=1.0)
    p.eval()
    yc = t.randn(2, 1, 48)
    z, preds = p.sample(2, None, yc, top_k=1, get_preds=True)
    out.update(npd(p.state_dict(), "c."))
    out.update({"c.y_cond": yc.numpy(), "c.z": z.numpy(), "c.preds": preds.numpy()})
    save("autoregressive", **out)


# ----------------------------------------------------------------------------------------
TINY_VQVAE = Hyperparams(levels=3, downs_t=(2, 2, 2), strides_t=(2, 2, 2), emb_width=16, l_bins=64,
                         hvqvae_multipliers=(2, 1, 1), width=8, depth=2, m_conv=1.0, dilation_growth_rate=3,
                         sample_length=4608, sr=22050)
TINY_LABELS = Hyperparams(y_bins=(10, 100), t_bins=64, max_bow_genre_size=1, min_duration=0.1, max_duration=600.0,
                          sr=22050)
TINY_TOP = Hyperparams(level=2, n_ctx=48, prior_width=32, prior_depth=16, heads=2, attn_order=12, blocks=8,
                       init_scale=1.0, labels=True, labels_v3=True, use_tokens=True, n_tokens=16, n_vocab=79,
                       single_enc_dec=True, prime_loss_fraction=0.4)
TINY_TOP.update(TINY_LABELS)
TINY_UP = Hyperparams(n_ctx=128, prior_width=32, prior_depth=6, heads=1, attn_order=2, blocks=8, init_scale=1.0,
                      cond_width=16, cond_depth=4, cond_dilation_growth_rate=3, cond_dilation_cycle=2, cond_c_res=1,
                      labels=True, y_bins=(10, 100), t_bins=128, max_bow_genre_size=5, min_duration=0.1,
                      max_duration=600.0, sr=22050, use_tokens=False)
TINY_UP0 = Hyperparams(level=0)
TINY_UP0.update(TINY_UP)
TINY_UP1 = Hyperparams(level=1, cond_res_scale=True)
TINY_UP1.update(TINY_UP)
# separated encoder-decoder top prior (the prior_5b_lyrics structure: lyric encoder + cross-attention layers, merged_decoder)
TINY_SEP = Hyperparams(level=2, n_ctx=48, prior_width=32, prior_depth=10, heads=2, attn_order=8, blocks=8, init_scale=1.0,
                       labels=True, labels_v3=True, use_tokens=True, n_tokens=16, n_vocab=79, single_enc_dec=False,
                       merged_decoder=True, prime_loss_fraction=0.4, prime_width=16, prime_depth=3, prime_heads=2,
                       prime_attn_order=2, prime_blocks=4, prime_init_scale=1.0)
TINY_SEP.update(TINY_LABELS)
HPARAMS_REGISTRY.update(tiny_vqvae=TINY_VQVAE, tiny_top=TINY_TOP, tiny_up0=TINY_UP0, tiny_up1=TINY_UP1, tiny_sep=TINY_SEP)


def build_tiny():
    t.manual_seed(0)
    vq = make_vqvae(setup_hparams("tiny_vqvae", dict(restore_vqvae="")), "cpu")
    for blk in vq.bottleneck.level_blocks:
        blk.k.normal_()
    priors = []
    for i, name in enumerate(["tiny_up0", "tiny_up1", "tiny_top"]):
        t.manual_seed(10 + i)
        priors.append(make_prior(setup_hparams(name, dict(restore_prior="")), vq, "cpu"))
    return vq, priors


def make_labels(prior, n, seed):
    """labels dict as Labeller.get_batch_labels would return (data/labels.py:76-87), built from ids."""
    rng = np.random.RandomState(seed)
    ys, infos = [], []
    total_length, offset = 3 * 4608, 1152
    for i in range(n):
        full_tokens = rng.randint(1, 79, size=40 + 7 * i).tolist() if prior.n_tokens > 0 else []
        genre_ids = rng.randint(0, 10, size=1 if prior.labels_v3 else 1 + i % 3).tolist()
        artist_id = int(rng.randint(0, 100))
        if prior.n_tokens > 0:
            from jukebox.data.labels import get_relevant_lyric_tokens
            tokens, _ = get_relevant_lyric_tokens(full_tokens, prior.n_tokens, total_length, offset, prior.sample_length)
        else:
            tokens = []
        ys.append(prior.labeller.get_y_from_ids(artist_id, genre_ids, tokens, total_length, offset))
        infos.append(dict(artist="n/a", genre="n/a", lyrics="", full_tokens=full_tokens))
    return dict(y=t.from_numpy(np.stack(ys)).long(), info=infos)


def gen_vqvae_and_priors():
    vq, priors = build_tiny()
    hps_dump = {}
    for nm in ["tiny_vqvae", "tiny_up0", "tiny_up1", "tiny_top", "tiny_sep"]:
        kw = dict(restore_vqvae="") if nm == "tiny_vqvae" else dict(restore_prior="")
        H = setup_hparams(nm, kw)
        hps_dump[nm] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in H.items()}
    import json
    with open(os.path.join(HERE, "tiny_hps.json"), "w") as f:
        json.dump(hps_dump, f, indent=0, sort_keys=True)

    # ---- VQ-VAE encode / decode ----
    t.manual_seed(1)
    x = 2 * t.rand(2, 4608, 1) - 1
    zs = vq.encode(x, bs_chunks=2)
    out = npd(vq.state_dict(), "sd.")
    out["x"] = x.numpy()
    for l in range(3):
        out[f"z{l}"] = zs[l].numpy()
        out[f"xd{l}"] = vq.decode(zs[l:], start_level=l, bs_chunks=2).numpy()
    # encoder pre-quantisation distances' top-2 gap, to let tests apply the near-tie rule
    save("vqvae", **out)

    # ---- per-prior sampling (greedy) ----
    n = 3
    out = {}
    labels = [make_labels(p, n, 50 + i) for i, p in enumerate(priors)]
    for i, p in enumerate(priors):
        out.update(npd(p.state_dict(), f"p{i}."))
        out[f"p{i}.labels_y"] = labels[i]["y"].numpy()
        for j, info in enumerate(labels[i]["info"]):
            out[f"p{i}.full_tokens{j}"] = np.array(info["full_tokens"], dtype=np.int64)
    top, up1, up0 = priors[2], priors[1], priors[0]
    # top level: ancestral window, then a primed window (z given), chunked prefill
    y = top.get_y(labels[2], 0)
    z_a = top.sample(n, z=t.zeros(n, 0, dtype=t.long), y=y, top_k=1, chunk_size=5)
    out["top.y0"] = y.numpy()
    out["top.z_ancestral"] = z_a.numpy()
    # the same window through the inner model, keeping raw (merged-vocab) tokens and logits
    x_cond, y_cond, prime = top.get_cond(None, y)
    zz, xc = top.prior_preprocess([prime], [None, x_cond])
    raw_z, raw_preds = top.prior.primed_sample(n, zz, xc, y_cond, top_k=1, chunk_size=5, get_preds=True)
    out.update({"top.x_cond": x_cond.numpy(), "top.y_cond": y_cond.numpy(), "top.raw_z": raw_z.numpy(),
                "top.raw_preds": raw_preds.numpy()})
    y = top.get_y(labels[2], 24)
    z_p = top.sample(n, z=z_a[:, 24:].contiguous(), y=y, top_k=1, chunk_size=5)
    out["top.y24"] = y.numpy()
    out["top.z_primed"] = z_p.numpy()
    z_p16 = top.sample(n, z=z_a[:, 24:].contiguous(), y=y, top_k=1, chunk_size=5, fp16=True)
    out["top.z_primed16"] = z_p16.numpy()
    # partial window
    z_pt = top.sample(n, z=z_a[:, :10].contiguous(), y=top.get_y(labels[2], 0), top_k=1, chunk_size=5, sample_tokens=30)
    out["top.z_partial30"] = z_pt.numpy()
    # upsamplers: conditioned on random upper-level codes
    t.manual_seed(3)
    for nm, p, li in (("up1", up1, 1), ("up0", up0, 0)):
        zc = t.randint(0, 64, (n, p.n_ctx // p.cond_downsample))
        y = p.get_y(labels[li], 0)
        x_cond, y_cond, _ = p.get_cond([zc], y)
        z = p.sample(n, z=t.zeros(n, 0, dtype=t.long), z_conds=[zc], y=y, top_k=1)
        _, preds = p.prior.sample(n, x_cond, y_cond, None, top_k=1, get_preds=True)
        out[f"{nm}.preds"] = preds.numpy()
        zp = p.sample(n, z=z[:, :64].contiguous(), z_conds=[zc], y=y, top_k=1, chunk_size=32)
        out.update({f"{nm}.z_cond": zc.numpy(), f"{nm}.y": y.numpy(), f"{nm}.x_cond": x_cond.numpy(),
                    f"{nm}.y_cond": y_cond.numpy(), f"{nm}.z": z.numpy(), f"{nm}.z_primed": zp.numpy()})
    save("priors", **out)

    # ---- separated encoder-decoder prior: lyric encoder -> encoder_kv -> cross-attention decoder ----
    t.manual_seed(14)
    sep = make_prior(setup_hparams("tiny_sep", dict(restore_prior="")), vq, "cpu")
    # cross-attention c_proj is zero-initialised (transformer.py:131): randomise so the path is exercised
    for l in sep.prior.transformer._attn_mods:
        if l.attn_func == 6:
            t.nn.init.normal_(l.attn.c_proj.w, std=0.02)
            t.nn.init.normal_(l.mlp.c_proj.w, std=0.02)
    lab = make_labels(sep, n, 77)
    so = npd(sep.state_dict(), "sd.")
    so["labels_y"] = lab["y"].numpy()
    for j, info in enumerate(lab["info"]):
        so[f"full_tokens{j}"] = np.array(info["full_tokens"], dtype=np.int64)
    y = sep.get_y(lab, 0)
    x_cond, y_cond, prime = sep.get_cond(None, y)
    ekv = sep.get_encoder_kv(prime, fp16=False, sample=True)
    z_raw, preds = sep.prior.sample(n, x_cond, y_cond, ekv, top_k=1, get_preds=True)
    z_a = sep.sample(n, z=t.zeros(n, 0, dtype=t.long), y=y, top_k=1)
    z_p = sep.sample(n, z=z_a[:, :20].contiguous(), y=y, top_k=1, chunk_size=6)
    so.update({"y0": y.numpy(), "encoder_kv": ekv.numpy(), "x_cond": x_cond.numpy(), "y_cond": y_cond.numpy(),
               "preds": preds.numpy(), "z_raw": z_raw.numpy(), "z_ancestral": z_a.numpy(), "z_primed": z_p.numpy()})
    save("prior_sep", **so)

    # ---- end-to-end 3-level ancestral (window loop of sample.py, greedy) ----
    hps = Hyperparams(n_samples=n, sample_length=4608, hop_fraction=[0.5, 0.5, 0.125], sr=22050, name="unused")
    zs = [t.zeros(n, 0, dtype=t.long) for _ in range(3)]
    sk = [dict(temp=1.0, fp16=False, chunk_size=8, max_batch_size=2, top_k=1),
          dict(temp=1.0, fp16=False, chunk_size=8, max_batch_size=2, top_k=1),
          dict(temp=1.0, fp16=False, chunk_size=5, max_batch_size=2, top_k=1)]
    e2e = {}
    for level in (2, 1, 0):
        prior = priors[level]
        total_length = hps.sample_length // prior.raw_to_tokens
        hop = int(hps.hop_fraction[level] * prior.n_ctx)
        zs = ref_sample.sample_level(zs, labels[level], sk[level], level, prior, total_length, hop, hps)
        e2e[f"z{level}"] = zs[level].numpy()
        e2e[f"x{level}"] = prior.decode(zs[level:], start_level=level, bs_chunks=n).numpy()
        e2e[f"starts{level}"] = np.array(get_starts(total_length, prior.n_ctx, hop), dtype=np.int64)
    # ---- primed mode (sample.py:143-147): encode a 2304-sample prompt, then continue all levels to 4608 samples ----
    t.manual_seed(21)
    xprompt = 2 * t.rand(n, 2304, 1) - 1
    zs_p = priors[-1].encode(xprompt, start_level=0, end_level=3, bs_chunks=n)
    e2e["primed.x"] = xprompt.numpy()
    for level in (0, 1, 2):
        e2e[f"primed.z_prompt{level}"] = zs_p[level].numpy()
    for level in (2, 1, 0):
        prior = priors[level]
        total_length = hps.sample_length // prior.raw_to_tokens
        hop = int(hps.hop_fraction[level] * prior.n_ctx)
        zs_p = ref_sample.sample_level(zs_p, labels[level], sk[level], level, prior, total_length, hop, hps)
        e2e[f"primed.z{level}"] = zs_p[level].numpy()
    # ---- lyric alignment of the top level (align.py:15-83), on the codes just sampled ----
    from jukebox.align import get_alignment
    top.alignment_layer, top.alignment_head = 15, 1          # the tiny top prior's only prime layer
    ahps = Hyperparams(levels=3, hop_fraction=hps.hop_fraction)
    al = get_alignment(None, zs, labels[2], top, False, ahps)
    for j, a_ in enumerate(al):
        e2e[f"alignment{j}"] = np.asarray(a_, dtype=np.float32)
    save("e2e", **e2e)


# ----------------------------------------------------------------------------------------
def gen_misc():
    out = {}
    for i, (tl, n_ctx, hop) in enumerate([(6890, 6144, 768), (27560, 8192, 4096), (110240, 8192, 4096),
                                          (72, 48, 6), (288, 128, 64), (1152, 128, 64), (8192, 8192, 4096),
                                          (8200, 8192, 1024)]):
        out[f"starts{i}.args"] = np.array([tl, n_ctx, hop], dtype=np.int64)
        out[f"starts{i}"] = np.array(get_starts(tl, n_ctx, hop), dtype=np.int64)
    for i, (length, cs) in enumerate([(384, 32), (5398, 32), (34, 7), (1, 5), (64, 64)]):
        out[f"chunks{i}.args"] = np.array([length, cs], dtype=np.int64)
        out[f"chunks{i}"] = np.array(split_chunks(length, cs), dtype=np.int64)
    t.manual_seed(5)
    logits = t.randn(4, 1, 97)
    logits[0, 0, 3] = logits[0, 0, 5]  # a tie
    out["filter.logits"] = logits.numpy()
    for k in (1, 5, 97, 200):
        out[f"filter.top_k{k}"] = filter_logits(logits, top_k=k).numpy()
    save("misc", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["attention", "autoregressive", "models", "misc"]
    if "attention" in which:
        gen_attention()
    if "autoregressive" in which:
        gen_autoregressive()
    if "models" in which:
        gen_vqvae_and_priors()
    if "misc" in which:
        gen_misc()
