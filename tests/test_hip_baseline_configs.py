"""GPU suite: BASELINE.json's configs as parity cases at their real sizes (seeded random-init weights, oracle as the checker).

  config 1  small_vqvae encode + decode, sample_length = 262144, bs = 1, random audio
  config 3  1b_lyrics top-prior geometry (width 2048, 2 heads x 256, attn_order 12 with prime layers, 6144 + 384
            positions) at reduced depth: lyric prefill + decode, fp32 tokens / logits vs the oracle
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from jukebox_amd.hparams import setup_hparams  # noqa: E402


def test_config1_small_vqvae_encode_decode():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from jukebox_amd.make_models import make_vqvae
    from oracle.vqvae import VQVAE as OracleVQVAE
    hps = setup_hparams("small_vqvae", dict(sample_length=262144))
    torch.manual_seed(0)
    with torch.device("cuda"):
        vq = make_vqvae(hps, "cuda")
        for blk in vq.bottleneck.level_blocks:
            blk.k.normal_()
        x = 2 * torch.rand(1, 262144, 1) - 1
    zs = vq.encode(x, bs_chunks=1)
    assert [tuple(z.shape) for z in zs] == [(1, 8192), (1, 1024)]
    sd = {k: v.detach().cpu().numpy() for k, v in vq.state_dict().items()}
    ora = OracleVQVAE(sd, dict(hps))
    zs_ref = ora.encode(x.cpu().numpy())
    for l in range(2):
        got = zs[l].cpu().numpy()
        assert (got == zs_ref[l]).mean() > 0.995, l                       # argmin near-ties may flip a code
        xd = vq.decode([torch.from_numpy(zs_ref[k]).cuda() for k in range(l, 2)], start_level=l, bs_chunks=1)
        xr = ora.decode([zs_ref[k] for k in range(l, 2)], start_level=l)
        assert xd.shape == (1, 262144, 1)
        assert np.abs(xd.cpu().numpy() - xr).max() < 1e-3 * max(1.0, np.abs(xr).max())   # north_star: within 1e-3


def _sd_1b(rng, W, depth, bins, seq, scale=0.02):
    S = W // 4
    sd = {"x_emb.weight": rng.standard_normal((bins, W)) * 0.05, "pos_emb.pos_emb": rng.standard_normal((seq, W)) * 0.01}
    sd["x_out.weight"] = sd["x_emb.weight"]
    for d in range(depth):
        p = f"transformer._attn_mods.{d}."
        for nm, shp, sc in (("attn.c_attn.w", (W, 3 * S), scale), ("attn.c_proj.w", (S, W), scale), ("mlp.c_fc.w", (W, W), scale),
                            ("mlp.c_proj.w", (W, W), scale)):
            sd[p + nm] = rng.standard_normal(shp) * sc
        for nm, n in (("attn.c_attn.b", 3 * S), ("attn.c_proj.b", W), ("mlp.c_fc.b", W), ("mlp.c_proj.b", W)):
            sd[p + nm] = rng.standard_normal(n) * 0.01
        for ln in ("ln_0", "ln_1"):
            sd[p + ln + ".weight"] = 1 + 0.05 * rng.standard_normal(W)
            sd[p + ln + ".bias"] = 0.02 * rng.standard_normal(W)
    return {k: np.asarray(v, np.float32) for k, v in sd.items()}


def test_config3_1b_top_prior_geometry():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from jukebox_amd.engine import PriorEngine
    from oracle.autoregressive import ConditionalAutoregressive2D as OracleAR
    rng = np.random.default_rng(7)
    W, depth, bins, n_tok, n_ctx, blocks = 2048, 16, 2127, 384, 6144, 64      # layer 15 is a prime layer (attn_order 12)
    seq = n_ctx + n_tok
    sd = _sd_1b(rng, W, depth, bins, seq)
    N, n_total = 4, n_tok + 40
    prime = rng.integers(0, 79, (N, n_tok))
    xc = np.concatenate([np.zeros((N, n_tok, W), np.float32), (rng.standard_normal((N, n_ctx, W)) * 0.05).astype(np.float32)], 1)
    yc = (rng.standard_normal((N, 1, W)) * 0.05).astype(np.float32)
    ora = OracleAR(sd, "", (seq,), bins, W, depth, 2, attn_order=12, blocks=blocks, x_cond=True, y_cond=True, prime_len=n_tok)
    z_ref, p_ref = ora.primed_sample(N, prime, xc, yc, top_k=1, get_preds=True, chunk_size=128, sample_tokens=n_total)
    eng = PriorEngine({k: torch.from_numpy(v).cuda() for k, v in sd.items()}, "", n_batch=N, seq_len=seq, bins=bins, width=W,
                      depth=depth, heads=2, attn_order=12, blocks=blocks, prime_len=n_tok, y_cond=True, fp16=False,
                      want_preds=True, chunk_cap=128)
    eng.set_cond(torch.from_numpy(xc), torch.from_numpy(yc))
    eng.set_sampling(temp=1.0, top_k=1)
    eng.tokens[:, :n_tok] = torch.from_numpy(prime).cuda()
    eng.prefill(0, n_tok)
    eng.decode(n_tok, n_total - n_tok)
    torch.cuda.synchronize()
    preds, z = eng.preds.cpu().numpy()[:, :n_total], eng.tokens.cpu().numpy()[:, :n_total]
    assert np.abs(preds - p_ref).max() < 5e-4 * max(1.0, np.abs(p_ref).max())
    if not np.array_equal(z, z_ref):
        n, t = np.argwhere(z != z_ref)[np.argmin(np.argwhere(z != z_ref)[:, 1])]
        srt = np.sort(p_ref[n, t])
        assert srt[-1] - srt[-2] < 2e-3, "token mismatch outside a near-tie"
    # fp16 production path at the same geometry (MFMA QK^T, d_head = 256): in range, close to the fp32 stream
    eng16 = PriorEngine({k: torch.from_numpy(v).cuda() for k, v in sd.items()}, "", n_batch=N, seq_len=seq, bins=bins,
                        width=W, depth=depth, heads=2, attn_order=12, blocks=blocks, prime_len=n_tok, y_cond=True, fp16=True,
                        want_preds=True, chunk_cap=128)
    eng16.set_cond(torch.from_numpy(xc), torch.from_numpy(yc))
    eng16.set_sampling(temp=1.0, top_k=1)
    eng16.tokens[:, :n_tok] = torch.from_numpy(prime).cuda()
    eng16.prefill(0, n_tok)
    eng16.decode(n_tok, n_total - n_tok)
    torch.cuda.synchronize()
    p16 = eng16.preds.cpu().numpy()[:, n_tok - 1:n_tok + 1]
    assert np.abs(p16 - p_ref[:, n_tok - 1:n_tok + 1]).max() < 5e-2 * max(1.0, np.abs(p_ref).max())
