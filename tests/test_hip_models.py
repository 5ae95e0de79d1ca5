"""GPU suite, model level: the host mirror (make_models / SimplePrior / VQVAE / sample driver) running on the HIP
kernels against the reference's golden outputs (tests/golden), all in fp32 parity mode.
VQ-VAE bar: <= 1e-3 on reconstruction (BASELINE.json north_star); greedy tokens identical, near-tie rule when
reference logits are available."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import load_golden, sub_state  # noqa: E402
from jukebox_amd.hparams import Hyperparams, setup_hparams  # noqa: E402


@pytest.fixture(scope="module")
def models(tiny_hps):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from jukebox_amd.make_models import make_prior, make_vqvae
    vq_h = Hyperparams(tiny_hps["tiny_vqvae"])
    vq_h.downs_t, vq_h.strides_t = tuple(vq_h.downs_t), tuple(vq_h.strides_t)
    vq = make_vqvae(vq_h, "cuda")
    g = load_golden("vqvae")
    vq.load_state_dict({k: torch.from_numpy(v) for k, v in sub_state(g, "sd.").items()}, strict=True)
    gp = load_golden("priors")
    priors = []
    for i, nm in enumerate(("tiny_up0", "tiny_up1", "tiny_top")):
        h = Hyperparams(tiny_hps[nm])
        h.y_bins = tuple(h.y_bins)
        p = make_prior(h, vq, "cpu")
        p.load_state_dict({k: torch.from_numpy(v) for k, v in sub_state(gp, f"p{i}.").items()
                           if not k.startswith(("labels_y", "full_tokens"))}, strict=True)
        priors.append(p.cuda())
    return vq, priors


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_vqvae_decode_and_encode(models):
    vq, _ = models
    g = load_golden("vqvae")
    for l in range(3):
        xd = vq.decode([cu(g[f"z{k}"]) for k in range(l, 3)], start_level=l, bs_chunks=2).cpu().numpy()
        assert xd.shape == g[f"xd{l}"].shape
        assert np.abs(xd - g[f"xd{l}"]).max() < 1e-4, l            # bar is 1e-3
    zs = vq.encode(cu(g["x"]), bs_chunks=1)
    for l in range(3):
        assert (zs[l].cpu().numpy() == g[f"z{l}"]).mean() > 0.995     # argmin near-ties may flip a code


def test_top_prior_conditioning_and_sampling(models):
    _, (up0, up1, top) = models
    g = load_golden("priors")
    y0 = cu(g["top.y0"])
    x_cond, y_cond, prime = top.get_cond(None, y0)
    assert np.abs(x_cond.cpu().numpy() - g["top.x_cond"]).max() < 1e-6
    assert np.abs(y_cond.cpu().numpy() - g["top.y_cond"]).max() < 1e-6
    zz, xc = top.prior_preprocess([prime], [None, x_cond])
    z_raw, preds = top.prior.primed_sample(3, zz, xc, y_cond, top_k=1, get_preds=True)
    assert np.abs(preds.cpu().numpy() - g["top.raw_preds"]).max() < 2e-4
    assert np.array_equal(z_raw.cpu().numpy(), g["top.raw_z"])
    z = top.sample(3, z=torch.zeros(3, 0, dtype=torch.long, device="cuda"), y=y0, top_k=1, chunk_size=5)
    assert np.array_equal(z.cpu().numpy(), g["top.z_ancestral"])
    zp = top.sample(3, z=cu(g["top.z_ancestral"][:, 24:]), y=cu(g["top.y24"]), top_k=1, chunk_size=5)
    assert np.array_equal(zp.cpu().numpy(), g["top.z_primed"])
    zpt = top.sample(3, z=cu(g["top.z_ancestral"][:, :10]), y=y0, top_k=1, chunk_size=5, sample_tokens=30)
    assert np.array_equal(zpt.cpu().numpy(), g["top.z_partial30"])


def test_upsamplers_conditioner_and_sampling(models):
    _, (up0, up1, top) = models
    g = load_golden("priors")
    for nm, p in (("up1", up1), ("up0", up0)):
        zc, y = cu(g[f"{nm}.z_cond"]), cu(g[f"{nm}.y"])
        x_cond, y_cond, _ = p.get_cond([zc], y)
        ref = g[f"{nm}.x_cond"]
        assert np.abs(x_cond.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
        assert np.abs(y_cond.cpu().numpy() - g[f"{nm}.y_cond"]).max() < 1e-6
        z, preds = p.prior.sample(3, x_cond, y_cond, None, top_k=1, get_preds=True)
        assert np.abs(preds.cpu().numpy() - g[f"{nm}.preds"]).max() < 3e-4
        assert np.array_equal(z.cpu().numpy(), g[f"{nm}.z"])
        zp = p.sample(3, z=cu(g[f"{nm}.z"][:, :64]), z_conds=[zc], y=y, top_k=1, chunk_size=32)
        assert np.array_equal(zp.cpu().numpy(), g[f"{nm}.z_primed"])


def test_end_to_end_three_levels(models):
    """The reference's window loop (sample.py) over the tiny 3-level model, greedy: codes per level and decoded audio."""
    from jukebox_amd import sample as S
    vq, priors = models
    g, e = load_golden("priors"), load_golden("e2e")
    n = 3
    labels = [dict(y=cu(g[f"p{i}.labels_y"]),
                   info=[dict(full_tokens=list(map(int, g[f"p{i}.full_tokens{j}"]))) for j in range(n)]) for i in range(3)]
    hps = Hyperparams(n_samples=n, sample_length=4608, hop_fraction=[0.5, 0.5, 0.125], sr=22050, name="unused")
    sk = [dict(temp=1.0, fp16=False, chunk_size=8, max_batch_size=2, top_k=1),
          dict(temp=1.0, fp16=False, chunk_size=8, max_batch_size=2, top_k=1),
          dict(temp=1.0, fp16=False, chunk_size=5, max_batch_size=2, top_k=1)]
    zs = S.ancestral_sample(labels, sk, priors, hps, save=False)
    for p in priors:
        p.cuda()
    for level in (2, 1, 0):
        got, want = zs[level].cpu().numpy(), e[f"z{level}"]
        assert got.shape == want.shape
        # a lower level conditions on the upper one: compare it only where the upper levels agreed
        assert (got == want).mean() > 0.98, (level, (got == want).mean())
    assert np.array_equal(zs[2].cpu().numpy(), e["z2"])
    x0 = S._sample.last_audio[0].cpu().numpy()
    if np.array_equal(zs[0].cpu().numpy(), e["z0"]):
        assert np.abs(x0 - e["x0"]).max() < 1e-4


@pytest.mark.parametrize("func", [0, 1, 2, 3, 7])
def test_factored_attention_module_chunks(func):
    """The module-level forward(sample=True) in the reference's ragged chunk schedule (its check_chunks analogue)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from jukebox_amd.transformer.factored_attention import FactoredAttention
    g = load_golden("attention")
    sd = sub_state(g, f"f{func}.")
    x, y_chunks = sd.pop("x"), sd.pop("y_chunks")
    sd.pop("y_full"); sd.pop("encoder_kv", None)
    L = x.shape[1]
    att = FactoredAttention(32, L, 64, 2, mask=True, attn_func=func, blocks=8, prime_len=24 if func == 7 else None)
    att.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    att = att.cuda().eval()
    xs = cu(x)
    ys, pos = [], 0
    for c in list(g["chunks"]) + [11] * 100:
        if pos >= L:
            break
        c = int(min(c, L - pos))
        ys.append(att(xs[:, pos:pos + c].contiguous(), sample=True))
        pos += c
        att.check_cache(2, pos, False)
    got = torch.cat(ys, 1).cpu().numpy()
    assert np.abs(got - y_chunks).max() < 5e-6


def test_pipelined_levels_equal_sequential(models):
    """The level pipeline (one stream per level) draws exactly the tokens of the reference's sequential level loop,
    with random (non-greedy) sampling in fp16."""
    from jukebox_amd import sample as S
    vq, priors = models
    g = load_golden("priors")
    n = 3
    labels = [dict(y=cu(g[f"p{i}.labels_y"]),
                   info=[dict(full_tokens=list(map(int, g[f"p{i}.full_tokens{j}"]))) for j in range(n)]) for i in range(3)]
    sk = [dict(temp=0.99, fp16=True, chunk_size=8, max_batch_size=3) for _ in range(3)]
    outs = []
    # pipeline_chunk: decode steps between two publications of a window's codes to the level below (0: whole windows);
    # 5 does not divide the windows, so a lower window starts in the middle of an upper one
    # recheck: the lowest level's window in plain chunks of that many steps while upper levels still run, the sampler asked
    # again between them (ConditionalAutoregressive2D._decode_window; 512 in production -- longer than these windows)
    from jukebox_amd.prior.autoregressive import ConditionalAutoregressive2D as AR
    for pipe, chunk, bs, recheck in ((False, 0, 3, 512), (True, 0, 3, 512), (True, 5, 3, 512), (True, 256, 3, 512), (False, 0, 2, 512),
                                     (True, 0, 3, 7)):
        hps = Hyperparams(n_samples=n, sample_length=4608, hop_fraction=[0.5, 0.5, 0.125], sr=22050, name="unused",
                          keep_priors_resident=True, pipeline_levels=pipe, pipeline_chunk=chunk, seed=5)
        AR.PIPE_RECHECK_STEPS = recheck
        try:
            zs = S.ancestral_sample(labels, [dict(k, max_batch_size=bs) for k in sk], priors, hps, save=False)
        finally:
            AR.PIPE_RECHECK_STEPS = 512
        outs.append([z.cpu().numpy() for z in zs])
    for other in outs[1:]:            # the last run splits the batch 2 + 1: draws are keyed by the global sample index
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
    # another seed gives another sample; no seed gives a fresh one per job (the reference's unseeded torch RNG)
    hps.seed = 6
    z6 = S.ancestral_sample(labels, sk, priors, hps, save=False)
    assert not np.array_equal(z6[0].cpu().numpy(), outs[0][0])


def test_alignment_matches_reference(models):
    """jukebox/align.py:get_alignment on the codes of the golden 3-level run (fp32): every item's stitched
    (total_length, n_lyric_characters) attention matrix."""
    from jukebox_amd.align import get_alignment
    vq, priors = models
    g, e = load_golden("priors"), load_golden("e2e")
    top = priors[2]
    top.alignment_layer, top.alignment_head = 15, 1
    labels = dict(y=cu(g["p2.labels_y"]), info=[dict(full_tokens=list(map(int, g[f"p2.full_tokens{j}"]))) for j in range(3)])
    zs = [cu(e["z0"]), cu(e["z1"]), cu(e["z2"])]
    hps = Hyperparams(levels=3, hop_fraction=[0.5, 0.5, 0.125])
    al = get_alignment(None, zs, labels, top, False, hps)
    top.cuda()
    for j in range(3):
        assert al[j].shape == e[f"alignment{j}"].shape
        assert np.abs(al[j] - e[f"alignment{j}"]).max() < 2e-6


def test_primed_mode_and_data_dump(models, tmp_path):
    """sample.py primed mode: encode a prompt with the VQ-VAE, continue every level (greedy) -- and the data.pth.tar
    dump / load_codes round trip (sample.py:116,164-175)."""
    import os
    from jukebox_amd import sample as S
    vq, priors = models
    g, e = load_golden("priors"), load_golden("e2e")
    n = 3
    labels = [dict(y=cu(g[f"p{i}.labels_y"]),
                   info=[dict(full_tokens=list(map(int, g[f"p{i}.full_tokens{j}"]))) for j in range(n)]) for i in range(3)]
    sk = [dict(temp=1.0, fp16=False, chunk_size=8, max_batch_size=2, top_k=1),
          dict(temp=1.0, fp16=False, chunk_size=8, max_batch_size=2, top_k=1),
          dict(temp=1.0, fp16=False, chunk_size=5, max_batch_size=2, top_k=1)]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        hps = Hyperparams(n_samples=n, sample_length=4608, hop_fraction=[0.5, 0.5, 0.125], sr=22050, name="run",
                          keep_priors_resident=True)
        for p in priors:
            p.alignment_layer = None
        zs = S.primed_sample(cu(e["primed.x"]), labels, sk, priors, hps, save=True)
        agree = [(zs[l].cpu().numpy() == e[f"primed.z{l}"]).mean() for l in range(3)]
        prompt_ok = all((zs[l][:, :e[f"primed.z_prompt{l}"].shape[1]].cpu().numpy() == e[f"primed.z_prompt{l}"]).mean() > 0.99
                        for l in range(3))
        assert prompt_ok
        if all(np.array_equal(zs[l][:, :e[f"primed.z_prompt{l}"].shape[1]].cpu().numpy(), e[f"primed.z_prompt{l}"]) for l in range(3)):
            assert np.array_equal(zs[2].cpu().numpy(), e["primed.z2"])
            assert min(agree) > 0.98, agree
        data = torch.load("run/level_0/data.pth.tar", map_location="cpu", weights_only=False)
        assert set(data) == {"zs", "labels", "sampling_kwargs", "x"} and data["x"].shape == (n, 4608, 1)
        assert os.path.exists("run/level_0/item_0.wav")
        zs2 = S.load_codes("run/level_2/data.pth.tar", 2304, priors, hps)
        assert [z.shape[1] for z in zs2] == [576, 144, 36] and torch.equal(zs2[2].cpu(), zs[2][:, :36].cpu())
    finally:
        os.chdir(cwd)


def test_separated_encoder_decoder_prior(models, tiny_hps):
    """prior_5b_lyrics structure on the HIP path: lyric encoder (only_encode prefill) -> projection + LayerNorm ->
    cross-attention layers (attn_func 6) in prefill and decode, merged_decoder head."""
    from jukebox_amd.make_models import make_prior
    vq, _ = models
    g = load_golden("prior_sep")
    h = Hyperparams(tiny_hps["tiny_sep"])
    h.y_bins = tuple(h.y_bins)
    sep = make_prior(h, vq, "cpu")
    sd = {k: torch.from_numpy(v) for k, v in sub_state(g, "sd.").items()}
    assert set(sd) == set(sep.state_dict())
    sep.load_state_dict(sd, strict=True)
    sep = sep.cuda()
    y0 = cu(g["y0"])
    x_cond, y_cond, prime = sep.get_cond(None, y0)
    ekv = sep.get_encoder_kv(prime, fp16=False, sample=True)
    assert np.abs(ekv.cpu().numpy() - g["encoder_kv"]).max() < 5e-5
    z, preds = sep.prior.sample(3, x_cond, y_cond, ekv, top_k=1, get_preds=True)
    assert np.abs(preds.cpu().numpy() - g["preds"]).max() < 3e-4
    assert np.array_equal(z.cpu().numpy(), g["z_raw"])
    za = sep.sample(3, z=torch.zeros(3, 0, dtype=torch.long, device="cuda"), y=y0, top_k=1)
    assert np.array_equal(za.cpu().numpy(), g["z_ancestral"])
    zp = sep.sample(3, z=cu(g["z_ancestral"][:, :20]), y=y0, top_k=1, chunk_size=6)
    assert np.array_equal(zp.cpu().numpy(), g["z_primed"])
    # fp16 path runs and stays in range (rounding-level agreement is covered at the transformer level)
    z16 = sep.sample(3, z=torch.zeros(3, 0, dtype=torch.long, device="cuda"), y=y0, top_k=1, fp16=True)
    assert (z16.cpu().numpy() == g["z_ancestral"]).mean() > 0.5


def test_top_prior_fp16_against_reference_fp16(models):
    """SimplePrior.sample(fp16=True) against the reference's own fp16 run (CPU half arithmetic): greedy streams agree
    except where half rounding flips a near-tie (~1 %/step expected at these logit gaps)."""
    _, (_, _, top) = models
    g = load_golden("priors")
    zp = top.sample(3, z=cu(g["top.z_ancestral"][:, 24:]), y=cu(g["top.y24"]), top_k=1, chunk_size=5, fp16=True)
    got, want = zp.cpu().numpy(), g["top.z_primed16"]
    assert got.shape == want.shape and np.array_equal(got[:, :24], want[:, :24])     # primed part is copied through
    assert (got == want).mean() > 0.85, (got == want).mean()


def test_teacher_forced_losses(models, tiny_hps):
    """SimplePrior.z_forward (prior.py:312-347) on the HIP prefill path: loss, bits per token of the lyric and music parts
    and the logits of a given code sequence against the reference (tests/golden/forward.npz), fp32."""
    from jukebox_amd.make_models import make_prior
    vq, (up0, up1, top) = models
    g, f = load_golden("priors"), load_golden("forward")
    gs = load_golden("prior_sep")
    h = Hyperparams(tiny_hps["tiny_sep"])
    h.y_bins = tuple(h.y_bins)
    sep = make_prior(h, vq, "cpu")
    sep.load_state_dict({k: torch.from_numpy(v) for k, v in sub_state(gs, "sd.").items()}, strict=True)
    sep = sep.cuda()
    cases = (("top", top, g["top.z_ancestral"], [], g["top.y0"]),
             ("up0", up0, g["up0.z"], [g["up0.z_cond"]], g["up0.y"]),
             ("up1", up1, g["up1.z"], [g["up1.z_cond"]], g["up1.y"]),
             ("sep", sep, gs["z_ancestral"][:1], [], gs["y0"][:1]))
    for tag, prior, z, z_conds, y in cases:
        loss, m = prior.z_forward(cu(z), [cu(c) for c in z_conds], cu(y), fp16=False, get_preds=True)
        assert np.abs(m["preds"].cpu().numpy() - f[f"{tag}.preds"]).max() < 3e-4, tag
        for k in ("bpd", "prime_loss", "gen_loss"):
            assert abs(float(m[k]) - float(f[f"{tag}.{k}"])) < 1e-4, (tag, k)
        assert abs(float(loss) - float(f[f"{tag}.loss"])) < 1e-4, tag


def test_timed_job_tokens():
    """The job bench.py times, as a job: 1b_lyrics at its real widths, contexts, heads and conditioners (upsamplers 1920 wide /
    one 480-channel head / 8192 tokens, top prior 2048 wide / 2 heads / 6144 + 384 tokens, the 5b VQ-VAE), 16 samples, 6 s of
    audio (the shortest the upsamplers take, sample.py:183) at temp 0.99 in fp16 -- at reduced DEPTH (12 / 12 / 16 layers; the
    schedule does not know the depth) -- sampled twice:
      * as the bench does: the level pipeline (a host thread and a stream per level, partial windows published to the level
        below), level 0 switching to software-pipelined launches with operand-order hand-offs in the middle of the window in
        which the upper levels finish, the in-situ comparison of the two launch forms, the pair released at the end;
      * the reference's schedule (sample.py:90-121): one level after the other, whole windows, plain launch chain.
    Same seed, so the codes of EVERY level must be identical (the draw of a position is a pure function of seed, level, sample
    and position); and a second pipelined job in the same process reproduces the first (what bench.py asserts over its timed
    steps through `breakdown.step_digests`)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from jukebox_amd import sample as S
    from jukebox_amd.make_models import MODELS, make_prior, make_vqvae
    sr, n = 44100, 16
    sample_length = int(6.0 * sr) // 128 * 128
    names = MODELS["1b_lyrics"]
    torch.manual_seed(0)
    with torch.device("cuda"):
        vq = make_vqvae(setup_hparams(names[0], dict(sample_length=sample_length, restore_vqvae="")), "cuda")
        for blk in vq.bottleneck.level_blocks:
            blk.k.normal_()
        priors = [make_prior(setup_hparams(nm, dict(restore_prior="", prior_depth=d)), vq, "cuda") for nm, d in zip(names[1:], (12, 12, 16))]
    assert priors[0].prior.pipeline_candidate and priors[0].prior.width == 1920 and priors[2].prior.width == 2048
    rng = np.random.RandomState(0)
    labels = []
    for p in priors:
        yb = p.y_emb.bow_genre_emb.bins, p.y_emb.artist_emb.bins
        items = [dict(artist_id=int(rng.randint(1, yb[1])), genre_ids=[int(rng.randint(1, yb[0]))],
                      full_tokens=rng.randint(1, 79, size=1500).tolist() if p.n_tokens > 0 else [], total_length=180 * sr, offset=0)
                 for _ in range(n)]
        labels.append(p.labeller.get_batch_labels_from_ids(items, "cuda"))
    sk = S.default_sampling_kwargs("1b_lyrics")

    def job(pipelined):
        hps = Hyperparams(n_samples=n, sample_length=sample_length, hop_fraction=[0.5, 0.5, 0.125], sr=sr, name="unused",
                          keep_priors_resident=True, pipeline_levels=pipelined, pipeline_launches=pipelined, seed=0)
        priors[0].prior.pipeline_report = None
        zs = S.ancestral_sample(labels, sk, priors, hps, save=False, device="cuda")
        torch.cuda.synchronize()
        return [z.cpu().numpy() for z in zs], priors[0].prior.pipeline_report

    z_pipe, report = job(True)
    assert report is not None and report["pipelined_ms"], "level 0 never tried its pipelined launches: the test did not test the bench's schedule"
    assert not any(e.pipelined or e.pipeline_resident for p in priors for e in p.prior._engines.values()), "a pair of streams outlived the job"
    z_seq, rep_seq = job(False)
    assert rep_seq is None                                   # the sequential plain-chain schedule ran no pipelined launch
    z_again, _ = job(True)
    print("timed-job tokens: level 0 in situ", report, "; tokens per level", [z.shape for z in z_pipe])
    for l in range(3):
        assert z_pipe[l].shape == (n, sample_length // priors[l].raw_to_tokens)
        assert np.array_equal(z_pipe[l], z_seq[l]), f"level {l}: the pipelined schedule and the sequential plain chain drew different codes"
        assert np.array_equal(z_pipe[l], z_again[l]), f"level {l}: two pipelined jobs with one seed drew different codes"
    assert len(np.unique(z_pipe[0])) > 100                   # (a sampled stream, not a constant)
    for p in priors:
        p.cpu()
