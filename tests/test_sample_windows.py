"""CPU suite: the window / level / sharding logic of jukebox_amd.sample with a fake prior, following the
reference's own test (jukebox/tests/test_sample.py:13-141: a DummyPrior emitting position-determined tokens,
3 levels, n_ctx = 8192, hops n/2, n/2, n/8) -- plus the sharded driver on 2 gloo ranks.

The fake token at absolute position p of sample s on level l is f(l, s, p); the fake prior derives p from the
label offset it is handed (prior.get_y) and checks every conditioning input it receives, so a wrong window start,
conditioning length, upper-level slice or shard boundary fails loudly."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch as t

from conftest import ROOT
from jukebox_amd import sample as S
from jukebox_amd.hparams import Hyperparams

BINS = 2048
LOG = []            # (level, window start) in call order, all priors


def f(level, sample_ids, pos):
    """(n,) sample ids x (T,) positions -> (n, T) tokens."""
    return (sample_ids.view(-1, 1) * 7919 + pos.view(1, -1) * (level + 3) + level * 101) % BINS


class DummyPrior:
    def __init__(self, level, levels, n_ctx, raw_to_tokens, cond_downsample):
        self.level, self.levels, self.n_ctx = level, levels, n_ctx
        self.raw_to_tokens, self.cond_downsample = raw_to_tokens, cond_downsample
        self.x_cond = level != levels - 1
        self.n_tokens = 0
        self.calls = []
        self.kwargs_seen = []
        self.after_publish = None

    def to(self, device):
        return self

    def cpu(self):
        return self

    def get_y(self, labels, start):
        y = labels["y"].clone()
        y[:, 1] = y[:, 1] + int(start * self.raw_to_tokens)
        return y

    def get_z_conds(self, zs, start, end):
        if not self.x_cond:
            return None
        cd = self.cond_downsample
        assert start % cd == end % cd == 0
        z_cond = zs[self.level + 1][:, start // cd:end // cd]
        assert z_cond.shape[1] == self.n_ctx // cd
        return [z_cond]

    def sample(self, n_samples, z=None, z_conds=None, y=None, sample_tokens=None, sample_base=0, **kw):
        ids = y[:, 3]
        start = int(y[0, 1]) // self.raw_to_tokens
        assert (y[:, 1] == y[0, 1]).all()
        n_tok = self.n_ctx if sample_tokens is None else sample_tokens
        want = f(self.level, ids, t.arange(start, start + n_tok))
        # primed tokens must be exactly the tokens of this absolute range
        assert z.shape[1] < n_tok and t.equal(z, want[:, :z.shape[1]]), "conditioning tokens misaligned"
        if self.x_cond:
            cd = self.cond_downsample
            up = f(self.level + 1, ids, t.arange(start // cd, (start + self.n_ctx) // cd))
            assert t.equal(z_conds[0], up), "upper-level conditioning misaligned"
        # global sample index bookkeeping of the sharded driver: artist id column carries the global index
        assert t.equal(ids, t.arange(int(ids[0]), int(ids[0]) + n_samples))
        assert int(ids[0]) >= sample_base
        self.calls.append((start, z.shape[1], n_tok))
        self.kwargs_seen.append(dict(kw, sample_base=sample_base))
        LOG.append((self.level, start))
        tap = getattr(self, "window_tap", None)
        if tap is not None:                      # the level pipeline's partial-window publication (SimplePrior._decode_tap)
            every, cb = tap
            for lo in range(z.shape[1], n_tok, every):
                hi = min(lo + every, n_tok)
                cb(lo, hi, want[:, lo:hi])
                if self.after_publish is not None:
                    self.after_publish(self, start + hi)
        return want

    def decode(self, zs, start_level=None, bs_chunks=1):
        return zs[0].float().unsqueeze(-1)


def make_setup(n_samples, top_tokens, n_ctxs=(8192, 8192, 8192)):
    levels = 3
    raw = [8, 32, 128]
    priors = [DummyPrior(l, levels, n_ctxs[l], raw[l], 4 if l < 2 else None) for l in range(levels)]
    hps = Hyperparams(n_samples=n_samples, sample_length=top_tokens * raw[2], hop_fraction=[0.5, 0.5, 0.125], sr=44100,
                      name="unused")
    y = t.zeros((n_samples, 5), dtype=t.long)
    y[:, 0] = 10 ** 9
    y[:, 3] = t.arange(n_samples)
    labels = [dict(y=y.clone(), info=[dict(full_tokens=[])] * n_samples) for _ in range(levels)]
    sk = [dict(max_batch_size=3), dict(max_batch_size=3), dict(max_batch_size=2)]
    return priors, hps, labels, sk


def check_levels(zs, n_samples, top_tokens):
    ids = t.arange(n_samples)
    for level, mult in ((2, 1), (1, 4), (0, 16)):
        assert zs[level].shape == (n_samples, top_tokens * mult)
        assert t.equal(zs[level], f(level, ids, t.arange(top_tokens * mult)))


def test_ancestral_windows_three_levels():
    priors, hps, labels, sk = make_setup(n_samples=5, top_tokens=8192 + 3 * 1024)
    zs = S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
    check_levels(zs, 5, 8192 + 3 * 1024)
    # top level: hop n/8 -> windows at 0, 1024, 2048, 3072; each later window primes on n_ctx - hop tokens
    top_calls = sorted(set(priors[2].calls))
    assert [c[0] for c in top_calls] == [0, 1024, 2048, 3072]
    assert all(c[1] == (0 if c[0] == 0 else 8192 - 1024) for c in top_calls)
    # lower levels: hop n/2
    assert sorted(set(c[0] for c in priors[1].calls))[:3] == [0, 4096, 8192]


@pytest.mark.parametrize("chunk", [0, 256, 1000])
def test_pipelined_levels_on_cpu(chunk):
    """The level pipeline (one host thread per level, partial-window publication every `pipeline_chunk` steps) visits the
    same windows with the same conditioning as the sequential loop; with publication on, a lower level starts while
    the upper level's first window is still being sampled."""
    import threading
    top = 8192 + 2 * 1024
    priors, hps, labels, sk = make_setup(n_samples=3, top_tokens=top)
    hps.keep_priors_resident, hps.pipeline_levels, hps.pipeline_chunk = True, True, chunk
    sk[2]["max_batch_size"] = 3              # partial publication needs the window in one sub-batch
    started = {1: threading.Event(), 0: threading.Event()}
    seen = {}

    def after_publish(prior, n_done):
        # the top level's first window has published the codes level 1's first window needs: wait for level 1 to start
        # before finishing the window (only possible when publication is partial)
        if prior.level == 2 and n_done >= 2048 and n_done < 8192 and "l1" not in seen:
            seen["l1"] = started[1].wait(timeout=20)
        if prior.level == 1 and n_done >= 2048 and n_done < 8192 and "l0" not in seen:
            seen["l0"] = started[0].wait(timeout=20)

    for p in priors:
        p.after_publish = after_publish if chunk else None
        orig = p.sample

        def wrapped(*a, _orig=orig, _p=p, **k):
            if _p.level in started:
                started[_p.level].set()
            return _orig(*a, **k)
        p.sample = wrapped
    zs = S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
    check_levels(zs, 3, top)
    if chunk:
        assert seen == {"l1": True, "l0": True}, seen
    # same windows as the sequential schedule
    ref_priors, ref_hps, _, _ = make_setup(n_samples=3, top_tokens=top)
    S.ancestral_sample(labels, sk, ref_priors, ref_hps, save=False, device="cpu")
    for a, b in zip(priors, ref_priors):
        assert sorted(a.calls) == sorted(b.calls)


def test_level_pipeline_gives_pipelined_launches_to_the_lowest_level_once_it_is_alone():
    """The level pipeline's say on software-pipelined launches (sample._sample_levels_pipelined): ONE level -- the lowest whose
    model can have them (`prior.prior.pipeline_candidate`: an upsampler) -- is asked, and is told "not now" while any other level
    still samples, "yes" once it has the GPU to itself; every other level keeps the plain chain (False).  Nothing of it outlives
    the job (release_pipeline, pipeline_launches back to None).  (Round 5's two pipelined levels side by side -- regimes, a
    rendezvous, a bound -- measured 67.53 s against 67.21 s and were removed in round 6.)"""
    top = 8192 + 1024
    priors, hps, labels, sk = make_setup(n_samples=2, top_tokens=top)
    hps.keep_priors_resident, hps.pipeline_levels, hps.pipeline_chunk = True, True, 256
    sk[2]["max_batch_size"] = 3

    class FakeAR:
        def __init__(self, candidate):
            self.pipeline_candidate, self.pipeline_launches, self.released = candidate, None, 0
        def release_pipeline(self):
            self.released += 1

    seen = {0: [], 1: [], 2: []}
    for p in priors:
        p.prior = FakeAR(p.level != 2)
        orig = p.sample

        def wrapped(*a, _orig=orig, _p=p, **k):
            w = _p.prior.pipeline_launches
            seen[_p.level].append(w() if callable(w) else w)
            out = _orig(*a, **k)
            seen[_p.level].append(w() if callable(w) else w)
            return out
        p.sample = wrapped
    zs = S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
    check_levels(zs, 2, top)
    assert all(x is False for x in seen[2] + seen[1]), (seen[2][:4], seen[1][:4])      # never candidates for the launches
    assert set(seen[0]) <= {False, True} and seen[0][0] is False and seen[0][-1] is True and seen[0] == sorted(seen[0])
    for p in priors:
        assert p.prior.pipeline_launches is None and p.prior.released >= 2      # job start + job end
    # hps.pipeline_launches = False: nobody is asked
    priors, hps, labels, sk = make_setup(n_samples=2, top_tokens=top)
    hps.keep_priors_resident, hps.pipeline_levels, hps.pipeline_chunk, hps.pipeline_launches = True, True, 256, False
    sk[2]["max_batch_size"] = 3
    said = []
    for p in priors:
        p.prior = FakeAR(p.level != 2)
        orig = p.sample

        def wrapped0(*a, _orig=orig, _p=p, **k):
            said.append(_p.prior.pipeline_launches)
            return _orig(*a, **k)
        p.sample = wrapped0
    S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
    assert said and all(x is False for x in said)


def test_pipelined_levels_refuse_a_total_length_below_a_lower_context():
    """A job shorter than a lower level's context (here level 1: 8192 tokens = 2048 top-level codes, the top level has 757)
    fails in the sequential loop on get_z_conds' length assertion; the level pipeline must fail too, not wait for upper-level
    codes that will never come."""
    priors, hps, labels, sk = make_setup(n_samples=2, top_tokens=757)
    with pytest.raises(AssertionError):
        S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
    priors, hps, labels, sk = make_setup(n_samples=2, top_tokens=757)
    hps.keep_priors_resident, hps.pipeline_levels = True, True
    with pytest.raises(AssertionError, match="shorter than this level's context"):
        S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")


def test_primed_continue_and_partial_window():
    """continue_sample from existing codes (sample.py:131-134) and a short top level (sample_partial_window)."""
    n, top = 4, 8192 + 2048
    priors, hps, labels, sk = make_setup(n, top)
    ids = t.arange(n)
    zs0 = [f(2 - i, ids, t.arange(1500 * m)) for i, m in ((2, 16), (1, 4), (0, 1))]     # levels 0,1,2 prefixes
    zs = S.continue_sample([z.clone() for z in zs0], labels, sk, priors, hps, save=False, device="cpu")
    check_levels(zs, n, top)
    # partial window: total length shorter than n_ctx on the top level only
    priors, hps, labels, sk = make_setup(3, 2048 * 4 // 4, n_ctxs=(8192, 8192, 8192 * 2))
    zs = [t.zeros(3, 0, dtype=t.long) for _ in range(3)]
    zs = S.sample_level(zs, labels[2], dict(sk[2]), 2, priors[2], 2048, int(0.125 * priors[2].n_ctx), hps)
    assert t.equal(zs[2], f(2, t.arange(3), t.arange(2048)))


def test_upsample_mode_keeps_top_level():
    n, top = 3, 8192
    priors, hps, labels, sk = make_setup(n, top)
    ids = t.arange(n)
    zs = [t.zeros(n, 0, dtype=t.long), t.zeros(n, 0, dtype=t.long), f(2, ids, t.arange(top))]
    zs = S.upsample(zs, labels, sk, priors, hps, save=False, device="cpu")
    check_levels(zs, n, top)
    assert priors[2].calls == []


def test_sharded_sampler_two_gloo_ranks(tmp_path):
    """n_samples = 5 over 2 ranks (3 + 2): every rank samples its slice, codes are all-gathered per level."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import sys, torch as t
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        from jukebox_amd.utils.dist_utils import setup_dist_from_env
        from jukebox_amd.utils import dist_adapter as dist
        from jukebox_amd import sample as S
        import test_sample_windows as T
        rank, _, _ = setup_dist_from_env("gloo")
        priors, hps, labels, sk = T.make_setup(5, 8192 + 1024)
        if rank != 0:
            for lab in labels: lab["y"] = None            # only rank 0 knows the labels
        labels, _ = S.broadcast_conditioning(labels)
        zs = S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
        T.check_levels(zs, 5, 8192 + 1024)
        lo = 0 if rank == 0 else 3
        assert all(c for c in priors[0].calls)
        print("rank", rank, "ok", len(priors[0].calls))
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


class _DummyLabeller:
    def get_batch_labels(self, metas, device="cpu"):
        n = len(metas)
        y = t.zeros((n, 5), dtype=t.long)
        y[:, 0] = t.tensor([m["total_length"] for m in metas])
        y[:, 1] = t.tensor([m["offset"] for m in metas])
        y[:, 3] = t.arange(n)
        return dict(y=y.to(device), info=[dict(full_tokens=[]) for _ in metas])


@pytest.mark.parametrize("mode", ["ancestral", "continue", "upsample", "primed"])
def test_save_samples_modes(mode, tmp_path, monkeypatch):
    """save_samples / run plumbing (sample.py:178-271): preset metas tiled to the batch, mode dispatch, prompt-length
    rounding to the top level's hop, codes file and prompt audio loading -- with the fake priors standing in for models."""
    import numpy as np
    from scipy.io import wavfile
    from jukebox_amd import make_models
    n, top = 3, 8192 + 1024
    priors, hps, labels, sk = make_setup(n, top)
    for p in priors:
        p.labeller = _DummyLabeller()
        p.encode = lambda x, start_level=0, end_level=3, bs_chunks=1: [
            f(l, t.arange(x.shape[0]), t.arange(x.shape[1] // (8 * 4 ** l))) for l in range(start_level, end_level)]
    monkeypatch.setattr(S, "default_sampling_kwargs", lambda model: [dict(max_batch_size=3) for _ in range(3)])

    def fake_make_model(model, device, h, levels=None):
        h.sample_length = top * 128
        return None, priors
    monkeypatch.setattr(make_models, "make_model", fake_make_model)
    hps = Hyperparams(n_samples=n, sr=44100, total_sample_length_in_seconds=600, hop_fraction=[0.5, 0.5, 0.125],
                      name=str(tmp_path / "out"), levels=3)
    prompt_s = 2000 * 128 / 44100 + 0.001                 # rounds down to 2000 top-level tokens
    ids = t.arange(n)
    sample_hps = Hyperparams(mode=mode, prompt_length_in_seconds=None, codes_file=None, audio_file=None)
    if mode in ("continue", "upsample"):
        given = 3000 if mode == "continue" else top
        zs0 = [f(l, ids, t.arange(given * 4 ** (2 - l))) for l in range(3)]
        if mode == "upsample":
            zs0 = [zs0[0][:, :0], zs0[1][:, :0], zs0[2]]      # only the top level exists
        t.save(dict(zs=zs0), tmp_path / "codes.pth.tar")
        sample_hps.codes_file = str(tmp_path / "codes.pth.tar")
        if mode == "continue":
            sample_hps.prompt_length_in_seconds = prompt_s
    if mode == "primed":
        rng = np.random.default_rng(0)
        wav = (rng.uniform(-0.5, 0.5, (2000 * 128 + 4000, 2)) * 32767).astype(np.int16)
        for i in range(2):
            wavfile.write(tmp_path / f"p{i}.wav", 44100, wav)
        sample_hps.audio_file = f"{tmp_path}/p0.wav,{tmp_path}/p1.wav"
        sample_hps.prompt_length_in_seconds = prompt_s
        x = S.load_prompts(sample_hps.audio_file.split(","), 2000 * 128, hps, device="cpu")
        assert tuple(x.shape) == (n, 2000 * 128, 1) and x.dtype == t.float32
        assert np.allclose(x[0, :, 0].numpy(), wav[:2000 * 128].astype(np.float32).mean(1) / 32768.0, atol=1e-7)
        assert t.equal(x[0], x[2])                        # files repeat to fill the batch
    zs = S.save_samples("1b_lyrics", "cpu", hps, sample_hps)
    check_levels(zs, n, top)
    if mode == "continue":                                # the codes were cut to the prompt length before continuing
        assert min(c[1] for c in priors[2].calls) == 2000
    if mode == "upsample":
        assert priors[2].calls == []
    with pytest.raises(ValueError, match="Unknown sample mode"):
        S.save_samples("1b_lyrics", "cpu", hps, Hyperparams(mode="nope"))


def test_command_line_arguments_parse_like_fire():
    """`python -m jukebox_amd.sample --model=... --hop_fraction=0.5,0.5,0.125` (the reference's fire.Fire(run) command line)."""
    kw = S._argv_kwargs(["--model=5b_lyrics", "--name=sample_5b", "--levels=3", "--sample_length_in_seconds=20",
                         "--total_sample_length_in_seconds", "180", "--sr=44100", "--n_samples=6",
                         "--hop_fraction=0.5,0.5,0.125", "--mode=primed", "--audio_file=a.wav,b.wav",
                         "--prompt_length_in_seconds=12.5"])
    assert kw == dict(model="5b_lyrics", name="sample_5b", levels=3, sample_length_in_seconds=20,
                      total_sample_length_in_seconds=180, sr=44100, n_samples=6, hop_fraction=(0.5, 0.5, 0.125),
                      mode="primed", audio_file="a.wav,b.wav", prompt_length_in_seconds=12.5)
    with pytest.raises(AssertionError):
        S._argv_kwargs(["model=1b_lyrics"])


def test_seed_and_absolute_positions_reach_the_sampler():
    """hps.seed is handed to every prior.sample call together with the window's start (pos_base) and the sub-batch's
    global sample offset (sample_base): the draw of (level, sample, absolute position) does not depend on the window
    schedule or the batch split.  Without hps.seed a fresh seed is drawn once per job (all levels share it)."""
    priors, hps, labels, sk = make_setup(n_samples=5, top_tokens=8192 + 1024)
    hps.seed = 1234
    S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
    for level, p in enumerate(priors):
        assert p.kwargs_seen and all(k["seed"] == 1234 for k in p.kwargs_seen)
        starts = [c[0] for c in p.calls]
        assert [k["pos_base"] for k in p.kwargs_seen] == starts
        bs = sk[level]["max_batch_size"]
        assert sorted({k["sample_base"] for k in p.kwargs_seen}) == list(range(0, 5, bs))
    priors, hps, labels, sk = make_setup(n_samples=2, top_tokens=8192)
    S.ancestral_sample(labels, sk, priors, hps, save=False, device="cpu")
    seeds = {k["seed"] for p in priors for k in p.kwargs_seen}
    assert len(seeds) == 1 and isinstance(next(iter(seeds)), int)
    priors2, hps2, labels2, sk2 = make_setup(n_samples=2, top_tokens=8192)
    S.ancestral_sample(labels2, sk2, priors2, hps2, save=False, device="cpu")
    assert {k["seed"] for p in priors2 for k in p.kwargs_seen} != seeds          # unseeded jobs differ, like the reference
