"""Sampling driver with the reference's control flow (jukebox/sample.py:17-279): windows and hops per level,
batch splitting, ancestral / continue / upsample / primed modes, per-level decode and `data.pth.tar` dumps.

Extensions for MI355X nodes: `n_samples` is sharded across the ranks of one node (one process per GPU); rank 0
broadcasts the label matrices (and primed codes) over RCCL, every rank samples its contiguous slice with per-sample
random streams, and the generated codes are all-gathered once per level.  The reference runs
the whole job redundantly on every rank with identical seeds (sample.py:110-113)."""
import contextlib
import os
import time

import torch as t

from .data.labels import EmptyLabeller
from .hparams import Hyperparams
from .utils import dist_adapter as dist
from .utils.dist_utils import broadcast_tensor, gather_shards, print_once, shard_range
from .utils.sample_utils import get_starts, split_batch
from .utils.torch_utils import empty_cache


def sample_partial_window(zs, labels, sampling_kwargs, level, prior, tokens_to_sample, hps):
    """sample.py:17-29."""
    z = zs[level]
    n_ctx = prior.n_ctx
    current_tokens = z.shape[1]
    if current_tokens < n_ctx - tokens_to_sample:
        sampling_kwargs["sample_tokens"] = current_tokens + tokens_to_sample
        start = 0
    else:
        sampling_kwargs["sample_tokens"] = n_ctx
        start = current_tokens - n_ctx + tokens_to_sample
    return sample_single_window(zs, labels, sampling_kwargs, level, prior, start, hps)


def sample_single_window(zs, labels, sampling_kwargs, level, prior, start, hps):
    """sample.py:31-78."""
    n_samples = hps.n_samples
    n_ctx = prior.n_ctx
    end = start + n_ctx
    z = zs[level][:, start:end]
    sample_tokens = sampling_kwargs.get("sample_tokens", end - start)
    conditioning_tokens, new_tokens = z.shape[1], sample_tokens - z.shape[1]
    print_once(f"Sampling {sample_tokens} tokens for [{start},{start + sample_tokens}]. Conditioning on {conditioning_tokens} tokens")
    if new_tokens <= 0:
        return zs
    z_conds = prior.get_z_conds(zs, start, end)
    y = prior.get_y(labels, start)
    if not hps.get("keep_priors_resident", False):
        empty_cache()       # sample.py:47; releasing cached blocks synchronises the whole device, which would stall the
                            # other levels' streams in resident / pipelined mode (and 288 GB make it unnecessary)
    kwargs = dict(sampling_kwargs)
    max_batch_size = kwargs.pop("max_batch_size")
    sample_base = kwargs.pop("sample_base", 0)
    z_list = split_batch(z, n_samples, max_batch_size)
    z_conds_list = split_batch(z_conds, n_samples, max_batch_size)
    y_list = split_batch(y, n_samples, max_batch_size)
    z_samples, done = [], 0
    for z_i, z_conds_i, y_i in zip(z_list, z_conds_list, y_list):
        z_conds_i = None if z_conds_i is None else [zc.contiguous() for zc in z_conds_i]
        z_samples.append(prior.sample(n_samples=z_i.shape[0], z=z_i.contiguous(), z_conds=z_conds_i, y=y_i,
                                      sample_base=sample_base + done, pos_base=start, **kwargs))
        done += z_i.shape[0]
    z = t.cat(z_samples, dim=0)
    z_new = z[:, -new_tokens:]
    zs[level] = t.cat([zs[level], z_new], dim=1)
    if z.is_cuda:
        # the window's conditioner ran on the f16-split conv kernels: fail here, not a job later, if it was handed activations
        # outside the half range (a host-coherent flag: no wait, no queue touched)
        from . import hip_ops
        hip_ops.check_split_overflow(wait=False)
    return zs


def sample_level(zs, labels, sampling_kwargs, level, prior, total_length, hop_length, hps):
    """sample.py:81-88."""
    print_once(f"Sampling level {level}")
    if total_length >= prior.n_ctx:
        for start in get_starts(total_length, prior.n_ctx, hop_length):
            zs = sample_single_window(zs, labels, sampling_kwargs, level, prior, start, hps)
    else:
        zs = sample_partial_window(zs, labels, sampling_kwargs, level, prior, total_length, hps)
    return zs


def _level_plan(prior, zs_level_len, total_length, hop_length):
    """[(start, sample_tokens or None)] for one level -- the windows sample_level would visit."""
    if total_length >= prior.n_ctx:
        return [(start, None) for start in get_starts(total_length, prior.n_ctx, hop_length)]
    n_ctx, cur = prior.n_ctx, zs_level_len
    if cur < n_ctx - total_length:
        return [(0, cur + total_length)]
    return [(cur - n_ctx + total_length, n_ctx)]


def _sample_levels_pipelined(zs_local, labels, sampling_kwargs, priors, sample_levels, hps, local_hps, lo, hi, device):
    """Same windows, same tokens as the sequential level loop, scheduled as a pipeline: every level runs on its own
    HIP stream (driven by its own host thread) and starts a window as soon as the upper-level codes that window is
    conditioned on exist (prior.get_z_conds needs zs[level+1][start/cd : end/cd]).  The decode step is latency-bound
    and uses at most half of the CUs, so the three levels overlap almost for free.  Tokens are a pure function of
    (seed, level, global sample index, absolute position), so the schedule does not change the result."""
    import threading
    cond = threading.Condition()
    progress = {l: int(z.shape[1]) for l, z in enumerate(zs_local)}
    # Codes of every sampled level as they appear (written once, at absolute positions): a window publishes its tokens
    # every `pipeline_chunk` decode steps, so the level below starts on a PARTIAL upper window -- its first window needs
    # only the first n_ctx / cond_downsample upper codes, not the upper level's whole first window.
    chunk = int(hps.get("pipeline_chunk", 256))
    zbuf = {}
    for l in sample_levels:
        zbuf[l] = t.zeros((zs_local[l].shape[0], hps.sample_length // priors[l].raw_to_tokens), dtype=t.long, device=device)
        zbuf[l][:, :zs_local[l].shape[1]] = zs_local[l]
    ready_event = {}
    errors = []
    finished = set()                                       # levels whose codes are complete (their streams drained)
    levels = sorted(sample_levels, reverse=True)
    on_gpu = str(device).startswith("cuda")            # on CPU (host-logic tests) the schedule runs without streams
    current = torch_cuda_current_stream(device) if on_gpu else None
    # One stream per level.  Measured on MI355X / ROCm 7.2 (tools/bench_concurrent.py): two normal-priority streams can
    # land on one hardware queue and serialise completely, whereas torch's pooled high-priority streams plus one
    # normal-priority stream ran three decode chains concurrently at 1.3x the single-chain step time.  The two
    # upsampler levels (the long poles) therefore get high priority, the top level normal priority.
    # Tried and measured in round 2 (HISTORY.md section 5a): confining the levels' token loops to disjoint CU sets
    # (hipExtStreamCreateWithCUMask, tools/cu_mask_probe.py) -- the long pole runs faster while the others decode, but they
    # decode longer, and the job time does not move; prefilling the next window in a second engine while the current one
    # decodes -- the prefill's bandwidth slows the latency-bound chain by what it saves.  Neither is in the tree.
    order = sorted(levels)                                 # lowest level first
    lowest = min(sample_levels)
    prios = [-1, -1, 0, 0]      # (round 2: every assignment of priorities to the levels measured the same job time)
    stream_of = {level: t.cuda.Stream(device=device, priority=prios[min(i, 3)]) if on_gpu else None
                 for i, level in enumerate(order)}

    def new_event(stream=None):
        if not on_gpu:
            return None
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(device) if stream is None else stream)
        return ev

    def worker(level):
        try:
            prior = priors[level]
            stream = stream_of[level]
            cur = (lambda: t.cuda.current_stream(device)) if on_gpu else (lambda: None)
            if on_gpu:
                stream.wait_stream(current)
            with (t.cuda.stream(stream) if on_gpu else contextlib.nullcontext()):
                total_length = hps.sample_length // prior.raw_to_tokens
                hop_length = int(hps.hop_fraction[level] * prior.n_ctx)
                kw = dict(sampling_kwargs[level])
                kw["sample_base"] = lo
                lab = _shard_labels(labels[level], lo, hi)
                started = getattr(_sample, "level_start", None)
                if callable(started):
                    started(level)
                for start, sample_tokens in _level_plan(prior, zs_local[level].shape[1], total_length, hop_length):
                    if prior.x_cond and (level + 1) in sample_levels:
                        need = (start + prior.n_ctx) // prior.cond_downsample
                        with cond:
                            cond.wait_for(lambda: progress[level + 1] >= need or errors or (level + 1) in finished)
                            if errors:
                                return
                            # the upper level is complete and still too short for this window: the sequential loop fails
                            # the same way (prior.get_z_conds asserts the conditioning length, prior.py:158-166)
                            assert progress[level + 1] >= need, (
                                f"level {level}: the window at {start} is conditioned on {need} codes of level {level + 1}, which "
                                f"has only {progress[level + 1]} -- sample_length is shorter than this level's context")
                            ev = ready_event.get(level + 1)
                        if ev is not None:
                            cur().wait_event(ev)
                    k = dict(kw)
                    if sample_tokens is not None:
                        k["sample_tokens"] = sample_tokens
                    view = list(zs_local)
                    if (level + 1) in zbuf:
                        view[level + 1] = zbuf[level + 1]          # only [start/cd, end/cd) is read: published above
                    known = int(zs_local[level].shape[1])
                    # partial windows are published for the level BELOW; the lowest level sampled has no consumer and
                    # decodes a window in one call
                    tapped = local_hps.n_samples <= k["max_batch_size"] and chunk > 0 and (level - 1) in sample_levels

                    def publish(lo, hi, tok, start=start, known=known):
                        # window-relative music tokens [lo, hi), all new (the primed part is never decoded)
                        assert start + lo >= known
                        zbuf[level][:, start + lo:start + hi] = tok
                        ev = new_event()
                        with cond:
                            progress[level] = start + hi
                            ready_event[level] = ev
                            cond.notify_all()

                    prior.window_tap = (chunk, publish) if tapped else None
                    t_w = time.perf_counter()
                    try:
                        out = sample_single_window(view, lab, k, level, prior, start, local_hps)
                    finally:
                        prior.window_tap = None
                        timeline.append((level, start, round(t_w - t_job, 3), round(time.perf_counter() - t_job, 3)))
                    new_len = int(out[level].shape[1])
                    if not tapped:
                        zbuf[level][:, known:new_len] = out[level][:, known:new_len]
                    ev = new_event(cur())
                    with cond:
                        zs_local[level] = out[level]
                        progress[level] = new_len
                        ready_event[level] = ev
                        cond.notify_all()
                callback = getattr(_sample, "level_done", None)
                if on_gpu:
                    cur().synchronize()
                # this level's phase is over: its engines' pairs of streams go BEFORE the level counts as finished -- an idle pair
                # slows the other levels' plain chains, and the library admits one pipelined engine per process
                level_ar = getattr(prior, "prior", None)
                if level_ar is not None and callable(getattr(level_ar, "release_pipeline", None)):
                    level_ar.release_pipeline()
                with cond:
                    finished.add(level)
                    cond.notify_all()
                if callable(callback):
                    callback(level)
                # This level's audio (VQVAE.decode of its codes, sample.py:105) right away, on the level's stream, while the
                # levels below are still sampling -- the upper levels' decodes leave the end of the job.  (The codes of the
                # levels above are complete: this level has consumed all of them.)
                if on_gpu and hps.get("decode_audio_early", True):
                    with cond:
                        zs_now = list(zs_local)
                    early_audio[level] = prior.decode(zs_now[level:], start_level=level, bs_chunks=max(1, zs_now[level].shape[0]))
                    cur().synchronize()
        except BaseException as e:          # noqa: BLE001 -- re-raised in the caller
            with cond:
                errors.append(e)
                cond.notify_all()

    # Software-pipelined launches for the lowest level (the long pole of the job) ONLY WHILE IT RUNS ALONE.  A pipelined engine's
    # waiting launches keep compute units half occupied, and whatever runs next to them crawls -- measured with everything that
    # could be in the way removed (profiles/r05_pipe_concurrent_two_engines.log, upsampler geometry, ms per step): alone 1.53
    # pipelined / 1.77 plain; two plain chains side by side 2.09 each; a pipelined engine next to a plain chain 5.9 / 6.8.
    # (Round 5 also built TWO pipelined levels side by side -- a lean attention kernel, two owners in the library, regimes and
    # a rendezvous here: 2.02 ms per step each against 2.09, 67.53 s against 67.21 s for the default on the 6-second job; it
    # did not pay and was removed in round 6: HISTORY.md sections 4.5 / 4.6.)  The engine is asked before every window and every 512
    # steps of a window on the plain chain; its pair of streams is made when the launches are switched on and released when
    # they go off or the job ends: two more hardware queues in the process -- even idle ones -- slowed the concurrent levels'
    # plain chains 2.5x (profiles/r04_pipe_in_job.log).
    if _want_pipelined_launches(hps):
        cands = [l for l in sorted(sample_levels) if getattr(getattr(priors[l], "prior", None), "pipeline_candidate", False)][:1]
        for l in cands:
            priors[l].prior.pipeline_launches = lambda l=l: all(m in finished for m in sample_levels if m != l)
    early_audio = {}
    _sample_levels_pipelined.early_audio = early_audio
    # (level, window start, seconds into the job at which the window's sampling began / ended) per window: diagnostics
    timeline, t_job = [], time.perf_counter()
    _sample_levels_pipelined.timeline = timeline
    threads = [threading.Thread(target=worker, args=(l,), name=f"level{l}") for l in levels]
    try:
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if on_gpu:
            t.cuda.synchronize(device)
    finally:
        # nothing of the pipelined launches outlives the job: the next job's upper levels run plain chains again, and an idle
        # pair of streams in the process slows those (the reference's loop is re-entrant too: jukebox/sample.py:90-121)
        _release_pipelines(priors, sample_levels)
    if errors:
        raise errors[0]
    return zs_local


def torch_cuda_current_stream(device):
    return t.cuda.current_stream(device)


def _shard_labels(labels, lo, hi):
    return dict(y=labels["y"][lo:hi].contiguous(), info=labels["info"][lo:hi])


def _release_pipelines(priors, levels):
    """Every engine of these levels back on the plain chain, its pair of streams and graphs released (release_pipeline)."""
    for level in levels:
        ar = getattr(priors[level], "prior", None)
        if ar is not None:
            ar.pipeline_launches = None
            if callable(getattr(ar, "release_pipeline", None)):
                ar.release_pipeline()


def _want_pipelined_launches(hps):
    """Software-pipelined launches of the decode step (-24 % per token step for an engine that has the
    GPU to itself, bit-identical tokens) are the sampler's default for a level that runs alone; hps.pipeline_launches = False
    or JB_PIPELINE_LAUNCHES=0 keeps the plain launch chain."""
    if os.environ.get("JB_PIPELINE_LAUNCHES", "") == "0":
        return False
    return bool(hps.get("pipeline_launches", True))


def _sample(zs, labels, sampling_kwargs, priors, sample_levels, hps, save=True, device="cuda"):
    """sample.py:91-121 with n_samples sharded over the ranks.  `zs`, `labels` describe ALL hps.n_samples samples
    (identical on every rank -- see broadcast_conditioning); returns the full zs on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(hps.n_samples, rank, world)
    # One seed per job (hps.seed, `--seed` on the command line); without one rank 0 draws it from the OS, as the reference's
    # unseeded torch RNG gives different audio on every run.  The draw of (level, sample, absolute position) is a pure
    # function of it, whatever the sharding, the window schedule or the batch split.
    seed = hps.get("seed", None)
    if seed is None:
        seed = int.from_bytes(os.urandom(7), "little")
        if world > 1:
            seed = int(broadcast_tensor(t.tensor([seed], dtype=t.long, device=device), 0)[0].item())
    sampling_kwargs = [dict(kw, seed=int(kw.get("seed", seed))) for kw in sampling_kwargs]
    local_hps = Hyperparams(hps)
    local_hps.n_samples = hi - lo
    zs_local = [z[lo:hi].contiguous() for z in zs]
    xs = {}
    alignments = None
    pipelined = bool(hps.get("pipeline_levels", False)) and hps.get("keep_priors_resident", False) \
        and len(sample_levels) > 1 and local_hps.n_samples > 0
    # Software-pipelined launches (one engine per process at a time): the lowest level sampled has by far the most token
    # steps (x4 per level) and is the long pole of the job; when the levels run one after the other, each in its turn.
    ar = lambda p: getattr(p, "prior", None)               # the autoregressive model that owns the engine
    _release_pipelines(priors, sample_levels)              # a job starts with no pair of streams in the process
    for level in sample_levels:
        if ar(priors[level]) is not None:
            ar(priors[level]).pipeline_launches = False if pipelined else None   # (the level pipeline sets the lowest level's)
    if pipelined:
        for level in sample_levels:
            priors[level].to(device)
        zs_local = _sample_levels_pipelined(zs_local, labels, sampling_kwargs, priors, sample_levels, hps, local_hps, lo, hi,
                                            device)
    for level in reversed(sample_levels):
        prior = priors[level]
        prior.to(device)                         # sample.py:95 (prior.cuda()); binds the HIP engine lazily
        empty_cache()
        assert hps.sample_length % prior.raw_to_tokens == 0, \
            f"Expected sample_length {hps.sample_length} to be multiple of {prior.raw_to_tokens}"
        total_length = hps.sample_length // prior.raw_to_tokens
        hop_length = int(hps.hop_fraction[level] * prior.n_ctx)
        kw = dict(sampling_kwargs[level])
        kw["sample_base"] = lo
        if local_hps.n_samples > 0 and not pipelined:
            if callable(getattr(_sample, "level_start", None)):
                _sample.level_start(level)
            if ar(prior) is not None:
                ar(prior).pipeline_launches = _want_pipelined_launches(hps)     # one level at a time: the GPU is this level's
            try:
                zs_local = sample_level(zs_local, _shard_labels(labels[level], lo, hi), kw, level, prior, total_length,
                                        hop_length, local_hps)
            finally:
                if ar(prior) is not None:
                    ar(prior).pipeline_launches = False        # the next level's engine may take them over
                    ar(prior).release_pipeline()               # (split-batch tails included)
        if not hps.get("keep_priors_resident", False):
            prior.cpu()                          # sample.py:104: drops the engine's device copies
            empty_cache()
        zs[level] = gather_shards(zs_local[level], hps.n_samples)
        early = getattr(_sample_levels_pipelined, "early_audio", {}) if pipelined else {}
        x_local = early.pop(level) if level in early else \
            prior.decode(zs_local[level:], start_level=level, bs_chunks=max(1, zs_local[level].shape[0]))
        xs[level] = x_local
        _sample.level_done = getattr(_sample, "level_done", None)
        if callable(_sample.level_done) and not pipelined:
            _sample.level_done(level)
        if save:
            x = gather_shards(x_local, hps.n_samples)
            if rank == 0:
                logdir = f"{hps.name}/level_{level}"
                os.makedirs(logdir, exist_ok=True)
                t.save(dict(zs=zs, labels=labels, sampling_kwargs=sampling_kwargs, x=x), f"{logdir}/data.pth.tar")
                save_wav(logdir, x, hps.sr)
                top = priors[-1]
                if alignments is None and top is not None and top.n_tokens > 0 and \
                        not isinstance(top.labeller, EmptyLabeller) and getattr(top, "alignment_layer", None) is not None \
                        and len(zs[-1][0]) > 0:
                    # sample.py:118-120 + align.py:85-97
                    from .align import get_alignment
                    ahps = Hyperparams(levels=len(priors), hop_fraction=hps.hop_fraction)
                    alignments = get_alignment(x, zs, labels[-1], top, sampling_kwargs[-1]["fp16"], ahps, device=device)
                    if not hps.get("keep_priors_resident", False):
                        top.cpu()
                    t.save(dict(alignments=alignments), f"{logdir}/data_align.pth.tar")
                if "info" in labels[-1] and all("lyrics" in i for i in labels[-1]["info"]):
                    from .save_html import save_html                      # sample.py:120
                    save_html(logdir, x, zs, labels[-1], alignments, Hyperparams(levels=len(priors), sr=hps.sr))
    if t.device(device).type == "cuda":
        # the conv stacks (conditioners, VQ-VAE decoder) ran on the f16-split kernels: fail loudly if any of them was handed
        # activations outside the half range (a sticky device flag; one 4-byte read per job)
        from . import hip_ops
        hip_ops.check_split_overflow()
    _sample.last_audio = xs
    return zs


def save_wav(fname, aud, sr):
    """utils/audio_utils.py:142-146 (clamp to [-1, 1], one wav per item) via scipy (soundfile is not installed)."""
    from scipy.io import wavfile
    aud = t.clamp(aud, -1, 1).cpu().numpy()
    for i in range(aud.shape[0]):
        wavfile.write(f"{fname}/item_{i}.wav", sr, aud[i])


def broadcast_conditioning(labels, zs=None):
    """Rank 0's label matrices (and codes, for continue / upsample / primed modes) to every rank over RCCL."""
    for lab in labels:
        lab["y"] = broadcast_tensor(lab["y"], 0)
    if zs is not None:
        zs = [broadcast_tensor(z, 0) for z in zs]
    return labels, zs


def ancestral_sample(labels, sampling_kwargs, priors, hps, save=True, device="cuda"):
    """sample.py:124-128."""
    sample_levels = list(range(len(priors)))
    zs = [t.zeros(hps.n_samples, 0, dtype=t.long, device=device) for _ in range(len(priors))]
    return _sample(zs, labels, sampling_kwargs, priors, sample_levels, hps, save=save, device=device)


def continue_sample(zs, labels, sampling_kwargs, priors, hps, save=True, device="cuda"):
    """sample.py:131-134."""
    return _sample(zs, labels, sampling_kwargs, priors, list(range(len(priors))), hps, save=save, device=device)


def upsample(zs, labels, sampling_kwargs, priors, hps, save=True, device="cuda"):
    """sample.py:137-140."""
    return _sample(zs, labels, sampling_kwargs, priors, list(range(len(priors) - 1)), hps, save=save, device=device)


def primed_sample(x, labels, sampling_kwargs, priors, hps, save=True, device="cuda"):
    """sample.py:143-147."""
    zs = priors[-1].encode(x, start_level=0, end_level=len(priors), bs_chunks=x.shape[0])
    return _sample(zs, labels, sampling_kwargs, priors, list(range(len(priors))), hps, save=save, device=device)


def load_codes(codes_file, duration, priors, hps, device="cuda"):
    """sample.py:164-175: reload `zs` from a level's data.pth.tar, optionally truncated to `duration` samples."""
    data = t.load(codes_file, map_location="cpu", weights_only=False)
    zs = [z.to(device) for z in data["zs"]]
    assert zs[-1].shape[0] == hps.n_samples, f"Expected bs = {hps.n_samples}, got {zs[-1].shape[0]}"
    if duration is not None:
        top_raw_to_tokens = priors[-1].raw_to_tokens
        assert duration % top_raw_to_tokens == 0, f"Cut-off duration {duration} not an exact multiple of top_raw_to_tokens"
        assert duration // top_raw_to_tokens <= zs[-1].shape[1]
        zs = [z[:, :duration // prior.raw_to_tokens] for z, prior in zip(zs, priors)]
    return zs


def default_sampling_kwargs(model):
    """sample.py:231-241."""
    lower = dict(temp=0.99, fp16=True, chunk_size=32, max_batch_size=16)
    top = dict(temp=0.99, fp16=True, chunk_size=32, max_batch_size=16) if model == "1b_lyrics" else \
        dict(temp=0.99, fp16=True, chunk_size=16, max_batch_size=3)
    return [dict(lower), dict(lower), top]


def load_audio(path, sr, duration, offset=0.0, mono=True):
    """utils/io.py load_audio for the prompt files of `primed` mode: (channels, duration) float32 in [-1, 1].
    WAV only (scipy; the reference decodes any container with PyAV, which is not installed); a file at another sample
    rate is resampled with scipy's polyphase filter, which is close to, not identical with, the reference's resampler."""
    import numpy as np
    from scipy.io import wavfile
    file_sr, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":                         # 8-bit PCM is unsigned
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim == 1:
        data = data[:, None]
    if file_sr != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(file_sr))
        data = resample_poly(data, int(sr) // g, int(file_sr) // g, axis=0).astype(np.float32)
    start = int(offset * sr)
    data = data[start:start + duration]
    assert data.shape[0] == duration, f"{path}: {data.shape[0]} samples after offset, {duration} needed for the prompt"
    if mono:
        data = data.mean(axis=1, keepdims=True)
    return np.ascontiguousarray(data.T)


def load_prompts(audio_files, duration, hps, device="cuda"):
    """sample.py:150-162: `duration` samples of each file as (n_samples, duration, 1), files repeated to fill the batch."""
    xs = [load_audio(f, sr=hps.sr, duration=duration, offset=0.0, mono=True).T for f in audio_files]
    while len(xs) < hps.n_samples:
        xs.extend(xs)
    xs = xs[:hps.n_samples]
    return t.stack([t.from_numpy(x) for x in xs]).to(device, non_blocking=True)


# sample.py:192-222 presets (artist / genre per item); the lyric sheets of the reference (jukebox/lyricdict.py) are not
# redistributed here -- pass `metas` to save_samples for real lyrics.  Default text: P. B. Shelley, 1818 (public domain).
_DEFAULT_LYRICS = ("I met a traveller from an antique land,\nWho said: Two vast and trunkless legs of stone\n"
                   "Stand in the desert. Near them, on the sand,\nHalf sunk a shattered visage lies")
_PRESETS = (("Alan Jackson", "Country"), ("Joe Bonamassa", "Blues Rock"), ("Frank Sinatra", "Classic Pop"),
            ("Ella Fitzgerald", "Jazz"), ("Céline Dion", "Pop"))


def save_samples(model, device, hps, sample_hps, metas=None):
    """sample.py:178-262: build the models, label the batch, dispatch on sample_hps.mode
    (ancestral | continue | upsample | primed).  Returns the codes `zs` (the reference returns None)."""
    from .make_models import make_model
    print_once(str(hps))
    vqvae, priors = make_model(model, device, hps)
    assert hps.sample_length // priors[-2].raw_to_tokens >= priors[-2].n_ctx, \
        "Upsampling needs atleast one ctx in get_z_conds. Please choose a longer sample length"
    total_length = hps.total_sample_length_in_seconds * hps.sr
    if metas is None:
        metas = [dict(artist=a, genre=g, lyrics=_DEFAULT_LYRICS, total_length=total_length, offset=0) for a, g in _PRESETS]
    metas = list(metas)
    while len(metas) < hps.n_samples:
        metas.extend(metas)
    metas = metas[:hps.n_samples]
    labels = [prior.labeller.get_batch_labels(metas, device) for prior in priors]
    for label in labels:
        assert label["y"].shape[0] == hps.n_samples
    sampling_kwargs = default_sampling_kwargs(model)
    top_raw_to_tokens = priors[-1].raw_to_tokens
    prompt_s = sample_hps.get("prompt_length_in_seconds")
    mode = sample_hps.mode
    if mode == "ancestral":
        return ancestral_sample(labels, sampling_kwargs, priors, hps, device=device)
    if mode in ("continue", "upsample"):
        assert sample_hps.get("codes_file") is not None
        duration = None if prompt_s is None else (int(prompt_s * hps.sr) // top_raw_to_tokens) * top_raw_to_tokens
        zs = load_codes(sample_hps.codes_file, duration, priors, hps, device=device)
        fn = continue_sample if mode == "continue" else upsample
        return fn(zs, labels, sampling_kwargs, priors, hps, device=device)
    if mode == "primed":
        assert sample_hps.get("audio_file") is not None
        assert prompt_s is not None
        duration = (int(prompt_s * hps.sr) // top_raw_to_tokens) * top_raw_to_tokens
        x = load_prompts(sample_hps.audio_file.split(","), duration, hps, device=device)
        return primed_sample(x, labels, sampling_kwargs, priors, hps, device=device)
    raise ValueError(f"Unknown sample mode {mode}.")


def run(model, mode="ancestral", codes_file=None, audio_file=None, prompt_length_in_seconds=None, port=29500, **kwargs):
    """sample.py:264-271.  Ranks come from the torchrun environment (one process per GPU) instead of MPI; `port` is
    accepted for compatibility (the rendezvous port is torchrun's MASTER_PORT)."""
    from .utils.dist_utils import setup_dist_from_env
    rank, local_rank, device = setup_dist_from_env()
    hps = Hyperparams(**kwargs)
    sample_hps = Hyperparams(dict(mode=mode, codes_file=codes_file, audio_file=audio_file,
                                  prompt_length_in_seconds=prompt_length_in_seconds))
    with t.no_grad():
        return save_samples(model, device, hps, sample_hps)


def _argv_kwargs(argv):
    """`--key=value` / `--key value` pairs with Python-literal values (what fire.Fire(run) accepts for this script)."""
    import ast
    out, i = {}, 0
    while i < len(argv):
        assert argv[i].startswith("--"), f"expected --key=value, got {argv[i]}"
        if "=" in argv[i]:
            k, v = argv[i][2:].split("=", 1)
        else:
            k, v = argv[i][2:], argv[i + 1]
            i += 1
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
        i += 1
    return out


if __name__ == "__main__":
    import sys
    run(**_argv_kwargs(sys.argv[1:]))
