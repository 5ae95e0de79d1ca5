"""Where a level-0 window's non-decode time goes (a window that runs alone is 4096 x the decode step
+ ~0.5 s, of which the conditioner and the prefill micro-benchmarks explain 0.37 s).

One process: the level-0 upsampler of 1b_lyrics with seeded random weights, 16 samples; windows of 4096 primed tokens +
`--steps` decode steps through the sampler's own entry point (sample_single_window -> SimplePrior.sample ->
ConditionalAutoregressive2D.primed_sample), (a) untouched: wall time per window; (b) with a device synchronisation and a
host timer around every stage: what each stage costs on its own.  glue = (a) - sum of the GPU stages of (b).

Usage: python -u tools/window_glue.py [--steps 512] [--windows 3] [--plain]"""
import argparse
import collections
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jukebox_amd import sample as S  # noqa: E402
from jukebox_amd.hparams import Hyperparams, setup_hparams  # noqa: E402
from jukebox_amd.make_models import MODELS, make_prior, make_vqvae  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=512)
ap.add_argument("--windows", type=int, default=3)
ap.add_argument("--plain", action="store_true", help="plain launch chain instead of pipelined launches")
a = ap.parse_args()

dev = torch.device("cuda:0")
torch.manual_seed(0)
with torch.device(dev):
    vq = make_vqvae(setup_hparams(MODELS["1b_lyrics"][0], dict(sample_length=881920, restore_vqvae="")), dev)
    prior = make_prior(setup_hparams(MODELS["1b_lyrics"][1], dict(restore_prior="")), vq, dev)
N, n_ctx = 16, prior.n_ctx
rng = np.random.RandomState(0)
yb = prior.y_emb.bow_genre_emb.bins, prior.y_emb.artist_emb.bins
items = [dict(artist_id=int(rng.randint(1, yb[1])), genre_ids=[int(rng.randint(1, yb[0]))], full_tokens=[],
              total_length=180 * 44100, offset=0) for _ in range(N)]
labels = prior.labeller.get_batch_labels_from_ids(items, dev)
hps = Hyperparams(n_samples=N, keep_priors_resident=True)
kw = dict(temp=0.99, fp16=True, chunk_size=32, max_batch_size=16, seed=0, sample_tokens=n_ctx // 2 + a.steps)
ar = prior.prior
ar.pipeline_launches = not a.plain

stages = collections.OrderedDict()
sync_stages = False


def timed(name, fn):
    def wrapper(*args, **kwargs):
        if not sync_stages:
            return fn(*args, **kwargs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*args, **kwargs)
        torch.cuda.synchronize()
        stages.setdefault(name, []).append(time.perf_counter() - t0)
        return out
    return wrapper


# the stages of a window, outermost first (a stage's time contains the stages nested in it)
prior.get_z_conds = timed("get_z_conds", prior.get_z_conds)
prior.get_y = timed("get_y", prior.get_y)
prior.y_emb.forward = timed("get_cond: y_emb", prior.y_emb.forward)
prior.x_emb = timed("get_cond: x_emb (conditioner)", prior.x_emb)
ar.preprocess = timed("preprocess", ar.preprocess)
ar.postprocess = timed("postprocess (range asserts)", ar.postprocess)
_engine = ar.engine


def engine(*args, **kwargs):
    eng = _engine(*args, **kwargs)
    if not getattr(eng, "_glue_wrapped", False):
        eng._glue_wrapped = True
        eng._pipe_verdict = not a.plain                     # no in-situ comparison inside the timed windows
        eng.set_cond = timed("engine.set_cond (copy of x_cond)", eng.set_cond)
        eng.set_sampling = timed("engine.set_sampling", eng.set_sampling)
        eng.prefill = timed("engine.prefill", eng.prefill)
        eng.decode = timed("engine.decode", eng.decode)
    return eng


ar.engine = timed("engine lookup", engine)


def window(i):
    zs = [torch.randint(0, 2048, (N, n_ctx // 2), device=dev), torch.randint(0, 2048, (N, n_ctx), device=dev)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = S.sample_single_window(zs, labels, dict(kw), 0, prior, 0, hps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert out[0].shape[1] == n_ctx // 2 + a.steps
    return dt


window(-1)                                                  # engine creation, graph capture, pair of streams
print(f"launch form: {'pipelined' if ar.bound_engine().pipelined else 'plain chain'}; {a.steps} decode steps per window")
free = [window(i) for i in range(a.windows)]
print("window, untouched          :", " ".join(f"{x * 1e3:8.1f}" for x in free), "ms")
sync_stages = True
synced = [window(i) for i in range(a.windows)]
print("window, stage-synchronised :", " ".join(f"{x * 1e3:8.1f}" for x in synced), "ms")
for name, v in stages.items():
    per_window = len(v) // a.windows
    tot = [sum(v[w * per_window:(w + 1) * per_window]) for w in range(a.windows)]
    print(f"  {name:38s}", " ".join(f"{x * 1e3:8.2f}" for x in tot), "ms")
dec = np.mean([sum(stages["engine.decode"][w:w + 1]) for w in range(a.windows)])
gpu = dec + np.mean(stages["engine.prefill"]) + np.mean(stages["get_cond: x_emb (conditioner)"])
print(f"decode per step {dec / a.steps * 1e3:.4f} ms; conditioner + prefill + decode {gpu * 1e3:.1f} ms; "
      f"untouched window {np.mean(free) * 1e3:.1f} ms -> everything else {(np.mean(free) - gpu) * 1e3:.1f} ms per window")
