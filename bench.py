#!/usr/bin/env python3
"""Benchmark of the BASELINE.json metric: generated audio-seconds per wall-second for a 3-level ancestral sample
of `1b_lyrics` (top prior + 2 upsamplers + VQ-VAE decode), 20 s of audio, 16 samples per GPU, synthetic labels and
seeded random-init weights of the released architecture (no checkpoints are reachable offline).

    python bench.py --gpus 1 --steps 1 --warmup 0
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full pass of the hot path over one batch: all three levels for 16 samples per GPU (n_samples is sharded
over the ranks, weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from jukebox_amd import sample as S  # noqa: E402
from jukebox_amd.hparams import Hyperparams, setup_hparams  # noqa: E402
from jukebox_amd.make_models import MODELS, make_prior, make_vqvae  # noqa: E402
from jukebox_amd.utils import dist_adapter as dist  # noqa: E402
from jukebox_amd.utils.dist_utils import setup_dist_from_env  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)
try:
    BASELINE_METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except (OSError, KeyError, ValueError):
    BASELINE_METRIC = "generated audio sec/sec (3-level ancestral sample), 1b_lyrics at 1/2/4/8 MI355X"

TINY = dict(
    vqvae=dict(levels=3, downs_t=(2, 2, 2), strides_t=(2, 2, 2), emb_width=16, l_bins=64, hvqvae_multipliers=(2, 1, 1),
               width=8, depth=2, m_conv=1.0, dilation_growth_rate=3, sr=22050),
    up=dict(n_ctx=128, prior_width=64, prior_depth=6, heads=1, attn_order=2, blocks=8, cond_width=32, cond_depth=4,
            cond_dilation_growth_rate=3, cond_dilation_cycle=2, cond_c_res=1, labels=True, y_bins=(10, 100), t_bins=128,
            max_bow_genre_size=5, min_duration=0.1, max_duration=600.0, sr=22050),
    top=dict(level=2, n_ctx=48, prior_width=64, prior_depth=16, heads=2, attn_order=12, blocks=8, labels=True,
             labels_v3=True, use_tokens=True, n_tokens=16, n_vocab=79, single_enc_dec=True, y_bins=(10, 100), t_bins=64,
             max_bow_genre_size=1, min_duration=0.1, max_duration=600.0, sr=22050),
)


def build_models(model, sample_length, device):
    """Seeded random-init models of the named architecture, constructed directly in HBM."""
    torch.manual_seed(0)
    with torch.device(device):
        if model == "tiny":
            vq = make_vqvae(setup_hparams("teeny", dict(sample_length=sample_length, **TINY["vqvae"])), device)
            hs = [setup_hparams("teeny", dict(level=0, **TINY["up"])),
                  setup_hparams("teeny", dict(level=1, cond_res_scale=True, **TINY["up"])),
                  setup_hparams("teeny", TINY["top"])]
        else:
            names = MODELS[model]
            vq = make_vqvae(setup_hparams(names[0], dict(sample_length=sample_length, restore_vqvae="")), device)
            hs = [setup_hparams(n, dict(restore_prior="")) for n in names[1:]]
        for blk in vq.bottleneck.level_blocks:
            blk.k.normal_()                       # the default all-zero codebook would make every code identical
        priors = [make_prior(h, vq, device) for h in hs]
    return vq, priors


def synthetic_labels(priors, n_samples, total_length, device):
    """Label matrices as Labeller.get_batch_labels would produce, from synthetic ids (data/labels.py:76-87)."""
    rng = np.random.RandomState(0)
    labels = []
    for p in priors:
        yb = p.y_emb.bow_genre_emb.bins, p.y_emb.artist_emb.bins
        items = [dict(artist_id=int(rng.randint(1, yb[1])), genre_ids=[int(rng.randint(1, yb[0]))],
                      full_tokens=rng.randint(1, 79, size=1500).tolist() if p.n_tokens > 0 else [],
                      total_length=total_length, offset=0) for _ in range(n_samples)]
        labels.append(p.labeller.get_batch_labels_from_ids(items, device))
    return labels


def cpu_baseline(prior, n_batch, steps, total_decode_steps, audio_seconds):
    """CPU port of the decode step (oracle/torch_port.py: the oracle's algorithm on the torch CPU kernels the reference
    itself would run on; pinned to the numpy oracle by tests/test_oracle_golden.py) timed on this box's host cores on a
    bounded sample: `steps` decode steps of the level-0 upsampler's transformer at batch n_batch, fp32.  The thread count
    is chosen by a short sweep (16-row matmuls do not scale to hundreds of threads).  Calibration against the unmodified
    reference in the build container (8 cores, tests/golden/time_reference_cpu.py): reference 182 ms, this port 131 ms,
    numpy oracle 469 ms per step -- the port is the faster (conservative) stand-in."""
    from oracle.torch_port import TorchDecodeStack
    ar = prior.prior
    sd = {k: v.detach().float().cpu().numpy() for k, v in ar.transformer.state_dict().items()}
    tr = TorchDecodeStack(sd, "", n_in=ar.width, n_ctx=ar.input_dims, n_head=ar.heads, n_depth=ar.depth,
                          attn_order=ar.attn_order, blocks=ar.blocks, m_attn=ar.m_attn, n_batch=n_batch)
    x = np.random.default_rng(0).standard_normal((n_batch, 1, ar.width)).astype(np.float32)
    before = torch.get_num_threads()
    tr.forward(x)                                    # warm-up step (t = 0)
    best, threads = None, 1
    for nt in (8, 16, 32, 64, 128, 256):
        if nt > (os.cpu_count() or 1):
            break
        torch.set_num_threads(nt)
        tr.forward(x)
        t0 = time.perf_counter()
        tr.forward(x)
        dt1 = time.perf_counter() - t0
        if best is None or dt1 < best:
            best, threads = dt1, nt
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.forward(x)
    sec_per_step = (time.perf_counter() - t0) / steps
    torch.set_num_threads(before)
    value = audio_seconds / (sec_per_step * total_decode_steps)
    return dict(value=value, unit="audio_s/s", cores=int(threads), kind="port",
                sample=f"{steps} consecutive decode steps of the level-0 upsampler transformer (early positions) at batch "
                       f"{n_batch}, torch-CPU fp32 port of the oracle (oracle/torch_port.py): {sec_per_step * 1e3:.1f} ms/step, "
                       f"extrapolated over the {total_decode_steps} decode steps of the workload (prefill, conditioner and "
                       "VQ-VAE conv stacks are not charged to the CPU)")


def projection_roofline(eng, t0, n_steps):
    """Dominant kernel, timed in situ with HIP events on the launch stream: the LayerNorm-fused weight-streaming
    projections (attn.c_attn / mlp.c_fc) of an engine's decode step."""
    us, launches, abytes = eng.probe_projection(t0, n_steps)
    achieved = abytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
    # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), collected in
    # their own rocprofv3 runs on the same kernel/shapes (profiles/r01_pmc_dominant_kernel.json); null otherwise
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_dominant_kernel.json")))
        if int(pmc["algorithmic_bytes_per_launch"]) == int(abytes):
            traffic = int(pmc["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return dict(bound="hbm", kernel="gemv_lnf_kernel<f16> (LayerNorm-folded attn.c_attn / mlp.c_fc of the decode step)",
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=traffic, avg_launch_us=round(us, 3), launches_timed=launches, bytes_per_launch=int(abytes))


def roofline_only(a, device):
    """Level-0 upsampler only: 256 graph-replayed decode steps at t = 4096.. plus the in-situ probe.  This is the
    command whose rocprofv3 kernel-trace summary is committed under profiles/."""
    assert a.model != "tiny"
    torch.manual_seed(0)
    with torch.device(device):
        vq = make_vqvae(setup_hparams(MODELS[a.model][0], dict(sample_length=262144, restore_vqvae="")), device)
        prior = make_prior(setup_hparams(MODELS[a.model][1], dict(restore_prior="")), vq, device)
    N = a.samples_per_gpu
    ar = prior.prior
    eng = ar.engine(N, True)
    eng.set_cond(torch.randn(N, ar.input_dims, ar.width, device=device) * 0.01, torch.randn(N, 1, ar.width, device=device) * 0.01)
    eng.set_sampling(temp=0.99, seed=1)
    eng.decode(4096, 8)
    torch.cuda.synchronize()
    ts = time.perf_counter()
    eng.decode(4096, 256)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - ts) / 256 * 1e3
    out = dict(roofline=projection_roofline(eng, 4096, 16), level0_decode_ms_per_token_step=round(ms, 4),
               launches_per_token_step=eng.launches_per_step, weights_gb=round(eng.weight_bytes() / 1e9, 3),
               kv_cache_gb=round(eng.cache_bytes() / 1e9, 3))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--model", default="1b_lyrics")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--samples-per-gpu", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the levels strictly one after the other (the reference's order) instead of pipelining them")
    ap.add_argument("--roofline-only", action="store_true",
                    help="build only the level-0 upsampler, run a short decode burst and print the roofline block "
                         "(the command profiled under profiles/)")
    a = ap.parse_args()

    rank, local_rank, device = setup_dist_from_env()
    world = dist.get_world_size()
    assert world == a.gpus or world == 1, f"launched with {world} ranks but --gpus {a.gpus}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"

    if a.roofline_only:
        return roofline_only(a, device)
    tiny = a.model == "tiny"
    sr = 22050 if tiny else 44100
    hop = 64 if tiny else 128
    sample_length = int(a.seconds * sr) // hop * hop if not tiny else 4608
    vq, priors = build_models(a.model, sample_length, device)
    n_samples = a.samples_per_gpu * world
    hps = Hyperparams(n_samples=n_samples, sample_length=sample_length, hop_fraction=[0.5, 0.5, 0.125], sr=sr, name="bench",
                      keep_priors_resident=True, pipeline_levels=not a.no_pipeline)
    labels = synthetic_labels(priors, n_samples, 180 * sr if not tiny else 3 * 4608, device)
    sk = S.default_sampling_kwargs(a.model if not tiny else "1b_lyrics")
    audio_seconds_per_step = n_samples * sample_length / sr

    level_t, level_t0 = {}, {}

    def mark(level):
        torch.cuda.current_stream(device).synchronize()     # the level's own stream is already drained by the sampler
        level_t.setdefault(level, []).append(time.perf_counter())

    def mark_start(level):
        level_t0.setdefault(level, []).append(time.perf_counter())
    S._sample.level_done, S._sample.level_start = mark, mark_start

    def one_step():
        return S.ancestral_sample(labels, sk, priors, hps, save=False, device=device)

    # Warm-up passes are untimed: they run the same three-level job on 6 s of audio (every kernel, graph capture and
    # allocation of the timed step, 1/3 of its length) so that `--warmup W` does not cost W x 5 minutes.
    warm_len = min(sample_length, int(6.0 * sr) // hop * hop) if not tiny else sample_length
    warm_hps = Hyperparams(hps)
    warm_hps.sample_length = warm_len
    for _ in range(a.warmup):
        S.ancestral_sample(labels, sk, priors, warm_hps, save=False, device=device)
    level_t.clear()
    level_t0.clear()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        zs = one_step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if torch.distributed.get_backend() == "gloo" else device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    assert all(int(z.shape[1]) == sample_length // p.raw_to_tokens for z, p in zip(zs, priors))

    if rank != 0:
        return
    value = audio_seconds_per_step * a.steps / dt
    # per-level wall time of the last step
    breakdown = {"levels_pipelined": not a.no_pipeline}
    if a.steps == 1:
        for l in (2, 1, 0):          # seconds from the start of the step until level l had produced all its codes
            breakdown[f"level{l}_codes_done_at_s"] = round(level_t[l][-1] - t0, 3)
            if l in level_t0:        # tokens per second of the level while it was running (levels overlap when pipelined)
                busy = max(level_t[l][-1] - level_t0[l][-1], 1e-9)
                breakdown[f"level{l}_tokens_per_s"] = round(n_samples // world * (sample_length // priors[l].raw_to_tokens) / busy, 1)

    # dominant kernel, timed in situ with HIP events on the launch stream: the LayerNorm-fused weight-streaming
    # projections of the level-0 upsampler's decode step
    eng = next(iter(priors[0].prior._engines.values()))
    roofline = projection_roofline(eng, 4096 if not tiny else 64, 16 if not tiny else 8)
    # one decode step (graph replay) of the same engine
    torch.cuda.synchronize()
    ts = time.perf_counter()
    n_probe = 64 if not tiny else 16
    eng.decode(4096 if not tiny else 64, n_probe)
    torch.cuda.synchronize()
    breakdown["level0_decode_ms_per_token_step"] = round((time.perf_counter() - ts) / n_probe * 1e3, 4)
    breakdown["launches_per_token_step"] = eng.launches_per_step

    out = dict(metric=BASELINE_METRIC, value=round(value, 4), unit="audio_s/s",
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 1),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f16", data="synthetic",
               config=dict(workload=f"{a.model} full 3-level ancestral sample (top prior + 2 upsamplers + VQ-VAE decode), "
                                    f"{sample_length / sr:.2f} s audio at {sr} Hz, n_samples={n_samples} "
                                    f"({a.samples_per_gpu}/GPU), temp=0.99, fp16 activations and weights, "
                                    "hop_fraction=(0.5,0.5,0.125), random-init weights",
                           samples_per_gpu=a.samples_per_gpu, parallelism=f"sample-sharded x{world}"),
               roofline=roofline, breakdown=breakdown)
    if world == 1 and not a.no_cpu_baseline:
        total_steps = sum(sample_length // p.raw_to_tokens for p in priors)
        out["cpu_baseline"] = cpu_baseline(priors[0], a.samples_per_gpu, a.cpu_steps, total_steps, audio_seconds_per_step)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
