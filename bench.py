#!/usr/bin/env python3
"""Benchmark of the BASELINE.json metric: generated audio-seconds per wall-second for a 3-level ancestral sample
of `1b_lyrics` (top prior + 2 upsamplers + VQ-VAE decode), 20 s of audio, 16 samples per GPU, synthetic labels and
seeded random-init weights of the released architecture (no checkpoints are reachable offline).

    python bench.py --gpus N --steps K --warmup W          # N > 1: re-launches itself as N ranks under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full pass of the hot path over one batch: all three levels for 16 samples per GPU (n_samples is sharded
over the ranks, weak scaling) -- minutes of GPU time.  The run therefore works against a WALL BUDGET (JB_BENCH_BUDGET_S,
default 1400 s from process start, model build included): at most one short untimed warm-up pass, then full-length timed
steps while another one still fits, at least one and at most --steps.  The JSON line reports the steps actually timed
(`steps`, `ms_per_step`) next to the requested counts, so a driver command such as `--steps 20 --warmup 5` always
yields a line.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

T_PROC = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--model", default="1b_lyrics")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--samples-per-gpu", type=int, default=16)
    ap.add_argument("--max-batch-size", type=int, default=0,
                    help="rows per engine call (the reference's max_batch_size, sample.py:231-238: 16 -- a V100 limit); 0 = the "
                         "reference's value.  The headline stays at 16; larger values are the separate report SURVEY 8d asks for")
    ap.add_argument("--cpu-steps", type=int, default=32, help="decode steps of the fallback CPU port (kind 'port')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the levels strictly one after the other (the reference's order) instead of pipelining them")
    ap.add_argument("--roofline-only", action="store_true",
                    help="build only the level-0 upsampler, run a short decode burst and print the roofline block "
                         "(the command profiled under profiles/)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every step is a short sleep.  Exercises the launcher, the rank bootstrap (gloo on a "
                         "CPU-only host), the budgeted step loop and the JSON contract (tests/test_bench_contract.py)")
    return ap.parse_args(argv)


def _respawn_as_ranks(a, argv):
    """`python bench.py --gpus N` without an outer torchrun: become the launcher of N ranks (one per GPU)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, JB_BENCH_T0=repr(T_PROC))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


if __name__ == "__main__":
    _a = _parse_args()
    if _a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(_respawn_as_ranks(_a, sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from jukebox_amd import sample as S  # noqa: E402
from jukebox_amd.hparams import Hyperparams, setup_hparams  # noqa: E402
from jukebox_amd.make_models import MODELS, make_prior, make_vqvae  # noqa: E402
from jukebox_amd.utils import dist_adapter as dist  # noqa: E402
from jukebox_amd.utils.dist_utils import setup_dist_from_env, shard_range  # noqa: E402

_JSON_OUT = sys.stdout
BUDGET_S = float(os.environ.get("JB_BENCH_BUDGET_S", "1400"))
T_ORIGIN = float(os.environ.get("JB_BENCH_T0", repr(T_PROC)))


def budget_left():
    return BUDGET_S - (time.time() - T_ORIGIN)


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


def cpu_baseline_reference(seconds, n_batch, budget_s):
    """The UNMODIFIED reference on this box's host cores: oracle/time_reference.py in a subprocess (its import shim
    monkey-patches torch, so it cannot share a process with the GPU path) against the snapshot oracle/_ref made by
    oracle/make_ref.py.  Bounded: about 25 s of timed CPU work + the construction of the reference model.  Returns the
    cpu_baseline object (kind "reference") or None when the snapshot is absent or the leg fails."""
    script = os.path.join(ROOT, "oracle", "time_reference.py")
    if not os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "jukebox")) or not os.path.exists(script):
        return None
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, script, "--budget-s", str(budget_s), "--batch", str(n_batch), "--seconds", str(seconds)],
                           env=env, capture_output=True, text=True, timeout=max(60.0, 3.5 * budget_s))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        sys.stderr.write(f"cpu_baseline (reference) produced no JSON, rc={r.returncode}: {r.stderr[-400:]}\n")
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        sys.stderr.write(f"cpu_baseline (reference) failed: {e}\n")
    return None


def cpu_baseline_port(prior, n_batch, steps, total_decode_steps, audio_seconds):
    """Fallback when oracle/_ref is absent: CPU port of the decode step (oracle/torch_port.py: the oracle's algorithm on
    the torch CPU kernels, with preallocated k/v caches; pinned to the numpy oracle by tests/test_oracle_golden.py) timed
    on this box's host cores: `steps` decode steps of the level-0 upsampler's transformer at batch n_batch, fp32, early
    positions.  It is FASTER than the unmodified reference (no per-step cache re-copy: the reference's step grows from
    0.2 s at t=0 to 1.5 s at t=6144 on 8 cores, this port stays at ~0.13 s), so it flatters the CPU."""
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
                       "VQ-VAE conv stacks are not charged to the CPU; faster than the unmodified reference, see docstring)")


def projection_roofline(eng, t0, n_steps):
    """Dominant kernel, timed in situ with HIP events on the launch stream: the LayerNorm-fused weight-streaming
    projections (attn.c_attn / mlp.c_fc) of an engine's decode step."""
    us, launches, abytes = eng.probe_projection(t0, n_steps)
    achieved = abytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
    # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), collected in
    # their own rocprofv3 runs on the same kernel/shapes (profiles/r01_pmc_dominant_kernel.json); null otherwise
    traffic, source = None, None
    for name in ("r06_pmc_dominant_kernel_engine_form.json", "r05_pmc_dominant_kernel_wide.json", "r04_pmc_dominant_kernel_wide.json", "r03_pmc_dominant_kernel_wide.json", "r02_pmc_dominant_kernel.json",
                 "r01_pmc_dominant_kernel.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if int(pmc["algorithmic_bytes_per_launch"]) == int(abytes):
                traffic = int(pmc["traffic_bytes_per_launch"])
                source = f"profiles/{name} (separate rocprofv3 --pmc passes on the same kernel and shapes; NOT measured in this run)"
                break
        except (OSError, KeyError, ValueError):
            pass
    return dict(bound="hbm", kernel="gemv_lnf_kernel<f16> (LayerNorm-folded attn.c_attn / mlp.c_fc of the decode step)",
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=traffic, traffic_source=source, avg_launch_us=round(us, 3), launches_timed=launches,
                bytes_per_launch=int(abytes),
                timing="HIP events on the launch stream around a back-to-back burst of exactly these launches over all "
                       "layers (cold weights per launch, as in the real step)")


def engine_step_report(eng, t0, n_steps=256, pipelined=True):
    """ms per token step of one engine at position t0 (graph replay: plain chain, and software-pipelined launches where the
    engine has them), the algorithmic bytes of that step (jb_engine_step_bytes) and the fraction of the HBM peak."""
    eng.set_pipelined(False)
    eng.decode(t0, 8)
    plain = eng.timed_decode(t0, n_steps) * 1e3
    rep = dict(n_batch=eng.N, position=t0, steps_timed=n_steps, launches_per_step=eng.launches_per_step, ms_per_step_plain_chain=round(plain, 4))
    best = plain
    if pipelined and eng.set_pipelined(True):
        eng.decode(t0, 16)
        pipe = eng.timed_decode(t0, n_steps) * 1e3
        if not eng.pipe_error() and eng.pipelined:
            rep["ms_per_step_pipelined_launches"] = round(pipe, 4)
            best = min(best, pipe)
        eng.set_pipelined(False)
    gb = eng.step_bytes(t0 + n_steps // 2) / 1e9
    rep.update(ms_per_step=round(best, 4), algorithmic_gb_per_step=round(gb, 4), weights_gb=round(eng.weight_bytes() / 1e9, 3),
               frac_of_hbm_peak=round(gb / (best * 1e-3) / HBM_PEAK_GBS, 4))
    return rep


def other_configs(priors, n_batch, device):
    """BASELINE.json's configs 2, 3 and 5 as decode-step lines next to the headline (config 4): 256 graph-replayed token steps
    of each engine at the reference's batch sizes -- small_prior (hparams.py:210-220; N = 16, mid-window), the prior_1b_lyrics
    top prior of THIS job right behind its 384-token lyric prefill (hparams.py:165-188; N = 16), and the prior_5b_lyrics decoder
    (hparams.py:127-156: 4800 wide, 79 layers, 8 heads, attn_order 10, 512 encoder states; N = 3, random-init).  Outside the
    timed region; what tools/bench_engine.py prints, in the line the driver runs."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_engine import CFGS, random_state
    from jukebox_amd.engine import PriorEngine
    out = {}
    top = priors[2].prior.bound_engine()
    if top is not None and top.N == n_batch:
        top.prefill(0, priors[2].n_tokens)               # the lyric prefix of the job's last window is still in the token buffer
        out["config3_1b_lyrics_top_prior"] = engine_step_report(top, priors[2].n_tokens)
    for key, name, n, t0 in (("config2_small_prior", "small", 16, 4096), ("config5_5b_lyrics_top_prior", "5b", 3, 4096)):
        cfg = CFGS[name]
        eng = PriorEngine(random_state(cfg, device), "", n_batch=n, fp16=True, chunk_cap=64, **cfg)
        y = torch.randn(n, 1, cfg["width"], device=device) * 0.01 if cfg["y_cond"] else None
        eng.set_cond(torch.randn(n, cfg["seq_len"], cfg["width"], device=device) * 0.01, y)
        eng.set_sampling(temp=0.99, seed=1)
        if cfg.get("encoder_dims"):
            eng.set_encoder_kv(torch.randn(n, cfg["encoder_dims"], cfg["width"], device=device) * 0.1)
        out[key] = engine_step_report(eng, t0)
        eng.close()
        del eng
        torch.cuda.empty_cache()
    return out


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
    print(json.dumps(out), file=_JSON_OUT, flush=True)


def timed_steps(step_fn, requested, world, device, tail_reserve_s, sync):
    """Time full steps: barrier + device sync on both sides, MAX over ranks.  Every rank takes the same decision to go on
    (all-reduced), so the barrier count matches.  Returns (steps_timed, seconds, per_step_seconds, last_result)."""
    on_gpu = device.type == "cuda"

    def agree(flag):
        if world == 1:
            return flag
        tdev = device if torch.distributed.get_backend() != "gloo" else "cpu"
        f = torch.tensor([1 if flag else 0], dtype=torch.int32, device=tdev)
        torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MIN)
        return bool(int(f.item()))

    if world > 1:
        torch.distributed.barrier()
    sync()
    t0 = time.perf_counter()
    done, per_step, out = 0, [], None
    while True:
        ts = time.perf_counter()
        out = step_fn()
        sync()
        per_step.append(time.perf_counter() - ts)
        done += 1
        more = done < requested and budget_left() > 1.08 * max(per_step) + tail_reserve_s
        if not agree(more):
            break
    sync()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tdev = device if (on_gpu and torch.distributed.get_backend() != "gloo") else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    return done, dt, per_step, out


def dist_info(world, rank=0, device=None, per_step=None, n_samples_rank=None):
    """What a reviewer of a multi-GPU line needs: backend and its version, and per rank the device it ran on, the samples it
    held and its own step seconds (the line's `value` uses the MAX over ranks; the spread shows stragglers).  Collective:
    every rank calls it."""
    backend = torch.distributed.get_backend() if world > 1 else "none"
    mine = dict(rank=rank, host=os.uname().nodename, device=str(device), n_samples=n_samples_rank,
                step_seconds=[round(x, 3) for x in (per_step or [])])
    if device is not None and device.type == "cuda":
        pr = torch.cuda.get_device_properties(device)
        mine.update(device_name=pr.name, device_uuid=str(getattr(pr, "uuid", "")), hbm_gb=round(pr.total_memory / 2 ** 30, 1),
                    compute_units=pr.multi_processor_count)
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        torch.distributed.all_gather_object(ranks, mine)
    info = dict(world_size=world, backend=backend,
                launcher="torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ or world > 1 else "single process",
                ranks=ranks)
    if backend == "nccl":
        try:
            info["rccl_version"] = ".".join(map(str, torch.cuda.nccl.version()))
        except Exception as e:   # noqa: BLE001 -- version probing must never cost a bench line
            info["rccl_version"] = f"unknown ({e})"
    means = [sum(r["step_seconds"]) / len(r["step_seconds"]) for r in ranks if r and r["step_seconds"]]
    if means:
        info["rank_step_seconds_min_max"] = [round(min(means), 3), round(max(means), 3)]
    return info


def dry_run(a, rank, world, device):
    """No GPU work (see --dry-run): same launcher, bootstrap, budgeted loop and JSON layout, every step a sleep."""
    n_samples = a.samples_per_gpu * world
    audio = n_samples * a.seconds
    n_warm = 0
    if a.warmup > 0:                          # as the real run: ONE untimed pass whenever a warm-up is asked for, whatever the budget
        time.sleep(0.02)
        n_warm = 1
    done, dt, per_step, _ = timed_steps(lambda: time.sleep(0.02 * (1 + rank)), max(a.steps, 1), world, device, 0.0, lambda: None)
    lo, hi = shard_range(n_samples, rank, world)          # the sampler's own partition of the samples (sample.py)
    di = dist_info(world, rank, device, per_step, hi - lo)
    if rank != 0:
        return
    line = json.dumps(dict(metric=BASELINE_METRIC, value=round(audio * done / dt, 4), unit="audio_s/s", n_gpus=world, steps=done,
                          warmup=n_warm, steps_requested=a.steps, warmup_requested=a.warmup, ms_per_step=round(dt / done * 1e3, 1),
                          higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f16",
                          data="dry-run (no GPU work; launcher / rank bootstrap / step loop only)",
                          config=dict(workload="dry-run", samples_per_gpu=a.samples_per_gpu, n_samples=n_samples,
                                      parallelism=f"sample-sharded x{world}"),
                          dist=di, roofline=None, cpu_baseline=None))
    print(line, file=_JSON_OUT, flush=True)


def main():
    a = _parse_args()
    # stdout carries the ONE JSON line and nothing else: the samplers' progress messages go to stderr
    global _JSON_OUT
    _JSON_OUT, sys.stdout = sys.stdout, sys.stderr
    rank, local_rank, device = setup_dist_from_env()
    world = dist.get_world_size()
    assert world == a.gpus, f"running as {world} rank(s) but --gpus {a.gpus} (launch with torch.distributed.run or let bench.py spawn the ranks)"
    if a.dry_run:
        return dry_run(a, rank, world, device)
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
                      keep_priors_resident=True, pipeline_levels=not a.no_pipeline, seed=0)
    labels = synthetic_labels(priors, n_samples, 180 * sr if not tiny else 3 * 4608, device)
    sk = S.default_sampling_kwargs(a.model if not tiny else "1b_lyrics")
    if a.max_batch_size > 0:
        for k in sk:
            k["max_batch_size"] = a.max_batch_size
    audio_seconds_per_step = n_samples * sample_length / sr

    level_t, level_t0 = {}, {}

    def mark(level):
        torch.cuda.current_stream(device).synchronize()     # the level's own stream is already drained by the sampler
        level_t.setdefault(level, []).append(time.perf_counter())

    def mark_start(level):
        level_t0.setdefault(level, []).append(time.perf_counter())
    S._sample.level_done, S._sample.level_start = mark, mark_start

    step_t0 = []

    step_codes = []        # every timed step's codes (device tensors, a few MB each): digested AFTER the timed region

    def one_step():
        step_t0.append(time.perf_counter())
        zs = S.ancestral_sample(labels, sk, priors, hps, save=False, device=device)
        step_codes.append(zs)
        return zs

    # Warm-up (untimed): at most ONE pass of the same three-level job on 6 s of audio per sample -- every kernel, graph
    # capture and allocation of the timed step without its length (a full-length pass costs minutes).
    warm_len = min(sample_length, int(6.0 * sr) // hop * hop) if not tiny else sample_length
    warm_hps = Hyperparams(hps)
    warm_hps.sample_length = warm_len
    n_warm = 0
    if a.warmup > 0:
        # never dropped silently: a process's first job differs from every later one in what the earlier jobs left behind
        # (round 4: a pair of streams that survived the job made steps 2..n 13 % slower, and every builder-side line had
        # skipped the warm-up under a short budget)
        S.ancestral_sample(labels, sk, priors, warm_hps, save=False, device=device)
        n_warm = 1
    level_t.clear()
    level_t0.clear()
    cpu_leg_s = 0.0 if (world > 1 or a.no_cpu_baseline) else 120.0
    if world == 1 and not tiny:
        cpu_leg_s += 60.0                  # (the decode-step lines of BASELINE's other configs: other_configs)
    steps_done, dt, per_step, zs = timed_steps(one_step, max(a.steps, 1), world, device, 45.0 + cpu_leg_s,
                                               lambda: torch.cuda.synchronize())
    assert all(int(z.shape[1]) == sample_length // p.raw_to_tokens for z, p in zip(zs, priors))
    # The timed job's own tokens: every step samples the same job with the same seed, and the draw of (level, sample, position)
    # is a pure function of the seed -- whatever the launch form a window ran in, the moment the levels' threads met, the
    # in-situ comparison or a recovered time-out did to the schedule -- so every step must produce the SAME codes on every
    # level.  (tests/test_hip_models.py::test_timed_job_tokens holds the pipelined schedule to the sequential plain chain.)
    import hashlib
    step_digests = []
    for zs_i in step_codes:
        h = hashlib.sha256()
        for z in zs_i:
            h.update(z.to("cpu", torch.int64).contiguous().numpy().tobytes())
        step_digests.append(h.hexdigest())
    assert len(set(step_digests)) == 1, f"the timed steps did not produce the same codes: {step_digests}"
    del step_codes[:]
    di = dist_info(world, rank, device, per_step, a.samples_per_gpu)

    if rank != 0:
        return
    value = audio_seconds_per_step * steps_done / dt
    # per-level wall time of the LAST timed step
    t_last = step_t0[-1]
    breakdown = {"levels_pipelined": not a.no_pipeline, "step_seconds": [round(x, 2) for x in per_step],
                 "step_digests": step_digests, "step_digests_note": "SHA-256 over the codes of all levels of each timed step; asserted equal"}
    for l in (2, 1, 0):          # seconds from the start of the step until level l had produced all its codes
        if l in level_t:
            breakdown[f"level{l}_codes_done_at_s"] = round(level_t[l][-1] - t_last, 3)
        if l in level_t0 and l in level_t:   # tokens per second of the level while it ran (levels overlap when pipelined)
            busy = max(level_t[l][-1] - level_t0[l][-1], 1e-9)
            breakdown[f"level{l}_tokens_per_s"] = round(n_samples // world * (sample_length // priors[l].raw_to_tokens) / busy, 1)

    # dominant kernel, timed in situ with HIP events on the launch stream: the LayerNorm-folded weight-streaming
    # projections of the level-0 upsampler's decode step
    eng = priors[0].prior.bound_engine()
    roofline = projection_roofline(eng, 4096 if not tiny else 64, 16 if not tiny else 8)
    # one decode step (graph replay) of the same engine, against the algorithmic bytes of a step
    torch.cuda.synchronize()
    ts = time.perf_counter()
    n_probe = 64 if not tiny else 16
    eng.decode(4096 if not tiny else 64, n_probe)
    torch.cuda.synchronize()
    step_ms = plain_ms = (time.perf_counter() - ts) / n_probe * 1e3
    breakdown["level0_decode_ms_per_token_step_plain_chain"] = round(plain_ms, 4)
    # the job's own in-situ comparison of the two launch forms (ms per step of 384 pipelined / 128 plain steps of the first
    # window the level had to itself: ConditionalAutoregressive2D._decode); the sampler released the pair of streams at the end
    # of the job, so the form the level ran in is what that report says, and the step is timed here once more in that form
    report = getattr(priors[0].prior, "pipeline_report", None)
    kept = bool(report and report.get("kept"))
    breakdown["level0_launch_form"] = "pipelined" if kept else "plain chain"
    if report:
        breakdown["level0_in_situ_comparison_ms_per_step"] = report
    # every in-situ comparison of the last jobs, per level
    for l in (1, 0):
        reps = getattr(getattr(priors[l], "prior", None), "pipeline_reports", None)
        if reps:
            breakdown[f"level{l}_in_situ_comparisons"] = reps[-4:]
    if kept and eng.set_pipelined(True):
        eng.decode(4096, 16)
        step_ms = eng.timed_decode(4096, 256) * 1e3
        assert not eng.pipe_error()
        eng.set_pipelined(False)
    breakdown["level0_decode_ms_per_token_step"] = round(step_ms, 4)
    breakdown["launches_per_token_step"] = eng.launches_per_step
    tl = getattr(S._sample_levels_pipelined, "timeline", None)
    if tl and os.environ.get("JB_BENCH_TIMELINE") == "1":        # per-window schedule of the last step (diagnostics)
        breakdown["timeline"] = [list(x) for x in tl]
    step_bytes = eng.step_bytes(4096 if not tiny else 64)
    breakdown["level0_decode_step_algorithmic_gb"] = round(step_bytes / 1e9, 4)
    breakdown["level0_decode_step_frac_of_hbm_peak"] = round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    if world == 1 and not tiny and a.model == "1b_lyrics" and budget_left() > 200:
        try:
            breakdown["other_configs"] = other_configs(priors, a.samples_per_gpu, device)
        except Exception as e:   # noqa: BLE001 -- an auxiliary report must never cost the bench line
            breakdown["other_configs"] = dict(error=f"{type(e).__name__}: {e}")

    out = dict(metric=BASELINE_METRIC, value=round(value, 4), unit="audio_s/s",
               n_gpus=world, steps=steps_done, warmup=n_warm, steps_requested=a.steps, warmup_requested=a.warmup,
               ms_per_step=round(dt / steps_done * 1e3, 1),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f16", data="synthetic",
               config=dict(workload=f"{a.model} full 3-level ancestral sample (top prior + 2 upsamplers + VQ-VAE decode), "
                                    f"{sample_length / sr:.2f} s audio at {sr} Hz, n_samples={n_samples} "
                                    f"({a.samples_per_gpu}/GPU), temp=0.99, fp16 activations and weights, "
                                    "hop_fraction=(0.5,0.5,0.125), random-init weights",
                           samples_per_gpu=a.samples_per_gpu, max_batch_size=[k["max_batch_size"] for k in sk],
                           parallelism=f"sample-sharded x{world}",
                           wall_budget_s=BUDGET_S,
                           note="a step is the whole 3-level job; the run times as many full steps as fit the wall budget "
                                "(steps <= steps_requested), after at most one short warm-up pass"),
               dist=di, roofline=roofline, breakdown=breakdown)
    if world == 1 and not a.no_cpu_baseline:
        cb = None
        if not tiny and budget_left() > 90:
            cb = cpu_baseline_reference(sample_length / sr, a.samples_per_gpu, 45.0)
        if cb is None:
            total_steps = sum(sample_length // p.raw_to_tokens for p in priors)
            cb = cpu_baseline_port(priors[0], a.samples_per_gpu, a.cpu_steps if not tiny else 4, total_steps, audio_seconds_per_step)
        out["cpu_baseline"] = cb
        # BASELINE.md holds no published number for this metric, so vs_baseline stays null; the ratio to the reference timed
        # on this box's host cores in this run (north_star: >= 30x) is reported under its own name
        if cb and cb.get("value"):
            out["vs_cpu_baseline"] = round(value / cb["value"], 1)
    print(json.dumps(out), file=_JSON_OUT, flush=True)


if __name__ == "__main__":
    main()
    # leave the process group in order: a rank that simply exits while its peers are still inside a collective makes the
    # backend's worker threads abort ("terminate called without an active exception").  Only on the success path: a rank that
    # raised must not enter a barrier its peers are not in (they sit in another collective and would hang until the timeout).
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        try:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        except Exception:   # noqa: BLE001 -- shutting down
            pass
