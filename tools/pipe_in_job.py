"""Why are software-pipelined launches slow INSIDE the 3-level job?  (HISTORY.md section 4.2, finding 3.)

One process, one GPU call: a short 3-level job (1b_lyrics, 16 samples, --seconds of audio, level pipeline, level 0 pipelined
while it runs alone), then the job's own level-0 engine is timed in the state the job leaves the process in:
  A  the job's engine as it is (64- and 256-step calls, per-slot stamps), then its plain chain
  B  the same engine with a fresh pair of streams and fresh graphs
  D  a reference engine (random upsampler, short caches) created now
  E  the job's engine from a worker thread on a side stream
Usage: JB_PIPE_TIMEOUT_MS=50 python -u tools/pipe_in_job.py [--seconds 6]"""
import argparse
import os
import sys
import time

os.environ["JB_PIPE_DEBUG"] = "1"
os.environ.setdefault("JB_PIPE_TIMEOUT_MS", "100")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
import bench_engine as BE  # noqa: E402
import jukebox_amd.sample as S  # noqa: E402
from jukebox_amd.engine import PriorEngine  # noqa: E402
from jukebox_amd.hparams import Hyperparams  # noqa: E402


def ms_per_step(eng, t0, n, use_graph=True):
    """(ms per step until the GPU is done, ms per step the HOST spent enqueueing)"""
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.decode(t0, n, use_graph=use_graph)
    t_host = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, (t_host - t) / n * 1e3


def measure(tag, eng, t0, stamps=True):
    eng.decode(t0, 8, use_graph=True)                 # setup / capture outside the timed calls
    (a, ha), (b, hb) = ms_per_step(eng, t0, 64), ms_per_step(eng, t0, 256)
    per_step = (b * 256 - a * 64) / 192
    # host enqueue ~ total: the graph launches themselves are the bottleneck (hipGraphLaunch off its pre-built-packet path?)
    print(f"{tag}: pipelined={eng.pipelined}  64-step call {a:.3f} ms/step (host enqueue {ha:.3f}), 256-step call {b:.3f} ms/step "
          f"(host enqueue {hb:.3f}) -> {per_step:.3f} ms/step + {(a - per_step) * 64:.1f} ms per call; error word {eng.pipe_error()}",
          flush=True)
    n_long = min(1536, eng.T - t0)
    (d, hd) = ms_per_step(eng, t0, n_long)            # a call far longer than any hardware queue: does the rate hold when the rings are full?
    print(f"      {n_long}-step call {d:.3f} ms/step (host enqueue {hd:.3f})", flush=True)
    if eng.pipelined:
        # the same launches on the same pair of streams WITHOUT the graph executor: fast here and slow above -> hipGraphLaunch
        eng.decode(t0, 4, use_graph=False)
        (c, hc) = ms_per_step(eng, t0, 128, use_graph=False)
        print(f"      eager launches on the pair of streams: {c:.3f} ms/step (host enqueue {hc:.3f}); error word {eng.pipe_error()}", flush=True)
    if stamps and eng.pipelined:
        BE.report_stamps(eng, indent="      ")


def reference_engine(dev, seq_len=2048):
    cfg = dict(BE.CFGS["up"])
    cfg["seq_len"], cfg["blocks"] = seq_len, seq_len // 64       # block_ctx 64 as in the real upsampler, short caches
    sd = BE.random_state(cfg, dev)
    eng = PriorEngine(sd, "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
    eng.set_cond(torch.randn(16, seq_len, cfg["width"], device=dev) * 0.01,
                 torch.randn(16, 1, cfg["width"], device=dev) * 0.01 if cfg["y_cond"] else None)
    eng.set_sampling(temp=0.99, seed=1)
    return eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0, help="at least 5.95: level 1 needs a full context")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sr = 44100
    sample_length = int(a.seconds * sr) // 128 * 128

    vq, priors = bench.build_models("1b_lyrics", sample_length, dev)
    hps = Hyperparams(n_samples=16, sample_length=sample_length, hop_fraction=[0.5, 0.5, 0.125], sr=sr, name="pipe_in_job",
                      keep_priors_resident=True, pipeline_levels=True, seed=0)
    labels = bench.synthetic_labels(priors, 16, 180 * sr, dev)
    sk = S.default_sampling_kwargs("1b_lyrics")
    t = time.perf_counter()
    try:
        S.ancestral_sample(labels, sk, priors, hps, save=False, device=dev)
    except RuntimeError as e:
        print("job raised:", str(e)[:300])
    torch.cuda.synchronize()
    print(f"job: {time.perf_counter() - t:.1f} s; windows (level, start, began, ended):")
    for x in getattr(S._sample_levels_pipelined, "timeline", []):
        print("   ", [round(v, 2) if isinstance(v, float) else v for v in x])
    print(f"memory allocated {torch.cuda.memory_allocated() / 1e9:.1f} GB", flush=True)

    print("in-situ comparison of the launch forms (level 0, first window alone):", getattr(priors[0].prior, "pipeline_report", None))
    eng = priors[0].prior.bound_engine()
    print(f"job's level-0 engine: pipelined={eng.pipelined} error word {eng.pipe_error()}")
    if eng.pipelined:
        print("   stamps of the job's last pipelined step:")
        BE.report_stamps(eng, indent="      ")
    else:
        print("   enabling:", eng.set_pipelined(True))
    eng.clear_pipe_error()
    measure("A  job's engine as the job left it", eng, 4096)
    eng.clear_pipe_error()
    eng.set_pipelined(False)
    measure("A' job's engine, plain chain", eng, 4096, stamps=False)
    print("   fresh streams + graphs:", eng.set_pipelined(True, fresh=True))
    measure("B  job's engine, fresh pair of streams", eng, 4096)
    eng.clear_pipe_error()
    eng.set_pipelined(False)
    ref2 = reference_engine(dev)
    assert ref2.set_pipelined(True)
    measure("D  reference engine made now", ref2, 1024)
    ref2.set_pipelined(False)
    ref2.set_pipelined(False)
    measure("D' the same reference engine, plain chain", ref2, 1024, stamps=False)
    # the job's engine once more, from a thread of its own on a side stream (as the level pipeline calls it)
    import threading
    eng.set_pipelined(True)
    st = torch.cuda.Stream(device=dev)

    def worker():
        with torch.cuda.stream(st):
            measure("E  job's engine, side stream, worker thread", eng, 4096, stamps=False)
    th = threading.Thread(target=worker)
    th.start()
    th.join()


if __name__ == "__main__":
    main()
