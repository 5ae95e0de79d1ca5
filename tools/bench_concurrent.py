"""Two upsampler engines decoding concurrently on separate streams (the level pipeline's situation):
per-engine ms/step alone, together at equal priority, and with engine A on a high-priority stream."""
import sys, threading, time
import torch
sys.path.insert(0, ".")
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine

dev = torch.device("cuda:0")
cfg = CFGS["up"]
sd = random_state(cfg, dev)
engs = []
for i in range(3):
    e = PriorEngine(sd, "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
    e.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
    e.set_sampling(temp=0.99, seed=i)
    engs.append(e)
del sd
STEPS = 384


def run(eng, stream, out, key):
    with torch.cuda.stream(stream):
        eng.decode(4096, 8)
        stream.synchronize()
        t = time.perf_counter()
        eng.decode(4096, STEPS)
        stream.synchronize()
        out[key] = (time.perf_counter() - t) / STEPS * 1e3


def trial(prios, both=True):
    streams = [torch.cuda.Stream(priority=p) for p in prios]
    out = {}
    ths = [threading.Thread(target=run, args=(engs[i], streams[i], out, i)) for i in range(len(prios) if both else 1)]
    for th in ths: th.start()
    for th in ths: th.join()
    return out


fmt = lambda d: {k: round(v, 3) for k, v in sorted(d.items())}
print("A alone:", fmt(trial((0, 0), both=False)))
print("A+B (0,0):", fmt(trial((0, 0))))
print("A+B (-1,-1):", fmt(trial((-1, -1))))
print("A+B (-1,0):", fmt(trial((-1, 0))))
print("A+B+C (-1,0,0):", fmt(trial((-1, 0, 0))))
print("A+B+C (-1,-1,0):", fmt(trial((-1, -1, 0))))
from jukebox_amd import _lib as L
import ctypes as C
a_, b_ = C.c_int(), C.c_int(); L.lib().jb_stream_priority_range(C.byref(a_), C.byref(b_)); print("hip priority range least", a_.value, "greatest", b_.value)
def trial_ext(n):
    streams, raw = L.priority_streams([-1, -1, 0][:n])
    out = {}
    ths = [threading.Thread(target=run, args=(engs[i], streams[i], out, i)) for i in range(n)]
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize(); L.destroy_streams(raw)
    return out
print("ext streams (-1,-1):", fmt(trial_ext(2)))
print("ext streams (-1,-1,0):", fmt(trial_ext(3)))
print("ext streams (-1,-1,0) again:", fmt(trial_ext(3)))
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
