"""Three upsampler engines decoding side by side from three host threads, as the level pipeline runs them; engine 0 asks for
pipelined launches, engine 1 asks too and is refused (one pipelined engine per process), engine 2 is plain: ms per step
each, whether an engine has pipelined launches, and the error words."""
import sys, threading, time
import torch
sys.path.insert(0, ".")
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine

dev = torch.device("cuda:0")
cfg = dict(CFGS["up"], depth=int(sys.argv[1]) if len(sys.argv) > 1 else 12, seq_len=2048, blocks=32)
sd = random_state(cfg, dev)
engs = []
for i in range(3):
    e = PriorEngine(sd, "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
    e.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
    e.set_sampling(temp=0.99, seed=i)
    engs.append(e)
print("pipelined launches granted:", [e.set_pipelined(i < 2) for i, e in enumerate(engs)], flush=True)
STEPS = 256


def run(eng, stream, out, key):
    with torch.cuda.stream(stream):
        eng.decode(1024, 8)
        stream.synchronize()
        t = time.perf_counter()
        for c in range(2):
            eng.decode(1024 + c * STEPS, STEPS)
        stream.synchronize()
        out[key] = ((time.perf_counter() - t) / (2 * STEPS) * 1e3, eng.pipelined, eng.pipe_error())


for prios in ((-1, -1, 0), (0, 0, 0), (-1, 0, 0)):
    streams = [torch.cuda.Stream(priority=p) for p in prios]
    out = {}
    ths = [threading.Thread(target=run, args=(engs[i], streams[i], out, i)) for i in range(3)]
    for th in ths: th.start()
    for th in ths: th.join()
    print("priorities", prios, {k: (round(v[0], 3), v[1], v[2]) for k, v in sorted(out.items())}, flush=True)
    break      # an engine keeps the stream it was first used on
