"""Upsampler engines decoding side by side from their own host threads, as the level pipeline runs them: ms per step each,
whether an engine has pipelined launches, and the error words, for
  A  two engines on plain chains            B  two engines on pipelined launches (two owners per process: lean attention)
  C  one pipelined, one plain               D  one engine alone, pipelined / plain
Usage: python tools/pipe_concurrent.py [depth] [plain]        (plain: cases A and D-plain only, for a kernel trace)"""
import sys, threading, time
import torch
sys.path.insert(0, ".")
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine

dev = torch.device("cuda:0")
cfg = dict(CFGS["up"], depth=int(sys.argv[1]) if len(sys.argv) > 1 else 12, seq_len=2048, blocks=32)
sd = random_state(cfg, dev)
engs = []
for i in range(2):
    e = PriorEngine(sd, "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
    e.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
    e.set_sampling(temp=0.99, seed=i)
    engs.append(e)
STEPS = 384
streams = [torch.cuda.Stream(priority=-1) for _ in engs]


def run(i, out):
    eng, stream = engs[i], streams[i]
    with torch.cuda.stream(stream):
        eng.decode(1024, 8)
        stream.synchronize()
        t = time.perf_counter()
        for c in range(2):
            eng.decode(1024 + c * STEPS, STEPS)
        stream.synchronize()
        out[i] = (round((time.perf_counter() - t) / (2 * STEPS) * 1e3, 3), eng.pipelined, eng.pipe_error())


def case(name, modes):
    for e, m in zip(engs, modes):
        if m is not None:
            e.set_pipelined(bool(m))
    out = {}
    ths = [threading.Thread(target=run, args=(i, out)) for i, m in enumerate(modes) if m is not None]
    for th in ths: th.start()
    for th in ths: th.join()
    print(name, dict(sorted(out.items())), flush=True)
    for e in engs:
        e.set_pipelined(False)


from jukebox_amd import _lib as L
L.lib().jb_tune_attn_decode_wide_lean(1)         # two pipelined engines need the lean attention kernel
if len(sys.argv) > 2 and sys.argv[2] == "plain":
    # for a kernel trace (tools/overlap_account.py): plain chains only -- side by side, then one alone
    case("A two plain chains        ", (0, 0))
    case("D alone, plain            ", (0, None))
    sys.exit(0)
for rep in range(2):
    case("A two plain chains        ", (0, 0))
    case("B two pipelined engines   ", (1, 1))
    if rep == 0:
        case("C pipelined next to plain ", (1, 0))
    case("D alone, pipelined        ", (1, None))
    case("D alone, plain            ", (0, None))
