"""Two upsampler engines on plain chains side by side (levels 1 and 0 of the job while they overlap): ms per step of each on
(a) ordinary streams, (b) streams of different priority, (c) CU-masked streams on disjoint halves / a 3:1 split of every XCD's CUs.
    python tools/overlap_forms.py            # on the GPU box"""
import sys, threading, time
import torch
sys.path.insert(0, ".")
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine
from jukebox_amd import _lib as L

dev = torch.device("cuda:0")
cfg = CFGS["up"]
sd = random_state(cfg, dev)
engs = []
for i in range(2):
    e = PriorEngine(sd, "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
    e.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
    e.set_sampling(temp=0.99, seed=i)
    e.set_pipelined(False)
    engs.append(e)
del sd
STEPS = 512


def run(eng, stream, out, key, steps):
    with torch.cuda.stream(stream):
        eng.decode(4096, 8)
        stream.synchronize()
        t = time.perf_counter()
        eng.decode(4096, steps)
        stream.synchronize()
        out[key] = (time.perf_counter() - t) / steps * 1e3


def trial(streams, steps=(STEPS, STEPS)):
    out = {}
    ths = [threading.Thread(target=run, args=(engs[i], streams[i], out, i, steps[i])) for i in range(len(streams))]
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize()
    return {k: round(v, 3) for k, v in sorted(out.items())}


print("A alone, ordinary stream:", trial([torch.cuda.Stream()]))
for rep in range(2):
    print("A + B, ordinary streams:", trial([torch.cuda.Stream(), torch.cuda.Stream()]))
    print("A + B, priorities (-1, 0):", trial([torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]))
    for name, (ma, mb) in {"halves of every XCD (bits 0..127 | 128..255)": (range(128), range(128, 256)),
                           "3 : 1 (bits 0..191 | 192..255)": (range(192), range(192, 256)),
                           "even | odd bits (XCDs 0,2,4,6 | 1,3,5,7 -- if a mask may leave an XCD empty)": (range(0, 256, 2), range(1, 256, 2))}.items():
        sa, ra = L.cu_mask_stream(ma)
        sb, rb = L.cu_mask_stream(mb)
        print(f"A + B, CU-masked {name}:", trial([sa, sb]))
        print(f"   A alone on its mask:", trial([sa]))
        L.destroy_streams([ra, rb])
    print("A + B, ordinary streams again (after the masked streams were destroyed):", trial([torch.cuda.Stream(), torch.cuda.Stream()]))
