"""CU-masked streams: (1) where do mask bits land (XCC, CU census)?  (2) does confining the upper levels' decode chains to
their own CUs protect the level-0 chain from the 1.3x slowdown measured with three unmasked streams?

    python tools/cu_mask_probe.py            # on the GPU box"""
import sys, threading, time
from collections import Counter

import torch

sys.path.insert(0, ".")
from jukebox_amd import _lib as L
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine

dev = torch.device("cuda:0")


def census(stream, label):
    out = torch.zeros(2 * 2048, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        L.check(L.lib().jb_cu_census(2048, out.data_ptr(), stream.cuda_stream))
    stream.synchronize()
    o = out.cpu().numpy().reshape(-1, 2)
    xcc = o[:, 0] & 0xF
    cu = (o[:, 1] >> 8) & 0xF
    sh = (o[:, 1] >> 12) & 0x1
    se = (o[:, 1] >> 13) & 0x7
    units = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = Counter(k[0] for k in units)
    print(f"{label}: {len(units)} distinct (xcc, se, sh, cu); per XCC {dict(sorted(per_xcc.items()))}")


masks = {
    "bits 0..63": range(64),
    "bits 0..127": range(128),
    "every 4th bit": range(0, 256, 4),
    "bits 192..255": range(192, 256),
    "even bits": range(0, 256, 2),
}
census(torch.cuda.Stream(), "unmasked")
raws = []
for name, bits in masks.items():
    s, h = L.cu_mask_stream(bits)
    raws.append(h)
    census(s, name)

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
        out[key] = round((time.perf_counter() - t) / STEPS * 1e3, 3)


def trial(streams):
    out = {}
    ths = [threading.Thread(target=run, args=(engs[i], s, out, i)) for i, s in enumerate(streams)]
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize()
    return dict(sorted(out.items()))


# host-side cost of enqueuing one captured decode step (hipGraphLaunch of 5L+2 kernel nodes): if this is a large fraction
# of the 2 ms step, three host threads feeding three chains contend on the host, not on the GPU
torch.cuda.synchronize()
for n in (1, 2, 4, 8):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        engs[0].decode(4096, n)
        ts.append((time.perf_counter() - t) * 1e3)
        torch.cuda.synchronize()
    print(f"host time to enqueue decode({n} steps): {min(ts):.3f} ms  ({min(ts) / n:.3f} ms per step incl. fixed part)")

hp = lambda: torch.cuda.Stream(priority=-1)
print("A alone:", trial([hp()]))
print("A,B,C unmasked (-1,-1,0):", trial([hp(), hp(), torch.cuda.Stream()]))
for name, (ba, bb, bc) in {
    "A unmasked | B bits 128..191 | C bits 192..255": (None, range(128, 192), range(192, 256)),
    "A bits 0..127 | B 128..191 | C 192..255": (range(128), range(128, 192), range(192, 256)),
    "A unmasked | B,C every-4th-bit (offset 2,3)": (None, range(2, 256, 4), range(3, 256, 4)),
    "A bits%4 in {0,1} | B bits%4==2 | C bits%4==3": ([b for b in range(256) if b % 4 < 2], range(2, 256, 4), range(3, 256, 4)),
}.items():
    ss = []
    for bits in (ba, bb, bc):
        if bits is None:
            ss.append(hp())
        else:
            s, h = L.cu_mask_stream(bits)
            raws.append(h)
            ss.append(s)
    print(name, "->", trial(ss))
    print("   A alone on its stream:", trial(ss[:1]))
