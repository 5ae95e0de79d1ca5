"""Shader clock seen by the decode chain: a 20-us spin kernel (clock64 vs the 100-MHz wall clock) on an idle GPU, right
behind a burst of decode steps on the same stream, and between two bursts."""
import sys, time
import torch
sys.path.insert(0, ".")
from jukebox_amd import _lib as L
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine

dev = torch.device("cuda:0")
out = torch.zeros(2, dtype=torch.int64, device=dev)


def probe(label):
    L.check(L.lib().jb_clock_probe(out.data_ptr(), 2000, L.stream()))
    torch.cuda.synchronize()
    c, w = out.cpu().tolist()
    print(f"{label}: {c / (w * 10e-3):.0f} MHz shader clock ({c} cycles in {w * 10} ns)")


time.sleep(1.0)
probe("idle GPU (after 1 s of nothing)")
probe("second probe right after")
cfg = CFGS["up"]
eng = PriorEngine(random_state(cfg, dev), "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
eng.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
eng.set_sampling(temp=0.99, seed=1)
eng.decode(4096, 8)
torch.cuda.synchronize()
for rep in range(3):
    eng.decode(4096, 200)
    probe(f"behind 200 decode steps (rep {rep})")
