"""Debug: make / use / release the streams of the pipelined launches several times in one process; ms per step each time."""
import sys, time
import torch
sys.path.insert(0, ".")
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine
from jukebox_amd import _lib as L
if "--two-streams" in sys.argv:
    L.lib().jb_tune_pipeline(3)
dev = torch.device("cuda:0")
cfg = dict(CFGS["up"])
sd = random_state(cfg, dev)
eng = PriorEngine(sd, "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
eng.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
eng.set_sampling(temp=0.99, seed=1)
for rep in range(8):
    eng.set_pipelined(True)
    eng.decode(4096, 16)
    ms = eng.timed_decode(4096, 256) * 1e3
    err = eng.pipe_error()
    eng.set_pipelined(False)
    eng.decode(4096, 16)
    plain = eng.timed_decode(4096, 128) * 1e3
    print(f"creation {rep}: pipelined {ms:.3f} ms per step (pipe_error {err}), plain chain after the release {plain:.3f}")
