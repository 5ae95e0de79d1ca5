"""Do RCCL's communicator (its streams and hardware queues) and software-pipelined launches get along in one process?
One rank, backend "nccl" (= RCCL): an all-reduce and a barrier on the GPU first, then the upsampler step on the plain chain and with
pipelined launches, then again with an all-reduce between the decode calls -- what every rank of `bench.py --gpus N` does.  The
reference points are tools/bench_engine.py's numbers without a process group (1.86 / 1.56 ms per step)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="env://", rank=0, world_size=1)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x)
dist.barrier()
torch.cuda.synchronize()
print(f"process group up: backend {dist.get_backend()}, all_reduce -> {float(x[0])}", flush=True)

cfg = CFGS["up"]
eng = PriorEngine(random_state(cfg, dev), "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
eng.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
eng.set_sampling(temp=0.99, seed=1)


def timed(n, between=None):
    eng.decode(4096, 8)
    torch.cuda.synchronize()
    t = time.perf_counter()
    done = 0
    while done < n:
        eng.decode(4096 + done, 128)
        done += 128
        if between is not None:
            between()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"plain chain                         : {timed(512):.3f} ms per step", flush=True)
on = eng.set_pipelined(True)
print(f"pipelined launches switched on      : {on}", flush=True)
print(f"pipelined                           : {timed(512):.3f} ms per step   (timed out slot: {eng.pipe_error()}, still pipelined: {eng.pipelined})", flush=True)
print(f"pipelined, all_reduce every 128 steps: {timed(512, lambda: dist.all_reduce(x)):.3f} ms per step   (timed out slot: {eng.pipe_error()}, still pipelined: {eng.pipelined})", flush=True)
eng.set_pipelined(False)
print(f"plain again                         : {timed(512):.3f} ms per step", flush=True)
dist.barrier()
dist.destroy_process_group()
print("done")
