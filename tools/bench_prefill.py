"""Prefill (primed window) and conditioner timings at upsampler dimensions, N=16."""
import sys, time
import torch
sys.path.insert(0, ".")
from tools.bench_engine import CFGS, random_state
from jukebox_amd.engine import PriorEngine
from jukebox_amd.prior.conditioners import Conditioner

dev = torch.device("cuda:0")
cfg = CFGS["up"]
eng = PriorEngine(random_state(cfg, dev), "", n_batch=16, fp16=True, chunk_cap=int(sys.argv[1]) if len(sys.argv) > 1 else 2048, **cfg)
eng.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
eng.tokens.random_(0, cfg["bins"])
for n in (512, 4096):
    eng.prefill(0, n); torch.cuda.synchronize()
    t = time.perf_counter(); eng.prefill(0, n); torch.cuda.synchronize(); dt = time.perf_counter() - t
    fl = 16 * n * 1.59e9
    print(f"prefill {n} tokens x16: {dt*1e3:.1f} ms  ({fl/dt/1e12:.1f} TFLOP/s)")
with torch.device(dev):
    c = Conditioner(input_shape=(2048,), bins=2048, down_t=2, stride_t=2, out_width=1920, init_scale=0.4, zero_out=False,
                    res_scale=False, width=1024, depth=16, m_conv=1.0, dilation_growth_rate=3, dilation_cycle=8, checkpoint_res=1)
z = torch.randint(0, 2048, (16, 2048), device=dev)
c(z); torch.cuda.synchronize()
t = time.perf_counter(); c(z); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"conditioner x16: {dt*1e3:.1f} ms  ({16*0.93e12/dt/1e12:.1f} TFLOP/s fp32)")
