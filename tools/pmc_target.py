"""Minimal PMC target: eager launches of the LayerNorm-folded projection (mlp.c_fc shape of the level-0 upsampler,
K = J = 1920, 16 rows, fp16) over 48 distinct weight matrices (354 MB > the 256 MB Infinity Cache, so every launch
streams its weights from HBM as in the real decode step).  `--wide`: the c_attn shape of a wide-value layer instead
(J = 2S + W = 2880 columns: q | k | v' with v' = v.Wp, jukebox_amd.engine.wide_value_weights)."""
import sys

import torch
from jukebox_amd import hip_ops as H, _lib as L

if "--engine" in sys.argv:
    # round 6: the dominant kernel in the form the engine's own steps launch it (operand-order activation blocks, 16-byte
    # write-through stores: gemv_lnf_kernel<..., true> without its hand-shake) -- a back-to-back burst of the c_attn / c_fc
    # launches of all 72 layers of the upsampler engine at 16 samples, every launch on its own cold weights
    # (jb_engine_probe_projection: what bench.py's roofline object times)
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_engine import CFGS, random_state
    from jukebox_amd.engine import PriorEngine
    cfg = CFGS["up"]
    dev = torch.device("cuda:0")
    eng = PriorEngine(random_state(cfg, dev), "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
    eng.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
    eng.set_sampling(temp=0.99, seed=1)
    us, launches, abytes = eng.probe_projection(4096, 2)
    print("launches", launches, "algorithmic bytes per launch (mean)", abytes, "avg us", us)
    sys.exit(0)

dev = torch.device("cuda:0")
N, W = 16, 1920
x = torch.randn(N, W, device=dev, dtype=torch.float16)
g, b = torch.ones(W, device=dev), torch.zeros(W, device=dev)
bias = torch.zeros(W, device=dev)
out = torch.empty(N, W, device=dev, dtype=torch.float16)
S = W // 4
JA = 2 * S + W if "--wide" in sys.argv else 3 * S          # columns of the c_attn launch
# the decode step's form of the projection: LayerNorm folded into the weights (hip_ops.FoldedLN -> gemv_lnf_kernel)
ws = [H.FoldedLN(torch.randn(W, W, device=dev) * 0.02, bias, g, b, torch.float16) for _ in range(28)]
wa = [H.FoldedLN(torch.randn(W, JA, device=dev) * 0.02, torch.zeros(JA, device=dev), g, b, torch.float16) for _ in range(28)]
out3 = torch.empty(N, JA, device=dev, dtype=torch.float16)
torch.cuda.synchronize()
for rep in range(4):
    for w, w3 in zip(ws, wa):          # the two shapes of the dominant kernel, alternating as in a layer
        H.gemv(x, None, ln_fold=w3, out=out3)
        H.gemv(x, None, ln_fold=w, act=L.ACT_QUICK_GELU, out=out)
torch.cuda.synchronize()
# a wide c_attn launch is credited with the reference's c_attn + the attn.c_proj it absorbed (W x 4S), as bench.py does
JC = 4 * S if "--wide" in sys.argv else 3 * S
print("launches", 8 * len(ws), "algorithmic bytes per launch (mean)",
      0.5 * ((W * W + W * JC) * 2 + 2 * N * W * 2 + N * (W + JC) * 2))
