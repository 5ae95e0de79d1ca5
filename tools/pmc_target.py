"""Minimal PMC target: eager launches of the LayerNorm-folded projection (mlp.c_fc shape of the level-0 upsampler,
K = J = 1920, 16 rows, fp16) over 48 distinct weight matrices (354 MB > the 256 MB Infinity Cache, so every launch
streams its weights from HBM as in the real decode step)."""
import torch
from jukebox_amd import hip_ops as H, _lib as L

dev = torch.device("cuda:0")
N, W = 16, 1920
x = torch.randn(N, W, device=dev, dtype=torch.float16)
g, b = torch.ones(W, device=dev), torch.zeros(W, device=dev)
bias = torch.zeros(W, device=dev)
out = torch.empty(N, W, device=dev, dtype=torch.float16)
S = W // 4
# the decode step's form of the projection: LayerNorm folded into the weights (hip_ops.FoldedLN -> gemv_lnf_kernel)
ws = [H.FoldedLN(torch.randn(W, W, device=dev) * 0.02, bias, g, b, torch.float16) for _ in range(28)]
wa = [H.FoldedLN(torch.randn(W, 3 * S, device=dev) * 0.02, torch.zeros(3 * S, device=dev), g, b, torch.float16) for _ in range(28)]
bias3 = torch.zeros(3 * S, device=dev)
out3 = torch.empty(N, 3 * S, device=dev, dtype=torch.float16)
torch.cuda.synchronize()
for rep in range(4):
    for w, w3 in zip(ws, wa):          # the two shapes of the dominant kernel, alternating as in a layer
        H.gemv(x, None, ln_fold=w3, out=out3)
        H.gemv(x, None, ln_fold=w, act=L.ACT_QUICK_GELU, out=out)
torch.cuda.synchronize()
print("launches", 8 * len(ws), "algorithmic bytes per launch (mean)",
      0.5 * ((W * W + W * 3 * S) * 2 + 2 * N * W * 2 + N * (W + 3 * S) * 2))
