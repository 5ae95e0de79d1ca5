"""Isolated kernel timings: N identical dependent launches replayed from a graph (us per launch)."""
import time
import torch
from jukebox_amd import hip_ops as H, _lib as L

dev = torch.device("cuda:0")
N, W, S = 16, 1920, 480
dt = torch.float16
x = torch.randn(N, W, device=dev, dtype=dt)
xs = torch.randn(N, S, device=dev, dtype=dt)
g, b = torch.ones(W, device=dev), torch.zeros(W, device=dev)
w_attn = H.pack_conv1d_w(torch.randn(W, 3 * S, device=dev) * 0.02, dt)
w_proj = H.pack_conv1d_w(torch.randn(S, W, device=dev) * 0.02, dt)
w_fc = H.pack_conv1d_w(torch.randn(W, W, device=dev) * 0.02, dt)
bias = torch.zeros(3 * W, device=dev)
T = 8192
kc = torch.randn(N, T, S, device=dev, dtype=dt)
vc = torch.randn(N, T, S, device=dev, dtype=dt)
t_dev = torch.tensor([4500], dtype=torch.int32, device=dev)
out_w = torch.empty(N, W, device=dev, dtype=dt)
out_s = torch.empty(N, S, device=dev, dtype=dt)
out_3s = torch.empty(N, 3 * S, device=dev, dtype=dt)


def bench(name, fn, reps=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(reps):
                fn()
        gr.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        print(f"{name:40s} {(time.perf_counter() - t) / (5 * reps) * 1e6:7.2f} us")


bench("gemv plain K=1920 J=1920", lambda: H.gemv(x, w_fc, bias=bias[:W], out=out_w))
bench("gemv plain+res K=1920 J=1920", lambda: H.gemv(x, w_fc, bias=bias[:W], res=x, out=out_w))
bench("gemv plain K=480 J=1920 (+res)", lambda: H.gemv(xs, w_proj, bias=bias[:W], res=x, out=out_w))
bench("gemv LN K=1920 J=1440", lambda: H.gemv(x, w_attn, bias=bias[:3 * S], ln=(g, b), out=out_3s))
bench("gemv LN+gelu K=1920 J=1920", lambda: H.gemv(x, w_fc, bias=bias[:W], ln=(g, b), act=L.ACT_QUICK_GELU, out=out_w))
for thr, kb in ((512, -1), (512, 4)):
    L.lib().jb_tune_attn_decode(thr, kb)
    for f in (1, 2, 3):
        bench(f"attn_decode func {f} t=4500 thr={thr} kb={kb}", lambda f=f: H.attn_decode(f, xs, kc, vc, 1, 64, t_dev, T))
