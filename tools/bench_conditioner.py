"""Conditioner of upsampler_level_0 (2048 upper-level codes -> 8192 x 1920, conv width 1024, 2 x 16 dilated residual blocks),
16 samples: the conv stack on the exact-fp32 matrix instruction against the f16-split form (jb_gemm_args.w_split) -- time
per call and the largest difference between the outputs."""
import sys
import time

import torch

sys.path.insert(0, ".")
from jukebox_amd.prior.conditioners import Conditioner
from jukebox_amd.vqvae import rows

dev = torch.device("cuda:0")
torch.manual_seed(0)
with torch.device(dev):
    c = Conditioner(input_shape=(2048,), bins=2048, down_t=2, stride_t=2, out_width=1920, init_scale=0.4, zero_out=False,
                    res_scale=False, width=1024, depth=16, m_conv=1.0, dilation_growth_rate=3, dilation_cycle=8, checkpoint_res=1)
c.eval()
z = torch.randint(0, 2048, (16, 2048), device=dev)


def drop_packed(m):
    for mod in m.modules():
        mod.__dict__.pop("_packed_cache", None)


def run(split, reps=3):
    rows.SPLIT_F16 = split
    drop_packed(c)
    with torch.no_grad():
        out = c(z)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            out = c(z)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
    return out, min(ts)


ref, t_exact = run(False)
print(f"conditioner x16, exact fp32 MFMA : {t_exact * 1e3:7.1f} ms  ({16 * 0.93 / t_exact:.0f} TFLOP/s)")
out, t = run(True)
d = float((out - ref).abs().max() / ref.abs().max())
print(f"conditioner x16, f16 split       : {t * 1e3:7.1f} ms  ({16 * 0.93 / t:.0f} TFLOP/s)   max |split - exact| / max |exact| = {d:.2e}")
