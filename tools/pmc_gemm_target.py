"""PMC / timing target for the prefill GEMM: the library's jb_gemm on the prefill's dominant shape (M = 16 x 4096 rows, K = J = 1920, fp16,
bias + quick_gelu) through gemm_8phase_kernel and, with --glds, gemm_glds_kernel; --time prints HIP-event timings of both and of
torch.matmul (hipBLASLt: the plain library GEMM without the epilogue) on the same operands.
Usage (counters in passes of their own, as MI355X_MICROARCH.md prescribes):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out -- python tools/pmc_gemm_target.py
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d out -- python tools/pmc_gemm_target.py"""
import sys

import torch
from jukebox_amd import hip_ops as H, _lib as L

dev = torch.device("cuda:0")
M, K, J = 65536, 1920, 1920
torch.manual_seed(0)
A = torch.randn(M, K, device=dev, dtype=torch.float16)
W = (torch.randn(K, J, device=dev) * 0.02).half()
b = torch.randn(J, device=dev)
pw = H.pack_conv1d_w(W, torch.float16)
out = torch.empty(M, J, device=dev, dtype=torch.float16)
if "--glds" in sys.argv:
    L.lib().jb_tune_gemm_8phase(-1)


def timed(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flop = 2.0 * M * K * J
if "--time" in sys.argv:
    for name, tiles in (("gemm_8phase_kernel", 512), ("gemm_glds_kernel", -1)):
        L.lib().jb_tune_gemm_8phase(tiles)
        us = timed(lambda: H.gemm(A, pw, bias=b, act=L.ACT_QUICK_GELU, out=out))
        print(f"{name:22s} bias + quick_gelu epilogue   {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")
    L.lib().jb_tune_gemm_8phase(512)
    us = timed(lambda: torch.matmul(A, W, out=out))
    print(f"{'torch.matmul':22s} (hipBLASLt, no epilogue)     {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")
    Wt = W.t().contiguous()
    us = timed(lambda: torch.matmul(A, Wt.t(), out=out))
    print(f"{'torch.matmul':22s} (W stored [J][K])            {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")
else:
    for _ in range(6):
        H.gemm(A, pw, bias=b, act=L.ACT_QUICK_GELU, out=out)
    torch.cuda.synchronize()
