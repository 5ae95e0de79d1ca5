import ctypes as C, torch, os
import jukebox_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libjb_timing.so")
from jukebox_amd import hip_ops as H
lib = L.lib(); lib.jb_set_dbg.argtypes = [C.c_void_p]
dev = torch.device("cuda:0"); N, W, S = 16, 1920, 480; dt = torch.float16
x = torch.randn(N, W, device=dev, dtype=dt); g, b = torch.ones(W, device=dev), torch.zeros(W, device=dev)
w = H.pack_conv1d_w(torch.randn(W, W, device=dev) * 0.02, dt); bias = torch.zeros(W, device=dev)
out = torch.empty(N, W, device=dev, dtype=dt)
dbg = torch.zeros(32, dtype=torch.int64, device=dev); lib.jb_set_dbg(dbg.data_ptr())
names = ["issue", "stage->lds", "sync1", "ln-phase", "sync2", "mainloop", "sync3", "epilogue"]
for label, kw in (("LN", dict(ln=(g, b))), ("plain", dict())):
    for rep in range(3):
        torch.cuda.synchronize(); dbg.zero_()
        for _ in range(20): H.gemv(x, w, bias=bias, out=out, **kw)   # warm clocks
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(-1, 2)
        cyc = d[:8, 0]; wall = d[:8, 1]
        valid = [i for i in range(8) if cyc[i] != 0]
        line = f"{label} rep{rep}: "
        for a, bb in zip(valid[:-1], valid[1:]):
            line += f"{names[bb]}={cyc[bb]-cyc[a]}c/{(wall[bb]-wall[a])*10}ns  "
        tot_c, tot_w = cyc[valid[-1]] - cyc[valid[0]], (wall[valid[-1]] - wall[valid[0]]) * 10
        print(line, f"| total {tot_c} cyc, {tot_w} ns -> {tot_c / max(tot_w,1):.2f} GHz")
