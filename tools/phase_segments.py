"""Per-segment account of a pipelined phase of the decode step: the measurement build of the library
(-DJB_PIPE_SEGMENTS: `python -m jukebox_amd.csrc.build --segments`) stamps, in workgroup 0 of every launch,
    flags seen -> barrier behind the poll -> operand fragments landed -> arithmetic retired, partials in LDS -> LDS exchange
    barrier -> stores issued -> stores drained -> own ticket returned -> (the launch's last workgroup) published -> the next
    launch sees the flags
and this tool averages them per kind of launch over the 72 layers and several steps of the upsampler engine (N = 16).

    JB_LIB_SEGMENTS=1 JB_PIPE_DEBUG=1 python tools/phase_segments.py [--calls 8] [--steps 64] [--md out.md]

Workgroup 0 of the measurement build WAITS where the product build does not have to (all operands landed before the first
MFMA; the LDS writes retired before the barrier), so its own timeline is the account's; the other workgroups run as in the
product build and the step time printed at the end says how much the stamping costs."""
import argparse
import os
import sys
import time

os.environ.setdefault("JB_LIB_SEGMENTS", "1")
os.environ.setdefault("JB_PIPE_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_engine import CFGS, random_state  # noqa: E402
from jukebox_amd.engine import PriorEngine  # noqa: E402

SEGMENTS = [("flags seen -> barrier behind the poll", 1, 4), ("-> operand fragments landed", 4, 5),
            ("-> arithmetic retired, partials in LDS", 5, 6), ("-> LDS exchange barrier passed", 6, 7),
            ("-> epilogue done, stores issued", 7, 3), ("-> stores drained (vmcnt 0)", 3, 8),
            ("-> barrier + own ticket returned", 8, 9), ("-> last workgroup of the launch published", 9, 2)]
KINDS = ["c_attn", "attention", "c_fc", "mlp.c_proj"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=8)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--t0", type=int, default=4096)
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    cfg = CFGS["up"]
    dev = torch.device("cuda:0")
    eng = PriorEngine(random_state(cfg, dev), "", n_batch=16, fp16=True, chunk_cap=64, **cfg)
    eng.set_cond(torch.randn(16, cfg["seq_len"], cfg["width"], device=dev) * 0.01, torch.randn(16, 1, cfg["width"], device=dev) * 0.01)
    eng.set_sampling(temp=0.99, seed=1)
    assert eng.set_pipelined(True), "this engine has no pipelined launches"
    eng.decode(a.t0, 16)
    torch.cuda.synchronize()
    rows = {k: [] for k in range(4)}          # per kind: list of per-launch segment vectors (us)
    hand = {k: [] for k in range(4)}          # published -> the consumer's wave 0 sees the flags; whole phase
    ms, cal = [], []
    for call in range(a.calls):
        t = time.perf_counter()
        eng.decode(a.t0 + 16 + call * a.steps, a.steps)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t) / a.steps * 1e3)
        assert not eng.pipe_error(), "a pipelined wait timed out"
        st = eng.pipe_stamps().astype(np.float64) * 0.01          # us of the 100 MHz clock
        n = st.shape[0]
        cal.append(st[8:n - 3, 11] - st[8:n - 3, 10])
        for k in range(4):
            idx = np.arange(8 + k, n - 3, 4)
            rows[k].append(np.stack([st[idx, b] - st[idx, a_] for _, a_, b in SEGMENTS], 1))
            hand[k].append(np.stack([st[idx + 1, 1] - st[idx, 2], st[idx + 1, 1] - st[idx, 1], st[idx, 1] - st[idx, 0]], 1))
    lines = []
    head = "| segment (workgroup 0 of the launch) | " + " | ".join(KINDS) + " |"
    lines += [head, "|---|" + "---|" * 4]
    seg = [np.concatenate(rows[k]).mean(0) for k in range(4)]
    hd = [np.concatenate(hand[k]).mean(0) for k in range(4)]
    for i, (name, _, _) in enumerate(SEGMENTS):
        lines.append(f"| {name} | " + " | ".join(f"{seg[k][i]:.2f}" for k in range(4)) + " |")
    lines.append("| -> the NEXT launch's wave 0 sees the flags | " + " | ".join(f"{hd[k][0]:.2f}" for k in range(4)) + " |")
    lines.append("| **flags seen -> the next launch sees its flags (the phase)** | " + " | ".join(f"**{hd[k][1]:.2f}**" for k in range(4)) + " |")
    lines.append("| (poll entered this long before the flags were seen) | " + " | ".join(f"{hd[k][2]:.2f}" for k in range(4)) + " |")
    lines.append(f"| (two stamps back to back: every segment above contains about this much of the stamping itself) | {np.concatenate(cal).mean():.2f} | | | |")
    out = "\n".join(lines)
    print(out)
    frag = os.environ.get("JB_PIPE_FRAG", "1")
    tail = (f"\nus, means over {a.calls} steps x 72 layers; upsampler engine, N = 16, t = {a.t0}..; completion protocol 1, "
            f"operand-order hand-offs {frag}; step by the host clock {np.mean(ms):.3f} ms "
            f"({'measurement' if os.environ.get('JB_LIB_SEGMENTS') == '1' else 'product'} build)")
    print(tail)
    if a.md:
        with open(a.md, "a") as f:
            f.write(out + "\n" + tail + "\n\n")


if __name__ == "__main__":
    main()
