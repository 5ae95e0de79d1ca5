#!/usr/bin/env python3
"""The arithmetic bench.py times, pinned to the reference's own: the UNMODIFIED reference's ConditionalAutoregressive2D at the
upsampler geometry (width 1920, depth 72, one head, 128 blocks of 64) run with fp16=True -- the reference's sampling default
(sample.py:239-241; transformer.py:169-192 `x.half()`, ops.py:14-24 LayerNorm in fp32 cast back, ops.py:97-101 `w.type_as(x)`
addmm in half, factored_attention.py:82-108 softmax in fp32) -- on CPU in the build container, zero GPU minutes:

    python tests/golden/gen_fp16_window.py upsampler_fp16

  1. primed_sample(fp16=True, top_k=1, get_preds=True, chunk_size=32): t0 = 4096 primed tokens, then n_steps = 4096 greedy
     steps (autoregressive.py:251-359) -- a whole level-0 window of the 20-second job as the sampler runs it (sample.py:17-88
     with hop 4096).  Stored: the stream; for EVERY sampled position the eight largest fp16 logits with their indices; the
     fp16 logits in full for the last 64 primed positions, the first 128 and the last 128 sampled ones.
  2. the SAME model in fp32 teacher-forced on that stream (primed_sample over the stream's first t0 + n_steps - 1 tokens): the
     fp32 logits of the same positions in full, and per sampled position its arg-max, top-1 / top-2 gap and max / mean
     |fp16 - fp32| -- the reference's own half-precision error is the scale the GPU test's gate is read against.

Weights and inputs: tests/full_size_cases.py (FP16 table; numpy PCG64 streams, rebuilt bit for bit by the GPU test)."""
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refshim  # noqa: E402

refshim.install()

import torch as t  # noqa: E402
from jukebox.prior.autoregressive import ConditionalAutoregressive2D  # noqa: E402
import full_size_cases as FS  # noqa: E402

def main():
    tag = sys.argv[1]
    case = FS.FP16[tag]
    t.set_grad_enabled(False)
    t.set_num_threads(int(os.environ.get("JB_GEN_THREADS", "6")))
    N, W, t0 = case["N"], case["W"], case["t0"]
    n_steps = int(os.environ.get("JB_GEN_STEPS", str(case["n_steps"])))
    t0 = int(os.environ.get("JB_GEN_T0", str(t0)))
    case = dict(case, t0=t0)
    p = ConditionalAutoregressive2D((case["seq"],), case["bins"], width=W, depth=case["depth"], heads=case["heads"],
                                    attn_order=case["attn_order"], blocks=case["blocks"], m_attn=0.25, m_mlp=1.0,
                                    x_cond=True, y_cond=True, prime_len=case["prime_len"])
    p.eval()
    sd = FS.state_dict(case)
    print(p.load_state_dict({k: t.from_numpy(v) for k, v in sd.items()}, strict=True), len(sd), "tensors", flush=True)
    ins = [FS.sample_inputs(case, n) for n in range(N)]
    x_cond = t.from_numpy(np.stack([i[1] for i in ins]))
    y_cond = t.from_numpy(np.stack([i[2] for i in ins]))
    prefix = t.from_numpy(np.stack([i[0] for i in ins]).astype(np.int64))
    t_start = time.time()
    stop = threading.Event()

    def beat():
        while not stop.wait(300):
            print(f"  ... {time.time() - t_start:6.0f} s", flush=True)
    threading.Thread(target=beat, daemon=True).start()
    T = t0 + n_steps
    z16, p16 = p.primed_sample(N, prefix, x_cond, y_cond, fp16=True, temp=1.0, top_k=1, get_preds=True, chunk_size=32,
                               sample_tokens=T)
    z16 = z16.view(N, -1)
    assert z16.shape == (N, T) and (z16[:, :t0] == prefix).all() and p16.shape == (N, T, case["bins"]) and p16.dtype == t.float32
    assert (p16[:, t0:].argmax(-1) == z16[:, t0:]).all(), "greedy token is the arg-max of the recorded logits"
    print(f"fp16 pass: {time.time() - t_start:.0f} s", flush=True)
    z32, p32 = p.primed_sample(N, z16[:, :T - 1].contiguous(), x_cond, y_cond, fp16=False, temp=1.0, top_k=1, get_preds=True,
                               chunk_size=32, sample_tokens=T)
    stop.set()
    assert p32.shape == p16.shape
    print(f"both passes: {time.time() - t_start:.0f} s", flush=True)
    d = (p16[:, t0:] - p32[:, t0:]).abs()
    agree = (p16[:, t0:].argmax(-1) == p32[:, t0:].argmax(-1)).float().mean()
    print("reference fp16 vs reference fp32 on the fp16 stream: max |dlogit| %.4f mean %.5f top-1 agreement %.4f (logit std %.3f)"
          % (d.max(), d.mean(), agree, p32[:, t0:].std()), flush=True)
    full = FS.fp16_full_positions(case, n_steps)                      # positions whose logit rows are kept in full
    top8 = p16[:, t0:].topk(8, dim=-1)
    top2 = p32[:, t0:].topk(2, dim=-1)
    out = FS.golden_path(tag)
    np.savez_compressed(out, z=z16.numpy().astype(np.int16), t0=np.int64(t0), n_steps=np.int64(n_steps), full_pos=full,
                        logits_fp16=p16[:, full].numpy().astype(np.float32), logits_fp32=p32[:, full].numpy().astype(np.float32),
                        top8_val=top8.values.numpy().astype(np.float32), top8_idx=top8.indices.numpy().astype(np.int16),
                        arg32=top2.indices[..., 0].numpy().astype(np.int16),
                        gap32=(top2.values[..., 0] - top2.values[..., 1]).numpy().astype(np.float32),
                        err_max=d.amax(-1).numpy().astype(np.float32), err_mean=d.mean(-1).numpy().astype(np.float32))
    print("wrote", out, os.path.getsize(out) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main()
