#!/usr/bin/env python3
"""BASELINE configs 3 and 4 as whole-window goldens: the UNMODIFIED reference's ConditionalAutoregressive2D on the two
geometries that carry the 20-second job, fp32, greedy (top_k = 1), EVERY token of a window -- CPU hours in the build
container, zero GPU minutes, run once:

    python tests/golden/gen_whole_window.py upsampler_whole        # sample()        (autoregressive.py:199-249), 8192 tokens
    python tests/golden/gen_whole_window.py 1b_lyrics_top_whole    # primed_sample() (autoregressive.py:251-359), 384 lyric
                                                                   # tokens prefilled in chunks of 32, then 6144 music tokens

Weights and inputs: tests/full_size_cases.py (state_dict / sample_inputs of the case: numpy PCG64 streams, rebuilt bit for
bit by the GPU test on the box, where /root/reference does not exist).  Saved per case: the token stream, and per sampled
position the reference's top-1 logit, its gap to the runner-up and the runner-up's index (the near-tie rule of the GPU test
needs to know where the reference itself was within rounding of another token), plus the first four logit rows in full."""
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
    case = FS.WHOLE[tag]
    t.set_grad_enabled(False)
    t.set_num_threads(int(os.environ.get("JB_GEN_THREADS", "6")))
    n_tok = int(os.environ.get("JB_GEN_TOKENS", str(case["seq"])))
    N, W, t0 = case["N"], case["W"], case["t0"]
    p = ConditionalAutoregressive2D((case["seq"],), case["bins"], width=W, depth=case["depth"], heads=case["heads"],
                                    attn_order=case["attn_order"], blocks=case["blocks"], m_attn=0.25, m_mlp=1.0,
                                    x_cond=True, y_cond=True, prime_len=case["prime_len"])
    p.eval()
    sd = FS.state_dict(case)
    print(p.load_state_dict({k: t.from_numpy(v) for k, v in sd.items()}, strict=True), len(sd), "tensors", flush=True)
    ins = [FS.sample_inputs(case, n) for n in range(N)]
    x_cond = t.from_numpy(np.stack([i[1] for i in ins]))
    y_cond = t.from_numpy(np.stack([i[2] for i in ins]))
    t_start = time.time()
    stop = threading.Event()

    def beat():
        while not stop.wait(300):
            print(f"  ... {time.time() - t_start:6.0f} s", flush=True)
    threading.Thread(target=beat, daemon=True).start()
    if t0 == 0:
        z, preds = p.sample(N, x_cond, y_cond, fp16=False, temp=1.0, top_k=1, get_preds=True, sample_tokens=n_tok)
    else:
        prefix = t.from_numpy(np.stack([i[0] for i in ins]).astype(np.int64))
        z, preds = p.primed_sample(N, prefix, x_cond, y_cond, fp16=False, temp=1.0, top_k=1, get_preds=True, chunk_size=32,
                                   sample_tokens=n_tok)
        assert (z.view(N, -1)[:, :t0] == prefix).all()
    stop.set()
    z = z.view(N, -1)
    print(f"sampled {tuple(z.shape)} in {time.time() - t_start:.0f} s", flush=True)
    # preds row i is the distribution of token i (teacher-forced over the prefix): keep the sampled part
    preds = preds[:, t0:] if t0 else preds
    assert preds.shape[1] == n_tok - t0
    top2 = preds.topk(2, dim=-1)
    assert (top2.indices[..., 0] == z[:, t0:]).all(), "greedy token is the arg-max of the recorded logits"
    gap = (top2.values[..., 0] - top2.values[..., 1]).numpy().astype(np.float32)
    out = FS.golden_path(tag) if n_tok == case["seq"] else FS.golden_path(f"{tag}_{n_tok}")
    np.savez_compressed(out, z=z.numpy().astype(np.int16), gap=gap, runner_up=top2.indices[..., 1].numpy().astype(np.int16),
                        top1=top2.values[..., 0].numpy().astype(np.float32), first_logits=preds[:, :4].numpy().astype(np.float32),
                        t0=np.int64(t0))
    print("wrote", out, os.path.getsize(out) // 1024, "KiB; near-ties (gap < 1e-3):", int((gap < 1e-3).sum()), flush=True)


if __name__ == "__main__":
    main()
