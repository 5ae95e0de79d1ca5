#!/usr/bin/env python3
"""BASELINE config 2 as a golden: the UNMODIFIED reference's ConditionalAutoregressive2D.sample (autoregressive.py:199-249)
on small_prior (hparams.py:210-219: width 1024, depth 48, 1 head, attn_order 2, blocks 64, n_ctx 8192), 16 samples, fp32,
greedy (top_k = 1), ALL 8192 tokens -- about an hour of CPU in the build container (8 cores), run once:

    python tests/golden/gen_small_prior_full.py            # writes tests/golden/small_prior_full.npz

Weights: tests/golden/seeded_weights.prior_state(seed 2024, ...) -- the GPU test rebuilds them from the same seed.
Saved: the token stream z (16 x 8192), and per position the reference's top-1 / top-2 logit GAP with the runner-up's index
(the near-tie rule of the GPU test needs to know where the reference itself was within rounding of another token).
"""
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

import torch as t  # noqa: E402
from jukebox.hparams import setup_hparams  # noqa: E402
from jukebox.prior.autoregressive import ConditionalAutoregressive2D  # noqa: E402
from seeded_weights import prior_state  # noqa: E402

SEED, N = 2024, 16


def main():
    t.set_grad_enabled(False)
    t.set_num_threads(int(os.environ.get("JB_GEN_THREADS", "6")))
    n_tok = int(os.environ.get("JB_GEN_TOKENS", "8192"))
    h = setup_hparams("small_prior", dict(restore_prior=""))
    assert (h.n_ctx, h.prior_width, h.prior_depth, h.heads, h.attn_order, h.blocks) == (8192, 1024, 48, 1, 2, 64)
    bins = 1024                                                        # small_vqvae.l_bins (hparams.py:191-208)
    p = ConditionalAutoregressive2D((h.n_ctx,), bins, width=h.prior_width, depth=h.prior_depth, heads=h.heads,
                                    attn_order=h.attn_order, blocks=h.blocks, init_scale=h.init_scale,
                                    m_attn=h.m_attn, m_mlp=h.m_mlp, x_cond=False, y_cond=False)
    p.eval()
    sd = prior_state(SEED, h.prior_width, h.prior_depth, bins, h.n_ctx)
    missing = p.load_state_dict({k: t.from_numpy(v) for k, v in sd.items()}, strict=True)
    print("loaded", len(sd), "tensors", missing, flush=True)
    t0 = time.time()
    stop = threading.Event()

    def beat():
        while not stop.wait(120):
            print(f"  ... {time.time() - t0:6.0f} s", flush=True)
    threading.Thread(target=beat, daemon=True).start()
    z, preds = p.sample(N, None, None, fp16=False, temp=1.0, top_k=1, get_preds=True, sample_tokens=n_tok)
    stop.set()
    print(f"sampled {tuple(z.shape)} in {time.time() - t0:.0f} s", flush=True)
    top2 = preds.topk(2, dim=-1)
    assert (top2.indices[..., 0] == z.view(N, -1)).all(), "greedy token is the arg-max of the recorded logits"
    gap = (top2.values[..., 0] - top2.values[..., 1]).numpy().astype(np.float32)
    out = os.path.join(HERE, "small_prior_full.npz" if n_tok == 8192 else f"small_prior_{n_tok}.npz")
    np.savez_compressed(out, seed=np.int64(SEED), z=z.view(N, -1).numpy().astype(np.int16), gap=gap,
                        runner_up=top2.indices[..., 1].numpy().astype(np.int16),
                        top1=top2.values[..., 0].numpy().astype(np.float32),
                        first_logits=preds[:, :4].numpy().astype(np.float32))
    print("wrote", out, os.path.getsize(out) // 1024, "KiB; near-ties (gap < 1e-3):", int((gap < 1e-3).sum()))


if __name__ == "__main__":
    main()
