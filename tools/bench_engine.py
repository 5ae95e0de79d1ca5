"""Micro-benchmark of the decode engine at production dimensions with random weights:
ms per decode step (eager vs hipGraph) and achieved HBM GB/s against the algorithmic bytes of
SURVEY.md section 8d.  Usage: python tools/bench_engine.py [1b|up|small] [--steps K] [--t0 T] [--fp32]"""
import argparse
import os
import time

import torch

from jukebox_amd.engine import PriorEngine, attn_funcs

CFGS = {
    "1b": dict(seq_len=6528, bins=2127, width=2048, depth=72, heads=2, attn_order=12, blocks=64, prime_len=384, y_cond=True),
    "up": dict(seq_len=8192, bins=2048, width=1920, depth=72, heads=1, attn_order=2, blocks=128, y_cond=True),
    "5b": dict(seq_len=8192, bins=2048, width=4800, depth=79, heads=8, attn_order=10, blocks=128, y_cond=True,
               encoder_dims=512, add_cond_after=False),
    "small": dict(seq_len=8192, bins=1024, width=1024, depth=48, heads=1, attn_order=2, blocks=64, y_cond=False),
}


def random_state(cfg, dev, scale=0.02):
    W, D, B, T = cfg["width"], cfg["depth"], cfg["bins"], cfg["seq_len"]
    S = W // 4
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s, sc=scale: torch.randn(*s, device=dev, generator=g) * sc
    sd = {"x_emb.weight": r(B, W), "pos_emb.pos_emb": r(T, W, sc=0.01), "start_token": r(1, W, sc=0.01)}
    sd["x_out.weight"] = sd["x_emb.weight"]
    funcs = attn_funcs(cfg["attn_order"], D)
    for d in range(D):
        p = f"transformer._attn_mods.{d}."
        if funcs[d] == 6:
            sd[p + "attn.c_attn.w"], sd[p + "attn.c_attn.b"] = r(W, S), torch.zeros(S, device=dev)
            sd[p + "attn.c_enc_kv.w"], sd[p + "attn.c_enc_kv.b"] = r(W, 2 * S), torch.zeros(2 * S, device=dev)
        else:
            sd[p + "attn.c_attn.w"], sd[p + "attn.c_attn.b"] = r(W, 3 * S), torch.zeros(3 * S, device=dev)
        sd[p + "attn.c_proj.w"], sd[p + "attn.c_proj.b"] = r(S, W), torch.zeros(W, device=dev)
        sd[p + "mlp.c_fc.w"], sd[p + "mlp.c_fc.b"] = r(W, W), torch.zeros(W, device=dev)
        sd[p + "mlp.c_proj.w"], sd[p + "mlp.c_proj.b"] = r(W, W), torch.zeros(W, device=dev)
        for ln in ("ln_0", "ln_1"):
            sd[p + ln + ".weight"], sd[p + ln + ".bias"] = torch.ones(W, device=dev), torch.zeros(W, device=dev)
    return sd


def step_bytes(cfg, N, t, esz):
    """Algorithmic HBM bytes of one decode step at position t (weights once + k/v rows read + k/v written + logits head)."""
    W, D, B = cfg["width"], cfg["depth"], cfg["bins"]
    S, M = W // 4, W
    bc = cfg["seq_len"] // cfg["blocks"] if cfg.get("blocks") else 0
    pl = (cfg["prime_len"] // cfg["blocks"] + 1) * cfg["blocks"] if cfg.get("prime_len") else 0
    per_layer_w = (3 * W * S + S * W + 2 * W * M) * esz
    kv = 0
    for f in attn_funcs(cfg["attn_order"], D):
        ln = {0: t + 1, 1: t % bc + 1 if bc else 0, 2: t // bc + 1 if bc else 0, 3: bc if bc and t >= bc else 0,
              7: min(t + 1, pl), 6: cfg.get("encoder_dims", 0)}[f]
        kv += ln * S * 2 * esz * N + S * 2 * esz * N
    return D * per_layer_w + kv + B * W * 4


def report_stamps(eng, indent="    "):
    """JB_PIPE_DEBUG=1: where the last pipelined step spent its time, per kind of launch (workgroup 0's stamps)."""
    import numpy as np
    st = eng.pipe_stamps().astype(np.float64) * 0.01          # us: poll entered, producer seen, published (last workgroup), stores issued
    n = st.shape[0]
    entered, seen, published, issued = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
    per = getattr(eng, "_launches_per_layer", 4)
    kinds = ["c_attn", "attention", "c_fc", "c_proj"] if per == 4 else ["c_attn", "attention", "attn.c_proj", "c_fc", "mlp.c_proj"]
    for k in range(per):
        idx = np.arange(2 * per + k, n - 3, per)
        print(f"{indent}{kinds[k]:10s} consumer sees it - it saw its producer {np.mean(seen[idx + 1] - seen[idx]):6.2f} us = inputs seen -> "
              f"stores issued {np.mean(issued[idx] - seen[idx]):5.2f} + drain, tickets {np.mean(published[idx] - issued[idx]):5.2f} + "
              f"propagation {np.mean(seen[idx + 1] - published[idx]):5.2f} | poll entered {np.mean(seen[idx] - entered[idx]):5.2f} before the "
              f"producer was seen, {np.mean(entered[idx + 2] - published[idx]):6.2f} after its same-stream predecessor published")
    print(f"{indent}whole step by the stamps: {seen[n - 1] - seen[1]:8.1f} us for slots 1..{n - 1}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="up")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--t0", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--prio", type=int, default=None, help="run on a torch stream of this priority (-1 = high) instead of the default stream")
    ap.add_argument("--plain-first", type=int, default=0, help="run this many plain graph-replayed steps before switching the engine to pipelined launches")
    ap.add_argument("--noise-streams", type=int, default=0, help="touch this many extra torch streams first (hardware-queue pressure)")
    ap.add_argument("--pipelined", type=int, default=-1, help="1 / 0: software-pipelined launches on / off (default: the engine's choice)")
    ap.add_argument("--two-streams", action="store_true", help="single-head engines: the two-stream form of the pipelined step (jb_tune_pipeline(3)), not the three-stream one")
    ap.add_argument("--long-rows", action="store_true", help="5b: projections over 129..160 k-tiles on the 8-wave two-stage kernel (jb_tune_gemv_long(1))")
    ap.add_argument("--fat-attention", action="store_true", help="the fat form of the 480-channel wide-value attention kernel (jb_tune_attn_decode_wide_lean(0))")
    a = ap.parse_args()
    cfg = CFGS[a.model]
    dev = torch.device("cuda:0")
    from jukebox_amd import _lib as L
    if a.fat_attention:
        L.lib().jb_tune_attn_decode_wide_lean(0)
    if a.long_rows:
        L.lib().jb_tune_gemv_long(1)
    if a.two_streams:
        L.lib().jb_tune_pipeline(3)
    sd = random_state(cfg, dev)
    eng = PriorEngine(sd, "", n_batch=a.batch, fp16=not a.fp32, chunk_cap=64, **cfg)
    y = torch.randn(a.batch, 1, cfg["width"], device=dev) * 0.01 if cfg["y_cond"] else None
    xc = torch.randn(a.batch, cfg["seq_len"], cfg["width"], device=dev) * 0.01
    eng.set_cond(xc, y)
    eng.set_sampling(temp=0.99, seed=1)
    if cfg.get("encoder_dims"):
        eng.set_encoder_kv(torch.randn(a.batch, cfg["encoder_dims"], cfg["width"], device=dev) * 0.1)
    del sd
    if a.plain_first:
        eng.set_pipelined(False)
        eng.decode(a.t0, a.plain_first, use_graph=True)
        torch.cuda.synchronize()
    if a.pipelined >= 0:
        eng.set_pipelined(bool(a.pipelined))
    torch.cuda.synchronize()
    eng._launches_per_layer = (eng.launches_per_step - 2) // cfg["depth"]
    print(f"model={a.model} N={a.batch} dtype={'f32' if a.fp32 else 'f16'} weights={eng.weight_bytes() / 1e9:.2f} GB "
          f"kv={eng.cache_bytes() / 1e9:.2f} GB launches/step={eng.launches_per_step} pipelined={eng.pipelined}")
    noise = []
    for i in range(a.noise_streams):
        st = torch.cuda.Stream(device=dev, priority=-1 if i % 2 == 0 else 0)
        with torch.cuda.stream(st):
            noise.append((st, torch.zeros(1024, device=dev).add_(1)))
    torch.cuda.synchronize()
    if a.prio is not None:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=a.prio))
    for use_graph in ([False] if a.eager else [False, True]):
        eng.decode(a.t0, 8, use_graph=use_graph)      # warm-up (+ capture)
        torch.cuda.synchronize()
        t = time.perf_counter()
        eng.decode(a.t0, a.steps, use_graph=use_graph)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / a.steps
        b = eng.step_bytes(a.t0 + a.steps // 2)
        if use_graph and eng.pipelined and eng.pipe_error():
            print("  !! pipelined launch timed out waiting for slot", eng.pipe_error() - 1)
        if use_graph and eng.pipelined and os.environ.get("JB_PIPE_DEBUG"):
            report_stamps(eng)
        print(f"  graph={use_graph}: {dt * 1e3:.3f} ms/step  algorithmic {b / 1e9:.3f} GB/step -> {b / dt / 1e12:.2f} TB/s "
              f"({b / dt / 8e12 * 100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
