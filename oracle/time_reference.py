"""CPU timing of the UNMODIFIED reference for bench.py's `cpu_baseline` leg (kind "reference").  Test infrastructure:
run as a subprocess by bench.py (the import shim monkey-patches torch, so it must not share a process with the GPU
path); never imported by `jukebox_amd/`.

    python oracle/time_reference.py [--budget-s 45] [--batch 16] [--seconds 20]

What is timed, on this box's host cores, fp32, batch 16: the reference's own modules (`jukebox.make_models.make_prior`
-> `SimplePrior.prior` = `ConditionalAutoregressive2D`, `Transformer`, `FactoredAttention`) from the snapshot under
`oracle/_ref` (made by oracle/make_ref.py) or `/root/reference`, under the shim of tests/golden/refshim.py.
  * decode: the body of the token loop of `ConditionalAutoregressive2D.sample` (jukebox/prior/autoregressive.py:222-236:
    get_emb, check_cache, transformer(sample=True), x_out, temperature, filter_logits, Categorical.sample) of the level-0
    upsampler at positions t in {0, 1/4, 1/2, 3/4} * n_ctx.  The reference can only reach position t by running t steps,
    so for t > 0 every layer's cache is set to the state the reference itself would hold there (`sample_t = t`, key /
    value of `_suff_cache_len()` rows, factored_attention.py:328-353) filled with random values -- timing does not
    depend on the values.
  * prefill: one 32-token chunk of `primed_sample`'s chunked conditioning pass (autoregressive.py:289-318,
    chunk_size=32 as in sample.py:233) at mid-window.
These are integrated over the window plan of the bench workload (decode steps and primed tokens per level for the
given audio length; the top level is charged at the upsampler's per-step cost, which under-charges the CPU: it is 2048
wide instead of 1920), the conditioner and VQ-VAE conv stacks are not charged.  The result is an integrated ESTIMATE of
the reference's audio-seconds per second, as SURVEY.md section 8d prescribes for configs that cannot finish on a CPU.
Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def window_plan(total, n_ctx, hop):
    """(decode steps, primed tokens) of one level: first window from scratch, later windows primed with n_ctx - hop
    tokens (jukebox/utils/sample_utils.py:14-22, jukebox/sample.py:31-78)."""
    if total < n_ctx:
        return total, 0
    starts = []
    for start in range(0, total - n_ctx + hop, hop):
        if start + n_ctx >= total:
            start = total - n_ctx
        starts.append(start)
    decode, primed, have = 0, 0, 0
    for s in starts:
        new = s + n_ctx - have
        decode += new
        primed += n_ctx - new if have else 0
        have = s + n_ctx
    return decode, primed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget-s", type=float, default=45.0)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--sr", type=int, default=44100)
    ap.add_argument("--steps", type=int, default=4, help="decode steps per position (upper bound; the budget wins)")
    a = ap.parse_args()
    t_start = time.perf_counter()

    ref_root = os.path.join(HERE, "_ref")
    if not os.path.isdir(os.path.join(ref_root, "jukebox")):
        ref_root = os.environ.get("JUKEBOX_REFERENCE", "/root/reference")
    os.environ["JUKEBOX_REFERENCE"] = ref_root
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
    os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, ROOT)
    import refshim                     # reads JUKEBOX_REFERENCE (set above)
    refshim.install()
    import torch
    from jukebox.hparams import setup_hparams
    from jukebox.make_models import MODELS, make_prior, make_vqvae
    from jukebox.transformer.ops import filter_logits

    cores = os.cpu_count() or 1
    torch.manual_seed(0)
    torch.set_num_threads(min(cores, 32))          # model construction; the decode thread count is swept below
    log = lambda msg: print(f"[time_reference +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)
    sample_length = int(a.seconds * a.sr) // 128 * 128
    vq = make_vqvae(setup_hparams(MODELS["1b_lyrics"][0], dict(sample_length=sample_length, restore_vqvae="")), "cpu")
    prior = make_prior(setup_hparams("upsampler_level_0", dict(restore_prior="")), vq, "cpu")
    ar = prior.prior
    N, T, W = a.batch, ar.input_dims, ar.width
    build_s = time.perf_counter() - t_start
    log(f"reference upsampler built ({cores} cores available)")
    x_cond = torch.zeros(N, T, W)
    y_cond = torch.zeros(N, 1, W)
    g = torch.Generator().manual_seed(1)

    pool = {}                                   # one filled (N, T, n_state) buffer per distinct n_state: caches are views of it

    def set_position(t):
        """Every layer's sampling state as the reference holds it after t tokens."""
        ar.transformer.del_cache()
        if t == 0:
            return
        for blk in ar.transformer._attn_mods:
            att = blk.attn
            att.sample_t = t
            n = att._suff_cache_len()
            if att.n_state not in pool:
                pool[att.n_state] = torch.full((N, T, att.n_state), 0.01)   # values do not matter for timing
            # views: the reference's first _append_cache (t.cat) makes its own copy, exactly as in a real run
            att.cache["key"] = pool[att.n_state][:, :n]
            att.cache["value"] = pool[att.n_state][:, :n]

    def decode_steps(t0, n):
        """n iterations of the loop body of ConditionalAutoregressive2D.sample starting at position t0."""
        x = None if t0 == 0 else torch.randint(0, ar.bins, (N, 1), generator=g)
        for sample_t in range(t0, t0 + n):
            x, cond = ar.get_emb(sample_t, N, x, x_cond, y_cond)
            ar.transformer.check_cache(N, sample_t, False)
            x = ar.transformer(x, encoder_kv=None, sample=True, fp16=False)
            if ar.add_cond_after_transformer:
                x = x + cond
            x = ar.x_out(x)
            x = x / 0.99
            x = filter_logits(x, top_k=0, top_p=0.0)
            x = torch.distributions.Categorical(logits=x).sample()

    with torch.no_grad():
        # thread count: 16-row matmuls do not scale to hundreds of threads, and late in a window the reference's step is
        # dominated by its cache re-copies (t.cat) and the attention over thousands of keys, which may scale differently:
        # sweep at t = 1 AND at t = T/2, take the count with the smallest sum of the two
        set_position(0)
        decode_steps(0, 1)
        sweep, best, threads = {}, None, min(cores, 32)
        for nt in sorted({c for c in (8, 16, 32, 64) if c <= cores}):
            torch.set_num_threads(nt)
            set_position(0)
            decode_steps(0, 1)
            t0 = time.perf_counter()
            decode_steps(1, 2)
            dt_early = (time.perf_counter() - t0) / 2
            set_position(T // 2)
            decode_steps(T // 2, 1)                    # untimed: first touch of the caches
            t0 = time.perf_counter()
            decode_steps(T // 2 + 1, 1)
            dt_late = time.perf_counter() - t0
            sweep[nt] = (round(dt_early * 1e3, 1), round(dt_late * 1e3, 1))
            log(f"{nt} threads: {dt_early * 1e3:.0f} ms per decode step at t=1, {dt_late * 1e3:.0f} ms at t={T // 2}")
            if best is None or dt_early + dt_late < best:
                best, threads = dt_early + dt_late, nt
            if time.perf_counter() - t_start > 0.45 * a.budget_s:
                break
        torch.set_num_threads(threads)

        positions = [0, T // 4, T // 2, 3 * T // 4]
        per_pos_budget = max(1.0, (a.budget_s - (time.perf_counter() - t_start) - 3.0) / (len(positions) + 1))
        step_ms = {}
        for t in positions:
            set_position(t)
            decode_steps(t, 1)                         # untimed: first touch of the new caches
            n, t0 = 0, time.perf_counter()
            while n < a.steps and (n < 2 or time.perf_counter() - t0 < per_pos_budget):
                decode_steps(t + 1 + n, 1)
                n += 1
            step_ms[t] = (time.perf_counter() - t0) / n * 1e3
            log(f"t={int(t)}: {step_ms[t]:.0f} ms per step ({n} steps)")
        # one prefill chunk of primed_sample (chunk_size 32) at mid-window
        t = T // 2
        set_position(t)
        xp = torch.randn(N, 32, W, generator=g) * 0.02
        t0 = time.perf_counter()
        ar.transformer.check_cache(N, t, False)
        ar.transformer(xp, encoder_kv=None, sample=True, fp16=False)
        chunk_ms = (time.perf_counter() - t0) * 1e3
        ar.transformer.del_cache()

    # integrate over the workload: tokens per level, windows and hops as bench.py runs them
    hops = {0: 0.5, 1: 0.5, 2: 0.125}
    n_ctx = {0: 8192, 1: 8192, 2: 6144}
    raw_to_tokens = {0: 8, 1: 32, 2: 128}
    decode_total, primed_total = 0, 0
    for level in (0, 1, 2):
        d, p = window_plan(sample_length // raw_to_tokens[level], n_ctx[level], int(hops[level] * n_ctx[level]))
        decode_total += d
        primed_total += p
    mean_step_s = sum(step_ms.values()) / len(step_ms) / 1e3
    cpu_seconds = decode_total * mean_step_s + (primed_total / 32.0) * chunk_ms / 1e3
    audio_seconds = N * sample_length / a.sr
    out = dict(value=audio_seconds / cpu_seconds, unit="audio_s/s", cores=int(threads), cores_available=int(cores),
               kind="reference",
               sample=(f"unmodified reference modules ({'oracle/_ref' if ref_root.endswith('_ref') else ref_root}) under the CPU shim, "
                       f"fp32, batch {N}: decode-loop body of ConditionalAutoregressive2D.sample for the level-0 upsampler at "
                       f"t in {[int(p) for p in positions]} ({', '.join(f'{v:.1f}' for v in step_ms.values())} ms/step) and one 32-token "
                       f"primed_sample chunk at t={T // 2} ({chunk_ms:.0f} ms); integrated over {decode_total} decode steps + "
                       f"{primed_total} primed tokens of the {a.seconds:g}-s 3-level job (conditioner / VQ-VAE not charged)"),
               thread_sweep_ms_at_t1_and_mid_window={str(k): v for k, v in sweep.items()},
               ms_per_decode_step=round(mean_step_s * 1e3, 2), ms_per_prefill_chunk32=round(chunk_ms, 1),
               decode_steps=decode_total, primed_tokens=primed_total, build_s=round(build_s, 1),
               cpu_seconds_estimated=round(cpu_seconds, 1), wall_s=round(time.perf_counter() - t_start, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
