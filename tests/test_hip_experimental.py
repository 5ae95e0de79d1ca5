"""GPU suite, opt-in (JB_EXPERIMENTAL=1): paths that are in the tree but not yet the default.

jb_gemv_pair / jb_engine_cfg.fused_pairs: the decode step with 3 launches per layer."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("JB_EXPERIMENTAL") != "1", reason="experimental paths: set JB_EXPERIMENTAL=1")]

from conftest import load_golden, sub_state  # noqa: E402
from oracle import ops as O  # noqa: E402


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def h16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("N,K0,K1,J", [(16, 1920, 480, 1920), (16, 1920, 1920, 1440), (3, 64, 32, 48), (16, 2048, 2048, 1536)])
def test_gemv_pair(N, K0, K1, J):
    """(out_a, out_b) = (res + in1.Wa + ba, gelu(LN(out_a).W + b)) in one launch against the two-step computation with
    the reference's rounding points; repeated launches with advancing epochs; the error flag stays clear."""
    from jukebox_amd import _lib as L, hip_ops as H
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.default_rng(N + K0 + K1)
    f16 = torch.float16
    res = h16(rng.standard_normal((N, K0)).astype(np.float32) * 1.5 + rng.standard_normal((N, 1)).astype(np.float32))
    in1 = h16(rng.standard_normal((N, K1)).astype(np.float32))
    Wa = h16((rng.standard_normal((K1, K0)) / np.sqrt(K1)).astype(np.float32))
    ba = (0.1 * rng.standard_normal(K0)).astype(np.float32)
    W = h16((rng.standard_normal((K0, J)) / np.sqrt(K0)).astype(np.float32))
    b = (0.1 * rng.standard_normal(J)).astype(np.float32)
    g = (1 + 0.2 * rng.standard_normal(K0)).astype(np.float32)
    be = (0.2 * rng.standard_normal(K0)).astype(np.float32)
    want_a = h16(res + h16(in1 @ Wa + h16(ba)))
    want_b = O.quick_gelu(h16(h16(O.layer_norm(want_a, g, be)) @ W + h16(b)), fp16=True)
    folded = H.FoldedLN(dev(W), dev(b), dev(g), dev(be), f16)
    pair = H.FusedPair(folded, dev(Wa), dev(ba), f16)
    pwa = H.pack_conv1d_w(dev(Wa), f16)
    stats = torch.zeros((K0 // 16, 16, 2), dtype=torch.int64, device="cuda")
    epoch = torch.ones(1, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(3):
        got_a, got_b = H.gemv_pair(dev(in1, f16), pwa, dev(ba), dev(res, f16), dev(res, f16), folded, pair, stats, epoch, err,
                                   act=L.ACT_QUICK_GELU)
        torch.cuda.synchronize()
        assert int(err.item()) == 0
        ga, gb = got_a.float().cpu().numpy(), got_b.float().cpu().numpy()
        assert np.abs(ga - want_a).max() <= 2e-3 * max(1.0, np.abs(want_a).max())
        assert np.abs(gb - want_b).max() <= 6e-3 * max(1.0, np.abs(want_b).max())
        epoch += 1


def test_engine_fused_pairs_fp16():
    """The 3-launch-per-layer decode step against the default fp16 engine on a seeded model (n_state must be a multiple
    of 32, so the golden 64-wide model does not qualify): same logits within the fp16 band while the greedy streams
    agree; graph replay == eager launches; the error flag stays clear."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from tools.bench_engine import random_state
    from jukebox_amd.engine import PriorEngine
    cfg = dict(seq_len=96, bins=160, width=256, depth=5, heads=2, attn_order=2, blocks=8, y_cond=False)
    sd = random_state(cfg, torch.device("cuda"), scale=0.05)
    for d in range(cfg["depth"]):                       # non-trivial LayerNorm parameters and biases
        for ln in ("ln_0", "ln_1"):
            sd[f"transformer._attn_mods.{d}.{ln}.weight"] = 1 + 0.2 * torch.randn(256, device="cuda")
            sd[f"transformer._attn_mods.{d}.{ln}.bias"] = 0.1 * torch.randn(256, device="cuda")
        for nm, n in (("attn.c_attn.b", 192), ("attn.c_proj.b", 256), ("mlp.c_fc.b", 256), ("mlp.c_proj.b", 256)):
            sd[f"transformer._attn_mods.{d}.{nm}"] = 0.05 * torch.randn(n, device="cuda")
    outs = []
    for fused in (False, True):
        eng = PriorEngine(sd, "", n_batch=5, fp16=True, want_preds=True, fused_pairs=fused, **cfg)
        assert eng.fused_pairs == fused and eng.launches_per_step == (3 * 5 + 6 if fused else 5 * 5 + 5)
        eng.set_cond(None, None)
        eng.set_sampling(temp=1.0, top_k=1)
        for use_graph in (False, True):
            eng.decode(0, 96, use_graph=use_graph)
            torch.cuda.synchronize()
            assert int(eng.pair_error.item()) == 0
            outs.append((eng.preds.cpu().numpy().copy(), eng.tokens.cpu().numpy().copy()))
    (p0, z0), (_, z0g), (p1, z1), (p1g, z1g) = outs
    assert np.array_equal(z0, z0g) and np.array_equal(z1, z1g) and np.array_equal(p1, p1g)      # graph == eager
    diverged = (z0 != z1).any(0)
    n_ok = int(np.argmax(diverged)) if diverged.any() else 96
    assert n_ok >= 8 and np.abs(p0[:, :n_ok] - p1[:, :n_ok]).max() < 3e-2 * max(1.0, np.abs(p0).max())
    assert (z0 == z1).mean() > 0.8


@pytest.mark.parametrize("func", [0, 1, 3, 7])
@pytest.mark.parametrize("H_,d,bc", [(1, 480, 128), (2, 64, 8), (1, 120, 8)])
def test_attn_prefill_v2(func, H_, d, bc):
    """jb_tune_attn_prefill_v2(1): 4-wave workgroups sharing vector-staged K/V tiles, against fp32 math on the half
    operands (same bar as tests/test_hip_kernels.py::test_attn_prefill)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from jukebox_amd import _lib as L, hip_ops as H
    from test_hip_kernels import _np_attention
    rng = np.random.default_rng(func * 10 + d + 1)
    N, prime_r = 2, 24
    T = 520 if bc == 128 else 120
    S = H_ * d
    cap = prime_r if func == 7 else T
    K = h16(rng.standard_normal((N, cap, S)).astype(np.float32))
    V = h16(rng.standard_normal((N, cap, S)).astype(np.float32))
    kc, vc = dev(K, torch.float16), dev(V, torch.float16)
    spans = ((0, 40), (0, 1), (5, 7), (13, 50), (64, 56), (37, 83)) if bc == 8 else ((0, 512), (100, 300), (384, 136), (7, 65))
    L.lib().jb_tune_attn_prefill_v2(1)
    try:
        for t0, nq in spans:
            q = h16(rng.standard_normal((N, nq, S)).astype(np.float32))
            got = H.attn_prefill(func, dev(q, torch.float16), kc, vc, H_, bc, t0).float().cpu().numpy()
            want = _np_attention(func, q, K, V, H_, bc, prime_r, list(range(t0, t0 + nq)), False)
            assert np.abs(got - want).max() < 6e-3 * max(1.0, np.abs(want).max()), (func, t0, nq)
    finally:
        L.lib().jb_tune_attn_prefill_v2(0)
