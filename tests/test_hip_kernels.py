"""GPU suite, kernel level: every C-ABI entry point against a numpy restatement on seeded inputs.
Tolerances: fp32 paths 2e-5 relative to the output scale (summation order only); fp16 paths are
checked against fp32 math on the half-rounded operands with the half-output rounding bound."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402
from oracle.transformer import allowed_keys, decode_key_index  # noqa: E402
from oracle.vqvae import conv1d_nct, conv_transpose1d_nct  # noqa: E402


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from jukebox_amd import hip_ops
    from jukebox_amd import _lib
    _lib.lib()          # fail loudly if the extension is missing
    return hip_ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def h16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def relerr(got, want):
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))


DT = [("f32", torch.float32, 2e-5), ("f16", torch.float16, 3e-3)]


@pytest.mark.parametrize("name,dt,tol", DT)
@pytest.mark.parametrize("M,K,J", [(300, 70, 50), (16, 32, 16), (1000, 257, 130), (5, 3, 1)])
def test_gemm_plain(H, name, dt, tol, M, K, J):
    rng = np.random.default_rng(M + K + J)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, J)).astype(np.float32)
    b = rng.standard_normal(J).astype(np.float32)
    if dt == torch.float16:
        A, W, bq = h16(A), h16(W), h16(b)
    else:
        bq = b
    pw = H.pack_conv1d_w(dev(W), dt)
    out = H.gemm(dev(A, dt), pw, bias=dev(b)).float().cpu().numpy()
    want = A @ W + bq
    assert relerr(out, want) < tol


@pytest.mark.parametrize("name,dt,tol", DT)
def test_gemm_epilogues(H, name, dt, tol):
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(1)
    M, K, J = 200, 64, 48
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (0.2 * rng.standard_normal((K, J))).astype(np.float32)
    b = rng.standard_normal(J).astype(np.float32)
    R = rng.standard_normal((M, J)).astype(np.float32)
    f16 = dt == torch.float16
    if f16:
        A, W, R = h16(A), h16(W), h16(R)
    r = (lambda x: h16(x)) if f16 else (lambda x: x)
    pw = H.pack_conv1d_w(dev(W), dt)
    # quick_gelu
    got = H.gemm(dev(A, dt), pw, bias=dev(b), act=L.ACT_QUICK_GELU).float().cpu().numpy()
    want = O.quick_gelu(r(A @ W + r(b)), fp16=f16)
    assert relerr(got, want) < tol
    # residual (res_scale 1 and 0.25), input relu
    got = H.gemm(dev(A, dt), pw, bias=dev(b), res=dev(R, dt), pre_relu=True).float().cpu().numpy()
    want = r(R + r(np.maximum(A, 0) @ W + r(b)))
    assert relerr(got, want) < tol
    got = H.gemm(dev(A, dt), pw, bias=dev(b), res=dev(R, dt), res_scale=0.25, act=L.ACT_RELU).float().cpu().numpy()
    want = r(R + r(0.25 * np.maximum(r(A @ W + r(b)), 0)))
    assert relerr(got, want) < tol


@pytest.mark.parametrize("name,dt,tol", DT)
@pytest.mark.parametrize("M,K,J", [(1300, 256, 264), (600, 96, 72), (256, 32, 128), (2048, 1920, 480)])
def test_gemm_lds_flat(H, name, dt, tol, M, K, J):
    """The LDS-staged kernel jb_gemm takes for flat problems (prefill projections), forced on at any size: row and
    column tails, every epilogue, and the q / k-cache / v-cache split at a cache offset."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(M + K)
    f16 = dt == torch.float16
    r = (lambda x: h16(x)) if f16 else (lambda x: x)
    A = r(rng.standard_normal((M, K)).astype(np.float32))
    W = r((rng.standard_normal((K, J)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(J).astype(np.float32)
    R = r(rng.standard_normal((M, J)).astype(np.float32))
    pw = H.pack_conv1d_w(dev(W), dt)
    L.lib().jb_tune_gemm_lds(1)
    L.lib().jb_tune_gemm_glds(-1)                         # this test is about the register-staged kernel
    try:
        got = H.gemm(dev(A, dt), pw, bias=dev(b), act=L.ACT_QUICK_GELU).float().cpu().numpy()
        assert relerr(got, O.quick_gelu(r(A @ W + r(b)), fp16=f16)) < tol
        got = H.gemm(dev(A, dt), pw, bias=dev(b), res=dev(R, dt)).float().cpu().numpy()
        assert relerr(got, r(R + r(A @ W + r(b)))) < tol
        L.lib().jb_tune_gemm_lds(-1)                      # same call through the direct-from-L1 kernel
        old = H.gemm(dev(A, dt), pw, bias=dev(b), res=dev(R, dt)).float().cpu().numpy()
        assert np.array_equal(got, old)                   # same k order, same rounding points
        L.lib().jb_tune_gemm_lds(1)
        if J % 3 == 0:
            n_seq, S, cap, t0 = 4, J // 3, M // 4 + 9, 5
            kc = torch.zeros((n_seq, cap, S), dtype=dt, device="cuda")
            vc = torch.zeros((n_seq, cap, S), dtype=dt, device="cuda")
            q = H.gemm_qkv(dev(A, dt), pw, dev(b), n_seq, M // 4, S, kc, vc, t0).float().cpu().numpy()
            want = r(A @ W + r(b))
            assert relerr(q, want[:, :S]) < tol
            kcn = kc.float().cpu().numpy()[:, t0:t0 + M // 4].reshape(M, S)
            vcn = vc.float().cpu().numpy()[:, t0:t0 + M // 4].reshape(M, S)
            assert relerr(kcn, want[:, S:2 * S]) < tol and relerr(vcn, want[:, 2 * S:]) < tol
            assert float(kc[:, :t0].abs().max()) == 0 and float(kc[:, t0 + M // 4:].abs().max()) == 0
    finally:
        L.lib().jb_tune_gemm_lds(1024)
        L.lib().jb_tune_gemm_glds(256)


@pytest.mark.parametrize("M,K,J", [(1300, 256, 264), (600, 96, 72), (256, 32, 128), (2048, 1920, 480), (4096, 480, 1920),
                                   (3000, 160, 200), (8192, 1920, 1440)])
def test_gemm_glds(H, M, K, J):
    """The prefill's fp16 GEMM since round 4 (gemm_glds_kernel: 128 x 128 tile, both operands by LDS-DMA into a swizzled LDS
    image): bit-identical to the register-staged LDS kernel and to the direct-from-L1 kernel (same MFMA, same k order per
    output element, same rounding points) on ragged rows / columns, odd and even numbers of k-tiles, with every epilogue
    (bias + quick_gelu, bias + residual, the q / k-cache / v-cache split) and with sequences that are NOT packed in the input
    (rows of a sequence contiguous, sequences a pitch apart: the per-sample v.Wp of the wide-value cache as one launch)."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(M + K + J)
    dt = torch.float16
    A = h16(rng.standard_normal((M, K)).astype(np.float32))
    W = h16((rng.standard_normal((K, J)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(J).astype(np.float32)
    R = h16(rng.standard_normal((M, J)).astype(np.float32))
    pw = H.pack_conv1d_w(dev(W), dt)
    Ad, Rd, bd = dev(A, dt), dev(R, dt), dev(b)
    n_seq, t = 4, M // 4
    pitch = t + 7
    Ap = torch.zeros((n_seq * pitch, K), dtype=dt, device="cuda")
    Ap.view(n_seq, pitch, K)[:, :t] = Ad[:n_seq * t].view(n_seq, t, K)

    def run():
        out = [H.gemm(Ad, pw, bias=bd, act=L.ACT_QUICK_GELU), H.gemm(Ad, pw, bias=bd, res=Rd),
               H.gemm(Ap, pw, n_seq=n_seq, t_in=t, in_seq_pitch=pitch)]
        if J % 3 == 0:
            S, cap, t0 = J // 3, t + 9, 5
            kc = torch.zeros((n_seq, cap, S), dtype=dt, device="cuda")
            vc = torch.zeros((n_seq, cap, S), dtype=dt, device="cuda")
            out += [H.gemm_qkv(Ad[:n_seq * t], pw, bd, n_seq, t, S, kc, vc, t0), kc, vc]
        return [o.float().cpu().numpy() for o in out]

    try:
        L.lib().jb_tune_gemm_glds(1)
        new = run()
        L.lib().jb_tune_gemm_glds(-1)
        L.lib().jb_tune_gemm_lds(1)
        lds = run()
        L.lib().jb_tune_gemm_lds(-1)
        l1 = run()
    finally:
        L.lib().jb_tune_gemm_lds(1024)
        L.lib().jb_tune_gemm_glds(256)
    for a, b_, c in zip(new, lds, l1):
        assert np.array_equal(a, b_) and np.array_equal(a, c)
    assert relerr(new[0], O.quick_gelu(h16(A @ W + h16(b)), fp16=True)) < 4e-3
    assert relerr(new[1], h16(R + h16(A @ W + h16(b)))) < 4e-3
    assert relerr(new[2], h16(A[:n_seq * t] @ W)) < 4e-3


@pytest.mark.parametrize("M,K,J", [(1300, 256, 264), (4096, 1920, 1440), (2048, 1920, 480), (777, 128, 1000), (8192, 1920, 1920),
                                   (3000, 384, 200), (4096, 480, 1920), (1000, 160, 200), (2500, 224, 264), (700, 4800, 520)])
def test_gemm_8phase(H, M, K, J):
    """The prefill's big projections since round 5 (gemm_8phase_kernel: 256 x 256 tile on 8 waves, two LDS stages of four
    half-tiles filled by LDS-DMA, the two wave rows half a phase apart; K >= 128, the k-tiles beyond a multiple of 128 straight
    from L1 behind the loop: K = 480, 160, 224, 4800): bit-identical to gemm_glds_kernel
    and to the register-staged kernel on ragged rows / columns (partial tiles in both directions, fewer tiles than compute
    units), one to fifteen loop iterations, every epilogue, sequences a pitch apart -- and the same answer on every one of
    several launches (a missed ordering between an LDS-DMA request and an operand read shows as a rare wrong tile)."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(M + K + J)
    dt = torch.float16
    A = h16(rng.standard_normal((M, K)).astype(np.float32))
    W = h16((rng.standard_normal((K, J)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(J).astype(np.float32)
    R = h16(rng.standard_normal((M, J)).astype(np.float32))
    pw = H.pack_conv1d_w(dev(W), dt)
    Ad, Rd, bd = dev(A, dt), dev(R, dt), dev(b)
    n_seq, t = 4, M // 4
    pitch = t + 7
    Ap = torch.zeros((n_seq * pitch, K), dtype=dt, device="cuda")
    Ap.view(n_seq, pitch, K)[:, :t] = Ad[:n_seq * t].view(n_seq, t, K)

    def run():
        out = [H.gemm(Ad, pw, bias=bd, act=L.ACT_QUICK_GELU), H.gemm(Ad, pw, bias=bd, res=Rd),
               H.gemm(Ap, pw, n_seq=n_seq, t_in=t, in_seq_pitch=pitch)]
        if J % 3 == 0:
            S, cap, t0 = J // 3, t + 9, 5
            kc = torch.zeros((n_seq, cap, S), dtype=dt, device="cuda")
            vc = torch.zeros((n_seq, cap, S), dtype=dt, device="cuda")
            out += [H.gemm_qkv(Ad[:n_seq * t], pw, bd, n_seq, t, S, kc, vc, t0), kc, vc]
        return [o.float().cpu().numpy() for o in out]

    try:
        L.lib().jb_tune_gemm_glds(1)
        L.lib().jb_tune_gemm_8phase(1)
        new = [run() for _ in range(4)]
        L.lib().jb_tune_gemm_8phase(-1)
        glds = run()
        L.lib().jb_tune_gemm_glds(-1)
        L.lib().jb_tune_gemm_lds(1)
        lds = run()
    finally:
        L.lib().jb_tune_gemm_lds(1024)
        L.lib().jb_tune_gemm_glds(256)
        L.lib().jb_tune_gemm_8phase(512)
    for rep in new:
        for a, b_, c in zip(rep, glds, lds):
            assert np.array_equal(a, b_) and np.array_equal(a, c)
    assert relerr(new[0][0], O.quick_gelu(h16(A @ W + h16(b)), fp16=True)) < 4e-3
    assert relerr(new[0][1], h16(R + h16(A @ W + h16(b)))) < 4e-3
    assert relerr(new[0][2], h16(A[:n_seq * t] @ W)) < 4e-3


def test_conv_stack_ops_fp32(H):
    """Dilated k=3 conv, strided k=4 conv and transposed conv on channels-last rows vs torch-free numpy NCT."""
    rng = np.random.default_rng(2)
    N, Ci, Co, T = 3, 24, 40, 200
    x = rng.standard_normal((N, Ci, T)).astype(np.float32)
    xr = dev(np.transpose(x, (0, 2, 1)).reshape(N * T, Ci))
    for dil in (1, 3, 27, 243):
        w = (0.2 * rng.standard_normal((Co, Ci, 3))).astype(np.float32)
        b = rng.standard_normal(Co).astype(np.float32)
        want = conv1d_nct(x, w, b, 1, dil, dil)
        pw = H.pack_conv_taps(dev(w), torch.float32)
        got = H.gemm(xr, pw, bias=dev(b), n_seq=N, t_in=T, shifts=(-dil, 0, dil)).cpu().numpy()
        got = np.transpose(got.reshape(N, T, Co), (0, 2, 1))
        assert relerr(got, want) < 2e-5, dil
    # strided conv k=4 s=2 p=1 (EncoderConvBlock)
    w = (0.2 * rng.standard_normal((Co, Ci, 4))).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    want = conv1d_nct(x, w, b, 2, 1, 1)
    pw = H.pack_conv_taps(dev(w), torch.float32)
    got = H.gemm(xr, pw, bias=dev(b), n_seq=N, t_in=T, t_out=T // 2, in_stride=2, shifts=(-1, 0, 1, 2)).cpu().numpy()
    got = np.transpose(got.reshape(N, T // 2, Co), (0, 2, 1))
    assert relerr(got, want) < 2e-5
    # transposed conv k=4 s=2 p=1 (DecoderConvBock): even outputs taps (1,3) at shifts (0,-1); odd taps (0,2) at (+1,0)
    w = (0.2 * rng.standard_normal((Ci, Co, 4))).astype(np.float32)
    want = conv_transpose1d_nct(x, w, b, 2, 1)
    pw = H.pack_conv_taps(dev(w), torch.float32, transposed=True)
    out = torch.empty((N * 2 * T, Co), dtype=torch.float32, device="cuda")
    H.gemm(xr, H.tap_view(pw, [1, 3]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(0, -1),
           out_stride=2, out_offset=0, out_rows_per_seq=2 * T)
    H.gemm(xr, H.tap_view(pw, [0, 2]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(1, 0),
           out_stride=2, out_offset=1, out_rows_per_seq=2 * T)
    got = np.transpose(out.cpu().numpy().reshape(N, 2 * T, Co), (0, 2, 1))
    assert relerr(got, want) < 2e-5


def _conv1d_f64(x, w, b, stride, padding, dilation):
    """conv1d_nct's arithmetic (oracle/vqvae.py) in float64: the yardstick for kernels that claim fp32 accuracy."""
    N, Ci, T = x.shape
    Co, _, k = w.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (padding, padding)))
    To = (T + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    out = np.zeros((N, Co, To))
    for tap in range(k):
        out += np.matmul(w[:, :, tap].astype(np.float64), xp[:, :, tap * dilation: tap * dilation + (To - 1) * stride + 1: stride])
    return out + b.astype(np.float64)[None, :, None]


def _conv_transpose1d_f64(x, w, b, stride, padding):
    N, Ci, T = x.shape
    _, Co, k = w.shape
    full = np.zeros((N, Co, (T - 1) * stride + k))
    for tap in range(k):
        full[:, :, tap: tap + (T - 1) * stride + 1: stride] += np.matmul(w[:, :, tap].T.astype(np.float64), x.astype(np.float64))
    To = (T - 1) * stride - 2 * padding + k
    return full[:, :, padding: padding + To] + b.astype(np.float64)[None, :, None]


@pytest.mark.parametrize("Ci,Co,T,scale", [(64, 40, 200, 1.0), (1024, 136, 300, 1.0), (96, 72, 130, 3000.0), (32, 16, 70, 1e-3)])
def test_conv_stack_ops_f16_split(H, Ci, Co, T, scale):
    """The fp32 conv stacks on the f16 matrix cores (jb_gemm_args.w_split: both operands as hi + 2^-11 lo pairs of halves,
    three f16 MFMAs per k-tile, fp32 accumulation) against float64 arithmetic and against the exact-fp32 kernel: dilated
    k=3 convolutions in the Resnet1D form (input ReLU, residual, dilations beyond the sequence), the strided and the
    transposed convolution, ragged row / column tiles, activations of a few thousand and of a few thousandths (the low
    halves go subnormal there).  Bar: within 2e-6 of the output scale of float64 -- the accumulation's own rounding, as for
    the exact-fp32 kernel (first GPU run: 1.0e-7 ... 5.8e-7, profiles/r04_split_kernel_tests.log)."""
    rng = np.random.default_rng(Ci + Co)
    N = 3
    x = (scale * rng.standard_normal((N, Ci, T))).astype(np.float32)
    xr = dev(np.transpose(x, (0, 2, 1)).reshape(N * T, Ci))
    x64 = x.astype(np.float64)
    worst = 0.0
    for dil in (1, 9, 243):
        w = (rng.standard_normal((Co, Ci, 3)) / np.sqrt(3 * Ci)).astype(np.float32)
        b = rng.standard_normal(Co).astype(np.float32)
        R = (scale * rng.standard_normal((N * T, Co))).astype(np.float32)
        want = _conv1d_f64(np.maximum(x64, 0), w, b, 1, dil, dil)
        want = R.astype(np.float64) + 0.5 * np.transpose(want, (0, 2, 1)).reshape(N * T, Co)
        outs = []
        for split in (True, False):
            pw = H.pack_conv_taps(dev(w), torch.float32, split=split)
            assert pw.split == split and pw.data.numel() == H.pack_conv_taps(dev(w), torch.float32).data.numel()
            outs.append(H.gemm(xr, pw, bias=dev(b), res=dev(R), res_scale=0.5, n_seq=N, t_in=T, shifts=(-dil, 0, dil),
                               pre_relu=True).cpu().numpy().astype(np.float64))
        e_split, e_exact = relerr(outs[0], want), relerr(outs[1], want)
        worst = max(worst, e_split)
        assert e_split < 2e-6 and e_exact < 2e-6, (dil, e_split, e_exact)
    # strided conv k=4 s=2 p=1
    w = (rng.standard_normal((Co, Ci, 4)) / np.sqrt(4 * Ci)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    want = _conv1d_f64(x64, w, b, 2, 1, 1)
    pw = H.pack_conv_taps(dev(w), torch.float32, split=True)
    got = H.gemm(xr, pw, bias=dev(b), n_seq=N, t_in=T, t_out=T // 2, in_stride=2, shifts=(-1, 0, 1, 2)).cpu().numpy()
    got = np.transpose(got.reshape(N, T // 2, Co), (0, 2, 1))
    worst = max(worst, relerr(got, want))
    assert relerr(got, want) < 2e-6
    # transposed conv k=4 s=2 p=1 through tap views of the split image
    w = (rng.standard_normal((Ci, Co, 4)) / np.sqrt(2 * Ci)).astype(np.float32)
    want = _conv_transpose1d_f64(x64, w, b, 2, 1)
    pw = H.pack_conv_taps(dev(w), torch.float32, transposed=True, split=True)
    out = torch.empty((N * 2 * T, Co), dtype=torch.float32, device="cuda")
    H.gemm(xr, H.tap_view(pw, [1, 3]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(0, -1),
           out_stride=2, out_offset=0, out_rows_per_seq=2 * T)
    H.gemm(xr, H.tap_view(pw, [0, 2]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(1, 0),
           out_stride=2, out_offset=1, out_rows_per_seq=2 * T)
    got = np.transpose(out.cpu().numpy().reshape(N, 2 * T, Co), (0, 2, 1))
    worst = max(worst, relerr(got, want))
    assert relerr(got, want) < 2e-6
    print("f16-split conv stack Ci=%d scale=%g: worst error %.2e of the output scale" % (Ci, scale, worst))


@pytest.mark.parametrize("Ci,Co,T", [(64, 72, 150), (1024, 256, 520)])
def test_conv_stack_presplit_equals_the_split_inside_the_gemm(H, Ci, Co, T):
    """jb_gemm_args.a_split (round 5): the activations of a w_split problem split once, by a pass of their own, into f16 hi / lo
    images that the GEMM loads as finished operands.  Same operands, same MFMAs in the same order: the outputs must EQUAL those
    of the split inside the GEMM bit for bit -- dilated k=3 with the input ReLU and a residual (dilations beyond the
    sequence), the strided and the transposed convolution, sequences a pitch apart with rubbish between them (which the pass
    must not look at: it would raise the range flag) -- and the range flag must come up from the pass as it does from the GEMM
    (a value beyond the half range; a NaN under the input ReLU)."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(Ci + T)
    N = 3
    x = (3.0 * rng.standard_normal((N * T, Ci))).astype(np.float32)
    xr = dev(x)
    pitch = T + 5
    xp = torch.full((N * pitch, Ci), float("nan"), dtype=torch.float32, device="cuda")       # rubbish between the sequences
    xp.view(N, pitch, Ci)[:, :T] = xr.view(N, T, Ci)
    w3 = (rng.standard_normal((Co, Ci, 3)) / np.sqrt(3 * Ci)).astype(np.float32)
    w4 = (rng.standard_normal((Co, Ci, 4)) / np.sqrt(4 * Ci)).astype(np.float32)
    wt = (rng.standard_normal((Ci, Co, 4)) / np.sqrt(2 * Ci)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    R = rng.standard_normal((N * T, Co)).astype(np.float32)
    p3 = H.pack_conv_taps(dev(w3), torch.float32, split=True)
    p4 = H.pack_conv_taps(dev(w4), torch.float32, split=True)
    pt = H.pack_conv_taps(dev(wt), torch.float32, transposed=True, split=True)

    def run():
        outs = []
        for dil in (1, 27, 243):
            outs.append(H.gemm(xr, p3, bias=dev(b), res=dev(R), res_scale=0.5, n_seq=N, t_in=T, shifts=(-dil, 0, dil), pre_relu=True))
        outs.append(H.gemm(xp, p3, bias=dev(b), n_seq=N, t_in=T, shifts=(-3, 0, 3), in_seq_pitch=pitch))
        outs.append(H.gemm(xr, p4, bias=dev(b), n_seq=N, t_in=T, t_out=T // 2, in_stride=2, shifts=(-1, 0, 1, 2)))
        out = torch.empty((N * 2 * T, Co), dtype=torch.float32, device="cuda")
        H.gemm(xr, H.tap_view(pt, [1, 3]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(0, -1), out_stride=2, out_offset=0,
               out_rows_per_seq=2 * T)
        H.gemm(xr, H.tap_view(pt, [0, 2]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(1, 0), out_stride=2, out_offset=1,
               out_rows_per_seq=2 * T)
        outs.append(out)
        H.check_split_overflow()                           # nothing out of range, the rubbish rows were not looked at
        return [o.cpu().numpy() for o in outs]

    try:
        L.lib().jb_tune_gemm_presplit(1)
        pre = run()
        L.lib().jb_tune_gemm_presplit(0)
        inside = run()
        for a, c in zip(pre, inside):
            assert np.array_equal(a, c)
        for on in (1, 0):
            L.lib().jb_tune_gemm_presplit(on)
            for bad, relu in ((7e4, False), (float("nan"), True), (-7e4, False)):
                xb = xr.clone()
                xb[T + 3, 5] = bad
                H.gemm(xb, p3, n_seq=N, t_in=T, shifts=(-1, 0, 1), pre_relu=relu)
                with pytest.raises(L.JukeboxHipError):
                    H.check_split_overflow()
            xb = xr.clone()
            xb[T + 3, 5] = -7e4                            # what the input ReLU clips cannot overflow
            H.gemm(xb, p3, n_seq=N, t_in=T, shifts=(-1, 0, 1), pre_relu=True)
            H.check_split_overflow()
    finally:
        L.lib().jb_tune_gemm_presplit(1)
        L.lib().jb_gemm_split_overflow(1)


@pytest.mark.parametrize("wscale", [1e-5, 1e-3, 1.0, 3e4, 1e7])
def test_conv_stack_f16_split_is_invariant_to_the_weights_scale(H, wscale):
    """A trained layer's weights can sit far below a half's normal range (|w| < 6e-5: hi halves subnormal, 11 bits gone) or
    above its largest value; the split image therefore holds s * w for the power of two s that puts max |w| into [128, 256)
    and the kernel multiplies the accumulators by 1 / s (jb_gemm_args.w_split_unscale).  Scaling the weights by ANY factor must
    leave the error against float64 where it is at O(1) weights (<= 2e-6 of the output scale: the fp32 accumulation's own
    rounding), with a 2^20 spread of magnitudes inside one matrix, for the dilated k=3 form with the input ReLU and for the
    transposed convolution through tap views."""
    rng = np.random.default_rng(7)
    N, Ci, Co, T = 2, 256, 72, 150
    x = rng.standard_normal((N, Ci, T)).astype(np.float32)
    xr = dev(np.transpose(x, (0, 2, 1)).reshape(N * T, Ci))
    w = (rng.standard_normal((Co, Ci, 3)) / np.sqrt(3 * Ci)).astype(np.float32)
    w[:, ::7] *= np.float32(2.0 ** -20)                                    # some input channels a million times weaker
    w = (w * np.float32(wscale)).astype(np.float32)
    want = np.transpose(_conv1d_f64(np.maximum(x.astype(np.float64), 0), w, np.zeros(Co, np.float32), 1, 9, 9), (0, 2, 1)).reshape(N * T, Co)
    pw = H.pack_conv_taps(dev(w), torch.float32, split=True)
    assert 128.0 <= np.abs(w).max() / pw.unscale < 256.0
    got = H.gemm(xr, pw, n_seq=N, t_in=T, shifts=(-9, 0, 9), pre_relu=True).cpu().numpy().astype(np.float64)
    e = relerr(got, want)
    # ... and each weak channel alone: its contribution is not lost in the strong ones' rounding of the IMAGE (the split keeps
    # 2^-22 relative precision per weight down to 2^-22 of the largest)
    x_weak = np.zeros_like(x)
    x_weak[:, ::7] = x[:, ::7]
    want_w = np.transpose(_conv1d_f64(np.maximum(x_weak.astype(np.float64), 0), w, np.zeros(Co, np.float32), 1, 9, 9), (0, 2, 1)).reshape(N * T, Co)
    got_w = H.gemm(dev(np.transpose(x_weak, (0, 2, 1)).reshape(N * T, Ci)), pw, n_seq=N, t_in=T, shifts=(-9, 0, 9),
                   pre_relu=True).cpu().numpy().astype(np.float64)
    e_weak = relerr(got_w, want_w)
    wt = (rng.standard_normal((Ci, Co, 4)) / np.sqrt(2 * Ci)).astype(np.float32) * np.float32(wscale)
    b = (rng.standard_normal(Co) * wscale).astype(np.float32)
    want_t = _conv_transpose1d_f64(x.astype(np.float64), wt, b, 2, 1)
    pwt = H.pack_conv_taps(dev(wt), torch.float32, transposed=True, split=True)
    out = torch.empty((N * 2 * T, Co), dtype=torch.float32, device="cuda")
    H.gemm(xr, H.tap_view(pwt, [1, 3]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(0, -1), out_stride=2, out_offset=0,
           out_rows_per_seq=2 * T)
    H.gemm(xr, H.tap_view(pwt, [0, 2]), bias=dev(b), out=out, n_seq=N, t_in=T, t_out=T, shifts=(1, 0), out_stride=2, out_offset=1,
           out_rows_per_seq=2 * T)
    e_t = relerr(np.transpose(out.cpu().numpy().reshape(N, 2 * T, Co), (0, 2, 1)), want_t)
    H.check_split_overflow()
    print("f16-split conv, weights x %g: error %.2e (dilated k=3), %.2e (the 2^-20 channels alone), %.2e (transposed) of the "
          "output scale" % (wscale, e, e_weak, e_t))
    assert e < 2e-6 and e_weak < 2e-6 and e_t < 2e-6


def test_gemm_split_reports_activations_outside_the_half_range(H):
    """The f16-split path cannot represent |x| > 65504: a launch that sees such an activation (or a NaN) raises the sticky device
    flag behind jb_gemm_split_overflow, hip_ops.check_split_overflow turns it into an exception and clears it; values up to the
    largest half, and rows that are only out of range BEFORE the input ReLU, do not."""
    from jukebox_amd import _lib as L
    H.check_split_overflow()                                         # (clean start)
    w = torch.randn(16, 64, 3, device="cuda") * 0.05
    pw = H.pack_conv_taps(w, torch.float32, split=True)
    x = torch.randn(2 * 50, 64, device="cuda")
    x[7, 3] = 65504.0
    x[9, 5] = -3.0e5
    H.gemm(x, pw, n_seq=2, t_in=50, shifts=(-1, 0, 1), pre_relu=True)   # the negative outlier dies in the ReLU
    H.check_split_overflow()
    H.gemm(x, pw, n_seq=2, t_in=50, shifts=(-1, 0, 1))
    torch.cuda.synchronize()                 # the flag is a host-coherent word: the call reads what FINISHED launches have stored
    assert L.lib().jb_gemm_split_overflow(0) == 1 and L.lib().jb_gemm_split_overflow(0) == 1      # sticky until reset
    with pytest.raises(L.JukeboxHipError):
        H.check_split_overflow()
    assert L.lib().jb_gemm_split_overflow(0) == 0
    x[9, 5] = float("nan")
    H.gemm(x, pw, n_seq=2, t_in=50, shifts=(-1, 0, 1))
    with pytest.raises(L.JukeboxHipError):
        H.check_split_overflow()
    H.gemm(x, pw, n_seq=2, t_in=50, shifts=(-1, 0, 1), pre_relu=True)   # fmaxf(NaN, 0) = 0: judged BEFORE the ReLU (torch's relu keeps a NaN)
    with pytest.raises(L.JukeboxHipError):
        H.check_split_overflow()
    x[9, 5] = -float("inf")                                             # relu(-inf) = 0: fine under the input ReLU
    H.gemm(x, pw, n_seq=2, t_in=50, shifts=(-1, 0, 1), pre_relu=True)
    H.check_split_overflow()
    x[9, 5] = 0.0
    x[3, 1] = 70000.0                                                   # too large for a half, not clipped by the ReLU
    H.gemm(x, pw, n_seq=2, t_in=50, shifts=(-1, 0, 1), pre_relu=True)
    assert L.lib().jb_gemm_split_overflow(0) in (0, 1)                  # (not waited for: whatever has finished)
    with pytest.raises(L.JukeboxHipError):
        H.check_split_overflow()                                        # waits, reads, clears
    H.check_split_overflow(wait=False)
    # only the workgroups of the FIRST column block look (every row passes through them): a wide layer (several column blocks)
    # reports an outlier in any row, once
    w2 = torch.randn(192, 64, 3, device="cuda") * 0.05
    pw2 = H.pack_conv_taps(w2, torch.float32, split=True)
    x2 = torch.randn(2 * 400, 64, device="cuda")
    H.gemm(x2, pw2, n_seq=2, t_in=400, shifts=(-3, 0, 3))
    H.check_split_overflow()
    for row in (0, 399, 400, 657, 799):
        x2[row, 17] = 1.0e6
        H.gemm(x2, pw2, n_seq=2, t_in=400, shifts=(-3, 0, 3))
        with pytest.raises(L.JukeboxHipError):
            H.check_split_overflow()
        x2[row, 17] = 0.5
    # weights have no range limit: the image holds s * w (s a power of two that puts max |w| into [128, 256)), checked finite
    big = H.pack_conv_taps(w * 1e7, torch.float32, split=True)
    assert big.unscale > 1.0 and 128.0 <= float((w * 1e7).abs().max()) / big.unscale < 256.0
    with pytest.raises(AssertionError):
        H.pack_conv_taps(w * float("inf"), torch.float32, split=True)


def test_gemm_split_refuses_what_it_cannot_take(H):
    """w_split is for fp32 problems with K a multiple of 32 (jb_gemm says so instead of computing something else)."""
    w = torch.randn(16, 48, 3, device="cuda")
    with pytest.raises(AssertionError):
        H.pack_conv_taps(w, torch.float32, split=True)
    pw = H.pack_conv_taps(torch.randn(16, 64, 3, device="cuda"), torch.float32, split=True)
    pw.K = 48                                                # a caller lying about K
    with pytest.raises(RuntimeError):
        H.gemm(torch.randn(30, 48, device="cuda"), pw, n_seq=1, t_in=30, shifts=(-1, 0, 1))


@pytest.mark.parametrize("name,dt,tol", DT)
@pytest.mark.parametrize("rows,K,J", [(16, 256, 96), (3, 100, 50), (40, 64, 16), (64, 512, 130)])
def test_gemv_ln_epilogues(H, name, dt, tol, rows, K, J):
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(rows + K)
    f16 = dt == torch.float16
    r = (lambda x: h16(x)) if f16 else (lambda x: x)
    x = r(rng.standard_normal((rows, K)).astype(np.float32) * 2 + 0.5)
    W = r((0.1 * rng.standard_normal((K, J))).astype(np.float32))
    b = rng.standard_normal(J).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    be = (0.1 * rng.standard_normal(K)).astype(np.float32)
    R = r(rng.standard_normal((rows, J)).astype(np.float32))
    pw = H.pack_conv1d_w(dev(W), dt)
    # plain + bias
    got = H.gemv(dev(x, dt), pw, bias=dev(b)).float().cpu().numpy()
    assert relerr(got, r(x @ W + r(b))) < tol
    # LN + gelu
    got = H.gemv(dev(x, dt), pw, bias=dev(b), ln=(dev(g), dev(be)), act=L.ACT_QUICK_GELU).float().cpu().numpy()
    xn = r(O.layer_norm(x, g, be))
    want = O.quick_gelu(r(xn @ W + r(b)), fp16=f16)
    assert relerr(got, want) < tol
    # residual
    got = H.gemv(dev(x, dt), pw, bias=dev(b), res=dev(R, dt)).float().cpu().numpy()
    assert relerr(got, r(R + r(x @ W + r(b)))) < tol


@pytest.mark.parametrize("name,dt,tol", DT)
@pytest.mark.parametrize("rows,K,J", [(16, 1920, 1440), (16, 256, 96), (32, 2048, 64), (3, 512, 132), (20, 1024, 20), (1, 32, 4),
                                      (3, 4800, 272), (16, 4800, 64), (3, 5120, 32)])
def test_gemv_ln_folded(H, name, dt, tol, rows, K, J):
    """Folded LayerNorm (jb_gemv_args.ln_fold_c1): rstd*(x.W' - mean*c1) + b' against LayerNorm -> Conv1D done the
    reference's way (ops.py:14-24,97-101), on rows with a mean several times their spread."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(rows * 7 + K)
    f16 = dt == torch.float16
    r = (lambda x: h16(x)) if f16 else (lambda x: x)
    x = r(rng.standard_normal((rows, K)).astype(np.float32) * 1.5 + rng.standard_normal((rows, 1)).astype(np.float32) * 4)
    W = r((0.1 * rng.standard_normal((K, J))).astype(np.float32))
    b = rng.standard_normal(J).astype(np.float32)
    g = (1 + 0.2 * rng.standard_normal(K)).astype(np.float32)
    be = (0.2 * rng.standard_normal(K)).astype(np.float32)
    if K >= 4800 and not f16:
        assert not H.ln_fold_supported(dt, K, J, rows)       # 300+ fp32 k-tiles: the fp32 engine normalises explicitly
        return
    assert H.ln_fold_supported(dt, K, J, rows)               # K = 4800 (5b_lyrics): 150 k-tiles on 16 waves
    f = H.FoldedLN(dev(W), dev(b), dev(g), dev(be), dt)
    got = H.gemv(dev(x, dt), None, ln_fold=f, act=L.ACT_QUICK_GELU).float().cpu().numpy()
    x64 = x.astype(np.float64)
    m, v = x64.mean(-1, keepdims=True), x64.var(-1, keepdims=True)
    pre = ((x64 - m) / np.sqrt(v + 1e-5) * g + be) @ W.astype(np.float64) + b
    exact = pre / (1 + np.exp(-1.702 * pre))
    want = O.quick_gelu(r(r(O.layer_norm(x, g, be)) @ W + r(b)), fp16=f16)
    assert relerr(got, want) < tol
    # the folded form skips the rounding of the normalised rows, so it is at least as close to exact arithmetic
    assert relerr(got, exact) < max(tol, 1.5 * relerr(want, exact))
    # and agrees with the in-kernel normalisation path of the same library
    if rows * K * (2 if f16 else 4) < 100 * 1024:          # the in-kernel path stages the rows in LDS
        cls = H.gemv(dev(x, dt), H.pack_conv1d_w(dev(W), dt), bias=dev(b), ln=(dev(g), dev(be)), act=L.ACT_QUICK_GELU)
        assert relerr(got, cls.float().cpu().numpy()) < tol


@pytest.mark.parametrize("rows,K,J", [(3, 4800, 272), (16, 4800, 64), (3, 4800, 4800), (8, 5120, 48), (3, 4128, 32)])
def test_gemv_long_rows_on_8_waves(H, rows, K, J):
    """jb_tune_gemv_long(1): projections over 129 .. 160 k-tiles (5b_lyrics, K = 4800) on 8-wave workgroups that walk their k-tiles
    through two register stages (gemv_long_kernel) instead of the 16-wave kernels -- the folded-LayerNorm form and the plain form with
    bias + residual, against the same references and bars as the 16-wave kernels, and against those kernels' own outputs (another
    summation order of the eight / sixteen partial tiles: equal to output rounding)."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(rows * 11 + K + J)
    dt, tol = torch.float16, 3e-3
    x = h16(rng.standard_normal((rows, K)).astype(np.float32) * 1.5 + rng.standard_normal((rows, 1)).astype(np.float32) * 4)
    W = h16((rng.standard_normal((K, J)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(J).astype(np.float32)
    g = (1 + 0.2 * rng.standard_normal(K)).astype(np.float32)
    be = (0.2 * rng.standard_normal(K)).astype(np.float32)
    R = h16(rng.standard_normal((rows, J)).astype(np.float32))
    f = H.FoldedLN(dev(W), dev(b), dev(g), dev(be), dt)
    pw = H.pack_conv1d_w(dev(W), dt)
    outs = {}
    try:
        for long_rows in (0, 1):
            L.lib().jb_tune_gemv_long(long_rows)
            folded = H.gemv(dev(x, dt), None, ln_fold=f, act=L.ACT_QUICK_GELU).float().cpu().numpy()
            plain = H.gemv(dev(x, dt), pw, bias=dev(b), res=dev(R, dt)).float().cpu().numpy()
            outs[long_rows] = (folded, plain)
    finally:
        L.lib().jb_tune_gemv_long(1)             # the default
    want_f = O.quick_gelu(h16(h16(O.layer_norm(x, g, be)) @ W + h16(b)), fp16=True)
    want_p = h16(R + h16(x @ W + h16(b)))
    for long_rows in (0, 1):
        assert relerr(outs[long_rows][0], want_f) < tol and relerr(outs[long_rows][1], want_p) < tol, long_rows
    assert relerr(outs[1][0], outs[0][0]) < 1e-3 and relerr(outs[1][1], outs[0][1]) < 1e-3
    assert not np.array_equal(outs[1][1], np.zeros_like(outs[1][1]))


def test_gemv_ln_folded_rejects_unsupported_shapes(H):
    from jukebox_amd import _lib as L
    assert not H.ln_fold_supported(torch.float16, 100, 64, 16)        # K not a whole number of k-tiles
    assert not H.ln_fold_supported(torch.float16, 8192, 64, 16)       # fragments of a row do not fit in registers
    assert H.ln_fold_supported(torch.float16, 4800, 3600, 3)          # 5b_lyrics: 150 k-tiles on 16 waves
    assert not H.ln_fold_supported(torch.float16, 4800, 3600, 17)     # ... which take one 16-row tile
    assert not H.ln_fold_supported(torch.float32, 256, 64, 48)        # more than 32 rows
    x = torch.zeros((16, 128), dtype=torch.float16, device="cuda")
    f = H.FoldedLN(torch.zeros((128, 64), device="cuda"), torch.zeros(64, device="cuda"), torch.ones(128, device="cuda"),
                   torch.zeros(128, device="cuda"), torch.float16)
    with pytest.raises(L.JukeboxHipError, match="ln_fold_c1 excludes"):
        a = L.GemvArgs()
        out = torch.empty((16, 64), dtype=torch.float16, device="cuda")
        a.dtype, a.x, a.ldx, a.n_rows = L.F16, x.data_ptr(), 128, 16
        a.W, a.bias, a.K, a.J, a.out, a.ldo = f.pw.ptr, f.bias.data_ptr(), 128, 64, out.data_ptr(), 64
        a.ln_fold_c1, a.ln_gamma, a.ln_beta = f.c1.data_ptr(), f.c1.data_ptr(), f.c1.data_ptr()
        import ctypes as C
        L.check(L.lib().jb_gemv(C.byref(a), L.stream()))


@pytest.mark.parametrize("name,dt,tol", DT)
def test_gemv_qkv_append(H, name, dt, tol):
    import ctypes as C
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(5)
    N, K, S, cap, t = 5, 64, 24, 9, 6
    f16 = dt == torch.float16
    r = (lambda x: h16(x)) if f16 else (lambda x: x)
    x = r(rng.standard_normal((N, K)).astype(np.float32))
    W = r((0.1 * rng.standard_normal((K, 3 * S))).astype(np.float32))
    b = rng.standard_normal(3 * S).astype(np.float32)
    pw = H.pack_conv1d_w(dev(W), dt)
    q = torch.zeros((N, S), dtype=dt, device="cuda")
    kc = torch.zeros((N, cap, S), dtype=dt, device="cuda")
    vc = torch.zeros((N, cap, S), dtype=dt, device="cuda")
    for tt in (t, cap + 3):                     # second call: beyond the cache capacity -> no k/v write
        t_dev = torch.tensor([tt], dtype=torch.int32, device="cuda")
        xd, bd = dev(x, dt), dev(b)
        a = L.GemvArgs()
        a.dtype = L.dtype_code(dt)
        a.x, a.ldx, a.n_rows = xd.data_ptr(), K, N
        a.W, a.bias, a.K, a.J = pw.ptr, bd.data_ptr(), K, 3 * S
        a.out, a.ldo = q.data_ptr(), S
        a.qkv_split, a.S, a.kcache, a.vcache, a.cache_cap, a.t_dev = 1, S, kc.data_ptr(), vc.data_ptr(), cap, t_dev.data_ptr()
        L.check(L.lib().jb_gemv(C.byref(a), L.stream()))
        torch.cuda.synchronize()
    want = r(x @ W + r(b))
    assert relerr(q.float().cpu().numpy(), want[:, :S]) < tol
    kcn, vcn = kc.float().cpu().numpy(), vc.float().cpu().numpy()
    assert relerr(kcn[:, t], want[:, S:2 * S]) < tol and relerr(vcn[:, t], want[:, 2 * S:]) < tol
    mask = np.ones(cap, bool); mask[t] = False
    assert np.all(kcn[:, mask] == 0) and np.all(vcn[:, mask] == 0)


@pytest.mark.parametrize("din,dout", [(torch.float32, torch.float32), (torch.float16, torch.float16),
                                       (torch.float32, torch.float16)])
def test_layernorm(H, din, dout):
    rng = np.random.default_rng(3)
    for rows, W in ((37, 200), (130, 1920), (5, 4096), (9, 203), (3, 4104)):     # vector path (W % 8 == 0, <= 4096) and scalar path
        x = (rng.standard_normal((rows, W)) * 3 + 1).astype(np.float32)
        if din == torch.float16:
            x = h16(x)
        g = (1 + 0.1 * rng.standard_normal(W)).astype(np.float32)
        b = (0.1 * rng.standard_normal(W)).astype(np.float32)
        got = H.layernorm(dev(x, din), dev(g), dev(b), out_dtype=dout).float().cpu().numpy()
        want = O.layer_norm(x, g, b)
        assert relerr(got, want) < (2e-3 if dout == torch.float16 else 1e-5), (rows, W)


def _np_attention(func, q, K, V, H_, bc, prime_r, qpos, fp16):
    """numpy reference: per query position, softmax over the closed-form key set."""
    N, nq, S = q.shape
    d = S // H_
    out = np.zeros_like(q)
    sc2 = (1.0 / math.sqrt(math.sqrt(d))) ** 2
    r = (lambda x: h16(x)) if fp16 else (lambda x: x)
    for i, p in enumerate(qpos):
        idx = decode_key_index(func, p, bc, prime_r)
        if idx is None:
            continue
        for h in range(H_):
            qs = q[:, i, h * d:(h + 1) * d]
            ks, vs = K[:, idx, h * d:(h + 1) * d], V[:, idx, h * d:(h + 1) * d]
            w = r(r(np.einsum("nd,nkd->nk", qs, ks)) * np.float32(sc2))
            pr = r(O.softmax(w, -1))
            out[:, i, h * d:(h + 1) * d] = r(np.einsum("nk,nkd->nd", pr, vs))
    return out


@pytest.mark.parametrize("name,dt,tol", [("f32", torch.float32, 1e-5), ("f16", torch.float16, 4e-3)])
@pytest.mark.parametrize("func", [0, 1, 2, 3, 7])
@pytest.mark.parametrize("H_,d", [(2, 16), (1, 120), (2, 256), (2, 64)])
def test_attn_decode(H, name, dt, tol, func, H_, d):
    rng = np.random.default_rng(func * 10 + d)
    N, T, bc, prime_r = 3, 96, 8, 24
    S = H_ * d
    fp16 = dt == torch.float16
    r = (lambda x: h16(x)) if fp16 else (lambda x: x)
    cap = prime_r if func == 7 else T
    K = r(rng.standard_normal((N, cap, S)).astype(np.float32))
    V = r(rng.standard_normal((N, cap, S)).astype(np.float32))
    kc, vc = dev(K, dt), dev(V, dt)
    for t in (0, 1, 7, 8, 9, 23, 24, 40, 95):
        q = r(rng.standard_normal((N, 1, S)).astype(np.float32))
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        got = H.attn_decode(func, dev(q[:, 0], dt), kc, vc, H_, bc, t_dev, T).float().cpu().numpy()
        want = _np_attention(func, q, K, V, H_, bc, prime_r, [t], fp16)[:, 0]
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), (func, t)


@pytest.mark.parametrize("func", [0, 1, 2, 3])
def test_attn_decode_5b_head_size(H, func):
    """fp16 MFMA decode attention at 5b_lyrics' head size: 8 heads of 150 channels (hparams.py:127-153) -- not a multiple of
    the 32-wide k-tile, heads start at odd multiples of 4 bytes; block_ctx 64."""
    rng = np.random.default_rng(50 + func)
    N, T, bc, H_, d = 3, 400, 64, 8, 150
    S = H_ * d
    K, V = h16(rng.standard_normal((N, T, S)).astype(np.float32)), h16(rng.standard_normal((N, T, S)).astype(np.float32))
    kc, vc = dev(K, torch.float16), dev(V, torch.float16)
    for t in (0, 1, 63, 64, 70, 200, 399):
        q = h16(rng.standard_normal((N, 1, S)).astype(np.float32))
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        got = H.attn_decode(func, dev(q[:, 0], torch.float16), kc, vc, H_, bc, t_dev, T).float().cpu().numpy()
        want = _np_attention(func, q, K, V, H_, bc, None, [t], True)[:, 0]
        assert np.abs(got - want).max() < 4e-3 * max(1.0, np.abs(want).max()), (func, t)


@pytest.mark.parametrize("name,dt,tol", [("f32", torch.float32, 1e-5), ("f16", torch.float16, 4e-3)])
@pytest.mark.parametrize("H_,d,keys", [(8, 150, 512), (4, 80, 512), (2, 16, 24), (1, 480, 100)])
def test_attn_cross_attention(H, name, dt, tol, H_, d, keys):
    """attn_func 6 (decode_attn, factored_attention.py:226-228): every query -- the single one of a decode step and each of
    a prefill chunk -- reads ALL encoder positions, no mask, whatever its own position.  8 x 150 channels over 512 keys is
    prior_5b_lyrics' cross-attention (hparams.py:127-153: n_tokens 512), 4 x 80 its lyric encoder's head size."""
    rng = np.random.default_rng(600 + d)
    N, S = 3, H_ * d
    fp16 = dt == torch.float16
    r = (lambda x: h16(x)) if fp16 else (lambda x: x)
    K = r(rng.standard_normal((N, keys, S)).astype(np.float32))
    V = r(rng.standard_normal((N, keys, S)).astype(np.float32))
    kc, vc = dev(K, dt), dev(V, dt)
    sc2 = np.float32(1.0 / math.sqrt(d))

    def ref(q, rounded):
        rr = r if rounded else (lambda x: x)
        out = np.zeros_like(q)
        for h in range(H_):
            sl = slice(h * d, (h + 1) * d)
            w = rr(rr(np.einsum("nqd,nkd->nqk", q[..., sl], K[..., sl])) * sc2)
            out[..., sl] = rr(np.einsum("nqk,nkd->nqd", rr(O.softmax(w, -1)), V[..., sl]))
        return out

    for t in (0, 5, 700):
        q = r(rng.standard_normal((N, 1, S)).astype(np.float32))
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        got = H.attn_decode(6, dev(q[:, 0], dt), kc, vc, H_, 64, t_dev, 8192).float().cpu().numpy()
        want = ref(q, fp16)[:, 0]
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), ("decode", t)
    for t0, nq in ((0, 40), (100, 7), (640, 130)):
        q = r(rng.standard_normal((N, nq, S)).astype(np.float32))
        got = H.attn_prefill(6, dev(q, dt), kc, vc, H_, 64, t0).float().cpu().numpy()
        want = ref(q, False)
        assert np.abs(got - want).max() < (6e-3 if fp16 else 2e-5) * max(1.0, np.abs(want).max()), ("prefill", t0, nq)


@pytest.mark.parametrize("name,dt,tol", [("f32", torch.float32, 2e-5), ("f16", torch.float16, 6e-3)])
@pytest.mark.parametrize("func", [1, 2, 3])
def test_attn_lyric_encoder_head_size(H, name, dt, tol, func):
    """The lyric encoder of prior_5b_lyrics (hparams.py:139-146: prime_width 1280, prime_heads 4 -> 4 heads of 80 channels,
    prime_blocks 32 over n_tokens 512 -> block_ctx 16, prime_attn_order 2): 80 is not a multiple of the 32-wide k-tile.
    Full-sequence prefill (the encoder is only ever run as one pass) and, for completeness, single decode queries."""
    rng = np.random.default_rng(800 + func)
    N, T, bc, H_, d = 3, 512, 16, 4, 80
    S = H_ * d
    fp16 = dt == torch.float16
    r = (lambda x: h16(x)) if fp16 else (lambda x: x)
    K = r(rng.standard_normal((N, T, S)).astype(np.float32))
    V = r(rng.standard_normal((N, T, S)).astype(np.float32))
    kc, vc = dev(K, dt), dev(V, dt)
    for t0, nq in ((0, 512), (0, 100), (37, 300)):
        q = r(rng.standard_normal((N, nq, S)).astype(np.float32))
        got = H.attn_prefill(func, dev(q, dt), kc, vc, H_, bc, t0).float().cpu().numpy()
        want = _np_attention(func, q, K, V, H_, bc, None, list(range(t0, t0 + nq)), False)
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), (func, t0, nq)
    for t in (0, 15, 16, 100, 511):
        q = r(rng.standard_normal((N, 1, S)).astype(np.float32))
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        got = H.attn_decode(func, dev(q[:, 0], dt), kc, vc, H_, bc, t_dev, T).float().cpu().numpy()
        want = _np_attention(func, q, K, V, H_, bc, None, [t], fp16)[:, 0]
        assert np.abs(got - want).max() < (4e-3 if fp16 else 1e-5) * max(1.0, np.abs(want).max()), (func, t)


@pytest.mark.parametrize("func", [1, 2, 3])
def test_attn_decode_upsampler_shape(H, func):
    """fp16 MFMA decode attention at the upsamplers' head size (1 head of 480) and block length 128 (one workgroup per
    sample)."""
    rng = np.random.default_rng(func)
    N, T, bc, d = 2, 512, 128, 480
    K, V = h16(rng.standard_normal((N, T, d)).astype(np.float32)), h16(rng.standard_normal((N, T, d)).astype(np.float32))
    kc, vc = dev(K, torch.float16), dev(V, torch.float16)
    for t in (0, 1, 127, 128, 130, 300, 511):
        q = h16(rng.standard_normal((N, 1, d)).astype(np.float32))
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        got = H.attn_decode(func, dev(q[:, 0], torch.float16), kc, vc, 1, bc, t_dev, T).float().cpu().numpy()
        want = _np_attention(func, q, K, V, 1, bc, None, [t], True)[:, 0]
        assert np.abs(got - want).max() < 4e-3 * max(1.0, np.abs(want).max()), (func, t)


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("N,K,S,keep_v", [(16, 1920, 480, False), (5, 256, 64, False), (16, 256, 64, True), (3, 128, 32, False)])
def test_gemv_qkv_wide_columns(H, N, K, S, keep_v, fold):
    """c_attn of a wide-value layer in the decode step (jb_gemv_args.vcache_wide): columns q | k | [v |] v' with v' rows of
    K = width elements appended to their own cache at *t_dev; explicit and folded LayerNorm."""
    rng = np.random.default_rng(N + K + S)
    f16 = torch.float16
    cap, t = 9, 6
    J = 2 * S + (S if keep_v else 0) + K
    x = h16(rng.standard_normal((N, K)).astype(np.float32))
    W = h16((rng.standard_normal((K, J)) / np.sqrt(K)).astype(np.float32))
    b = (0.1 * rng.standard_normal(J)).astype(np.float32)
    gam = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    bet = (0.1 * rng.standard_normal(K)).astype(np.float32)
    if fold and not H.ln_fold_supported(f16, K, J, N):
        pytest.skip("folded form not available for this shape")
    kc = torch.zeros((N, cap, S), dtype=f16, device="cuda")
    vc = torch.zeros((N, cap, S), dtype=f16, device="cuda") if keep_v else None
    vw = torch.zeros((N, cap, K), dtype=f16, device="cuda")
    t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
    if fold:
        f = H.FoldedLN(dev(W), dev(b), dev(gam), dev(bet), f16)
        q = H.gemv(dev(x, f16), None, ln_fold=f, qkv=(S, kc, vc, vw), t_dev=t_dev)
    else:
        q = H.gemv(dev(x, f16), H.pack_conv1d_w(dev(W), f16), bias=dev(b), ln=(dev(gam), dev(bet)), qkv=(S, kc, vc, vw), t_dev=t_dev)
    torch.cuda.synchronize()
    mu, var = x.mean(1, keepdims=True), x.var(1, keepdims=True)
    xn = (x - mu) / np.sqrt(var + 1e-5) * gam + bet
    want = xn @ W + b
    tol = 6e-3 * max(1.0, np.abs(want).max())
    assert tuple(q.shape) == (N, S)
    assert np.abs(q.float().cpu().numpy() - want[:, :S]).max() < tol
    assert np.abs(kc[:, t].float().cpu().numpy() - want[:, S:2 * S]).max() < tol
    v0 = 2 * S
    if keep_v:
        assert np.abs(vc[:, t].float().cpu().numpy() - want[:, v0:v0 + S]).max() < tol
        v0 += S
    assert np.abs(vw[:, t].float().cpu().numpy() - want[:, v0:]).max() < tol
    mask = np.ones(cap, bool); mask[t] = False
    assert (kc[:, mask] == 0).all() and (vw[:, mask] == 0).all() and (vc is None or (vc[:, mask] == 0).all())


@pytest.mark.parametrize("func", [0, 1, 2, 3, 7])
@pytest.mark.parametrize("d,W_,bc,T", [(480, 1920, 128, 1024), (64, 256, 8, 96), (32, 32, 8, 96), (256, 1024, 128, 8192)])
def test_attn_decode_wide(H, func, d, W_, bc, T):
    """Wide-value decode attention (jb_attn_decode_wide): with v' = v·Wp cached, res + (sum_k p_k v'_k + bp) equals the
    residual stream after jb_attn_decode + attn.c_proj (the five-launch form) up to the half rounding points, and equals
    the fp32 evaluation on the same half operands within the half output bound.  Upsampler geometry (1 x 480, width 1920,
    block length 128), block rows 0 and 63 of the transpose pattern, empty key sets (prev_block in block 0)."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(func * 100 + d + T)
    N = 5
    f16 = torch.float16
    prime_r = 448 if T >= 960 else 24
    cap = prime_r if func == 7 else T
    assert L.lib().jb_attn_decode_wide_supported(func, d, W_, bc, T) == 1
    assert L.lib().jb_attn_decode_wide_supported(6, d, W_, bc, T) == 0 and L.lib().jb_attn_decode_wide_supported(func, d + 8, W_, bc, T) == 0
    K = h16(rng.standard_normal((N, cap, d)).astype(np.float32))
    V = h16(rng.standard_normal((N, cap, d)).astype(np.float32))
    Wp = h16((rng.standard_normal((d, W_)) / np.sqrt(d)).astype(np.float32))
    bp = (0.1 * rng.standard_normal(W_)).astype(np.float32)
    Vw = h16(V @ Wp)                                                       # the rows prefill / c_attn would have cached
    res = h16(rng.standard_normal((N, W_)).astype(np.float32))
    kc, vc, vw = dev(K, f16), dev(V, f16), dev(Vw, f16)
    pw = H.pack_conv1d_w(dev(Wp), f16)
    sc2 = (1.0 / math.sqrt(math.sqrt(d))) ** 2
    ts = sorted({0, 1, bc - 1, bc, bc + 1, 2 * bc + 3, T // 2 + 5, T - bc - 1, T - 1})
    for t in ts:
        q = h16(rng.standard_normal((N, 1, d)).astype(np.float32))
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        got_t = H.attn_decode_wide(func, dev(q[:, 0], f16), kc, vw, dev(res, f16), dev(bp), bc, t_dev, T)
        if d == 480:
            # the lean form (query through LDS, <= 168 registers per lane: what two pipelined engines side by side need) and the
            # fat form (the default) are the same arithmetic in the same order
            L.lib().jb_tune_attn_decode_wide_lean(1)
            try:
                lean = H.attn_decode_wide(func, dev(q[:, 0], f16), kc, vw, dev(res, f16), dev(bp), bc, t_dev, T)
            finally:
                L.lib().jb_tune_attn_decode_wide_lean(0)
            assert torch.equal(got_t, lean), (func, t, "lean and fat kernels differ")
        got = got_t.float().cpu().numpy()
        idx = decode_key_index(func, t, bc, prime_r if func == 7 else None)
        if idx is None:
            a = np.zeros((N, W_), np.float32)
        else:
            w = h16(h16(np.einsum("nd,nkd->nk", q[:, 0], K[:, idx])) * np.float32(sc2))
            a = np.einsum("nk,nkw->nw", h16(O.softmax(w, -1)), Vw[:, idx])
        want = h16(res + h16(a + h16(bp)))
        scale = max(1.0, np.abs(want).max())
        assert np.abs(got - want).max() < 4e-3 * scale, (func, t)
        att = H.attn_decode(func, dev(q[:, 0], f16), kc, vc, 1, bc, t_dev, T)
        five = H.gemv(att, pw, bias=dev(bp), res=dev(res, f16)).float().cpu().numpy()
        assert np.abs(got - five).max() < 8e-3 * scale, (func, t)


def _merge_parts(parts, ml):
    """numpy restatement of the log-sum-exp merge gemv_merge_kernel performs on jb_attn_decode_split's output."""
    N, P, S = parts.shape
    Hn = ml.shape[1]
    d = S // Hn
    out = np.zeros((N, S), np.float32)
    for n in range(N):
        for h in range(Hn):
            m, l = ml[n, h, :, 0], ml[n, h, :, 1]
            if not (l > 0).any():
                continue
            w = np.where(l > 0, l * np.exp(m - m[l > 0].max()), 0.0)
            out[n, h * d:(h + 1) * d] = (w[:, None] * parts[n, :, h * d:(h + 1) * d]).sum(0) / w.sum()
    return out


@pytest.mark.parametrize("func", [0, 1, 2, 3, 7])
@pytest.mark.parametrize("H_,d,bc,T,n_parts", [(1, 480, 128, 1024, 4), (2, 256, 96, 960, 3), (2, 64, 8, 96, 2), (1, 32, 8, 96, 1),
                                               (1, 480, 128, 8192, 4)])
def test_attn_decode_split_and_merge(H, func, H_, d, bc, T, n_parts):
    """Key-split decode attention (jb_attn_decode_split) + the merging attn.c_proj (jb_gemv with x_parts): the merged
    partial states equal the one-workgroup attention, and the projection fed by them equals the projection of that
    attention output -- at the released head sizes (1 x 480 upsamplers, 2 x 256 top prior), at block rows 0 and 63 of
    the transpose pattern (T = 8192, bc = 128), for empty splits (early positions, prev_block in block 0)."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(func * 100 + d + T)
    N, S = 5, H_ * d
    prime_r = 448 if T >= 960 else 24
    cap = prime_r if func == 7 else T
    f16 = torch.float16
    K = h16(rng.standard_normal((N, cap, S)).astype(np.float32))
    V = h16(rng.standard_normal((N, cap, S)).astype(np.float32))
    kc, vc = dev(K, f16), dev(V, f16)
    W = h16((rng.standard_normal((S, 208)) / np.sqrt(S)).astype(np.float32))
    b = (0.1 * rng.standard_normal(208)).astype(np.float32)
    res = h16(rng.standard_normal((N, 208)).astype(np.float32))
    pw = H.pack_conv1d_w(dev(W), f16)
    max_keys = {0: T, 1: bc, 2: (T + bc - 1) // bc, 3: bc, 7: cap}[func]
    # by default only key sets longer than one pass of the 8-wave kernel are split
    assert (L.lib().jb_attn_decode_split_parts(L.F16, d, max_keys) >= 1) == (max_keys > 128)
    assert L.lib().jb_attn_decode_split_parts(L.F16, d + 8, max_keys) == 0 and L.lib().jb_attn_decode_split_parts(L.F32, d, 4096) == 0
    ts = sorted({0, 1, bc - 1, bc, bc + 1, 2 * bc + 3, T // 2 + 5, T - bc - 1, T - 1})
    for t in ts:
        q = h16(rng.standard_normal((N, 1, S)).astype(np.float32))
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        parts, ml = H.attn_decode_split(func, dev(q[:, 0], f16), kc, vc, H_, bc, t_dev, max_keys, n_parts)
        want = _np_attention(func, q, K, V, H_, bc, prime_r if func == 7 else None, [t], True)[:, 0]
        merged = _merge_parts(parts.float().cpu().numpy(), ml.cpu().numpy())
        assert np.isfinite(merged).all()
        assert np.abs(merged - want).max() < 5e-3 * max(1.0, np.abs(want).max()), (func, t)
        one = H.attn_decode(func, dev(q[:, 0], f16), kc, vc, H_, bc, t_dev, T)
        assert np.abs(merged - one.float().cpu().numpy()).max() < 3e-3 * max(1.0, np.abs(want).max()), (func, t)
        got = H.gemv(None, pw, bias=dev(b), res=dev(res, f16), parts=(parts, ml)).float().cpu().numpy()
        ref = H.gemv(dev(h16(merged), f16), pw, bias=dev(b), res=dev(res, f16)).float().cpu().numpy()
        want_p = h16(res + h16(h16(merged) @ W + h16(b)))
        assert np.abs(got - want_p).max() < 6e-3 * max(1.0, np.abs(want_p).max()), (func, t)
        assert np.abs(got - ref).max() < 4e-3 * max(1.0, np.abs(want_p).max()), (func, t)


@pytest.mark.parametrize("name,dt,tol", DT)
def test_gemv_second_output(H, name, dt, tol):
    """jb_gemv out2 / add2: the fp32 copy `float(out) + cond[:, t]` the last mlp.c_proj of the decode step hands to the
    logits head (autoregressive.py:226-227)."""
    rng = np.random.default_rng(5)
    N, K, J, T = 16, 256, 192, 12
    fp16 = dt == torch.float16
    r = (lambda x: h16(x)) if fp16 else (lambda x: x)
    x, W = r(rng.standard_normal((N, K)).astype(np.float32)), r((rng.standard_normal((K, J)) / 16).astype(np.float32))
    b, res = (0.1 * rng.standard_normal(J)).astype(np.float32), r(rng.standard_normal((N, J)).astype(np.float32))
    cond = rng.standard_normal((N, T, J)).astype(np.float32)
    pw = H.pack_conv1d_w(dev(W), dt)
    want = r(res + r(x @ W + r(b)))
    for t, add in ((0, True), (7, True), (11, False)):
        out2 = torch.zeros((N, J), dtype=torch.float32, device="cuda")
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        got = H.gemv(dev(x, dt), pw, bias=dev(b), res=dev(res, dt), out2=out2, add2=dev(cond) if add else None, t_dev=t_dev)
        g = got.float().cpu().numpy()
        assert relerr(g, want) < tol
        want2 = g + (cond[:, t] if add else 0)
        assert np.abs(out2.cpu().numpy() - want2).max() < 1e-6 * max(1.0, np.abs(want2).max())


@pytest.mark.parametrize("H_,d,bc,T,spans", [(1, 480, 16, 2100, ((5, 2000), (0, 2048))), (2, 64, 8, 1200, ((3, 1100),)),
                                             (1, 480, 64, 8192, ((4096, 2048),))])
def test_attn_prefill_transpose_on_the_tiled_kernel(H, H_, d, bc, T, spans):
    """Transpose pattern on the 4-wave prefill kernel (round 4): a workgroup owns 64 consecutive MEMBERS of one residue class
    mod block_ctx; classes with more than 64 members in the chunk take several workgroups (125 / 138 members here), the last
    case is the upsamplers' second prefill chunk (positions 4096 .. 6143, 32 members per class, up to 96 keys each).  Against
    fp32 math on the half operands, and the one-wave kernel for the same call."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(T + d)
    N, S = 2, H_ * d
    K = h16(rng.standard_normal((N, T, S)).astype(np.float32))
    V = h16(rng.standard_normal((N, T, S)).astype(np.float32))
    kc, vc = dev(K, torch.float16), dev(V, torch.float16)
    for t0, nq in spans:
        q = h16(rng.standard_normal((N, nq, S)).astype(np.float32))
        got = H.attn_prefill(2, dev(q, torch.float16), kc, vc, H_, bc, t0).float().cpu().numpy()
        L.lib().jb_tune_attn_prefill_v2(0)
        try:
            one_wave = H.attn_prefill(2, dev(q, torch.float16), kc, vc, H_, bc, t0).float().cpu().numpy()
        finally:
            L.lib().jb_tune_attn_prefill_v2(1)
        want = _np_attention(2, q, K, V, H_, bc, None, list(range(t0, t0 + nq)), False)
        assert np.abs(got - want).max() < 6e-3 * max(1.0, np.abs(want).max()), (t0, nq)
        assert np.abs(got - one_wave).max() < 6e-3 * max(1.0, np.abs(want).max()), (t0, nq)


@pytest.mark.parametrize("v2", [1, 0])
@pytest.mark.parametrize("func", [0, 1, 2, 3, 7])
@pytest.mark.parametrize("H_,d,bc", [(1, 480, 128), (2, 64, 8), (1, 120, 8)])
def test_attn_prefill_v2(H, func, H_, d, bc, v2):
    """fp16 prefill attention: the default 4-wave kernel sharing vector-staged K/V tiles (jb_tune_attn_prefill_v2(1)) and
    the one-wave kernel, against fp32 math on the half operands."""
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(func * 10 + d + 1)
    N, prime_r = 2, 24
    T = 520 if bc == 128 else 120
    S = H_ * d
    cap = prime_r if func == 7 else T
    K = h16(rng.standard_normal((N, cap, S)).astype(np.float32))
    V = h16(rng.standard_normal((N, cap, S)).astype(np.float32))
    kc, vc = dev(K, torch.float16), dev(V, torch.float16)
    spans = ((0, 40), (0, 1), (5, 7), (13, 50), (64, 56), (37, 83)) if bc == 8 else ((0, 512), (100, 300), (384, 136), (7, 65))
    L.lib().jb_tune_attn_prefill_v2(v2)
    try:
        for t0, nq in spans:
            q = h16(rng.standard_normal((N, nq, S)).astype(np.float32))
            got = H.attn_prefill(func, dev(q, torch.float16), kc, vc, H_, bc, t0).float().cpu().numpy()
            want = _np_attention(func, q, K, V, H_, bc, prime_r, list(range(t0, t0 + nq)), False)
            assert np.abs(got - want).max() < 6e-3 * max(1.0, np.abs(want).max()), (func, t0, nq)
    finally:
        L.lib().jb_tune_attn_prefill_v2(1)


@pytest.mark.parametrize("name,dt,tol", [("f32", torch.float32, 2e-5), ("f16", torch.float16, 6e-3)])
@pytest.mark.parametrize("func", [0, 1, 2, 3, 7])
@pytest.mark.parametrize("H_,d", [(2, 16), (1, 120), (2, 64)])
def test_attn_prefill(H, name, dt, tol, func, H_, d):
    rng = np.random.default_rng(func * 10 + d + 1)
    N, T, bc, prime_r = 2, 120, 8, 24
    S = H_ * d
    fp16 = dt == torch.float16
    r = (lambda x: h16(x)) if fp16 else (lambda x: x)
    cap = prime_r if func == 7 else T
    K = r(rng.standard_normal((N, cap, S)).astype(np.float32))
    V = r(rng.standard_normal((N, cap, S)).astype(np.float32))
    kc, vc = dev(K, dt), dev(V, dt)
    for t0, nq in ((0, 40), (0, 1), (5, 7), (13, 50), (64, 56), (37, 83)):
        q = r(rng.standard_normal((N, nq, S)).astype(np.float32))
        got = H.attn_prefill(func, dev(q, dt), kc, vc, H_, bc, t0).float().cpu().numpy()
        # online softmax rounds unnormalised probabilities in half mode: compare against fp32 math on the
        # half operands with a half-precision tolerance
        want = _np_attention(func, q, K, V, H_, bc, prime_r, list(range(t0, t0 + nq)), False)
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), (func, t0, nq)


def test_sampler(H):
    rng = np.random.default_rng(11)
    N, bins, T = 64, 97, 8
    logits = rng.standard_normal((N, bins)).astype(np.float32) * 2
    logits[0, 3] = logits[0, 5] = logits[0].max() + 1        # tie -> lowest index
    ld = dev(logits)
    tokens = torch.zeros((N, T), dtype=torch.int64, device="cuda")
    t_dev = torch.tensor([2], dtype=torch.int32, device="cuda")
    preds = torch.zeros((N, T, bins), dtype=torch.float32, device="cuda")
    # greedy
    H.sample_logits(ld, H.make_sample_params(temp=0.7, top_k=1), tokens, t_dev, preds)
    got = tokens[:, 2].cpu().numpy()
    assert np.array_equal(got, logits.argmax(1)) and got[0] == 3
    assert np.array_equal(preds[:, 2].cpu().numpy(), logits)
    # top-k / nucleus: the set of tokens that can be drawn EQUALS the kept set of filter_logits (ops.py:113-142).  Rows are
    # built so that every kept entry has probability >= ~4 %: over 64 rows x 40 seeds x 8 positions a kept entry that never
    # shows up, or a dropped one that does, fails the test.
    K2 = 24
    base = rng.standard_normal((N, K2)).astype(np.float32) * 0.3
    base[:, 8:] -= 6.0                                          # a clear tail
    l2 = dev(base)
    tok2 = torch.zeros((N, T), dtype=torch.int64, device="cuda")
    for kw in (dict(top_k=5), dict(top_k=8), dict(top_p=0.6), dict(top_p=0.9)):
        seen = np.zeros((N, K2), bool)
        for seed in range(40):
            for t in range(T):
                t_dev.fill_(t)
                H.sample_logits(l2, H.make_sample_params(temp=1.0, seed=seed, **kw), tok2, t_dev)
            g = tok2.cpu().numpy()
            for t in range(T):
                seen[np.arange(N), g[:, t]] = True
        kept = np.isfinite(O.filter_logits(base, **kw))
        assert np.array_equal(seen, kept), kw
    t_dev.fill_(2)
    # categorical frequencies: identical rows, different sample ids -> softmax(logits / temp)
    row = rng.standard_normal(16).astype(np.float32)
    ld2 = dev(np.tile(row, (64, 1)))
    counts = np.zeros(16)
    for seed in range(200):
        H.sample_logits(ld2, H.make_sample_params(temp=0.9, seed=seed), tokens, t_dev)
        counts += np.bincount(tokens[:, 2].cpu().numpy(), minlength=16)
    p = O.softmax(row / np.float32(0.9))
    n = counts.sum()
    z = (counts - n * p) / np.sqrt(n * p * (1 - p) + 1e-9)
    assert np.abs(z).max() < 5.0, z
    # the production vocabulary sizes take the same kernel (2048 upsamplers, 2127 = 2048 + 79 top prior: n2 = 4096)
    for bins2 in (2048, 2127):
        lg = rng.standard_normal((16, bins2)).astype(np.float32) * 3
        tk = torch.zeros((16, T), dtype=torch.int64, device="cuda")
        H.sample_logits(dev(lg), H.make_sample_params(temp=1.0, top_k=1), tk, t_dev)
        assert np.array_equal(tk[:, 2].cpu().numpy(), lg.argmax(1))
        H.sample_logits(dev(lg), H.make_sample_params(temp=0.99, seed=3), tk, t_dev)
        g = tk[:, 2].cpu().numpy()
        assert ((0 <= g) & (g < bins2)).all() and (lg[np.arange(16), g] > lg.max(1) - 12).all()


def test_sampler_streams(H):
    """The uniform behind a draw is keyed by (seed, stream_id = level, global sample index, absolute position = pos_base + t):
    windows, levels, samples and seeds never share it, and the same key always gives the same token (sharded or
    re-windowed runs reproduce).  Identical flat rows make the token a monotone function of the uniform."""
    N, bins, T = 32, 1024, 16
    ld = torch.zeros((N, bins), dtype=torch.float32, device="cuda")

    def draw(t, **kw):
        tok = torch.zeros((N, T), dtype=torch.int64, device="cuda")
        t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
        H.sample_logits(ld, H.make_sample_params(temp=1.0, **kw), tok, t_dev)
        return tok[:, t].cpu().numpy()

    a = draw(3, seed=7, sample_base=0, pos_base=0, stream_id=0)
    assert np.array_equal(a, draw(3, seed=7, sample_base=0, pos_base=0, stream_id=0))
    assert len(set(a.tolist())) > N // 2                                       # samples differ
    for other in (dict(seed=8), dict(stream_id=1), dict(pos_base=4096), dict(sample_base=N)):
        kw = dict(seed=7, sample_base=0, pos_base=0, stream_id=0)
        kw.update(other)
        assert (draw(3, **kw) != a).mean() > 0.9, other
    # absolute position: window start 4096 + position 3 == window start 4000 + position 99 ... (same key, same draw)
    assert np.array_equal(draw(3, seed=7, pos_base=4096), draw(9, seed=7, pos_base=4090))
    # global sample index: rows 8.. of a run with sample_base 0 == rows 0.. of the shard that starts at 8
    assert np.array_equal(a[8:], draw(3, seed=7, sample_base=8)[:N - 8])


def test_sample_step_tail(H):
    """jb_sample_step: token, embedding of the next position written by the drawing workgroup, counter advanced once."""
    import ctypes as C
    from jukebox_amd import _lib as L
    rng = np.random.default_rng(2)
    N, bins, W, T = 16, 200, 64, 10
    logits = rng.standard_normal((N, bins)).astype(np.float32) * 3
    x_emb = rng.standard_normal((bins, W)).astype(np.float32)
    pos = rng.standard_normal((T, W)).astype(np.float32)
    cond = rng.standard_normal((N, T, W)).astype(np.float32)
    ld, xe, pe, cd = dev(logits), dev(x_emb), dev(pos), dev(cond)
    params = H.make_sample_params(temp=1.0, top_k=1)
    for dt in (torch.float16, torch.float32):
        for t in (0, 4, T - 1):
            tokens = torch.zeros((N, T), dtype=torch.int64, device="cuda")
            t_dev = torch.tensor([t], dtype=torch.int32, device="cuda")
            ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
            x_next = torch.full((N, W), 7.0, dtype=dt, device="cuda")
            L.check(L.lib().jb_sample_step(ld.data_ptr(), N, bins, params.data_ptr(), tokens.data_ptr(), tokens.stride(0),
                                           t_dev.data_ptr(), None, 0, L.dtype_code(dt), x_next.data_ptr(), xe.data_ptr(),
                                           pe.data_ptr(), cd.data_ptr(), cond.shape[1] * W, W, W, T, ticket.data_ptr(), L.stream()))
            torch.cuda.synchronize()
            tok = tokens[:, t].cpu().numpy()
            assert np.array_equal(tok, logits.argmax(1))
            assert int(t_dev.item()) == t + 1 and int(ticket.item()) == 0
            if t + 1 < T:
                want = x_emb[tok] + pos[t + 1] + cond[:, t + 1]
                got = x_next.float().cpu().numpy()
                assert np.abs(got - want).max() < (4e-3 if dt == torch.float16 else 1e-6) * max(1.0, np.abs(want).max())
            else:
                assert (x_next == 7.0).all()


def test_vq(H):
    rng = np.random.default_rng(12)
    bins, emb, M = 64, 16, 1000
    cb = rng.standard_normal((bins, emb)).astype(np.float32)
    codes = rng.integers(0, bins, (4, 250))
    got = H.vq_gather(dev(codes), dev(cb)).cpu().numpy()
    assert np.array_equal(got, cb[codes])
    x = rng.standard_normal((M, emb)).astype(np.float32)
    pw = H.pack_linear_w(dev(cb), torch.float32)          # logical [k=emb][j=bins] = cb^T
    xk = H.gemm(dev(x), pw)
    got = H.vq_argmin(dev(x), xk, dev(cb)).cpu().numpy()
    dist = (x ** 2).sum(-1, keepdims=True) - 2 * x @ cb.T + (cb ** 2).sum(-1)[None]
    want = dist.argmin(-1)
    agree = got == want
    # disagreements only at near-ties of the distance
    srt = np.sort(dist, 1)
    assert agree.mean() > 0.99 and np.all((srt[~agree, 1] - srt[~agree, 0]) < 1e-4)
