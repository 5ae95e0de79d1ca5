// Decode step with THREE launches per layer (fp16 engines, n_batch <= 16) -- the projections that follow a residual add
// are taken from the operands of the add, and their LayerNorm is applied by the CONSUMER:
//
//   x_b = x_a + a.Wp + bp                                   (attn.c_proj + residual)
//   LN1(x_b).Wf + bf = rstd_b * (x_b.W'f - mean_b * c1f) + b'f                      W'f = diag(g1).Wf, c1f = colsum(W'f)
//   x_b.W'f          = x_a.W'f + a.(Wp.W'f) + bp.W'f                                (no x_b needed)
//   mean_b           = (sum(x_a) + a.rowsum(Wp) + sum(bp)) / W                      (no x_b needed either)
//
// so ONE launch ("A") produces the x_b tiles (with per-tile partial row sums of x_b and x_b^2) AND, from the same inputs,
// u_f = x_b.W'f - mean_b * c1f for every c_fc column.  Only rstd_b is missing, and it is a per-row scalar: the next launch
// ("B") sums the partials (a few KB), forms h = gelu(rstd_b * u_f + b'f) while it loads its operands, and produces
// x_a' = x_b + h.Wp2 + bp2 (tiles + partial sums) together with u_qkv = x_a'.W'a of the NEXT layer (fp32), again from the
// operands (x_b, h).  The attention launch ("C", attention.hip) finishes q, k, v = rstd_a' * (u_qkv - mean_a' * c1a) + b'a for
// its own sample from the partials, appends k / v to the cache and attends.  No workgroup ever waits for another workgroup of
// its launch (round 1's fused pair did, and gained nothing: an in-launch hand-off costs what a kernel boundary costs);
// the price is 21 % more weight bytes (the product matrix Wp2.W'a) in a chain that is latency-bound, not bandwidth-bound,
// and ~1 us of vector-ALU work per launch B for the on-the-fly gelu.
//
// Numerics: same rounding points as the folded-LayerNorm path (jb_gemv ln_fold_c1) except that u_f is stored in half
// before it is scaled (one extra half rounding on the c_fc pre-activation) and that the product matrices are rounded to
// half as matrices; the approximate mean differs from the mean of the stored half row by ~1e-5 of its magnitude.
// tests/test_hip_baseline_configs.py holds the path to the distance-to-fp32 of the reference-ordered fp16 engine.
#include "common.h"

namespace {

constexpr int NW = 8, E = 8, KT = 32;
typedef f16x8 V;

__device__ __forceinline__ void issue_fence() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ V ones_frag() {
    V v;
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = (f16)1.0f;
    return v;
}

// Sum over the 16 columns of a tile of one value per (row, column) thread: thread (wave r = tid >> 6 < 4, lane l) holds
// column (l >> 4) * 4 + r of row l & 15.  Result valid in s_red[row * 2 + which] after the call (32 entries).
__device__ __forceinline__ void tile_row_stats(float v, bool valid, float* s_part /* [4][16][2] */, float* s_red /* [32] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s1 = valid ? v : 0.f, s2 = valid ? v * v : 0.f;
    if (tid < 256) {
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if ((lane >> 4) == 0) { s_part[(wave * 16 + (lane & 15)) * 2] = s1; s_part[(wave * 16 + (lane & 15)) * 2 + 1] = s2; }
    }
    __syncthreads();
    if (tid < 32) {
        const int row = tid >> 1, which = tid & 1;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) tot += s_part[(w * 16 + row) * 2 + which];
        s_red[tid] = tot;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
struct FusedAParams {
    int n_rows, nkt0, nkt1, nA1, W;
    const f16* xa; int64_t ldx;
    const f16* att; int64_t lda;
    const f16* Wp; const float* bp; f16* xb; int64_t ldb; float* stats_b;
    const f16* Wfa; const float* kf; const float* c1f; const f16* wsum_p; float sum_bp; f16* uf; int64_t ldu; int M;
};

__global__ __launch_bounds__(512) void fused_a_kernel(FusedAParams p) {
    __shared__ __attribute__((aligned(16))) f32x4 s_acc[NW * 64];
    __shared__ float s_sum[NW * 16];
    __shared__ float s_part[4 * 16 * 2];
    __shared__ float s_red[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int rowc = min(c, p.n_rows - 1);
    const int erow = c, ecol = g * 4 + (wave & 3);            // this thread's output element (threads < 256)

    if ((int)blockIdx.x < p.nA1) {
        // ---- x_b tile = x_a + att . Wp + bp, partial row sums of the stored tile ----
        const int jt = blockIdx.x;
        const int kt0 = (wave * p.nkt1) / NW, kt1 = ((wave + 1) * p.nkt1) / NW;
        constexpr int NF = 8;                                  // n_state <= 2048
        V xf[NF], wf[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int kt = min(kt0 + i, p.nkt1 - 1);
            xf[i] = ld_frag<f16>(p.att + (int64_t)rowc * p.lda + kt * KT + g * E);
            wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(p.Wp + (((int64_t)jt * p.nkt1 + kt) * 64 + lane) * E));
        }
        const int j = jt * 16 + ecol;
        const float e_bias = p.bp[j];
        const float e_res = (float)p.xa[(int64_t)rowc * p.ldx + j];
        issue_fence();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NF; ++i) acc = jb_mfma(wf[i], keep_frag<f16>(kt0 + i < kt1, xf[i]), acc);
        s_acc[wave * 64 + lane] = acc;
        __syncthreads();
        float v = 0.f;
        const bool valid = tid < 256 && erow < p.n_rows;
        if (tid < 256) {
            const float* sa = reinterpret_cast<const float*>(s_acc);
#pragma unroll
            for (int w = 0; w < NW; ++w) v += sa[(w * 64 + lane) * 4 + (wave & 3)];
            v = jb_round<f16>(v + jb_round<f16>(e_bias));
            v = jb_round<f16>(e_res + v);
            if (valid) p.xb[(int64_t)erow * p.ldb + j] = (f16)v;
        }
        tile_row_stats(v, valid, s_part, s_red);
        if (tid < 32) p.stats_b[(int64_t)jt * 32 + tid] = s_red[tid];
        return;
    }

    // ---- u_f tile = x_a . W'f + att . (Wp . W'f) + bp . W'f - mean_b * c1f ----
    const int jt = blockIdx.x - p.nA1;
    const int nkt = p.nkt0 + p.nkt1;
    const int kt0 = (wave * nkt) / NW, kt1 = ((wave + 1) * nkt) / NW;
    constexpr int NF = 10;                                     // (W + n_state) / 32 / 8 <= 10: W <= 2048
    V xf[NF], wf[NF], sf[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int kt = min(kt0 + i, nkt - 1);
        const bool first = kt < p.nkt0;
        const f16* src = first ? p.xa + (int64_t)rowc * p.ldx + kt * KT : p.att + (int64_t)rowc * p.lda + (kt - p.nkt0) * KT;
        xf[i] = ld_frag<f16>(src + g * E);
        wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(p.Wfa + (((int64_t)jt * nkt + kt) * 64 + lane) * E));
        // row-sum operand: ones over the x_a part, rowsum(Wp) over the attention part (clamped address, selected below)
        sf[i] = ld_frag<f16>(p.wsum_p + max(kt - p.nkt0, 0) * KT + g * E);
    }
    const int j = jt * 16 + ecol, jc = min(j, p.M - 1);
    const float e_k = p.kf[jc], e_c1 = p.c1f[jc];
    issue_fence();
    const V ones = ones_frag();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const V xm = keep_frag<f16>(kt0 + i < kt1, xf[i]);
        acc = jb_mfma(wf[i], xm, acc);
        const bool first = min(kt0 + i, nkt - 1) < p.nkt0;
        V sv;
#pragma unroll
        for (int e = 0; e < E; ++e) sv[e] = first ? ones[e] : sf[i][e];
        a1 = jb_mfma(sv, xm, a1);                              // D[*][row] += sum_k s[k] x[row][k]
    }
    s_acc[wave * 64 + lane] = acc;
    if (g == 0) s_sum[wave * 16 + c] = a1[0];
    __syncthreads();
    if (tid < 256 && erow < p.n_rows && j < p.M) {
        const float* sa = reinterpret_cast<const float*>(s_acc);
        float v = 0.f, sm = p.sum_bp;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            v += sa[(w * 64 + lane) * 4 + (wave & 3)];
            sm += s_sum[w * 16 + erow];
        }
        const float mean = sm / (float)p.W;
        p.uf[(int64_t)erow * p.ldu + j] = (f16)(v + e_k - mean * e_c1);
    }
}

extern "C" int jb_fused_a(const jb_fused_a_args* a, void* stream) {
    JB_REQUIRE(a && a->xa && a->att && a->Wp && a->bp && a->xb && a->stats_b && a->Wfa && a->kf && a->c1f && a->wsum_p && a->uf,
               "null pointer");
    JB_REQUIRE(a->n_rows >= 1 && a->n_rows <= 16, "n_rows must be 1..16");
    JB_REQUIRE(a->W > 0 && a->W % 32 == 0 && a->W <= 2048 && a->S > 0 && a->S % 32 == 0 && a->S <= 2048 &&
                   a->W + a->S <= 2560 && a->M > 0 && a->M % 16 == 0, "unsupported dims (W, S multiples of 32; W <= 2048; W + S <= 2560)");
    JB_REQUIRE(a->ldx % 8 == 0 && a->lda % 8 == 0 && ((uintptr_t)a->xa % 16) == 0 && ((uintptr_t)a->att % 16) == 0 &&
                   ((uintptr_t)a->wsum_p % 16) == 0, "rows must be 16-byte aligned");
    FusedAParams p = {};
    p.n_rows = a->n_rows; p.nkt0 = a->W / 32; p.nkt1 = a->S / 32; p.nA1 = a->W / 16; p.W = a->W; p.M = a->M;
    p.xa = (const f16*)a->xa; p.ldx = a->ldx; p.att = (const f16*)a->att; p.lda = a->lda;
    p.Wp = (const f16*)a->Wp; p.bp = a->bp; p.xb = (f16*)a->xb; p.ldb = a->ldb; p.stats_b = a->stats_b;
    p.Wfa = (const f16*)a->Wfa; p.kf = a->kf; p.c1f = a->c1f; p.wsum_p = (const f16*)a->wsum_p; p.sum_bp = a->sum_bp;
    p.uf = (f16*)a->uf; p.ldu = a->ldu;
    fused_a_kernel<<<p.nA1 + a->M / 16, 512, 0, (hipStream_t)stream>>>(p);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
struct FusedBParams {
    int n_rows, nkt0, nktm, nB1, W, M, J2, n_stats; float ln_eps;
    const f16* xb; int64_t ldb;
    const f16* uf; int64_t ldu;
    const float* stats_b; const f16* bff;
    const f16* Wp2; const float* bp2; f16* xa_out; int64_t ldo; float* stats_a;
    float* out2; int64_t ldo2; const float* add2; int64_t add2_n, add2_t; const int* t_dev;
    const f16* Wfb; const float* ka; float* uq; int64_t ldq;
};

// h fragment from the un-normalised c_fc fragment: round(rstd * u + round(b'f)), then quick_gelu with the reference's half
// rounding points (ops.py:33-35) -- exactly the epilogue of the folded c_fc projection.
// The sigmoid uses the hardware exp2 / rcp (1 ulp-class fp32 results that are rounded to half right away): the fragment is
// formed on the critical path of every workgroup of the launch, and the libm expf / division cost 3x the instructions.
__device__ __forceinline__ V gelu_frag(V u, float rstd, V bias) {
    V h;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float pre = jb_round<f16>(rstd * (float)u[e] + (float)bias[e]);
        const float z = jb_round<f16>(1.702f * pre);
        const float sg = jb_round<f16>(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)));
        h[e] = (f16)(pre * sg);
    }
    return h;
}

// Launch B.  Workgroups [0, W/16): x_a' tiles (K = M, operand h).  Workgroups [W/16, W/16 + 3S/16): u_qkv tiles of the next
// layer (K = W + M, operands [x_b | h], fp32 output WITHOUT the mean term: the attention launch subtracts mean * c1a with
// the exact mean of the stored x_a' row -- fp32 has the headroom, a half store would not).  The k-tiles of a tile are dealt
// round-robin to the 8 waves (h tiles and x_b tiles separately), so every wave forms the same number of gelu fragments.
__global__ __launch_bounds__(512) void fused_b_kernel(FusedBParams p) {
    __shared__ __attribute__((aligned(16))) f32x4 s_acc[NW * 64];
    __shared__ float s_part[16 * 32];
    __shared__ float s_red[32];
    __shared__ float s_tot[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int rowc = min(c, p.n_rows - 1);
    const int erow = c, ecol = g * 4 + (wave & 3);
    const bool part1 = (int)blockIdx.x < p.nB1;

    // ---- every request of this workgroup, back to back: row-statistics partials, operands, weights, epilogue operands ----
    // partials of x_b: thread (rw = row * 2 + which = tid & 31, chunk = tid >> 5) sums tiles chunk, chunk + 16, ...
    float st = 0.f;
    {
        float tmp[16];
        const int rw = tid & 31, chunk = tid >> 5;
#pragma unroll
        for (int u = 0; u < 16; ++u) tmp[u] = p.stats_b[(int64_t)min(chunk + 16 * u, p.n_stats - 1) * 32 + rw];
#pragma unroll
        for (int u = 0; u < 16; ++u) st += (chunk + 16 * u < p.n_stats) ? tmp[u] : 0.f;
    }
    // k-tiles, dealt round-robin to the 8 waves: h tiles kh = wave + 8 i (operand = gelu of u_f), and -- second part only --
    // x_b tiles kx = wave + 8 i.  Two statically indexed groups, so the gelu loop has no per-fragment branch.
    const int nkt = part1 ? p.nktm : p.nkt0 + p.nktm;        // k-tiles of this workgroup's weight image
    const int h0 = part1 ? 0 : p.nkt0;                        // first image tile that multiplies h
    const int jt = part1 ? blockIdx.x : blockIdx.x - p.nB1;
    const f16* Wimg = part1 ? p.Wp2 : p.Wfb;
    constexpr int NFH = 8, NFX = 8;                           // M <= 2048, W <= 2048
    V hf[NFH], bf[NFH], wh[NFH], xf[NFX], wx[NFX];
#pragma unroll
    for (int i = 0; i < NFH; ++i) {
        const int kh = min(wave + NW * i, p.nktm - 1);         // tiles past the end are clamped here and zeroed below
        hf[i] = ld_frag<f16>(p.uf + (int64_t)rowc * p.ldu + kh * KT + g * E);
        bf[i] = ld_frag<f16>(p.bff + kh * KT + g * E);         // round16(b'f) of the fragment's 8 channels
        wh[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(Wimg + (((int64_t)jt * nkt + h0 + kh) * 64 + lane) * E));
    }
    if (!part1) {
#pragma unroll
        for (int i = 0; i < NFX; ++i) {
            const int kx = min(wave + NW * i, p.nkt0 - 1);
            xf[i] = ld_frag<f16>(p.xb + (int64_t)rowc * p.ldb + kx * KT + g * E);
            wx[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(Wimg + (((int64_t)jt * nkt + kx) * 64 + lane) * E));
        }
    }
    const int J = part1 ? p.W : p.J2;
    const int j = jt * 16 + ecol, jc = min(j, J - 1);
    const float e_a = part1 ? p.bp2[jc] : p.ka[jc];
    const float e_res = (float)p.xb[(int64_t)rowc * p.ldb + min(jc, p.W - 1)];
    issue_fence();
    float e_add2 = 0.f;
    if (part1 && p.add2) {                                     // last layer only: cond[:, t] for the logits head
        const int t = *p.t_dev;
        e_add2 = p.add2[(int64_t)rowc * p.add2_n + (int64_t)t * p.add2_t + jc];
    }

    // ---- rstd of the x_b rows from the partials ----
    s_part[(tid >> 5) * 32 + (tid & 31)] = st;                 // [chunk 0..15][row * 2 + which]
    __syncthreads();
    if (tid < 32) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += s_part[k * 32 + tid];
        s_tot[tid] = tot;
    }
    __syncthreads();
    const float inv_w = 1.0f / (float)p.W;
    const float mean_b = s_tot[rowc * 2] * inv_w;
    const float rstd_b = 1.0f / sqrtf(fmaxf(s_tot[rowc * 2 + 1] * inv_w - mean_b * mean_b, 0.f) + p.ln_eps);

    // ---- products ----
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!part1) {
#pragma unroll
        for (int i = 0; i < NFX; ++i) acc = jb_mfma(wx[i], keep_frag<f16>(wave + NW * i < p.nkt0, xf[i]), acc);
    }
#pragma unroll
    for (int i = 0; i < NFH; ++i) {
        const V hm = keep_frag<f16>(wave + NW * i < p.nktm, gelu_frag(hf[i], rstd_b, bf[i]));
        acc = jb_mfma(wh[i], hm, acc);
    }
    s_acc[wave * 64 + lane] = acc;
    __syncthreads();
    float v = 0.f;
    const bool valid = tid < 256 && erow < p.n_rows && j < J;
    if (tid < 256) {
        const float* sa = reinterpret_cast<const float*>(s_acc);
#pragma unroll
        for (int w = 0; w < NW; ++w) v += sa[(w * 64 + lane) * 4 + (wave & 3)];
    }
    if (part1) {
        if (tid < 256) {
            v = jb_round<f16>(v + jb_round<f16>(e_a));
            v = jb_round<f16>(e_res + v);
            if (valid) {
                p.xa_out[(int64_t)erow * p.ldo + j] = (f16)v;
                if (p.out2) p.out2[(int64_t)erow * p.ldo2 + j] = v + e_add2;
            }
        }
        tile_row_stats(v, valid, s_part, s_red);
        if (tid < 32) p.stats_a[(int64_t)jt * 32 + tid] = s_red[tid];
    } else if (valid) {
        p.uq[(int64_t)erow * p.ldq + j] = v + e_a;             // x_a' . W'a (+ bp2 . W'a): mean and rstd are applied by launch C
    }
}

extern "C" int jb_fused_b(const jb_fused_b_args* a, void* stream) {
    JB_REQUIRE(a && a->xb && a->uf && a->stats_b && a->bff16 && a->Wp2 && a->bp2 && a->xa_out && a->stats_a, "null pointer");
    JB_REQUIRE(a->n_rows >= 1 && a->n_rows <= 16, "n_rows must be 1..16");
    JB_REQUIRE(a->W > 0 && a->W % 32 == 0 && a->M > 0 && a->M % 32 == 0 && a->W + a->M <= 4096 && a->M <= 2048 && a->W <= 2048,
               "unsupported dims (W, M multiples of 32; W + M <= 4096)");
    JB_REQUIRE(a->n_stats >= 1 && a->n_stats <= 256, "n_stats must be 1..256");
    JB_REQUIRE(a->ldb % 8 == 0 && a->ldu % 8 == 0 && ((uintptr_t)a->xb % 16) == 0 && ((uintptr_t)a->uf % 16) == 0 &&
                   ((uintptr_t)a->bff16 % 16) == 0, "rows must be 16-byte aligned");
    JB_REQUIRE(!a->out2 || (a->ldo2 >= a->W && (!a->add2 || a->t_dev)), "bad second output");
    FusedBParams p = {};
    p.n_rows = a->n_rows; p.nkt0 = a->W / 32; p.nktm = a->M / 32; p.nB1 = a->W / 16; p.W = a->W; p.M = a->M; p.n_stats = a->n_stats;
    p.ln_eps = a->ln_eps;
    p.xb = (const f16*)a->xb; p.ldb = a->ldb; p.uf = (const f16*)a->uf; p.ldu = a->ldu; p.stats_b = a->stats_b;
    p.bff = (const f16*)a->bff16;
    p.Wp2 = (const f16*)a->Wp2; p.bp2 = a->bp2; p.xa_out = (f16*)a->xa_out; p.ldo = a->ldo; p.stats_a = a->stats_a;
    p.out2 = a->out2; p.ldo2 = a->ldo2; p.add2 = a->out2 ? a->add2 : nullptr; p.add2_n = a->add2_n_stride; p.add2_t = a->add2_t_stride;
    p.t_dev = a->t_dev;
    int n2 = 0;
    if (a->J2 > 0) {
        JB_REQUIRE(a->Wfb && a->ka && a->uq && a->J2 % 16 == 0 && a->ldq >= a->J2, "incomplete second part (u_qkv of the next layer)");
        p.J2 = a->J2; p.Wfb = (const f16*)a->Wfb; p.ka = a->ka; p.uq = a->uq; p.ldq = a->ldq;
        n2 = a->J2 / 16;
    }
    fused_b_kernel<<<p.nB1 + n2, 512, 0, (hipStream_t)stream>>>(p);
    JB_CHECK_LAUNCH();
    return JB_OK;
}
