// MFMA GEMM kernels for gfx950: weight packing, the tiled sequence/tap GEMM (prefill projections,
// conv stacks, logits over chunks) and the weight-streaming skinny GEMM of the decode step.
//
// Data layout.  Activations are channels-last rows [row][channel] (the reference's NTC).  Weights are
// re-laid once into MFMA fragment order  P[jt][kt][lane][e] = W[k = kt*KT + (lane>>4)*E + e][j = jt*16 + (lane&15)]
// (zero padded), so a wave's load of one 16x(KT) weight tile is 64 lanes x 16 B = 1 KiB contiguous and
// lands directly in the A-operand registers -- no LDS round trip for an operand that is streamed
// once (cdna_hip_programming.md "GEMV / M <= 16 decode weights").  The MFMA computes
// D[j][m] = sum_k W[k][j] * X[m][k], i.e. the output tile transposed: lane l holds row m = l&15 and
// the four consecutive columns j = (l>>4)*4 + r, which it stores as one 8/16-byte vector.
#include "common.h"

// ------------------------------------------------------------------------------------------------
template <typename TO>
__global__ void pack_weight_kernel(const void* __restrict__ src, int src_dtype, int64_t sk, int64_t sj, int K, int J,
                                   TO* __restrict__ dst, int nkt, int64_t total) {
    constexpr int E = Frag<TO>::E, KT = Frag<TO>::KT;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int e = (int)(idx % E);
    int lane = (int)((idx / E) % 64);
    int64_t tile = idx / (E * 64);
    int kt = (int)(tile % nkt);
    int jt = (int)(tile / nkt);
    int k = kt * KT + (lane >> 4) * E + e;
    int j = jt * 16 + (lane & 15);
    float v = 0.0f;
    if (k < K && j < J) v = jb_load_any(src, src_dtype, (int64_t)k * sk + (int64_t)j * sj);
    dst[idx] = (TO)v;
}

static inline void packed_dims(int K, int J, int dtype, int* nkt, int* njt, int* E) {
    int KT = dtype == JB_F16 ? 32 : 16;
    *E = dtype == JB_F16 ? 8 : 4;
    *nkt = (K + KT - 1) / KT;
    *njt = (J + 15) / 16;
}

extern "C" int64_t jb_packed_weight_bytes(int K, int J, int dtype) {
    int nkt, njt, E;
    packed_dims(K, J, dtype, &nkt, &njt, &E);
    return (int64_t)nkt * njt * 64 * E * (dtype == JB_F16 ? 2 : 4);
}

extern "C" int jb_pack_weight(const void* src, int src_dtype, int64_t stride_k, int64_t stride_j, int K, int J,
                              void* dst, int dst_dtype, void* stream) {
    JB_REQUIRE(src && dst && K > 0 && J > 0, "null pointer or empty matrix");
    JB_REQUIRE((src_dtype == JB_F32 || src_dtype == JB_F16) && (dst_dtype == JB_F32 || dst_dtype == JB_F16), "bad dtype");
    int nkt, njt, E;
    packed_dims(K, J, dst_dtype, &nkt, &njt, &E);
    int64_t total = (int64_t)nkt * njt * 64 * E;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dst_dtype == JB_F16)
        pack_weight_kernel<f16><<<grid, 256, 0, (hipStream_t)stream>>>(src, src_dtype, stride_k, stride_j, K, J, (f16*)dst, nkt, total);
    else
        pack_weight_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(src, src_dtype, stride_k, stride_j, K, J, (float*)dst, nkt, total);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------
// Operand fragment of one activation row: E consecutive channels from k0.
template <typename T>
__device__ __forceinline__ typename Frag<T>::vec load_row_frag(const T* __restrict__ row, bool valid, int k0, int K, bool vec) {
    constexpr int E = Frag<T>::E;
    typename Frag<T>::vec v = jb_zero_frag<T>();
    if (valid) {
        if (vec && k0 + E <= K) {
            v = *reinterpret_cast<const typename Frag<T>::vec*>(row + k0);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (k0 + e < K) v[e] = row[k0 + e];
        }
    }
    return v;
}

// Shared epilogue: bias -> round -> activation -> residual -> store (incl. the q / k-cache / v-cache split).
struct EpiParams {
    const float* bias;
    void* out; int64_t ldo;
    const void* res; int64_t ldr;
    int J, act; float res_scale;
    int qkv_split, S;
    void* kcache; void* vcache; int cache_cap;
    int vec_out;        // 4 consecutive columns may be stored as one vector
};

// vals[r] is the accumulator of column jb + r of output row `orow`; cache_row < 0 disables the k/v write.
template <typename T>
__device__ __forceinline__ void epilogue_store(const EpiParams& p, f32x4 acc, int64_t orow, int jb, int64_t cache_row) {
    float v[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int j = jb + r;
        ok[r] = j < p.J;
        float x = acc[r];
        if (ok[r]) {
            if (p.bias) x += jb_round<T>(p.bias[j]);
            x = jb_round<T>(x);
            x = jb_apply_act<T>(x, p.act);
            if (p.res) {
                float rr = (float)((const T*)p.res)[orow * p.ldr + j];
                x = (p.res_scale == 1.0f) ? jb_round<T>(rr + x) : jb_round<T>(rr + jb_round<T>(p.res_scale * x));
            }
        }
        v[r] = x;
    }
    if (!p.qkv_split) {
        T* dst = (T*)p.out + orow * p.ldo + jb;
        if (p.vec_out && ok[3]) {
            if constexpr (sizeof(T) == 2) {
                f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(dst) = o;
            } else {
                f32x4 o = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst) = o;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ok[r]) dst[r] = (T)v[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!ok[r]) continue;
            int j = jb + r;
            int part = j / p.S;
            int jj = j - part * p.S;
            if (part == 0) {
                ((T*)p.out)[orow * p.ldo + jj] = (T)v[r];
            } else if (cache_row >= 0) {
                T* c = (T*)(part == 1 ? p.kcache : p.vcache);
                c[cache_row * p.S + jj] = (T)v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct GemmParams {
    const void* A; int64_t lda;
    const void* W; int64_t tap_stride;
    int n_seq, t_in, t_out;
    int64_t in_seq_stride, out_seq_stride;
    int K, nkt, njt;
    int n_taps, in_stride; int shift[4];
    int out_stride, out_offset;
    int pre_relu, vec_a;
    int cache_t0;
    int64_t m_total;
    EpiParams epi;
};

// Block = 4 waves; wave w owns 64 rows (4 m-tiles) x 64 columns (4 j-tiles) of the 256 x 64 block tile.
// Weight fragments are shared by the 4 waves through L1/L2; activations are private to a wave.
template <typename T>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E, KT = Frag<T>::KT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t m_base = (int64_t)blockIdx.x * 256 + wave * 64;
    const int jt_base = blockIdx.y * 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    int n_idx[4], t_idx[4];
    bool mvalid[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int64_t q = m_base + mt * 16 + c;
        mvalid[mt] = q < p.m_total;
        int64_t qq = mvalid[mt] ? q : 0;
        n_idx[mt] = (int)(qq / p.t_out);
        t_idx[mt] = (int)(qq - (int64_t)n_idx[mt] * p.t_out);
    }
    if (m_base >= p.m_total) return;   // whole wave out of range (uniform per wave)

    const T* A = (const T*)p.A;
    for (int tap = 0; tap < p.n_taps; ++tap) {
        const T* arow[4];
        bool aval[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int tin = t_idx[mt] * p.in_stride + p.shift[tap];
            aval[mt] = mvalid[mt] && tin >= 0 && tin < p.t_in;
            arow[mt] = A + ((int64_t)n_idx[mt] * p.in_seq_stride + (aval[mt] ? tin : 0)) * p.lda;
        }
        const T* wtap = (const T*)p.W + (int64_t)tap * p.tap_stride + (int64_t)lane * E;
        for (int kt = 0; kt < p.nkt; ++kt) {
            V wf[4], af[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                int jtg = jt_base + jt;
                wf[jt] = jb_zero_frag<T>();
                if (jtg < p.njt) wf[jt] = *reinterpret_cast<const V*>(wtap + ((int64_t)jtg * p.nkt + kt) * (64 * E));
            }
            const int k0 = kt * KT + g * E;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                af[mt] = load_row_frag<T>(arow[mt], aval[mt], k0, p.K, p.vec_a);
                if (p.pre_relu) {
#pragma unroll
                    for (int e = 0; e < E; ++e) af[mt][e] = af[mt][e] > (T)0 ? af[mt][e] : (T)0;
                }
            }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[jt][mt] = jb_mfma(wf[jt], af[mt], acc[jt][mt]);
        }
    }

#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (!mvalid[mt]) continue;
        int64_t orow = (int64_t)n_idx[mt] * p.out_seq_stride + (int64_t)t_idx[mt] * p.out_stride + p.out_offset;
        int64_t cache_row = -1;
        if (p.epi.qkv_split) {
            int ct = p.cache_t0 + t_idx[mt];
            if (ct < p.epi.cache_cap) cache_row = (int64_t)n_idx[mt] * p.epi.cache_cap + ct;
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            int jb = (jt_base + jt) * 16 + g * 4;
            if (jb < p.epi.J) epilogue_store<T>(p.epi, acc[jt][mt], orow, jb, cache_row);
        }
    }
}

static inline bool aligned_to(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

extern "C" int jb_gemm(const jb_gemm_args* a, void* stream) {
    JB_REQUIRE(a && a->A && a->W && a->out, "null pointer");
    JB_REQUIRE(a->dtype == JB_F32 || a->dtype == JB_F16, "bad dtype");
    JB_REQUIRE(a->K > 0 && a->J > 0 && a->n_seq > 0 && a->t_out > 0 && a->t_in > 0, "empty problem");
    JB_REQUIRE(a->n_taps >= 1 && a->n_taps <= 4, "n_taps must be 1..4");
    JB_REQUIRE(!a->qkv_split || (a->S > 0 && a->J == 3 * a->S && a->kcache && a->vcache), "bad qkv split");
    const int esz = a->dtype == JB_F16 ? 2 : 4;
    const int E = a->dtype == JB_F16 ? 8 : 4;
    GemmParams p;
    p.A = a->A; p.lda = a->lda; p.W = a->W; p.tap_stride = a->tap_stride;
    p.n_seq = a->n_seq; p.t_in = a->t_in; p.t_out = a->t_out;
    p.in_seq_stride = a->in_seq_stride; p.out_seq_stride = a->out_seq_stride;
    p.K = a->K;
    int Edummy;
    packed_dims(a->K, a->J, a->dtype, &p.nkt, &p.njt, &Edummy);
    p.n_taps = a->n_taps; p.in_stride = a->in_stride;
    for (int i = 0; i < 4; ++i) p.shift[i] = a->shift[i];
    p.out_stride = a->out_stride; p.out_offset = a->out_offset;
    p.pre_relu = a->pre_relu;
    p.vec_a = (a->lda % E == 0) && aligned_to(a->A, 16);
    p.cache_t0 = a->cache_t0;
    p.m_total = (int64_t)a->n_seq * a->t_out;
    p.epi.bias = a->bias; p.epi.out = a->out; p.epi.ldo = a->ldo; p.epi.res = a->res; p.epi.ldr = a->ldr;
    p.epi.J = a->J; p.epi.act = a->act; p.epi.res_scale = a->res_scale;
    p.epi.qkv_split = a->qkv_split; p.epi.S = a->S; p.epi.kcache = a->kcache; p.epi.vcache = a->vcache;
    p.epi.cache_cap = a->cache_cap;
    p.epi.vec_out = !a->qkv_split && (a->ldo % 4 == 0) && aligned_to(a->out, 4 * esz);
    dim3 grid((unsigned)((p.m_total + 255) / 256), (unsigned)((p.njt + 3) / 4));
    if (a->dtype == JB_F16)
        gemm_kernel<f16><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    else
        gemm_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------
struct GemvParams {
    const void* x; int64_t ldx; int n_rows;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const void* W; int K, nkt;
    int vec_x;
    const int* t_dev;
    EpiParams epi;
};

// Decode-step GEMM: n_rows <= 16*MT activation rows against a K x J weight matrix that is read from
// HBM exactly once per launch.  One workgroup per 16-column tile; its 4 waves split the k-tiles, each
// streaming 1 KiB weight fragments straight into MFMA operands (non-temporal: a weight byte is used
// once per step), then the four partial 16x16 tiles are summed in a fixed order through LDS
// (deterministic; no atomics).  LayerNorm of the activation rows is recomputed per workgroup from
// L2 (the rows are <= 64 x W) and applied while building the B operand.
template <typename T, int MT>
__global__ __launch_bounds__(256) void gemv_kernel(GemvParams p) {
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E, KT = Frag<T>::KT;
    __shared__ float s_mean[16 * MT], s_rstd[16 * MT];
    __shared__ f32x4 s_acc[4][MT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int jt = blockIdx.x;
    const T* x = (const T*)p.x;
    const bool ln = p.ln_gamma != nullptr;

    // issue the first weight loads before the LayerNorm statistics so they overlap
    const int kt0 = (wave * p.nkt) / 4, kt1 = ((wave + 1) * p.nkt) / 4;
    const T* wbase = (const T*)p.W + ((int64_t)jt * p.nkt) * (64 * E) + (int64_t)lane * E;

    if (ln) {
        for (int r = wave; r < p.n_rows; r += 4) {
            const T* xr = x + (int64_t)r * p.ldx;
            float s = 0.f;
            for (int k = lane; k < p.K; k += 64) s += (float)xr[k];
            s = jb_wave_sum(s);
            const float mean = s / (float)p.K;
            float q = 0.f;
            for (int k = lane; k < p.K; k += 64) {
                float d = (float)xr[k] - mean;
                q += d * d;
            }
            q = jb_wave_sum(q);
            if (lane == 0) {
                s_mean[r] = mean;
                s_rstd[r] = 1.0f / sqrtf(q / (float)p.K + p.ln_eps);
            }
        }
        __syncthreads();
    }

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mean[MT], rstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 16 + c;
        mean[mt] = (ln && row < p.n_rows) ? s_mean[row] : 0.f;
        rstd[mt] = (ln && row < p.n_rows) ? s_rstd[row] : 0.f;
    }

#pragma unroll 4
    for (int kt = kt0; kt < kt1; ++kt) {
        V wf = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)kt * (64 * E)));
        const int k0 = kt * KT + g * E;
        float gam[E], bet[E];
        if (ln) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                int k = k0 + e;
                gam[e] = k < p.K ? p.ln_gamma[k] : 0.f;
                bet[e] = k < p.K ? p.ln_beta[k] : 0.f;
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int row = mt * 16 + c;
            V xf = load_row_frag<T>(x + (int64_t)row * p.ldx, row < p.n_rows, k0, p.K, p.vec_x);
            if (ln) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float y = ((float)xf[e] - mean[mt]) * rstd[mt] * gam[e] + bet[e];
                    xf[e] = (k0 + e < p.K) ? (T)y : (T)0;
                }
            }
            acc[mt] = jb_mfma(wf, xf, acc[mt]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s_acc[wave][mt][lane] = acc[mt];
    __syncthreads();
    if (wave != 0) return;

    int t = 0;
    if (p.epi.qkv_split) t = *p.t_dev;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 16 + c;
        if (row >= p.n_rows) continue;
        f32x4 v = s_acc[0][mt][lane];
        v += s_acc[1][mt][lane];
        v += s_acc[2][mt][lane];
        v += s_acc[3][mt][lane];
        int jb = jt * 16 + g * 4;
        int64_t cache_row = (p.epi.qkv_split && t < p.epi.cache_cap) ? (int64_t)row * p.epi.cache_cap + t : -1;
        if (jb < p.epi.J) epilogue_store<T>(p.epi, v, row, jb, cache_row);
    }
}

template <typename T>
static int launch_gemv(const GemvParams& p, int njt, hipStream_t s) {
    int mt = (p.n_rows + 15) / 16;
    switch (mt) {
        case 1: gemv_kernel<T, 1><<<njt, 256, 0, s>>>(p); break;
        case 2: gemv_kernel<T, 2><<<njt, 256, 0, s>>>(p); break;
        case 3: gemv_kernel<T, 3><<<njt, 256, 0, s>>>(p); break;
        default: gemv_kernel<T, 4><<<njt, 256, 0, s>>>(p); break;
    }
    return 0;
}

extern "C" int jb_gemv(const jb_gemv_args* a, void* stream) {
    JB_REQUIRE(a && a->x && a->W && a->out, "null pointer");
    JB_REQUIRE(a->dtype == JB_F32 || a->dtype == JB_F16, "bad dtype");
    JB_REQUIRE(a->n_rows >= 1 && a->n_rows <= 64, "n_rows must be 1..64");
    JB_REQUIRE(a->K > 0 && a->J > 0, "empty problem");
    JB_REQUIRE(!a->qkv_split || (a->S > 0 && a->J == 3 * a->S && a->kcache && a->vcache && a->t_dev), "bad qkv split");
    const int esz = a->dtype == JB_F16 ? 2 : 4;
    const int E = a->dtype == JB_F16 ? 8 : 4;
    GemvParams p;
    int njt, Edummy;
    packed_dims(a->K, a->J, a->dtype, &p.nkt, &njt, &Edummy);
    p.x = a->x; p.ldx = a->ldx; p.n_rows = a->n_rows;
    p.ln_gamma = a->ln_gamma; p.ln_beta = a->ln_beta; p.ln_eps = a->ln_eps;
    p.W = a->W; p.K = a->K;
    p.vec_x = (a->ldx % E == 0) && aligned_to(a->x, 16);
    p.t_dev = a->t_dev;
    p.epi.bias = a->bias; p.epi.out = a->out; p.epi.ldo = a->ldo; p.epi.res = a->res; p.epi.ldr = a->ldr;
    p.epi.J = a->J; p.epi.act = a->act; p.epi.res_scale = 1.0f;
    p.epi.qkv_split = a->qkv_split; p.epi.S = a->S; p.epi.kcache = a->kcache; p.epi.vcache = a->vcache;
    p.epi.cache_cap = a->cache_cap;
    p.epi.vec_out = !a->qkv_split && (a->ldo % 4 == 0) && aligned_to(a->out, 4 * esz);
    if (a->dtype == JB_F16) launch_gemv<f16>(p, njt, (hipStream_t)stream);
    else launch_gemv<float>(p, njt, (hipStream_t)stream);
    JB_CHECK_LAUNCH();
    return JB_OK;
}
