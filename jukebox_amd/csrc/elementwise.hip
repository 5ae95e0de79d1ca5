// Row-wise / gather kernels of the sampling path: LayerNorm, embedding assembly, logits sampling,
// VQ codebook gather and nearest-code search.  All HBM/L2-bound; one wave (64 lanes) per row where a
// row reduction is needed, coalesced along the channel dimension.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// One wave per row.  Rows whose width is a multiple of 8 are read once with 16-byte (f16) / 32-byte (f32) loads into
// registers (up to 8 vectors of 8 per lane: W <= 4096) and normalised from there -- two-pass statistics, one HBM read;
// other widths take the scalar loop.
template <typename T> struct Vec8 { T v[8]; };
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, TO* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t rows, int W, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const TI* xr = x + row * W;
    TO* yr = y + row * W;
    // 8-element vectors: 16 bytes for half, 32 for float -- the row bases must be aligned to the vector type used below
    if ((W & 7) == 0 && W <= 4096 && ((uintptr_t)x % (sizeof(TI) * 8)) == 0 && ((uintptr_t)y % (sizeof(TO) * 8)) == 0 &&
        ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0) {
        typedef Vec8<TI> __attribute__((aligned(sizeof(TI) * 8))) VI;
        typedef Vec8<TO> __attribute__((aligned(sizeof(TO) * 8))) VO;
        const int nvec = W >> 3;
        VI xv[8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                xv[i] = *reinterpret_cast<const VI*>(xr + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)xv[i].v[e];
            }
        }
        const float mean = jb_wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (lane + 64 * i < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xv[i].v[e] - mean; q += d * d; }
            }
        }
        const float rstd = 1.0f / sqrtf(jb_wave_sum(q) / (float)W + eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + v * 8), b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
                VO o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o.v[e] = (TO)(((float)xv[i].v[e] - mean) * rstd * g0[e] + b0[e]);
                    o.v[e + 4] = (TO)(((float)xv[i].v[e + 4] - mean) * rstd * g1[e] + b1[e]);
                }
                *reinterpret_cast<VO*>(yr + v * 8) = o;
            }
        }
        return;
    }
    float s = 0.f;
    for (int k = lane; k < W; k += 64) s += (float)xr[k];
    const float mean = jb_wave_sum(s) / (float)W;
    float q = 0.f;
    for (int k = lane; k < W; k += 64) {
        float d = (float)xr[k] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(jb_wave_sum(q) / (float)W + eps);
    for (int k = lane; k < W; k += 64) yr[k] = (TO)(((float)xr[k] - mean) * rstd * gamma[k] + beta[k]);
}

extern "C" int jb_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma,
                                const float* beta, int64_t rows, int width, float eps, void* stream) {
    JB_REQUIRE(x && y && gamma && beta, "null pointer");
    JB_REQUIRE(rows > 0 && width > 0, "empty input");
    dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == JB_F32 && y_dtype == JB_F32)
        layernorm_kernel<float, float><<<grid, 256, 0, s>>>((const float*)x, (float*)y, gamma, beta, rows, width, eps);
    else if (x_dtype == JB_F16 && y_dtype == JB_F16)
        layernorm_kernel<f16, f16><<<grid, 256, 0, s>>>((const f16*)x, (f16*)y, gamma, beta, rows, width, eps);
    else if (x_dtype == JB_F32 && y_dtype == JB_F16)
        layernorm_kernel<float, f16><<<grid, 256, 0, s>>>((const float*)x, (f16*)y, gamma, beta, rows, width, eps);
    else if (x_dtype == JB_F16 && y_dtype == JB_F32)
        layernorm_kernel<f16, float><<<grid, 256, 0, s>>>((const f16*)x, (float*)y, gamma, beta, rows, width, eps);
    else
        JB_REQUIRE(false, "bad dtype");
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_kernel(T* __restrict__ out, const int64_t* __restrict__ tokens, int64_t tok_stride,
                             const float* __restrict__ x_emb, const float* __restrict__ pos_emb,
                             const float* __restrict__ start, int64_t start_stride, const float* __restrict__ x_cond,
                             int64_t xc_n, int64_t xc_t, int W, int t0, const int* __restrict__ t_dev, int n_t) {
    const int c = blockIdx.x, n = blockIdx.y;
    const int t = (t_dev ? *t_dev : t0) + c;
    const float* src = (t == 0) ? start + (int64_t)n * start_stride
                                : x_emb + tokens[(int64_t)n * tok_stride + t - 1] * (int64_t)W;
    const float* pe = pos_emb + (int64_t)t * W;
    const float* cd = x_cond ? x_cond + (int64_t)n * xc_n + (int64_t)t * xc_t : nullptr;
    T* o = out + ((int64_t)n * n_t + c) * W;
    if ((W & 3) == 0) {
        for (int i = threadIdx.x * 4; i < W; i += blockDim.x * 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + i) + *reinterpret_cast<const f32x4*>(pe + i);
            if (cd) v += *reinterpret_cast<const f32x4*>(cd + i);
#pragma unroll
            for (int u = 0; u < 4; ++u) o[i + u] = (T)v[u];
        }
        return;
    }
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        float v = src[i] + pe[i];
        if (cd) v += cd[i];
        o[i] = (T)v;
    }
}

extern "C" int jb_embed(int out_dtype, void* out, const int64_t* tokens, int64_t tok_stride, const float* x_emb,
                        const float* pos_emb, const float* start, int64_t start_stride, const float* x_cond,
                        int64_t xc_n_stride, int64_t xc_t_stride, int n_batch, int width, int t0, const int* t_dev,
                        int n_t, void* stream) {
    JB_REQUIRE(out && tokens && x_emb && pos_emb && start, "null pointer");
    JB_REQUIRE(n_batch > 0 && width > 0 && n_t > 0, "bad dims");
    dim3 grid(n_t, n_batch);
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == JB_F16)
        embed_kernel<f16><<<grid, 256, 0, s>>>((f16*)out, tokens, tok_stride, x_emb, pos_emb, start, start_stride,
                                               x_cond, xc_n_stride, xc_t_stride, width, t0, t_dev, n_t);
    else
        embed_kernel<float><<<grid, 256, 0, s>>>((float*)out, tokens, tok_stride, x_emb, pos_emb, start, start_stride,
                                                 x_cond, xc_n_stride, xc_t_stride, width, t0, t_dev, n_t);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

template <typename T>
__global__ void final_add_kernel(const T* __restrict__ h, float* __restrict__ xf, int64_t xf_n, const float* __restrict__ x_cond,
                                 int64_t xc_n, int64_t xc_t, int W, int t0, const int* __restrict__ t_dev, int n_t) {
    const int c = blockIdx.x, n = blockIdx.y;
    const int t = (t_dev ? *t_dev : t0) + c;
    const int64_t row = (int64_t)n * n_t + c;
    const float* cd = x_cond ? x_cond + (int64_t)n * xc_n + (int64_t)t * xc_t : nullptr;
    float* o = xf + (int64_t)n * xf_n + (int64_t)c * W;
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        float v = (float)h[row * W + i];
        if (cd) v += cd[i];
        o[i] = v;
    }
}

extern "C" int jb_final_add(int h_dtype, const void* h, float* xf, int64_t xf_n_stride, const float* x_cond,
                            int64_t xc_n_stride, int64_t xc_t_stride, int n_batch, int width, int t0, const int* t_dev,
                            int n_t, void* stream) {
    JB_REQUIRE(h && xf, "null pointer");
    if (xf_n_stride == 0) xf_n_stride = (int64_t)n_t * width;
    dim3 grid(n_t, n_batch);
    hipStream_t s = (hipStream_t)stream;
    if (h_dtype == JB_F16)
        final_add_kernel<f16><<<grid, 256, 0, s>>>((const f16*)h, xf, xf_n_stride, x_cond, xc_n_stride, xc_t_stride, width, t0, t_dev, n_t);
    else
        final_add_kernel<float><<<grid, 256, 0, s>>>((const float*)h, xf, xf_n_stride, x_cond, xc_n_stride, xc_t_stride, width, t0, t_dev, n_t);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator: one independent uniform per (seed, stream, sample, absolute position).
// `stream` separates the levels of one job, `pos` is the ABSOLUTE token position (window start + position in the
// window), so successive windows and different levels never reuse a draw.
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
__device__ inline float philox_uniform(uint64_t seed, uint32_t stream, uint32_t sample, uint32_t pos) {
    uint32_t c0 = sample, c1 = pos, c2 = 0x4a756b65u ^ stream, c3 = 0x626f7821u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// 256-thread block primitives on wave shuffles: a wave-level step (6 cross-lane ops) plus one 4-entry LDS exchange.
// `scratch` holds 8 floats; every call ends with the block synchronised and scratch reusable.
__device__ __forceinline__ float wave_scan_incl(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
// inclusive prefix sum of v over the block in thread order; *total = sum over all threads
__device__ __forceinline__ float block_scan_incl(float v, float* scratch, float* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inc = wave_scan_incl(v);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    float base = 0.f, tot = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float sw = scratch[w];
        base += w < wave ? sw : 0.f;
        tot += sw;
    }
    __syncthreads();
    *total = tot;
    return inc + base;
}
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = jb_wave_sum(v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = jb_wave_max(v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    const float m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    __syncthreads();
    return m;
}
__device__ __forceinline__ int block_sum_int(int v, int* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    const int r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return r;
}

// What the decode step does after a token has been drawn (all null / zero for the stand-alone operator): the embedding
// of the NEXT position for this sample -- x_emb[token] + pos_emb[t+1] + x_cond[n][t+1] (get_emb, autoregressive.py:177-197)
// -- written by the workgroup that drew the token, and the position counter, advanced by the workgroup that finishes
// last (every workgroup has read *t_dev before it takes its ticket).  Two launches less per token step.
struct SampleTail {
    void* x_next; int x_dtype;
    const float* x_emb; const float* pos_emb; const float* x_cond; int64_t xc_n, xc_t;
    int W, seq_len;
    int* t_dev_w; unsigned* ticket;
};

// One workgroup (4 waves) per sample row: temperature, top-k / nucleus filter, categorical draw.
// PIPE: the last launch of a software-pipelined decode step (common.h, JbPipe): the logits are the producer launch's (read
// write-through after the wait); the next position's embedding and the counter are read by the FIRST launch of the next
// step, which is already waiting on the other stream: stored write-through, published at the end.
template <int PIPE>
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, int bins, int n2,
                                                     const jb_sample_params* __restrict__ params,
                                                     int64_t* __restrict__ tokens, int64_t tok_stride,
                                                     const int* __restrict__ t_dev, float* __restrict__ preds,
                                                     int64_t preds_n_stride, SampleTail tail, JbPipe pipe) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_x = sm;              // [bins] filtered, temperature-scaled logits
    __shared__ float s_f[8];
    __shared__ int s_i[8];
    __shared__ int s_pick, s_owner;
    const int n = blockIdx.x, tid = threadIdx.x;
    unsigned pipe_own = 0;
    if constexpr (PIPE == 1) pipe_own = jb_pipe_own(pipe);
    const int t = *t_dev;                                    // written by this launch slot's previous run: same stream
    const jb_sample_params P = *params;
    if constexpr (PIPE == 1) jb_pipe_wait(pipe, pipe_own);
    const float* row = logits + (int64_t)n * bins;
    float* pr = preds ? preds + (int64_t)n * preds_n_stride + (int64_t)t * bins : nullptr;
    for (int i = tid; i < bins; i += 256) {
        const float v = PIPE ? jb_ld_sc1(logits, (int64_t)n * bins + i) : row[i];
        if (pr) pr[i] = v;
        s_x[i] = v / P.temp;
    }
    if (tid == 0) { s_pick = 0x7fffffff; s_owner = 0x7fffffff; }
    __syncthreads();
    const int chunk = (bins + 255) / 256;
    const int lo = min(tid * chunk, bins), hi = min(lo + chunk, bins);     // this thread's contiguous index range

    const int top_k = min(P.top_k, bins);
    if (top_k == 1) {
        // greedy: the filtered set is {max}; lowest index wins a tie
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = lo; i < hi; ++i) {
            const float v = s_x[i];
            if (v > bv) { bv = v; bi = i; }
        }
        const float m = block_max(bv, s_f);
        if (bv == m && bi != 0x7fffffff) atomicMin(&s_pick, bi);
        __syncthreads();
    } else {
        if (top_k > 1 || P.top_p > 0.f) {
            // filter_logits (ops.py:113-142) without a sort: the threshold is found by bisection over the 32 bits of an
            // order-preserving integer key, every probe one block-wide count (top-k) or mass (nucleus) from registers --
            // wave shuffles plus a 4-entry LDS exchange per probe.
            //   top-k  : keep x  <=>  x >= k-th largest           = largest key T with #{key >= T} >= k
            //   nucleus: entry i >= 1 of the descending order is dropped iff the mass of the entries before it exceeds
            //            top_p, i.e. keep x  <=>  mass{y > x} <= top_p  <=>  key(x) >= smallest T with mass{key > T} <= top_p
            constexpr int CH = 16;                             // bins <= 4096
            unsigned key[CH];
            float ex[CH];
            float mloc = -INFINITY;
            for (int i = lo; i < hi; ++i) mloc = fmaxf(mloc, s_x[i]);
            const float mx = block_max(mloc, s_f);
            float tl = 0.f;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const bool in = lo + u < hi;
                const float v = in ? s_x[lo + u] : -INFINITY;
                const unsigned b = __float_as_uint(v);
                key[u] = in ? (b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u)) : 0u;      // ascending with the float order
                ex[u] = in ? expf(v - mx) : 0.f;
                tl += ex[u];
            }
            unsigned T;
            if (top_k > 1) {
                T = 0u;
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned cand = T | (1u << bit);
                    int cnt = 0;
#pragma unroll
                    for (int u = 0; u < CH; ++u) cnt += (lo + u < hi && key[u] >= cand) ? 1 : 0;
                    if (block_sum_int(cnt, s_i) >= top_k) T = cand;
                }
            } else {
                const float budget = P.top_p * block_sum(tl, s_f);
                T = 0xFFFFFFFFu;
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned cand = T & ~(1u << bit);
                    float above = 0.f;
#pragma unroll
                    for (int u = 0; u < CH; ++u) above += (lo + u < hi && key[u] > cand) ? ex[u] : 0.f;
                    if (block_sum(above, s_f) <= budget) T = cand;
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (lo + u < hi && key[u] < T) s_x[lo + u] = -INFINITY;
            __syncthreads();
        }
        // Categorical(logits).sample(): inverse CDF in index order.  Thread i owns the contiguous indices [lo, hi); the
        // block prefix sum of the per-thread masses locates the owner of the target, which walks its own chunk.
        float mloc = -INFINITY;
        for (int i = lo; i < hi; ++i) mloc = fmaxf(mloc, s_x[i]);
        const float m = block_max(mloc, s_f);
        float part = 0.f;
        int last_adm = -1;
        for (int i = lo; i < hi; ++i) {
            part += expf(s_x[i] - m);
            if (s_x[i] > -INFINITY) last_adm = i;
        }
        float total;
        const float inc = block_scan_incl(part, s_f, &total);
        const float u = philox_uniform(P.seed, (uint32_t)P.stream_id, (uint32_t)(P.sample_base + n), (uint32_t)(P.pos_base + t));
        const float target = u * total;
        // owner = first thread WITH MASS whose inclusive sum exceeds the target.  The shuffle-based prefix sums associate
        // differently per thread, so they are monotone only up to rounding: a thread whose whole chunk was filtered out
        // (part == 0) could otherwise win with nothing to pick from.
        if (inc > target && part > 0.f) atomicMin(&s_owner, tid);
        __syncthreads();
        if (tid == s_owner) {
            float cum = inc - part;
            int pick = last_adm;                               // rounding inside the chunk: its last admissible index
            for (int i = lo; i < hi; ++i) {
                cum += expf(s_x[i] - m);
                if (cum > target && s_x[i] > -INFINITY) { pick = i; break; }
            }
            if (pick >= 0) s_pick = pick;
        }
        __syncthreads();
        if (s_pick == 0x7fffffff) {
            // rounding pushed the target past the end: take the last admissible index of the row
            int la = last_adm;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) la = max(la, __shfl_xor(la, o, 64));
            if ((tid & 63) == 0) s_i[tid >> 6] = la;
            __syncthreads();
            if (tid == 0) s_pick = max(max(s_i[0], s_i[1]), max(s_i[2], s_i[3]));
            __syncthreads();
        }
    }
    const int tok = s_pick;
    if (tid == 0) tokens[(int64_t)n * tok_stride + t] = tok;
    if (tail.x_next && t + 1 < tail.seq_len) {
        const int W = tail.W;
        const float* src = tail.x_emb + (int64_t)tok * W;
        const float* pe = tail.pos_emb + (int64_t)(t + 1) * W;
        const float* cd = tail.x_cond ? tail.x_cond + (int64_t)n * tail.xc_n + (int64_t)(t + 1) * tail.xc_t : nullptr;
        if (PIPE && tail.x_dtype == JB_F16 && (W & 7) == 0) {
            // pipelined launches: 8 channels per thread, one 16-byte write-through store each (eight consecutive channels are
            // contiguous in operand order too: common.h)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            for (int i = tid * 8; i < W; i += 256 * 8) {
                f32x4 lo = *reinterpret_cast<const f32x4*>(src + i) + *reinterpret_cast<const f32x4*>(pe + i);
                f32x4 hi = *reinterpret_cast<const f32x4*>(src + i + 4) + *reinterpret_cast<const f32x4*>(pe + i + 4);
                if (cd) { lo += *reinterpret_cast<const f32x4*>(cd + i); hi += *reinterpret_cast<const f32x4*>(cd + i + 4); }
                const f16x8 o = {(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), jb_rsrc(tail.x_next),
                                                       (int)(((pipe.frag & JB_FRAG_OUT) ? (int64_t)jb_frag_el(n, i) : (int64_t)n * W + i) * 2), 0, 16);
            }
        } else
        for (int i = tid * 4; i < W; i += 256 * 4) {          // W % 4 == 0 is checked on the host
            f32x4 v = *reinterpret_cast<const f32x4*>(src + i) + *reinterpret_cast<const f32x4*>(pe + i);
            if (cd) v += *reinterpret_cast<const f32x4*>(cd + i);
            if (tail.x_dtype == JB_F16) {
                const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                if constexpr (PIPE) {
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), jb_rsrc(tail.x_next),
                                                          (int)(((pipe.frag & JB_FRAG_OUT) ? (int64_t)jb_frag_el(n, i) : (int64_t)n * W + i) * 2), 0, 16);
                } else {
                    *reinterpret_cast<f16x4*>((f16*)tail.x_next + (int64_t)n * W + i) = o;
                }
            } else {
                *reinterpret_cast<f32x4*>((float*)tail.x_next + (int64_t)n * W + i) = v;
            }
        }
    }
    if (tail.ticket && tid == 0) {
        const unsigned prev = atomicAdd(tail.ticket, 1u);
        if (prev == gridDim.x - 1) {                           // last workgroup: everyone has read *t_dev
            *tail.ticket = 0u;
            if constexpr (PIPE) jb_st_word(reinterpret_cast<unsigned*>(tail.t_dev_w), (unsigned)(t + 1)); else *tail.t_dev_w = t + 1;
        }
    }
    if constexpr (PIPE == 1) jb_pipe_publish(pipe, pipe_own);
}

static int launch_sample(const float* logits, int n_batch, int bins, const jb_sample_params* params, int64_t* tokens,
                         int64_t tok_stride, const int* t_dev, float* preds, int64_t preds_n_stride, const SampleTail& tail,
                         hipStream_t stream, const JbPipe* pipe = nullptr) {
    int n2 = 1;
    while (n2 < bins) n2 <<= 1;
    if (n2 > 4096) JB_UNSUPPORTED("vocabulary too large for the LDS sampler (bins <= 4096)");
    const size_t lds = (size_t)n2 * sizeof(float);
    if (pipe) {
        if (tail.x_dtype != JB_F16 || !tail.x_next) JB_UNSUPPORTED("a pipelined sampler launch takes the fp16 decode step's tail");
        if (pipe->slot >= 0 && pipe->proto >= 1 && n_batch < 8) JB_UNSUPPORTED("completion protocol 1 needs launches of >= 8 workgroups (>= 8 samples)");
        // (1: synchronised through the completion words; 2 = JB_PIPE_NO_SYNC: the same kernel form as a launch of a plain chain)
        if (pipe->slot >= 0)
            sample_kernel<1><<<n_batch, 256, lds, stream>>>(logits, bins, n2, params, tokens, tok_stride, t_dev, preds, preds_n_stride, tail, *pipe);
        else
            sample_kernel<2><<<n_batch, 256, lds, stream>>>(logits, bins, n2, params, tokens, tok_stride, t_dev, preds, preds_n_stride, tail, *pipe);
    } else {
        sample_kernel<0><<<n_batch, 256, lds, stream>>>(logits, bins, n2, params, tokens, tok_stride, t_dev, preds,
                                                             preds_n_stride, tail, JbPipe{nullptr, nullptr, nullptr, -1, -1, 0, nullptr});
    }
    JB_CHECK_LAUNCH();
    return JB_OK;
}

extern "C" int jb_sample_logits(const float* logits, int n_batch, int bins, const jb_sample_params* params,
                                int64_t* tokens, int64_t tok_stride, const int* t_dev, float* preds,
                                int64_t preds_n_stride, void* stream) {
    JB_REQUIRE(logits && params && tokens && t_dev, "null pointer");
    JB_REQUIRE(n_batch > 0 && bins > 0, "bad dims");
    SampleTail tail = {};
    return launch_sample(logits, n_batch, bins, params, tokens, tok_stride, t_dev, preds, preds_n_stride, tail, (hipStream_t)stream);
}

extern "C" int jb_sample_step(const float* logits, int n_batch, int bins, const jb_sample_params* params, int64_t* tokens,
                              int64_t tok_stride, int* t_dev, float* preds, int64_t preds_n_stride, int x_dtype, void* x_next,
                              const float* x_emb, const float* pos_emb, const float* x_cond, int64_t xc_n_stride,
                              int64_t xc_t_stride, int width, int seq_len, unsigned* ticket, void* stream) {
    return jb_sample_step_impl(logits, n_batch, bins, params, tokens, tok_stride, t_dev, preds, preds_n_stride, x_dtype, x_next, x_emb,
                               pos_emb, x_cond, xc_n_stride, xc_t_stride, width, seq_len, ticket, nullptr, stream);
}

int jb_sample_step_impl(const float* logits, int n_batch, int bins, const jb_sample_params* params, int64_t* tokens,
                        int64_t tok_stride, int* t_dev, float* preds, int64_t preds_n_stride, int x_dtype, void* x_next,
                        const float* x_emb, const float* pos_emb, const float* x_cond, int64_t xc_n_stride, int64_t xc_t_stride,
                        int width, int seq_len, unsigned* ticket, const JbPipe* pipe, void* stream) {
    JB_REQUIRE(logits && params && tokens && t_dev && x_next && x_emb && pos_emb && ticket, "null pointer");
    JB_REQUIRE(n_batch > 0 && bins > 0 && width > 0 && width % 4 == 0 && seq_len > 0, "bad dims (width must be a multiple of 4)");
    JB_REQUIRE(x_dtype == JB_F16 || x_dtype == JB_F32, "bad dtype");
    SampleTail tail = {};
    tail.x_next = x_next; tail.x_dtype = x_dtype; tail.x_emb = x_emb; tail.pos_emb = pos_emb; tail.x_cond = x_cond;
    tail.xc_n = xc_n_stride; tail.xc_t = xc_t_stride; tail.W = width; tail.seq_len = seq_len;
    tail.t_dev_w = t_dev; tail.ticket = ticket;
    return launch_sample(logits, n_batch, bins, params, tokens, tok_stride, t_dev, preds, preds_n_stride, tail, (hipStream_t)stream, pipe);
}

// ------------------------------------------------------------------------------------------------
__global__ void vq_gather_kernel(const int64_t* __restrict__ codes, const float* __restrict__ cb,
                                 float* __restrict__ out, int64_t total, int emb) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t r = i / emb;
    int c = (int)(i - r * emb);
    out[i] = cb[codes[r] * emb + c];
}

extern "C" int jb_vq_gather(const int64_t* codes, const float* codebook, float* out, int64_t n_codes, int emb_width,
                            int bins, void* stream) {
    JB_REQUIRE(codes && codebook && out, "null pointer");
    JB_REQUIRE(n_codes > 0 && emb_width > 0 && bins > 0, "bad dims");
    int64_t total = n_codes * emb_width;
    vq_gather_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(codes, codebook, out, total, emb_width);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// distance = ||x||^2 - 2 x.k + ||k||^2, first index of the minimum (bottleneck.py:112-119)
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ x, const float* __restrict__ xk,
                                                        const float* __restrict__ cb, int64_t* __restrict__ codes,
                                                        int64_t rows, int emb, int bins) {
    extern __shared__ __attribute__((aligned(16))) float s_k2[];
    for (int j = threadIdx.x; j < bins; j += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < emb; ++c) { float v = cb[(int64_t)j * emb + c]; s += v * v; }
        s_k2[j] = s;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < emb; c += 64) { float v = x[row * emb + c]; s += v * v; }
        const float x2 = jb_wave_sum(s);
        float bv = INFINITY; int bi = 0x7fffffff;
        for (int j = lane; j < bins; j += 64) {
            float d = x2 - 2.0f * xk[row * bins + j] + s_k2[j];
            if (d < bv) { bv = d; bi = j; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) codes[row] = bi;
    }
}

extern "C" int jb_vq_argmin(const float* x, const float* xk, const float* codebook, int64_t* codes, int64_t rows,
                            int emb_width, int bins, void* stream) {
    JB_REQUIRE(x && xk && codebook && codes, "null pointer");
    JB_REQUIRE(rows > 0 && emb_width > 0 && bins > 0, "bad dims");
    unsigned grid = (unsigned)((rows + 3) / 4);
    if (grid > 2048) grid = 2048;
    vq_argmin_kernel<<<grid, 256, (size_t)bins * sizeof(float), (hipStream_t)stream>>>(x, xk, codebook, codes, rows, emb_width, bins);
    JB_CHECK_LAUNCH();
    return JB_OK;
}
