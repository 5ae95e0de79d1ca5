// Row-wise / gather kernels of the sampling path: LayerNorm, embedding assembly, logits sampling,
// VQ codebook gather and nearest-code search.  All HBM/L2-bound; one wave (64 lanes) per row where a
// row reduction is needed, coalesced along the channel dimension.
#include "common.h"

// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, TO* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t rows, int W, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const TI* xr = x + row * W;
    float s = 0.f;
    for (int k = lane; k < W; k += 64) s += (float)xr[k];
    const float mean = jb_wave_sum(s) / (float)W;
    float q = 0.f;
    for (int k = lane; k < W; k += 64) {
        float d = (float)xr[k] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(jb_wave_sum(q) / (float)W + eps);
    TO* yr = y + row * W;
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
// Philox4x32-10 counter-based generator: one independent stream per (seed, sample, position).
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
__device__ inline float philox_uniform(uint64_t seed, uint32_t sample, uint32_t pos) {
    uint32_t c0 = sample, c1 = pos, c2 = 0x4a756b65u, c3 = 0x626f7821u;
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

// In-LDS bitonic sort, descending, n2 = power of two >= count (padding = -inf).
__device__ inline void bitonic_sort_desc(float* a, int n2) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    bool desc = (i & k) == 0;
                    float x = a[i], y = a[ixj];
                    if (desc ? (x < y) : (x > y)) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
}

// One workgroup per sample row.
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, int bins, int n2,
                                                     const jb_sample_params* __restrict__ params,
                                                     int64_t* __restrict__ tokens, int64_t tok_stride,
                                                     const int* __restrict__ t_dev, float* __restrict__ preds,
                                                     int64_t preds_n_stride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_x = sm;              // [bins] filtered, temperature-scaled logits
    float* s_sort = sm + n2;      // [n2] sort scratch (top-k / nucleus only)
    __shared__ float s_red[256];
    __shared__ int s_idx[256];
    __shared__ float s_thr;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int t = *t_dev;
    const jb_sample_params P = *params;
    const float* row = logits + (int64_t)n * bins;
    if (preds) {
        float* pr = preds + (int64_t)n * preds_n_stride + (int64_t)t * bins;
        for (int i = tid; i < bins; i += 256) pr[i] = row[i];
    }
    for (int i = tid; i < bins; i += 256) s_x[i] = row[i] / P.temp;
    __syncthreads();

    const int top_k = min(P.top_k, bins);
    if (top_k == 1) {
        // greedy: the filtered set is {max}; lowest index wins a tie
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < bins; i += 256) {
            float v = s_x[i];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
        s_red[tid] = bv; s_idx[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                float v = s_red[tid + o]; int ix = s_idx[tid + o];
                if (v > s_red[tid] || (v == s_red[tid] && ix < s_idx[tid])) { s_red[tid] = v; s_idx[tid] = ix; }
            }
            __syncthreads();
        }
        if (tid == 0) tokens[(int64_t)n * tok_stride + t] = s_idx[0];
        return;
    }
    if (top_k > 1 || P.top_p > 0.f) {
        for (int i = tid; i < n2; i += 256) s_sort[i] = i < bins ? s_x[i] : -INFINITY;
        __syncthreads();
        bitonic_sort_desc(s_sort, n2);
        if (tid == 0) {
            float thr;
            if (top_k > 1) {
                thr = s_sort[top_k - 1];                      // keep everything >= the k-th largest
            } else {
                // nucleus: sorted entry s is dropped iff the cumulative probability of entries < s exceeds top_p
                float mx = s_sort[0], tot = 0.f;
                for (int i = 0; i < bins; ++i) tot += expf(s_sort[i] - mx);
                float cum = 0.f; int keep = 1;
                for (int i = 0; i < bins - 1; ++i) {
                    cum += expf(s_sort[i] - mx) / tot;
                    if (cum > P.top_p) break;
                    keep = i + 2;
                }
                thr = s_sort[keep - 1];
            }
            s_thr = thr;
        }
        __syncthreads();
        const float thr = s_thr;
        for (int i = tid; i < bins; i += 256)
            if (s_x[i] < thr) s_x[i] = -INFINITY;
        __syncthreads();
    }

    // Categorical(logits).sample(): inverse CDF over contiguous per-thread chunks (index order)
    float m = -INFINITY;
    for (int i = tid; i < bins; i += 256) m = fmaxf(m, s_x[i]);
    s_red[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] = fmaxf(s_red[tid], s_red[tid + o]);
        __syncthreads();
    }
    m = s_red[0];
    __syncthreads();
    const int chunk = (bins + 255) / 256;
    const int lo = tid * chunk, hi = min(lo + chunk, bins);
    float part = 0.f;
    for (int i = lo; i < hi; ++i) part += expf(s_x[i] - m);
    s_red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        float total = 0.f;
        for (int i = 0; i < 256; ++i) total += s_red[i];
        const float u = philox_uniform(P.seed, (uint32_t)(P.sample_base + n), (uint32_t)t);
        const float target = u * total;
        float cum = 0.f;
        int c = 0;
        for (; c < 255; ++c) {
            if (cum + s_red[c] > target) break;
            cum += s_red[c];
        }
        int pick = -1, last_pos = 0;
        for (int i = c * chunk; i < min((c + 1) * chunk, bins); ++i) {
            float e = expf(s_x[i] - m);
            if (e > 0.f) last_pos = i;
            cum += e;
            if (cum > target) { pick = i; break; }
        }
        if (pick < 0) {
            // rounding pushed the target past the end: take the last admissible index
            pick = last_pos;
            for (int i = bins - 1; i >= 0; --i)
                if (s_x[i] > -INFINITY) { pick = i; break; }
        }
        tokens[(int64_t)n * tok_stride + t] = pick;
    }
}

extern "C" int jb_sample_logits(const float* logits, int n_batch, int bins, const jb_sample_params* params,
                                int64_t* tokens, int64_t tok_stride, const int* t_dev, float* preds,
                                int64_t preds_n_stride, void* stream) {
    JB_REQUIRE(logits && params && tokens && t_dev, "null pointer");
    JB_REQUIRE(n_batch > 0 && bins > 0, "bad dims");
    int n2 = 1;
    while (n2 < bins) n2 <<= 1;
    size_t lds = (size_t)2 * n2 * sizeof(float);
    if (lds > 128 * 1024) JB_UNSUPPORTED("vocabulary too large for the LDS sampler");
    if (lds > 64 * 1024)
        JB_HIP(hipFuncSetAttribute((const void*)sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    sample_kernel<<<n_batch, 256, lds, (hipStream_t)stream>>>(logits, bins, n2, params, tokens, tok_stride, t_dev,
                                                              preds, preds_n_stride);
    JB_CHECK_LAUNCH();
    return JB_OK;
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
