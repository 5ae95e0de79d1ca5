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
#include <mutex>

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

// The split image of an fp32 matrix (JB_F16_SPLIT, jb_gemm_args.w_split): part 0 = half(w), part 1 = half((w - part 0) * 2^11),
// each laid out like an f16 image.
__global__ void pack_weight_split_kernel(const void* __restrict__ src, int src_dtype, int64_t sk, int64_t sj, int K, int J,
                                         f16* __restrict__ dst, int nkt, int64_t total) {
    constexpr int E = Frag<f16>::E, KT = Frag<f16>::KT;
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
    const f16 hi = (f16)v;
    dst[idx] = hi;
    dst[total + idx] = (f16)((v - (float)hi) * 2048.0f);
}

static inline void packed_dims(int K, int J, int dtype, int* nkt, int* njt, int* E) {
    int KT = dtype == JB_F32 ? 16 : 32;
    *E = dtype == JB_F32 ? 4 : 8;
    *nkt = (K + KT - 1) / KT;
    *njt = (J + 15) / 16;
}

extern "C" int64_t jb_packed_weight_bytes(int K, int J, int dtype) {
    int nkt, njt, E;
    packed_dims(K, J, dtype, &nkt, &njt, &E);
    return (int64_t)nkt * njt * 64 * E * (dtype == JB_F32 ? 4 : 2) * (dtype == JB_F16_SPLIT ? 2 : 1);
}

extern "C" int jb_pack_weight(const void* src, int src_dtype, int64_t stride_k, int64_t stride_j, int K, int J,
                              void* dst, int dst_dtype, void* stream) {
    JB_REQUIRE(src && dst && K > 0 && J > 0, "null pointer or empty matrix");
    JB_REQUIRE((src_dtype == JB_F32 || src_dtype == JB_F16) &&
                   (dst_dtype == JB_F32 || dst_dtype == JB_F16 || dst_dtype == JB_F16_SPLIT), "bad dtype");
    int nkt, njt, E;
    packed_dims(K, J, dst_dtype, &nkt, &njt, &E);
    int64_t total = (int64_t)nkt * njt * 64 * E;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dst_dtype == JB_F16_SPLIT)
        pack_weight_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, src_dtype, stride_k, stride_j, K, J, (f16*)dst, nkt, total);
    else if (dst_dtype == JB_F16)
        pack_weight_kernel<f16><<<grid, 256, 0, (hipStream_t)stream>>>(src, src_dtype, stride_k, stride_j, K, J, (f16*)dst, nkt, total);
    else
        pack_weight_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(src, src_dtype, stride_k, stride_j, K, J, (float*)dst, nkt, total);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------
// Shared epilogue: bias -> round -> activation -> residual -> store (incl. the q / k-cache / v-cache split).
struct EpiParams {
    const float* bias;
    void* out; int64_t ldo;
    const void* res; int64_t ldr;
    int J, act; float res_scale;
    int qkv_split, S;
    void* kcache; void* vcache; int cache_cap;
    // column layout of a q/k/v split: q [0,S) | k [S,2S) | v [2S,2S+v_cols) (v_cols = S or 0) | v' [.., ..+v2w) -- v' is the
    // value already carried through attn.c_proj (single-head models, see jb_attn_decode_wide), cache rows of v2w elements
    void* vcache2; int v2w, v_cols;
    int vec_out;        // 4 consecutive columns may be stored as one vector
    // decode step, last layer: a second fp32 copy out2[row][j] = value + add2[row*add2_n + t*add2_t + j] (t = *t_dev) --
    // `x.float() + cond` before the logits head (autoregressive.py:226-227), fused into mlp.c_proj's epilogue
    float* out2; int64_t ldo2;
    const float* add2; int64_t add2_n, add2_t;
};

// Column j of a q/k/v(/v') projection goes to the query row or to row `cache_row` of the matching cache.
template <typename T>
__device__ __forceinline__ void qkv_store(const EpiParams& p, float x, int64_t orow, int j, int64_t cache_row) {
    if (j < p.S) { ((T*)p.out)[orow * p.ldo + j] = (T)x; return; }
    if (cache_row < 0) return;
    if (j < 2 * p.S) ((T*)p.kcache)[cache_row * p.S + (j - p.S)] = (T)x;
    else if (j < 2 * p.S + p.v_cols) ((T*)p.vcache)[cache_row * p.S + (j - 2 * p.S)] = (T)x;
    else ((T*)p.vcache2)[cache_row * p.v2w + (j - 2 * p.S - p.v_cols)] = (T)x;
}

// vals[r] is the accumulator of column jb + r of output row `orow`; cache_row < 0 disables the k/v write.
template <typename T>
__device__ __forceinline__ void epilogue_store(const EpiParams& p, f32x4 acc, int64_t orow, int jb, int64_t cache_row,
                                               const float* bias_pre = nullptr, const float* res_pre = nullptr) {
    float v[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int j = jb + r;
        ok[r] = j < p.J;
        float x = acc[r];
        if (ok[r]) {
            if (p.bias) x += jb_round<T>(bias_pre ? bias_pre[r] : p.bias[j]);
            x = jb_round<T>(x);
            x = jb_apply_act<T>(x, p.act);
            if (p.res) {
                float rr = res_pre ? res_pre[r] : (float)((const T*)p.res)[orow * p.ldr + j];
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
            qkv_store<T>(p, v[r], orow, jb + r, cache_row);
        }
    }
}

// One output element (decode GEMV epilogue, spread over all threads of the workgroup).
template <typename T>
__device__ __forceinline__ void epilogue_store1(const EpiParams& p, float x, int64_t orow, int j, int64_t cache_row,
                                                float bias_v, float res_v, float add2_v = 0.f) {
    if (p.bias) x += jb_round<T>(bias_v);
    x = jb_round<T>(x);
    x = jb_apply_act<T>(x, p.act);
    if (p.res) x = (p.res_scale == 1.0f) ? jb_round<T>(res_v + x) : jb_round<T>(res_v + jb_round<T>(p.res_scale * x));
    if (p.out2) p.out2[orow * p.ldo2 + j] = x + add2_v;
    if (!p.qkv_split) {
        ((T*)p.out)[orow * p.ldo + j] = (T)x;
    } else {
        qkv_store<T>(p, x, orow, j, cache_row);
    }
}

// The arithmetic of epilogue_store1 without the store.
template <typename T>
__device__ __forceinline__ float epilogue_value(const EpiParams& p, float x, float bias_v, float res_v) {
    if (p.bias) x += jb_round<T>(bias_v);
    x = jb_round<T>(x);
    x = jb_apply_act<T>(x, p.act);
    if (p.res) x = (p.res_scale == 1.0f) ? jb_round<T>(res_v + x) : jb_round<T>(res_v + jb_round<T>(p.res_scale * x));
    return x;
}
// LDS slot (in f32x4 units) of fragment lane `lane`'s accumulator inside a wave's 64-slot partial tile, pipelined launches:
// [row][column group] instead of [lane], see epi_coords<true>.
__device__ __forceinline__ int jb_pipe_tile_slot(int lane) { return ((lane & 15) << 2) | (lane >> 4); }

// Pipelined launches, thread = (row, column) of the tile (epi_coords<true>): the eight consecutive threads that hold columns
// jb .. jb+7 of output row `orow` hand them to the first, which stores them write-through in one 16-byte piece (f16; fp32: two);
// likewise the fp32 second output.  Quad broadcasts and a row shift by four lanes: seven DPP moves, no LDS.  Called by whole
// waves; jt16 = first column of the workgroup's tile (its destination region is uniform); frag_out: the output block is in
// operand order (common.h), where columns jb .. jb+7 of a row are contiguous as well.
template <typename T>
__device__ __forceinline__ void pipe_store8(const EpiParams& p, float x, float x2, int64_t orow, int jb, int jt16, int64_t cache_row, bool valid,
                                            bool frag_out) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const float a1 = jb_dpp<0x55>(x), a2 = jb_dpp<0xAA>(x), a3 = jb_dpp<0xFF>(x);                 // lanes 1, 2, 3 of the thread's quad
    const float a4 = jb_dpp<0x104>(x), a5 = jb_dpp<0x104>(a1), a6 = jb_dpp<0x104>(a2), a7 = jb_dpp<0x104>(a3);   // the next quad's (row_shl:4)
    float b[8] = {x2, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.out2) {
        b[1] = jb_dpp<0x55>(x2); b[2] = jb_dpp<0xAA>(x2); b[3] = jb_dpp<0xFF>(x2);
        b[4] = jb_dpp<0x104>(x2); b[5] = jb_dpp<0x104>(b[1]); b[6] = jb_dpp<0x104>(b[2]); b[7] = jb_dpp<0x104>(b[3]);
    }
    if (!valid || (threadIdx.x & 7)) return;
    if (p.out2) {
        const f32x4 lo = {b[0], b[1], b[2], b[3]}, hi = {b[4], b[5], b[6], b[7]};
        const int off = (int)((orow * p.ldo2 + jb) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), jb_rsrc(p.out2), off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), jb_rsrc(p.out2), off + 16, 0, 16);
    }
    T* base = (T*)p.out;
    int64_t el = (frag_out && !p.qkv_split) ? (int64_t)jb_frag_el((int)orow, jb) : orow * p.ldo + jb;     // (the query rows stay [row][channel])
    if (p.qkv_split && jt16 >= p.S) {
        if (cache_row < 0) return;
        if (jt16 < 2 * p.S) { base = (T*)p.kcache; el = cache_row * p.S + (jb - p.S); }
        else if (jt16 < 2 * p.S + p.v_cols) { base = (T*)p.vcache; el = cache_row * p.S + (jb - 2 * p.S); }
        else { base = (T*)p.vcache2; el = cache_row * p.v2w + (jb - 2 * p.S - p.v_cols); }
    }
    if constexpr (sizeof(T) == 2) {
        const f16x8 o = {(f16)x, (f16)a1, (f16)a2, (f16)a3, (f16)a4, (f16)a5, (f16)a6, (f16)a7};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), jb_rsrc(base), (int)(el * 2), 0, 16);
    } else {
        const f32x4 lo = {x, a1, a2, a3}, hi = {a4, a5, a6, a7};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), jb_rsrc(base), (int)(el * 4), 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), jb_rsrc(base), (int)(el * 4) + 16, 0, 16);
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
    unsigned* overflow;                  // gemm_split_kernel: the process's host-coherent "activation outside the half range" word
    float w_unscale;                     // gemm_split_kernel: 1 / (the power of two the weights were multiplied by when packed)
    const void* zero_line;               // gemm_split_glds_kernel: 128 bytes of zeros (what a row outside its sequence reads)
    EpiParams epi;
};

// Block = 4 waves; wave w owns 64 rows (4 m-tiles) x 64 columns (4 j-tiles) of the 256 x 64 block tile.
// Weight fragments are shared by the 4 waves through L1/L2; activations are private to a wave.
template <typename T, bool FAST>
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
        // a tap whose shifted rows all fall outside the sequence for this wave's 64 output rows contributes zeros: skip its
        // whole k-loop (the conditioner's dilations reach 2187 positions at sequence lengths of 2048 / 4096: up to two of
        // the three taps of a layer are out of range for every row, or for the rows near one end)
        if (!__any((int)(aval[0] || aval[1] || aval[2] || aval[3]))) continue;
        const T* wtap = (const T*)p.W + (int64_t)tap * p.tap_stride + (int64_t)lane * E;
        // Software-pipelined over the k-tiles (round 4): the 8 operand loads of tile kt + 1 are in flight while the 16 (f16) /
        // 64 (f32) MFMAs of tile kt issue, and the ReLU / validity selects are applied AFTER all loads of a tile have been
        // requested.  (Before: `if (pre_relu)` behind each row's load compiled to load -> s_waitcnt vmcnt(0) -> branch, four
        // dependent round trips per k-tile -- the conv stacks' fp32 kernel sat at 55 % of the fp32 MFMA peak.)  Two register
        // sets, the loop unrolled by two so that no operand array is indexed dynamically.  Same MFMAs in the same order.
        auto issue = [&](int kt, V (&wf)[4], V (&af)[4]) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int jtg = jt_base + jt;
                if (FAST) {
                    wf[jt] = ld_frag<T>(wtap + ((int64_t)min(jtg, p.njt - 1) * p.nkt + kt) * (64 * E));
                } else {
                    wf[jt] = jb_zero_frag<T>();
                    if (jtg < p.njt) wf[jt] = *reinterpret_cast<const V*>(wtap + ((int64_t)jtg * p.nkt + kt) * (64 * E));
                }
            }
            const int k0 = kt * KT + g * E;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (FAST) af[mt] = ld_frag<T>(arow[mt] + k0);
                else af[mt] = load_row_frag<T>(arow[mt], aval[mt], k0, p.K, p.vec_a);
            }
        };
        auto multiply = [&](V (&wf)[4], V (&af)[4]) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (FAST) af[mt] = keep_frag<T>(aval[mt], af[mt]);
                if (p.pre_relu) {
#pragma unroll
                    for (int e = 0; e < E; ++e) af[mt][e] = af[mt][e] > (T)0 ? af[mt][e] : (T)0;
                }
            }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[jt][mt] = jb_mfma(wf[jt], af[mt], acc[jt][mt]);
        };
        V wA[4], aA[4], wB[4], aB[4];
        issue(0, wA, aA);
        int kt = 0;
        for (; kt + 1 < p.nkt; kt += 2) {
            issue(kt + 1, wB, aB);
            multiply(wA, aA);
            issue(min(kt + 2, p.nkt - 1), wA, aA);      // unconditional (a clamped, unused re-load at the very end): a branch here
            multiply(wB, aB);                           // would make the compiler wait for ALL outstanding loads below
        }
        if (kt < p.nkt) multiply(wA, aA);
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

// The conv stacks (conditioner, VQ-VAE encoder / decoder: fp32 in the reference) on the f16 matrix cores at fp32 accuracy
// (jb_gemm_args.w_split).  Same block / wave tiling and the same tap logic as gemm_kernel<float>; a k-step is one f16 k-tile
// (32 channels).  Both operands are split x = hi + 2^-11 * lo with hi = half(x), lo = half((x - hi) * 2^11) -- the weights
// once, when they are packed (pack_weight_split_kernel), the activation fragments in registers after the step's loads have
// been requested -- and   w * a = w_hi*a_hi + 2^-11 * (w_hi*a_lo + w_lo*a_hi)   (+ 2^-22 w_lo*a_lo, dropped)
// goes through three v_mfma_f32_16x16x32_f16 with fp32 accumulation (every f16 product is exact in fp32); the 2^-11 terms
// have their own accumulators and join at the end.  The exact-fp32 instruction (v_mfma_f32_16x16x4_f32) has 1/16 of the f16
// rate: 3 instead of 8 instructions per 32 channels, each twice as fast.  Error against exact arithmetic: that of the fp32
// accumulation (1.5e-7 ... 5.8e-7 of the largest output at K = 3 x 64 ... 3 x 1024, as the fp32 kernel; tests/test_hip_kernels.py).
// Two workgroups per compute unit hide the operand latency; a second register set with the next k-step's operands in
// flight (one workgroup per compute unit) measured slower: 122 against 77 ms for the upsampler's conditioner at 16 samples,
// exact-fp32 kernel 133 ms (profiles/r04_bench_conditioner.log).
// Set (sticky) by gemm_split_kernel when an activation it was given does not fit a half (the hi part would be +-inf and the
// result garbage) or is a NaN.  The word lives in host-coherent memory (split_overflow_word): a launch that sees such a value
// stores 1 through the fabric, and jb_gemm_split_overflow reads it from the host without touching any queue -- the sampler
// looks after every window (a check that waited for the device would stall the other levels' streams), the tests after every case.
static unsigned* split_overflow_word() {
    static unsigned* word = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        unsigned* w = nullptr;
        if (hipHostMalloc((void**)&w, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess && w) {
            *w = 0u;
            word = w;
        }
    });
    return word;
}

// The activations of a w_split problem split ONCE (jb_gemm_args.a_split): gemm_split_kernel converts every activation
// fragment it multiplies -- 32 elements per lane and k-step, ~ 7 vector instructions each, again in every one of the J / 64
// column blocks -- and is bound by those instructions, not by the MFMAs (23 % of the f16 peak).  With a scratch buffer of the
// activations' size the same split (input ReLU, range check, hi = half(x), lo = half((x - hi) * 2^11): bit-identical operands)
// is done by one pass over the rows, and the GEMM (gemm_split_glds_kernel below) takes finished f16 operands through LDS-DMA.
__global__ void split_act_kernel(const float* __restrict__ A, int64_t lda, int n_seq, int t_in, int64_t in_seq_stride, int K,
                                 f16* __restrict__ img, int pre_relu, unsigned* overflow) {
    const int kv = K >> 3;                                     // 8-element pieces per row
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_seq * t_in * kv) return;
    const int64_t r = idx / kv;                                // row of the images: (sequence, position) packed
    const int k = (int)(idx - r * kv) << 3;
    const int64_t n = r / t_in;
    const float* src = A + (n * in_seq_stride + (r - n * t_in)) * lda + k;      // rows between two sequences are never looked at
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
    const float relu_floor = pre_relu ? 0.f : -INFINITY, neg_limit = pre_relu ? -INFINITY : -65504.0f;
    bool too_big = false;
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = e < 4 ? x0[e] : x1[e - 4];
        too_big = too_big || !(x <= 65504.0f && x >= neg_limit);          // judged before the ReLU: fmaxf(NaN, 0) is 0
        x = fmaxf(x, relu_floor);
        const f16 hh = (f16)x;
        h[e] = hh;
        l[e] = (f16)((x - (float)hh) * 2048.0f);
    }
    // image: per row, per 32-channel k-tile, one 128-byte line [hi 32 halves | lo 32 halves]
    f16* line = img + r * (2 * (int64_t)K) + (k >> 5) * 64 + (k & 31);
    *reinterpret_cast<f16x8*>(line) = h;
    *reinterpret_cast<f16x8*>(line + 32) = l;
    if (too_big) __hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256, 2) void gemm_split_kernel(GemmParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t m_base = (int64_t)blockIdx.x * 256 + wave * 64;
    const int jt_base = blockIdx.y * 4;

    f32x4 acc[4][4], acc2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

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

    const float* A = (const float*)p.A;
    bool too_big = false;                                       // an activation outside the half range was seen (g_split_overflow)
    const int64_t lo_image = (int64_t)p.njt * p.nkt * 512;      // f16 elements from a tap's hi image to its lo image
    const float relu_floor = p.pre_relu ? 0.f : -INFINITY;      // the input ReLU as a branch-free max
    const float neg_limit = p.pre_relu ? -INFINITY : -65504.0f;
    const bool check = blockIdx.y == 0;                         // (wave-uniform) this workgroup looks for activations outside the half range
    int64_t woff[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) woff[jt] = ((int64_t)min(jt_base + jt, p.njt - 1) * p.nkt) * 512 + (int64_t)lane * 8;
    for (int tap = 0; tap < p.n_taps; ++tap) {
        const float* arow[4];
        bool aval[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int tin = t_idx[mt] * p.in_stride + p.shift[tap];
            aval[mt] = mvalid[mt] && tin >= 0 && tin < p.t_in;
            arow[mt] = A + ((int64_t)n_idx[mt] * p.in_seq_stride + (aval[mt] ? tin : 0)) * p.lda + g * 8;
        }
        if (!__any((int)(aval[0] || aval[1] || aval[2] || aval[3]))) continue;      // (see gemm_kernel)
        const f16* whi = reinterpret_cast<const f16*>((const float*)p.W + (int64_t)tap * p.tap_stride);
        const f16* wlo = whi + lo_image;
        auto issue = [&](int ks, f16x8 (&wh)[4], f16x8 (&wl)[4], f32x4 (&ar)[4][2]) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                wh[jt] = ld_frag<f16>(whi + woff[jt] + (int64_t)ks * 512);
                wl[jt] = ld_frag<f16>(wlo + woff[jt] + (int64_t)ks * 512);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                ar[mt][0] = *reinterpret_cast<const f32x4*>(arow[mt] + ks * 32);
                ar[mt][1] = *reinterpret_cast<const f32x4*>(arow[mt] + ks * 32 + 4);
            }
        };
        auto multiply = [&](const f16x8 (&wh)[4], const f16x8 (&wl)[4], const f32x4 (&ar)[4][2]) {
            f16x8 ah[4], al[4];
            if (check) {
                // judged on the value as loaded: fmaxf(NaN, 0) is 0, so a NaN under the input ReLU would pass unseen (torch's relu
                // propagates it); what the ReLU clips anyway (x < 0) cannot overflow.  Only the workgroups of the first column
                // block look (a wave-uniform branch): every activation row passes through them too, and the kernel is bound by
                // the instructions it issues per element (the check in every workgroup: conditioner 78 -> 96 ms per window)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = ar[mt][e >> 2][e & 3];
                        too_big = too_big || (aval[mt] && !(x <= 65504.0f && x >= neg_limit));
                    }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = ar[mt][e >> 2][e & 3];
                    x = aval[mt] ? fmaxf(x, relu_floor) : 0.f;
                    const f16 h = (f16)x;
                    ah[mt][e] = h;
                    al[mt][e] = (f16)((x - (float)h) * 2048.0f);
                }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[jt][mt] = jb_mfma(wh[jt], ah[mt], acc[jt][mt]);
                    acc2[jt][mt] = jb_mfma(wh[jt], al[mt], acc2[jt][mt]);
                    acc2[jt][mt] = jb_mfma(wl[jt], ah[mt], acc2[jt][mt]);
                }
        };
        for (int ks = 0; ks < p.nkt; ++ks) {
            f16x8 wh[4], wl[4];
            f32x4 ar[4][2];
            issue(ks, wh, wl, ar);
            multiply(wh, wl, ar);
        }
    }

    if (__any((int)too_big) && lane == 0) __hip_atomic_store(p.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (!mvalid[mt]) continue;
        int64_t orow = (int64_t)n_idx[mt] * p.out_seq_stride + (int64_t)t_idx[mt] * p.out_stride + p.out_offset;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            int jb = (jt_base + jt) * 16 + g * 4;
            if (jb < p.epi.J) epilogue_store<float>(p.epi, (acc[jt][mt] + acc2[jt][mt] * (1.0f / 2048.0f)) * p.w_unscale, orow, jb, -1);
        }
    }
}

// LDS-staged GEMM for "flat" problems (one tap, unit strides: output row q reads input row q) -- the prefill
// projections.  gemm_kernel above feeds every MFMA operand straight from L1: 32 FLOP per L1 byte, so the texture path
// saturates at a few percent of the matrix cores.  Here a block of 4 waves owns a 256-row x 128-column tile; per
// k-tile the block copies the 256 x KT activation panel (rows padded by one 16-byte slot: conflict-free fragment
// reads) and the 8 packed weight fragments into LDS once, and each wave multiplies its 64 rows against all 128
// columns (4 x 8 = 32 accumulator tiles, 12 fragment reads per 32 MFMAs).  Global loads for k-tile kt+1 are issued
// before the MFMAs of k-tile kt and parked in the other LDS stage afterwards: one barrier per k-tile.
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_lds_kernel(GemmParams p) {
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E, KT = Frag<T>::KT;
    constexpr int BM = 256, BJ = 8;                    // rows, 16-column tiles per block
    constexpr int XP = KT + E;                         // LDS row pitch (elements)
    constexpr int SEG = KT / E;                        // 16-byte segments per activation row per k-tile (4)
    constexpr int XL = BM * SEG / 256;                 // activation vectors per thread per k-tile (4)
    constexpr int WL = BJ * 64 / 256;                  // weight vectors per thread per k-tile (2)
    constexpr int STAGE = BM * XP + BJ * 64 * E;       // elements per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    T* s_base = reinterpret_cast<T*>(s_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int jt0 = blockIdx.y * BJ;

    const T* xsrc[XL];
    int xdst[XL];
#pragma unroll
    for (int u = 0; u < XL; ++u) {
        const int idx = tid + u * 256, row = idx / SEG, seg = idx - row * SEG;
        const int64_t q = m0 + row < p.m_total ? m0 + row : p.m_total - 1;      // clamped: rows past the end are never stored
        xsrc[u] = (const T*)p.A + q * p.lda + seg * E;
        xdst[u] = row * XP + seg * E;
    }
    const T* wsrc[WL];
    int wdst[WL];
#pragma unroll
    for (int u = 0; u < WL; ++u) {
        const int idx = tid + u * 256, jt = idx >> 6, ln = idx & 63;
        wsrc[u] = (const T*)p.W + ((int64_t)min(jt0 + jt, p.njt - 1) * p.nkt * 64 + ln) * E;
        wdst[u] = BM * XP + (jt * 64 + ln) * E;
    }
    V xr[XL], wr[WL];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int u = 0; u < XL; ++u) xr[u] = ld_frag<T>(xsrc[u] + (int64_t)kt * KT);
#pragma unroll
        for (int u = 0; u < WL; ++u) wr[u] = ld_frag<T>(wsrc[u] + (int64_t)kt * (64 * E));
    };
    auto park = [&](int stage) {
        T* sb = s_base + stage * STAGE;
#pragma unroll
        for (int u = 0; u < XL; ++u) *reinterpret_cast<V*>(sb + xdst[u]) = xr[u];
#pragma unroll
        for (int u = 0; u < WL; ++u) *reinterpret_cast<V*>(sb + wdst[u]) = wr[u];
    };

    f32x4 acc[BJ][4];
#pragma unroll
    for (int a = 0; a < BJ; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    park(0);
    __syncthreads();
    for (int kt = 0; kt < p.nkt; ++kt) {
        const bool more = kt + 1 < p.nkt;
        if (more) fetch(kt + 1);
        const T* sb = s_base + (kt & 1) * STAGE;
        V af[4], wf[BJ];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const V*>(sb + (wave * 64 + mt * 16 + c) * XP + g * E);
#pragma unroll
        for (int jt = 0; jt < BJ; ++jt) wf[jt] = *reinterpret_cast<const V*>(sb + BM * XP + (jt * 64 + lane) * E);
#pragma unroll
        for (int jt = 0; jt < BJ; ++jt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[jt][mt] = jb_mfma(wf[jt], af[mt], acc[jt][mt]);
        if (more) park((kt + 1) & 1);
        __syncthreads();
    }

#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t q = m0 + wave * 64 + mt * 16 + c;
        if (q >= p.m_total) continue;
        const int n = (int)(q / p.t_out), t = (int)(q - (int64_t)n * p.t_out);
        int64_t cache_row = -1;
        if (p.epi.qkv_split) {
            const int ct = p.cache_t0 + t;
            if (ct < p.epi.cache_cap) cache_row = (int64_t)n * p.epi.cache_cap + ct;
        }
#pragma unroll
        for (int jt = 0; jt < BJ; ++jt) {
            const int jb = (jt0 + jt) * 16 + g * 4;
            if (jb < p.epi.J) epilogue_store<T>(p.epi, acc[jt][mt], q, jb, cache_row);
        }
    }
}

// The prefill projections (flat fp16 problems: one tap, unit strides) since round 4: 128 x 128 output tile, 64 channels per
// K-step, BOTH operands brought into LDS by LDS-DMA (`global_load_lds_dwordx4`, no register staging, no ds_write pass):
// activations as 8-row x 128-byte pieces into an XOR-swizzled image (gemm_glds_index.h: what the DMA deposits lane-linearly is
// what the operand reads expect, every ds_read_b128 bank-conflict free -- checked exhaustively on the CPU by
// tests/test_gemm_glds_index.py), weights straight from the packed MFMA-order image, one 1-KiB tile per instruction.  4 waves
// as 2 x 2, each 64 rows x 64 columns; ONE 32-KiB LDS stage and three workgroups per CU: the other resident workgroups cover
// a workgroup's load phase (measured, tools/gemm_glds_probe.hip on the prefill's shapes, M = 32768: 716 / 745 / 672 TFLOP/s
// at K = 1920 and J = 1440 / 1920 / 2880 against 520 / 526 / 534 for gemm_lds_kernel, 496 against 209 at K = 480; two LDS
// stages with one barrier per K-step: 620 / 632 / 594; a 256 x 256 tile on 8 waves: 726 / 704 / 710 and 387 at K = 480 --
// profiles/r04_gemm_glds_probe.log).  XCD-aware tile order: a 128-row panel's column tiles run on one XCD, whose L2 fetches
// the panel from HBM once.  Same MFMA, same k order per output element as gemm_lds_kernel: bit-identical results.
// Rows are (sequence, position) pairs with their own pitches on both sides, so the per-sample v·Wp of the wide-value
// cache is one launch (rows of a sample contiguous in the cache, samples cache_cap rows apart).
#include "gemm_glds_index.h"
__device__ __forceinline__ void jb_glds16(const void* src, unsigned char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
__global__ __launch_bounds__(256, 3) void gemm_glds_kernel(GemmParams p, int MB, int NB) {
    using namespace gi;
    using T = f16;
    extern __shared__ __attribute__((aligned(1024))) unsigned char s_glds[];
    int mp, nt;
    if (!tile_of_block((int)blockIdx.x, MB, NB, &mp, &nt)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wave_m = wave >> 1, wave_n = wave & 1;
    const int64_t m0 = (int64_t)mp * BM;
    const int jt0 = nt * BJT;
    const T* a_src[4];
    const T* w_src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int64_t q = m0 + (wave * 4 + u) * 8 + a_src_row(lane);
        q = q < p.m_total ? q : p.m_total - 1;                    // rows past the end are never stored
        const int64_t n = q / p.t_out, t = q - n * p.t_out;
        a_src[u] = (const T*)p.A + (n * p.in_seq_stride + t) * p.lda + a_src_seg(lane) * 8;
        const int jt = (wave * 4 + u) >> 1;
        w_src[u] = (const T*)p.W + ((int64_t)min(jt0 + jt, p.njt - 1) * p.nkt) * 512 + lane * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksteps = (p.nkt + 1) >> 1;
    for (int s = 0; s < ksteps; ++s) {
        const bool tail = 2 * s + 1 >= p.nkt;                    // odd number of k-tiles: the step's second half does not exist
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const T* src = a_src[u] + (int64_t)s * KSTEP;
            if (tail && a_src_seg(lane) >= 4) src -= 32;          // folded onto the first half (those slots are never read)
            jb_glds16(src, s_glds + (wave * 4 + u) * 1024);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = u & 1, kt = (tail && ks) ? 2 * s : 2 * s + ks;
            jb_glds16(w_src[u] + (int64_t)kt * 512, s_glds + A_BYTES + (wave * 4 + u) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's share of the step has landed ...
        __syncthreads();                                            // ... everybody's has
        const int n_ks = tail ? 1 : 2;
        for (int ks = 0; ks < n_ks; ++ks) {
            f16x8 af[4], wf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const f16x8*>(s_glds + a_byte(frag_row(wave_m, mt, lane), frag_seg(ks, lane)));
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const f16x8*>(s_glds + w_byte(wave_n * 4 + j, ks, lane));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[j][mt] = jb_mfma(wf[j], af[mt], acc[j][mt]);
        }
        __syncthreads();                                            // nobody still reads the stage the next step overwrites
    }
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t q = m0 + wave_m * 64 + mt * 16 + c;
        if (q >= p.m_total) continue;
        const int n = (int)(q / p.t_out), t = (int)(q - (int64_t)n * p.t_out);
        const int64_t orow = (int64_t)n * p.out_seq_stride + t;
        int64_t cache_row = -1;
        if (p.epi.qkv_split) {
            const int ct = p.cache_t0 + t;
            if (ct < p.epi.cache_cap) cache_row = (int64_t)n * p.epi.cache_cap + ct;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jb = (jt0 + wave_n * 4 + j) * 16 + g * 4;
            if (jb < p.epi.J) epilogue_store<T>(p.epi, acc[j][mt], orow, jb, cache_row);
        }
    }
}

// The prefill's big projections since round 5 (K >= 128, enough 256 x 256 tiles to fill the chip): the 8-phase K-loop of
// gemm_8phase.h -- one 8-wave workgroup per CU, 128 KiB of LDS in two stages of four half-tiles, the two wave rows half a phase
// apart so that one group's operand reads and LDS-DMA requests run under the other's MFMA cluster.  Tile order, row addressing
// (rows are (sequence, position) pairs with a pitch per sequence) and epilogue are gemm_glds_kernel's; same MFMA and k order per
// output element: bit-identical to the three older kernels (tests/test_hip_kernels.py::test_gemm_8phase).
#include "gemm_8phase.h"
// The conv stacks' GEMM on pre-split activations (jb_gemm_args.a_split; round 5).  split_act_kernel has written, per input row
// and 32-channel k-tile, one 128-byte line [hi 32 halves | lo 32 halves]; the weights' hi and lo images are 1-KiB MFMA-order
// tiles as ever.  That is gemm_glds_kernel's stage exactly -- a 128-row activation panel of 128-byte lines, 16 weight tiles --
// with "k-tile 0 / 1 of the K-step" read as "hi / lo of ONE k-tile": same LDS-DMA pieces, same XOR-swizzled image, same
// conflict-free operand reads (gemm_glds_index.h), and per stage 48 MFMAs per wave instead of 32:
//     acc += w_hi a_hi;   acc2 += w_hi a_lo;   acc2 += w_lo a_hi          (gemm_split_kernel's three, in its order)
// Taps: a stage belongs to one (tap, k-tile); the lanes' source rows are the tap's shifted rows, rows outside their sequence
// read a line of zeros; a tap that is out of range for all 128 rows of the tile is skipped (workgroup-uniform vote).
// Operands, MFMAs and their order per output element are gemm_split_kernel's: bit-identical (tests/test_hip_kernels.py).
__global__ __launch_bounds__(256, 2) void gemm_split_glds_kernel(GemmParams p, int MB, int NB) {
    using namespace gi;
    extern __shared__ __attribute__((aligned(1024))) unsigned char s_glds[];
    int mp, nt;
    if (!tile_of_block((int)blockIdx.x, MB, NB, &mp, &nt)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wave_m = wave >> 1, wave_n = wave & 1;
    const int64_t m0 = (int64_t)mp * BM;
    const int jt0 = nt * BJT;
    // the four activation pieces of this wave: (sequence, position) of the lane's row in each
    int pn[4], pt[4];
    bool pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t q = m0 + (wave * 4 + u) * 8 + a_src_row(lane);
        pv[u] = q < p.m_total;
        const int64_t qq = pv[u] ? q : 0;
        pn[u] = (int)(qq / p.t_out);
        pt[u] = (int)(qq - (int64_t)pn[u] * p.t_out);
    }
    const f16* img = (const f16*)p.A;                           // [n_seq * t_in][nkt][hi 32 | lo 32]
    const int64_t row_pitch = 2 * (int64_t)p.K;
    const int64_t lo_image = (int64_t)p.njt * p.nkt * 512;      // f16 elements from a tap's hi image to its lo image
    int64_t w_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ti = wave * 4 + u, jt = ti >> 1;              // stage tile (jt, ks): ks = 0 the hi image's tile, ks = 1 the lo image's
        w_off[u] = (ti & 1 ? lo_image : 0) + ((int64_t)min(jt0 + jt, p.njt - 1) * p.nkt) * 512 + lane * 8;
    }
    f32x4 acc[4][4], acc2[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[j][mt] = acc2[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < p.n_taps; ++tap) {
        const f16* a_src[4];
        int a_step[4];                                         // halves per k-tile: a line of the image, or the one line of zeros again
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tin = pt[u] * p.in_stride + p.shift[tap];
            const bool ok = pv[u] && tin >= 0 && tin < p.t_in;
            any = any || ok;
            a_src[u] = (ok ? img + ((int64_t)pn[u] * p.t_in + tin) * row_pitch : (const f16*)p.zero_line) + a_src_seg(lane) * 8;
            a_step[u] = ok ? 64 : 0;
        }
        if (!__syncthreads_or((int)any)) continue;             // (see gemm_kernel: dilations beyond the sequence)
        const f16* wtap = reinterpret_cast<const f16*>((const float*)p.W + (int64_t)tap * p.tap_stride);
        for (int kt = 0; kt < p.nkt; ++kt) {
#pragma unroll
            for (int u = 0; u < 4; ++u) jb_glds16(a_src[u] + (int64_t)kt * a_step[u], s_glds + (wave * 4 + u) * 1024);
#pragma unroll
            for (int u = 0; u < 4; ++u) jb_glds16(wtap + w_off[u] + (int64_t)kt * 512, s_glds + A_BYTES + (wave * 4 + u) * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's share of the stage has landed ...
            __syncthreads();                                            // ... everybody's has
            f16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                ah[mt] = *reinterpret_cast<const f16x8*>(s_glds + a_byte(frag_row(wave_m, mt, lane), frag_seg(0, lane)));
                al[mt] = *reinterpret_cast<const f16x8*>(s_glds + a_byte(frag_row(wave_m, mt, lane), frag_seg(1, lane)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(s_glds + w_byte(wave_n * 4 + j, 0, lane));
                wl[j] = *reinterpret_cast<const f16x8*>(s_glds + w_byte(wave_n * 4 + j, 1, lane));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[j][mt] = jb_mfma(wh[j], ah[mt], acc[j][mt]);
                    acc2[j][mt] = jb_mfma(wh[j], al[mt], acc2[j][mt]);
                    acc2[j][mt] = jb_mfma(wl[j], ah[mt], acc2[j][mt]);
                }
            __syncthreads();                                            // nobody still reads the stage the next step overwrites
        }
    }
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t q = m0 + wave_m * 64 + mt * 16 + c;
        if (q >= p.m_total) continue;
        const int n = (int)(q / p.t_out), t = (int)(q - (int64_t)n * p.t_out);
        const int64_t orow = (int64_t)n * p.out_seq_stride + (int64_t)t * p.out_stride + p.out_offset;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jb = (jt0 + wave_n * 4 + j) * 16 + g * 4;
            if (jb < p.epi.J) epilogue_store<float>(p.epi, (acc[j][mt] + acc2[j][mt] * (1.0f / 2048.0f)) * p.w_unscale, orow, jb, -1);
        }
    }
}

// Epilogue of gemm_8phase_kernel.  With ONE workgroup per CU nothing covers a tile's epilogue, and the MFMA result layout
// (a lane holds 4 consecutive columns of one row: 8 bytes, 16 rows per store instruction; 2-byte stores for a q / k / v
// split) is store-issue-bound.  So each wave first applies bias -> round -> activation in registers, parks its 128 x 64
// sub-tile as f16 in an LDS region of its own (rows 144 bytes apart: conflict-free 8-byte writes and 16-byte reads), and
// then moves whole 128-byte row segments: 16 bytes per lane, the residual read the same way and added in that form, the
// q / k-cache / v-cache destination chosen per 8-column piece.  Same operations in the same order per element as
// epilogue_store: bit-identical.  `fast` (host-checked: 8-column pieces never straddle a destination or an edge, every row
// 16-byte aligned) selects it; otherwise epilogue_store.
constexpr int G8_EPI_PITCH = 144;
constexpr int G8_EPI_WAVE_BYTES = 128 * G8_EPI_PITCH;
constexpr int G8_LDS_BYTES = 8 * G8_EPI_WAVE_BYTES > g8::LDS_BYTES ? 8 * G8_EPI_WAVE_BYTES : g8::LDS_BYTES;

__global__ __launch_bounds__(512, 1) void gemm_8phase_kernel(GemmParams p, int MB, int NB, int fast_epi) {
    using T = f16;
    extern __shared__ __attribute__((aligned(1024))) unsigned char s_glds[];
    int mp, nt;
    if (!gi::tile_of_block((int)blockIdx.x, MB, NB, &mp, &nt)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wave_m = wave >> 2, wave_n = wave & 3;
    const int64_t m0 = (int64_t)mp * g8::BM;
    const int jt0 = nt * g8::BJT;
    g8::Sources src;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int64_t q = m0 + g8::a_tile_row(h, (wave * 2 + u) * 8 + gi::a_src_row(lane));
            q = q < p.m_total ? q : p.m_total - 1;                    // rows past the end are never stored
            const int64_t n = q / p.t_out, t = q - n * p.t_out;
            src.a[h][u] = (const T*)p.A + (n * p.in_seq_stride + t) * p.lda + gi::a_src_seg(lane) * 8;
            const int ti = wave * 2 + u;
            src.w[h][u] = (const T*)p.W + ((int64_t)min(jt0 + g8::b_tile_jt(h, ti), p.njt - 1) * p.nkt + g8::b_tile_ks(ti)) * 512 + lane * 8;
        }
    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n_pairs = p.nkt >> 2;                     // loop iterations: two K-tiles = four 32-channel k-tiles each
    g8::mainloop(src, n_pairs * 2, s_glds, acc);
    const int g = lane >> 4, c = lane & 15;
    if (n_pairs * 4 < p.nkt) {
        // K is not a multiple of 128 (the upsamplers' K = 480 projections: 3 iterations + 3 k-tiles; 5b_lyrics' K = 4800: 37 + 2):
        // the last one to three k-tiles straight from L1 into the operand registers, as gemm_kernel does -- k ascending as ever
        const T* a_row[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            int64_t q = m0 + wave_m * 128 + mt * 16 + c;
            q = q < p.m_total ? q : p.m_total - 1;
            const int64_t n = q / p.t_out, t = q - n * p.t_out;
            a_row[mt] = (const T*)p.A + (n * p.in_seq_stride + t) * p.lda + g * 8;
        }
        const T* w_tile[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w_tile[j] = (const T*)p.W + ((int64_t)min(jt0 + wave_n * 4 + j, p.njt - 1) * p.nkt) * 512 + lane * 8;
        for (int kt = n_pairs * 4; kt < p.nkt; ++kt) {
            f16x8 wf[4], af[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = ld_frag<T>(w_tile[j] + (int64_t)kt * 512);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) af[mt] = ld_frag<T>(a_row[mt] + kt * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) acc[j][mt] = jb_mfma(wf[j], af[mt], acc[j][mt]);
        }
    }
    if (!fast_epi) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int64_t q = m0 + wave_m * 128 + mt * 16 + c;
            if (q >= p.m_total) continue;
            const int n = (int)(q / p.t_out), t = (int)(q - (int64_t)n * p.t_out);
            const int64_t orow = (int64_t)n * p.out_seq_stride + t;
            int64_t cache_row = -1;
            if (p.epi.qkv_split) {
                const int ct = p.cache_t0 + t;
                if (ct < p.epi.cache_cap) cache_row = (int64_t)n * p.epi.cache_cap + ct;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int jb = (jt0 + wave_n * 4 + j) * 16 + g * 4;
                if (jb < p.epi.J) epilogue_store<T>(p.epi, acc[j][mt], orow, jb, cache_row);
            }
        }
        return;
    }
    __syncthreads();                                   // every wave is done with the operand stages: the regions overlap them
    unsigned char* region = s_glds + wave * G8_EPI_WAVE_BYTES;
    const int col0 = (jt0 + wave_n * 4) * 16;          // first column of the wave's sub-tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.epi.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = jb_round<T>(p.epi.bias[min(col0 + j * 16 + g * 4 + r, p.epi.J - 1)]);
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[j][mt][r];
                if (p.epi.bias) x += bias[r];
                x = jb_round<T>(x);
                x = jb_apply_act<T>(x, p.epi.act);
                o[r] = (f16)x;
            }
            *reinterpret_cast<f16x4*>(region + (mt * 16 + c) * G8_EPI_PITCH + (j * 16 + g * 4) * 2) = o;
        }
    }
    // (a wave's LDS operations execute in order: its reads below see its own writes)
    const int seg = lane & 7, col = col0 + seg * 8;
    if (col >= p.epi.J) return;
    T* dst_base = (T*)p.epi.out;
    int64_t dst_ld = p.epi.ldo;
    int dst_col = col;
    bool to_cache = false;
    if (p.epi.qkv_split && col >= p.epi.S) {
        to_cache = true;
        if (col < 2 * p.epi.S) { dst_base = (T*)p.epi.kcache; dst_ld = p.epi.S; dst_col = col - p.epi.S; }
        else if (col < 2 * p.epi.S + p.epi.v_cols) { dst_base = (T*)p.epi.vcache; dst_ld = p.epi.S; dst_col = col - 2 * p.epi.S; }
        else { dst_base = (T*)p.epi.vcache2; dst_ld = p.epi.v2w; dst_col = col - 2 * p.epi.S - p.epi.v_cols; }
    }
    const int64_t q0 = m0 + wave_m * 128 + (lane >> 3);
    int n = (int)(q0 / p.t_out), t = (int)(q0 - (int64_t)n * p.t_out) - 8;      // (sequence, position) of the lane's rows: 8 apart
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int row = i * 8 + (lane >> 3);
        t += 8;
        while (t >= p.t_out) { t -= p.t_out; ++n; }
        if (q0 + i * 8 >= p.m_total) break;
        int64_t orow = (int64_t)n * p.out_seq_stride + t;
        f16x8 v = *reinterpret_cast<const f16x8*>(region + row * G8_EPI_PITCH + seg * 16);
        if (p.epi.res) {
            const f16x8 rr = *reinterpret_cast<const f16x8*>((const T*)p.epi.res + orow * p.epi.ldr + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (float)v[e], rv = (float)rr[e];
                v[e] = (f16)((p.epi.res_scale == 1.0f) ? jb_round<T>(rv + x) : jb_round<T>(rv + jb_round<T>(p.epi.res_scale * x)));
            }
        }
        if (to_cache) {
            const int ct = p.cache_t0 + t;
            if (ct >= p.epi.cache_cap) continue;
            orow = (int64_t)n * p.epi.cache_cap + ct;
        }
        *reinterpret_cast<f16x8*>(dst_base + orow * dst_ld + dst_col) = v;
    }
}

// problems of gemm_glds_kernel's kind with K a multiple of 128 and at least this many 256 x 256 output tiles take
// gemm_8phase_kernel (< 0: never); jb_tune_gemm_8phase
static int g_gemm_8phase_min_tiles = 512;
extern "C" void jb_tune_gemm_8phase(int min_tiles) { g_gemm_8phase_min_tiles = min_tiles; }

// the flat fp16 problems of >= this many rows take gemm_glds_kernel (< 0: never); jb_tune_gemm_glds
static int g_gemm_glds_min_rows = 256;
extern "C" void jb_tune_gemm_glds(int min_rows) { g_gemm_glds_min_rows = min_rows; }

// rows from which jb_gemm takes the LDS-staged kernel for flat problems (< 0: never); jb_tune_gemm_lds
static int g_gemm_lds_min_rows = 1024;
extern "C" void jb_tune_gemm_lds(int min_rows) { g_gemm_lds_min_rows = min_rows; }

static inline bool aligned_to(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// w_split problems that come with a scratch buffer (jb_gemm_args.a_split) split their activations once, in a pass of their own
// (0: never -- the in-kernel split); jb_tune_gemm_presplit
static int g_gemm_presplit = 1;
extern "C" void jb_tune_gemm_presplit(int on) { g_gemm_presplit = on; }

extern "C" int jb_gemm_split_overflow(int reset) {
    unsigned* w = split_overflow_word();
    if (!w) {
        jb_set_error("jb_gemm_split_overflow: no host-coherent memory for the flag");
        return JB_ERR_HIP;
    }
    const unsigned v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
    if (reset && v) __atomic_store_n(w, 0u, __ATOMIC_RELEASE);
    return v ? 1 : 0;
}

extern "C" int jb_gemm(const jb_gemm_args* a, void* stream) {
    JB_REQUIRE(a && a->A && a->W && a->out, "null pointer");
    JB_REQUIRE(a->dtype == JB_F32 || a->dtype == JB_F16, "bad dtype");
    JB_REQUIRE(a->K > 0 && a->J > 0 && a->n_seq > 0 && a->t_out > 0 && a->t_in > 0, "empty problem");
    JB_REQUIRE(a->n_taps >= 1 && a->n_taps <= 4, "n_taps must be 1..4");
    JB_REQUIRE(!a->qkv_split || (a->S > 0 && a->J == 3 * a->S && a->kcache && a->vcache), "bad qkv split");
    JB_REQUIRE(!a->w_split || (a->dtype == JB_F32 && !a->qkv_split && a->K % 32 == 0 && a->lda % 4 == 0 && aligned_to(a->A, 16)),
               "w_split takes fp32 problems with K a multiple of 32 and 16-byte aligned rows, no q/k/v split");
    const int esz = a->dtype == JB_F16 ? 2 : 4;
    const int E = a->dtype == JB_F16 ? 8 : 4;
    GemmParams p;
    p.A = a->A; p.lda = a->lda; p.W = a->W; p.tap_stride = a->tap_stride;
    p.n_seq = a->n_seq; p.t_in = a->t_in; p.t_out = a->t_out;
    p.in_seq_stride = a->in_seq_stride; p.out_seq_stride = a->out_seq_stride;
    p.K = a->K;
    int Edummy;
    packed_dims(a->K, a->J, a->w_split ? JB_F16_SPLIT : a->dtype, &p.nkt, &p.njt, &Edummy);
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
    p.epi.vcache2 = nullptr; p.epi.v2w = 0; p.epi.v_cols = a->S;
    p.epi.vec_out = !a->qkv_split && (a->ldo % 4 == 0) && aligned_to(a->out, 4 * esz);
    p.epi.out2 = nullptr; p.epi.ldo2 = 0; p.epi.add2 = nullptr; p.epi.add2_n = p.epi.add2_t = 0;
    dim3 grid((unsigned)((p.m_total + 255) / 256), (unsigned)((p.njt + 3) / 4));
    const int KT = a->dtype == JB_F16 ? 32 : 16;
    const bool fast = p.vec_a && (a->K % KT == 0);
    hipStream_t st = (hipStream_t)stream;
    p.overflow = nullptr;
    p.w_unscale = a->w_split_unscale != 0.f ? a->w_split_unscale : 1.0f;
    p.zero_line = nullptr;
    if (a->w_split) {
        p.overflow = split_overflow_word();
        JB_REQUIRE(p.overflow, "no host-coherent memory for the f16-split overflow flag");
        const int64_t img = (int64_t)a->n_seq * a->t_in * a->K;              // elements; the image holds 2 halves for each
        if (a->a_split && a->a_split_bytes >= img * 4 && aligned_to(a->a_split, 128) && g_gemm_presplit) {
            static const void* zero_lines[64] = {};
            static std::mutex zero_mutex;                       // (the levels' host threads may meet here on their first conv stack)
            int dev = 0;
            JB_HIP(hipGetDevice(&dev));
            JB_REQUIRE(dev >= 0 && dev < 64, "device index");
            {
                std::lock_guard<std::mutex> lock(zero_mutex);
                if (!zero_lines[dev]) {
                    void* z = nullptr;
                    JB_HIP(hipMalloc(&z, 128));
                    JB_HIP(hipMemset(z, 0, 128));
                    JB_HIP(hipDeviceSynchronize());
                    zero_lines[dev] = z;
                }
            }
            f16* image = (f16*)a->a_split;
            const int64_t pieces = img / 8;
            split_act_kernel<<<(unsigned)((pieces + 255) / 256), 256, 0, st>>>((const float*)a->A, a->lda, a->n_seq, a->t_in, a->in_seq_stride, a->K,
                                                                               image, a->pre_relu, p.overflow);
            JB_CHECK_LAUNCH();
            p.A = image;
            p.zero_line = zero_lines[dev];
            const int MB = (int)((p.m_total + gi::BM - 1) / gi::BM), NB = (p.njt + gi::BJT - 1) / gi::BJT;
            gemm_split_glds_kernel<<<(MB + 7) / 8 * 8 * NB, 256, gi::STAGE_BYTES, st>>>(p, MB, NB);
        } else {
            gemm_split_kernel<<<grid, 256, 0, st>>>(p);
        }
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
    const bool flat = a->n_taps == 1 && a->shift[0] == 0 && a->in_stride == 1 && a->out_stride == 1 && a->out_offset == 0 &&
                      a->t_in == a->t_out && a->in_seq_stride == a->t_in && a->out_seq_stride == a->t_out && !a->pre_relu;
    // one tap at unit strides, rows = (sequence, position) with a pitch per sequence on either side
    const bool seq_flat = a->n_taps == 1 && a->shift[0] == 0 && a->in_stride == 1 && a->out_stride == 1 && a->out_offset == 0 &&
                          a->t_in == a->t_out && !a->pre_relu;
    if (seq_flat && fast && a->dtype == JB_F16 && g_gemm_glds_min_rows >= 0 && p.m_total >= g_gemm_glds_min_rows &&
        (a->in_seq_stride * a->lda) % 8 == 0) {
        static bool configured[64] = {};
        int dev = 0;
        JB_HIP(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !configured[dev]) {      // a static request below 64 KiB needs no opt-in; kept for symmetry
            JB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       gi::STAGE_BYTES));
            configured[dev] = true;
        }
        const int MB8 = (int)((p.m_total + g8::BM - 1) / g8::BM), NB8 = (p.njt + g8::BJT - 1) / g8::BJT;
        if (a->K >= 128 && g_gemm_8phase_min_tiles >= 0 && (int64_t)MB8 * NB8 >= g_gemm_8phase_min_tiles) {
            static bool configured8[64] = {};
            if (dev >= 0 && dev < 64 && !configured8[dev]) {     // 144 KiB of dynamic LDS: above the 64-KiB default
                JB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_8phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           G8_LDS_BYTES));
                configured8[dev] = true;
            }
            // the line-wise epilogue: 8-column pieces (16 bytes) must neither straddle an edge nor a q / k / v boundary
            const EpiParams& e = p.epi;
            bool fast_epi = !e.out2 && e.J % 8 == 0 && aligned_to(e.out, 16) && e.ldo % 8 == 0 &&
                            (!e.res || (aligned_to(e.res, 16) && e.ldr % 8 == 0));
            if (e.qkv_split)
                fast_epi = fast_epi && e.S % 8 == 0 && e.v_cols % 8 == 0 && aligned_to(e.kcache, 16) &&
                           (e.v_cols == 0 || aligned_to(e.vcache, 16)) && (e.v2w == 0 || (e.v2w % 8 == 0 && aligned_to(e.vcache2, 16)));
            gemm_8phase_kernel<<<(MB8 + 7) / 8 * 8 * NB8, 512, G8_LDS_BYTES, st>>>(p, MB8, NB8, fast_epi ? 1 : 0);
            JB_CHECK_LAUNCH();
            return JB_OK;
        }
        const int MB = (int)((p.m_total + gi::BM - 1) / gi::BM), NB = (p.njt + gi::BJT - 1) / gi::BJT;
        gemm_glds_kernel<<<(MB + 7) / 8 * 8 * NB, 256, gi::STAGE_BYTES, st>>>(p, MB, NB);
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
    if (flat && fast && g_gemm_lds_min_rows >= 0 && p.m_total >= g_gemm_lds_min_rows) {
        dim3 lgrid((unsigned)((p.m_total + 255) / 256), (unsigned)((p.njt + 7) / 8));
        const size_t lds = 2 * (size_t)(256 * (KT + E) + 8 * 64 * E) * esz;
        if (a->dtype == JB_F16) gemm_lds_kernel<f16><<<lgrid, 256, lds, st>>>(p);
        else gemm_lds_kernel<float><<<lgrid, 256, lds, st>>>(p);
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
    if (a->dtype == JB_F16) {
        if (fast) gemm_kernel<f16, true><<<grid, 256, 0, st>>>(p); else gemm_kernel<f16, false><<<grid, 256, 0, st>>>(p);
    } else {
        if (fast) gemm_kernel<float, true><<<grid, 256, 0, st>>>(p); else gemm_kernel<float, false><<<grid, 256, 0, st>>>(p);
    }
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------
struct GemvParams {
    const void* x; int64_t ldx; int n_rows;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const float* ln_c1;                         // folded LayerNorm: column sums of the stored (gamma-scaled) weights
    const void* W; int K, nkt;
    int vec_x, lds_pitch, fast;
    long long* dbg;
    const int* t_dev;
    // key-split attention output as the operand (gemv_merge_kernel): x[n][k] = sum_s w_s(n, head(k)) * parts[n][s][k]
    const void* x_parts; const float* x_ml; int n_parts, n_head, d_head;
    JbPipe pipe;                                // software-pipelined launch (common.h); slot < 0: plain
    EpiParams epi;
};

// Decode-step GEMM: n_rows <= 16*MT activation rows against a K x J weight matrix that is read from
// HBM exactly once per launch.  One workgroup per 16-column tile; its NW waves split the k-tiles, each
// streaming 1 KiB weight fragments straight into MFMA operands (non-temporal: a weight byte is used
// once per step) in batches of WB loads issued back to back.  The kernel is latency-bound -- a whole
// projection is about one bandwidth-delay product and L2 is cold after every kernel boundary -- so it
// is organised as ONE dependent memory round trip: weights, activation rows, residual and bias are
// all requested up front; with LayerNorm (LNS) the rows are staged once into LDS, normalised there
// (fp32 statistics, two passes over LDS) and the B operands are read back with ds_read_b128.
// The NW partial 16x16 tiles are summed in a fixed order through LDS (deterministic; no atomics).
template <typename T>
__device__ __forceinline__ float frag_sum(typename Frag<T>::vec v) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < Frag<T>::E; ++e) s += (float)v[e];
    return s;
}

#ifdef JB_TIMING
#define JB_STAMP(i) do { if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) { p.dbg[2 * (i)] = clock64(); p.dbg[2 * (i) + 1] = wall_clock64(); } } while (0)
#else
#define JB_STAMP(i) do { } while (0)
#endif

// Epilogue operands of one thread's output elements, requested UNCONDITIONALLY and early.  A load behind `ptr ? ptr[i] : 0`
// compiles to a branch whose body ends in s_waitcnt vmcnt(0): every such operand becomes its own dependent memory round
// trip in front of the MFMAs (and the compiler sinks an unguarded late-use load behind them: the ISA of round 1's kernels
// had bias -> wait -> residual -> wait -> activations -> wait, three cold round trips where one suffices).  Absent
// operands are read through a valid dummy address (the weight image) and discarded by a select; jb_issue_fence() keeps the
// compiler from moving the requests below it (nothing waits there).
__device__ __forceinline__ void jb_issue_fence() { asm volatile("" ::: "memory"); }

// Which element of the MT 16x16 output tiles flat index i stands for: fragment lane l, register r of tile mt.  The plain
// kernels put consecutive threads on consecutive lanes (r = wave); ALT (pipelined launches, MT = 1) makes thread i the
// element (row i >> 4, column i & 15) of the tile -- 16 consecutive threads (one DPP row) hold one output row, so eight
// neighbours' columns can be gathered in the vector ALU and leave as ONE 16-byte write-through store instead of eight 2-byte
// ones (pipe_store8); the partial tiles are parked in LDS in that order (jb_pipe_tile_slot), so the sum over the waves reads
// word w * 256 + i: conflict free.
template <bool ALT> __device__ __forceinline__ void epi_coords(int i, int& mt, int& r, int& l) {
    mt = i >> 8;
    if (ALT) { r = i & 3; l = (((i & 15) >> 2) << 4) | ((i >> 4) & 15); } else { r = (i >> 6) & 3; l = i & 63; }
}

template <typename T, int EPT, int NT, bool ALT = false>
struct EpiOperands {
    float bias[EPT], res[EPT], c1[EPT];
    __device__ __forceinline__ void request(const GemvParams& p, int jt, int MT) {
        const float* bias_p = p.epi.bias ? p.epi.bias : reinterpret_cast<const float*>(p.W);
        const float* c1_p = p.ln_c1 ? p.ln_c1 : reinterpret_cast<const float*>(p.W);
        const T* res_p = p.epi.res ? (const T*)p.epi.res : reinterpret_cast<const T*>(p.W);
        const int64_t ldr = p.epi.res ? p.epi.ldr : 0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int i = threadIdx.x + u * NT;
            int emt, er, el;
            epi_coords<ALT>(i, emt, er, el);
            const int row = emt * 16 + (el & 15), j = blockIdx.x * 16 + (el >> 4) * 4 + er;
            const int jc = min(j, p.epi.J - 1), rc = min(row, p.n_rows - 1);
            bias[u] = bias_p[jc];
            c1[u] = c1_p[jc];
            res[u] = (float)res_p[(ALT && (p.pipe.frag & JB_FRAG_RES) && p.epi.res) ? (int64_t)jb_frag_el(rc, jc) : (int64_t)rc * ldr + jc];
        }
        (void)jt; (void)MT;
    }
    // the residual of this thread's element again, write-through (EPT = 1: the pipelined forms)
    __device__ __forceinline__ void request_res_sc1(const GemvParams& p) {
        static_assert(EPT == 1 || !ALT, "pipelined forms hold one element per thread");
        const T* res_p = p.epi.res ? (const T*)p.epi.res : reinterpret_cast<const T*>(p.W);
        const int64_t ldr = p.epi.res ? p.epi.ldr : 0;
        int emt, er, el;
        epi_coords<ALT>(threadIdx.x, emt, er, el);
        const int row = emt * 16 + (el & 15), j = blockIdx.x * 16 + (el >> 4) * 4 + er;
        const int jc = min(j, p.epi.J - 1), rc = min(row, p.n_rows - 1);
        res[0] = (float)jb_ld_sc1(res_p, ((p.pipe.frag & JB_FRAG_RES) && p.epi.res) ? (int64_t)jb_frag_el(rc, jc) : (int64_t)rc * ldr + jc);
    }
    __device__ __forceinline__ void finish(const GemvParams& p) {      // after the sums are ready: drop what was not there
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            bias[u] = p.epi.bias ? bias[u] : 0.f;
            c1[u] = p.ln_c1 ? c1[u] : 0.f;
            res[u] = p.epi.res ? res[u] : 0.f;
        }
    }
};

template <typename T, int MT, int NW, bool LNS, bool FAST, int NV, int PIPE = 0>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(GemvParams p) {
    static_assert(!PIPE || (!LNS && FAST), "pipelined launches: plain fast path only");
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E, KT = Frag<T>::KT;
    constexpr int WB = NW == 16 ? 10 : (MT <= 2 ? 8 : 4);   // weight (and activation) fragments in flight per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    f32x4* s_acc = reinterpret_cast<f32x4*>(s_dyn);                 // [NW][MT][64]
    T* s_x = reinterpret_cast<T*>(s_dyn + NW * MT * 64 * sizeof(f32x4));   // [16*MT][pitch] (LNS only)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform -> scalar branches
    const int g = lane >> 4, c = lane & 15;
    const int jt = blockIdx.x;
    const T* x = (const T*)p.x;
    const int pitch = p.lds_pitch;        // elements, multiple of E

    const int kt0 = (wave * p.nkt) / NW, kt1 = ((wave + 1) * p.nkt) / NW;
    const T* wbase = (const T*)p.W + ((int64_t)jt * p.nkt) * (64 * E) + (int64_t)lane * E;

    JB_STAMP(0);
    unsigned pipe_own = 0;
    if constexpr (PIPE == 1) pipe_own = jb_pipe_own(p.pipe);
    // ---- everything this workgroup needs from memory is requested here ----
    V wf[WB];
#pragma unroll
    for (int i = 0; i < WB; ++i) {
        if (FAST) {   // clamped, unconditional: tiles past kt1 are loaded but never multiplied
            wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)min(kt0 + i, p.nkt - 1) * (64 * E)));
        } else {
            wf[i] = jb_zero_frag<T>();
            if (kt0 + i < kt1) wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)(kt0 + i) * (64 * E)));
        }
    }
    // Epilogue operands of THIS thread's output elements (the MT 16x16 tiles are spread over all threads: flat index
    // i = (mt, r, l) is element r of fragment lane l of tile mt) and, on the plain fast path, the activation fragments of
    // the first (usually only) batch: everything is in flight before anything is waited for.
    constexpr int EPT = (MT * 256 + NW * 64 - 1) / (NW * 64);      // elements per thread
    EpiOperands<T, EPT, NW * 64, (PIPE != 0)> eo;
    eo.request(p, jt, MT);
    V xf0[(!LNS && FAST) ? WB : 1][MT];
    if constexpr (!LNS && FAST && !PIPE) {
#pragma unroll
        for (int i = 0; i < WB; ++i) {
            const int k0 = min(kt0 + i, p.nkt - 1) * KT + g * E;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf0[i][mt] = ld_frag<T>(x + (int64_t)min(mt * 16 + c, p.n_rows - 1) * p.ldx + k0);
        }
    }
    jb_issue_fence();
    if constexpr (PIPE == 1) {
        jb_pipe_wait(p.pipe, pipe_own);     // the weights and the constant epilogue operands are in flight
        // the residual rows are two launches old: behind the wait and write-through (in the three-stream form of the step,
        // engine.hip, their producer -- the attention -- is not this stream's predecessor)
        eo.request_res_sc1(p);
    }
    int t = 0;
    if (p.epi.qkv_split || p.epi.add2) t = PIPE ? (int)jb_ld_word(reinterpret_cast<const unsigned*>(p.t_dev)) : *p.t_dev;
    float e_add2[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = threadIdx.x + u * NW * 64;
        int emt, er, el;
        epi_coords<(PIPE != 0)>(i, emt, er, el);
        const int row = emt * 16 + (el & 15), j = jt * 16 + (el >> 4) * 4 + er;
        const int jc = min(j, p.epi.J - 1), rc = min(row, p.n_rows - 1);
        e_add2[u] = p.epi.add2 ? p.epi.add2[(int64_t)rc * p.epi.add2_n + (int64_t)t * p.epi.add2_t + jc] : 0.f;
    }

    if constexpr (LNS && FAST && NV > 0) {
        // Register path: a wave owns whole rows (r0 = wave, r1 = wave + NW, next round +2NW ...).  Lane l holds
        // vectors l, l+64, ... of each row (NV per row), so the fp32 mean / variance come from registers with two
        // wave reductions, gamma/beta are read as vectors from L2, and the normalised row is written to LDS once.
        constexpr int RW = NV <= 4 ? 2 : 1;                 // rows in flight per wave
        const int nvec = p.K / E;                           // whole vectors per row (K % KT == 0)
        for (int rb = wave; rb < p.n_rows; rb += RW * NW) {
            V xv[RW][NV];
            float gm[NV][E], bt[NV][E];
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const int r = min(rb + j * NW, p.n_rows - 1);
#pragma unroll
                for (int i = 0; i < NV; ++i) xv[j][i] = ld_frag<T>(x + (int64_t)r * p.ldx + (int64_t)min(lane + 64 * i, nvec - 1) * E);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k0 = min(lane + 64 * i, nvec - 1) * E;
#pragma unroll
                for (int e4 = 0; e4 < E; e4 += 4) {
                    f32x4 g4 = *reinterpret_cast<const f32x4*>(p.ln_gamma + k0 + e4);
                    f32x4 b4 = *reinterpret_cast<const f32x4*>(p.ln_beta + k0 + e4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { gm[i][e4 + u] = g4[u]; bt[i][e4 + u] = b4[u]; }
                }
            }
            float sm[RW], sq[RW], mean[RW], rstd[RW];
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                sm[j] = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) sm[j] += (lane + 64 * i < nvec) ? frag_sum<T>(xv[j][i]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                mean[j] = jb_wave_sum(sm[j]) / (float)p.K;
                sq[j] = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    float q = 0.f;
#pragma unroll
                    for (int e = 0; e < E; ++e) { float d = (float)xv[j][i][e] - mean[j]; q += d * d; }
                    sq[j] += (lane + 64 * i < nvec) ? q : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                rstd[j] = 1.0f / sqrtf(jb_wave_sum(sq[j]) / (float)p.K + p.ln_eps);
                const int r = rb + j * NW;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    V y;
#pragma unroll
                    for (int e = 0; e < E; ++e) y[e] = (T)(((float)xv[j][i][e] - mean[j]) * rstd[j] * gm[i][e] + bt[i][e]);
                    if (r < p.n_rows && lane + 64 * i < nvec) *reinterpret_cast<V*>(s_x + (int64_t)r * pitch + (int64_t)(lane + 64 * i) * E) = y;
                }
            }
        }
        __syncthreads();
    } else if constexpr (LNS) {
        // (1) all threads: raw rows + gamma/beta -> LDS, XB independent 16-byte requests per thread per round
        constexpr int XB = 8;
        const int vpr = pitch / E;                          // vectors per row
        const int total = p.n_rows * vpr;
        for (int base = threadIdx.x; base < total; base += NW * 64 * XB) {
            V tmp[XB];
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                int idx = base + u * NW * 64;
                int r = idx / vpr, k0 = (idx - r * vpr) * E;
                if (FAST) {   // K % KT == 0: every vector below K is whole; the pad vector (k0 == K) becomes zero
                    int rc = min(r, p.n_rows - 1);
                    tmp[u] = keep_frag<T>(k0 < p.K, ld_frag<T>(x + (int64_t)rc * p.ldx + min(k0, p.K - E)));
                } else {
                    tmp[u] = load_row_frag<T>(x + (int64_t)r * p.ldx, idx < total, k0, p.K, p.vec_x);
                }
            }
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                int idx = base + u * NW * 64;
                if (idx < total) {
                    int r = idx / vpr, k0 = (idx - r * vpr) * E;
                    *reinterpret_cast<V*>(s_x + (int64_t)r * pitch + k0) = tmp[u];
                }
            }
        }
        JB_STAMP(1);
        float* s_g = reinterpret_cast<float*>(s_x + (int64_t)p.n_rows * pitch);   // [pitch] gamma, then [pitch] beta
        float* s_b = s_g + pitch;
        for (int k = threadIdx.x; k < pitch; k += NW * 64) {
            s_g[k] = k < p.K ? p.ln_gamma[k] : 0.f;
            s_b[k] = k < p.K ? p.ln_beta[k] : 0.f;
        }
        __syncthreads();
        JB_STAMP(2);
        // (2) rows r0 = wave (+2NW, ...) and r1 = r0 + NW together: fp32 mean / variance and normalisation, all
        //     from LDS with 16-byte reads (gamma/beta as f32x4: a scalar read at this lane stride is 8-way conflicted)
        for (int r0 = wave; r0 < p.n_rows; r0 += 2 * NW) {
            const bool two = r0 + NW < p.n_rows;
            T* sa = s_x + (int64_t)r0 * pitch;
            T* sb = s_x + (int64_t)(two ? r0 + NW : r0) * pitch;
            float s0 = 0.f, s1 = 0.f;
            for (int k0 = lane * E; k0 < pitch; k0 += 64 * E) {
                s0 += frag_sum<T>(*reinterpret_cast<const V*>(sa + k0));
                s1 += frag_sum<T>(*reinterpret_cast<const V*>(sb + k0));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
            const float mean0 = s0 / (float)p.K, mean1 = s1 / (float)p.K;
            float q0 = 0.f, q1 = 0.f;
            for (int k0 = lane * E; k0 < pitch; k0 += 64 * E) {
                V va = *reinterpret_cast<const V*>(sa + k0), vb = *reinterpret_cast<const V*>(sb + k0);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float w = (FAST ? k0 < p.K : k0 + e < p.K) ? 1.f : 0.f;
                    float da = (float)va[e] - mean0, db = (float)vb[e] - mean1;
                    q0 += w * da * da;
                    q1 += w * db * db;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { q0 += __shfl_xor(q0, o, 64); q1 += __shfl_xor(q1, o, 64); }
            const float rstd0 = 1.0f / sqrtf(q0 / (float)p.K + p.ln_eps), rstd1 = 1.0f / sqrtf(q1 / (float)p.K + p.ln_eps);
            for (int k0 = lane * E; k0 < pitch; k0 += 64 * E) {
                V va = *reinterpret_cast<const V*>(sa + k0), vb = *reinterpret_cast<const V*>(sb + k0);
                float gm[E], bt[E];
#pragma unroll
                for (int e4 = 0; e4 < E; e4 += 4) {
                    f32x4 g4 = *reinterpret_cast<const f32x4*>(s_g + k0 + e4);
                    f32x4 b4 = *reinterpret_cast<const f32x4*>(s_b + k0 + e4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { gm[e4 + u] = g4[u]; bt[e4 + u] = b4[u]; }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = FAST ? (k0 < p.K) : (k0 + e < p.K);
                    va[e] = in ? (T)(((float)va[e] - mean0) * rstd0 * gm[e] + bt[e]) : (T)0;
                    vb[e] = in ? (T)(((float)vb[e] - mean1) * rstd1 * gm[e] + bt[e]) : (T)0;
                }
                *reinterpret_cast<V*>(sa + k0) = va;
                if (two) *reinterpret_cast<V*>(sb + k0) = vb;
            }
        }
        JB_STAMP(3);
        __syncthreads();
        JB_STAMP(4);
    }

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = kt0; kb < kt1; kb += WB) {
        if (kb != kt0) {
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                if (FAST) wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)min(kb + i, p.nkt - 1) * (64 * E)));
                else if (kb + i < kt1) wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)(kb + i) * (64 * E)));
            }
        }
        if (FAST) {
            // all activation fragments of the batch are requested before the first MFMA (the first batch's were requested
            // with the weights, above); tiles past kt1 are neutralised by zeroing their weight fragment (no branch anywhere)
            V xf[WB][MT];
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int k0 = min(kb + i, p.nkt - 1) * KT + g * E;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = min(mt * 16 + c, p.n_rows - 1);
                    if constexpr (LNS) {
                        xf[i][mt] = *reinterpret_cast<const V*>(s_x + (int64_t)row * pitch + k0);
                    } else if constexpr (PIPE) {
                        // the producer launch's rows (operand order: k-tile by k-tile, a contiguous KiB per wave request)
                        xf[i][mt] = jb_ld_frag_sc1<T>(x, (p.pipe.frag & JB_FRAG_X) ? (int64_t)min(kb + i, p.nkt - 1) * (64 * E) + lane * E
                                                                                   : (int64_t)row * p.ldx + k0);
                    } else {
                        if (kb == kt0) xf[i][mt] = xf0[i][mt];
                        else xf[i][mt] = ld_frag<T>(x + (int64_t)row * p.ldx + k0);
                    }
                }
            }
            if constexpr (PIPE) {
                // every fragment of the batch is requested before the first is waited for: the MFMA chain hangs on acc[0], so
                // none of it (and none of its waits) can be scheduled above the requests -- the compiler used to wait for the
                // first fragment with the last two or three requests still unissued: a second dependent round trip
                asm volatile("" : "+v"(acc[0]) :: "memory");
                JB_SEG_VM(p.pipe, 5);
            }
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const V w = keep_frag<T>(kb + i < kt1, wf[i]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = jb_mfma(w, xf[i][mt], acc[mt]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                if (kb + i >= kt1) break;
                const int k0 = (kb + i) * KT + g * E;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int row = mt * 16 + c;
                    V xf;
                    if (LNS) {
                        xf = jb_zero_frag<T>();
                        if (row < p.n_rows && k0 < pitch) xf = *reinterpret_cast<const V*>(s_x + (int64_t)row * pitch + k0);
                    } else {
                        xf = load_row_frag<T>(x + (int64_t)row * p.ldx, row < p.n_rows, k0, p.K, p.vec_x);
                    }
                    acc[mt] = jb_mfma(wf[i], xf, acc[mt]);
                }
            }
        }
    }
    JB_STAMP(5);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s_acc[(wave * MT + mt) * 64 + (PIPE ? jb_pipe_tile_slot(lane) : lane)] = acc[mt];
    if constexpr (PIPE) JB_SEG_LGKM(p.pipe, 6);
    __syncthreads();
    if constexpr (PIPE) JB_SEG(p.pipe, 7);
    JB_STAMP(6);
    eo.finish(p);
    if constexpr (PIPE) {
        // MT == 1: the 256 elements of the tile on the first four waves, thread i = (row i >> 4, column i & 15)
        if (wave < 4) {
            const float* sa = reinterpret_cast<const float*>(s_acc);
            const int row = threadIdx.x >> 4, j = jt * 16 + (threadIdx.x & 15);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += sa[w * 256 + threadIdx.x];
            const int64_t cache_row = (p.epi.qkv_split && t < p.epi.cache_cap) ? (int64_t)row * p.epi.cache_cap + t : -1;
            const float xo = epilogue_value<T>(p.epi, v, eo.bias[0], eo.res[0]);
            pipe_store8<T>(p.epi, xo, xo + e_add2[0], row, j, jt * 16, cache_row, row < p.n_rows && j < p.epi.J, (p.pipe.frag & JB_FRAG_OUT) != 0);
        }
        if constexpr (PIPE == 1) jb_pipe_publish(p.pipe, pipe_own);
        return;
    }
    {
        const float* sa = reinterpret_cast<const float*>(s_acc);
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int i = threadIdx.x + u * NW * 64;
            const int mt = i >> 8, r = (i >> 6) & 3, l = i & 63;
            const int row = mt * 16 + (l & 15), j = jt * 16 + (l >> 4) * 4 + r;
            if (i >= MT * 256 || row >= p.n_rows || j >= p.epi.J) continue;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += sa[((w * MT + mt) * 64 + l) * 4 + r];
            const int64_t cache_row = (p.epi.qkv_split && t < p.epi.cache_cap) ? (int64_t)row * p.epi.cache_cap + t : -1;
            epilogue_store1<T>(p.epi, v, row, j, cache_row, eo.bias[u], eo.res[u], e_add2[u]);
        }
    }
    JB_STAMP(7);
}

// Folded-LayerNorm projection (jb_gemv_args.ln_fold_c1): LN(x)·W + b = rstd·(x·W' - mean·c1) + b' with W' = diag(gamma)·W.
// The MFMA B operands are the RAW activation rows, read straight from global memory in fragment layout, so nothing is
// staged or normalised on the critical path.  The fp32 row statistics ride on the matrix cores too: per k-tile one extra
// MFMA against an all-ones fragment gives sum(x) of the 16 rows, and the fragment against itself gives the Gram matrix
// whose diagonal is sum(x^2) (f16 products are exact in the fp32 accumulator).  The per-wave partial sums join the
// partial tiles in the single LDS exchange before the epilogue, where mean / rstd meet the accumulators.
// A wave keeps all of its k-tiles' fragments in registers: needs ceil(nkt / NW) <= NF.
template <typename T, int MT, int NW, int NF, int PIPE = 0>
__global__ __launch_bounds__(NW * 64) void gemv_lnf_kernel(GemvParams p) {
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E, KT = Frag<T>::KT;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    f32x4* s_acc = reinterpret_cast<f32x4*>(s_dyn);                                   // [NW][MT][64]
    float* s_sum = reinterpret_cast<float*>(s_dyn + NW * MT * 64 * sizeof(f32x4));   // [NW][MT*16]
    float* s_sq = s_sum + NW * MT * 16;                                              // [NW][MT*16]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int jt = blockIdx.x;
    const T* x = (const T*)p.x;
    const int kt0 = (wave * p.nkt) / NW, kt1 = ((wave + 1) * p.nkt) / NW;
    const T* wbase = (const T*)p.W + ((int64_t)jt * p.nkt) * (64 * E) + (int64_t)lane * E;

    // ---- every request of this workgroup, issued back to back ----
    V xf[NF][MT];
    unsigned pipe_own = 0;
    if constexpr (PIPE == 1) pipe_own = jb_pipe_own(p.pipe);
    if constexpr (!PIPE) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int k0 = min(kt0 + i, p.nkt - 1) * KT + g * E;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                xf[i][mt] = ld_frag<T>(x + (int64_t)min(mt * 16 + c, p.n_rows - 1) * p.ldx + k0);
        }
    }
    V wf[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
        wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)min(kt0 + i, p.nkt - 1) * (64 * E)));
    constexpr int EPT = (MT * 256 + NW * 64 - 1) / (NW * 64);
    EpiOperands<T, EPT, NW * 64, (PIPE != 0)> eo;     // bias, column sums c1, residual: in flight with the weights
    eo.request(p, jt, MT);
    jb_issue_fence();
    if constexpr (PIPE) {
        // the weight stream is in flight; the rows are the producer launch's: wait for it (PIPE = 2: the kernel boundary has),
        // then read them write-through
        if constexpr (PIPE == 1) jb_pipe_wait(p.pipe, pipe_own);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int k0 = min(kt0 + i, p.nkt - 1) * KT + g * E;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                xf[i][mt] = jb_ld_frag_sc1<T>(x, (p.pipe.frag & JB_FRAG_X) ? (int64_t)min(kt0 + i, p.nkt - 1) * (64 * E) + lane * E
                                                                           : (int64_t)min(mt * 16 + c, p.n_rows - 1) * p.ldx + k0);
        }
        jb_issue_fence();                      // (all of them in flight before the first MFMA's wait)
        JB_SEG_VM(p.pipe, 5);
    }
    int t = 0;
    if (p.epi.qkv_split) t = PIPE ? (int)jb_ld_word(reinterpret_cast<const unsigned*>(p.t_dev)) : *p.t_dev;

    // ---- projection and row statistics, all on MFMA (tiles past kt1 are neutralised by zeroing the activations) ----
    V ones;
#pragma unroll
    for (int e = 0; e < E; ++e) ones[e] = (T)1.0f;
    f32x4 acc[MT], a1[MT], a2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = a1[mt] = a2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NF; ++i) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const V xm = keep_frag<T>(kt0 + i < kt1, xf[i][mt]);
            acc[mt] = jb_mfma(wf[i], xm, acc[mt]);      // D[j][row] += sum_k W'[k][j] x[row][k]
            a1[mt] = jb_mfma(ones, xm, a1[mt]);         // D[*][row] += sum_k x[row][k]
            a2[mt] = jb_mfma(xm, xm, a2[mt]);           // D[r'][row] += sum_k x[r'][k] x[row][k]
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        s_acc[(wave * MT + mt) * 64 + (PIPE ? jb_pipe_tile_slot(lane) : lane)] = acc[mt];
        if (g == 0) s_sum[wave * (MT * 16) + mt * 16 + c] = a1[mt][0];
        if (g == (c >> 2)) {                            // this lane holds the Gram diagonal of row c in register c & 3
            const int r = c & 3;
            s_sq[wave * (MT * 16) + mt * 16 + c] = r == 0 ? a2[mt][0] : (r == 1 ? a2[mt][1] : (r == 2 ? a2[mt][2] : a2[mt][3]));
        }
    }
    if constexpr (PIPE) JB_SEG_LGKM(p.pipe, 6);
    __syncthreads();
    if constexpr (PIPE) JB_SEG(p.pipe, 7);
    eo.finish(p);
    const float* sa = reinterpret_cast<const float*>(s_acc);
    if constexpr (PIPE) {
        // MT == 1: the 256 elements of the tile on the first four waves, thread i = (row i >> 4, column i & 15)
        if (wave < 4) {
            const int row = threadIdx.x >> 4, j = jt * 16 + (threadIdx.x & 15);
            float v = 0.f, sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                v += sa[w * 256 + threadIdx.x];
                sm += s_sum[w * 16 + row];
                sq += s_sq[w * 16 + row];
            }
            const float mean = sm / (float)p.K;
            const float var = fmaxf(sq / (float)p.K - mean * mean, 0.f);
            v = (v - mean * eo.c1[0]) / sqrtf(var + p.ln_eps);
            const int64_t cache_row = (p.epi.qkv_split && t < p.epi.cache_cap) ? (int64_t)row * p.epi.cache_cap + t : -1;
            const float xo = epilogue_value<T>(p.epi, v, eo.bias[0], eo.res[0]);
            pipe_store8<T>(p.epi, xo, xo, row, j, jt * 16, cache_row, row < p.n_rows && j < p.epi.J, (p.pipe.frag & JB_FRAG_OUT) != 0);
        }
        if constexpr (PIPE == 1) jb_pipe_publish(p.pipe, pipe_own);
        return;
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = threadIdx.x + u * NW * 64;
        const int mt = i >> 8, r = (i >> 6) & 3, l = i & 63;
        const int row = mt * 16 + (l & 15), j = jt * 16 + (l >> 4) * 4 + r;
        if (i >= MT * 256 || row >= p.n_rows || j >= p.epi.J) continue;
        float v = 0.f, sm = 0.f, sq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            v += sa[((w * MT + mt) * 64 + l) * 4 + r];
            sm += s_sum[w * (MT * 16) + row];
            sq += s_sq[w * (MT * 16) + row];
        }
        const float mean = sm / (float)p.K;
        const float var = fmaxf(sq / (float)p.K - mean * mean, 0.f);
        v = (v - mean * eo.c1[u]) / sqrtf(var + p.ln_eps);
        const int64_t cache_row = (p.epi.qkv_split && t < p.epi.cache_cap) ? (int64_t)row * p.epi.cache_cap + t : -1;
        epilogue_store1<T>(p.epi, v, row, j, cache_row, eo.bias[u], eo.res[u]);
    }
}

// Long rows (129 .. 160 k-tiles: 5b_lyrics, K = 4800) on workgroups that leave room.  The 16-wave kernels above keep every
// fragment of a tile in registers: one workgroup fills a compute unit, the 300 column tiles of a 4800-wide projection take two
// rounds on 256 compute units, and the next launch of a pipelined chain is never resident while this one streams (round 4:
// pipelined launches bit-identical and no faster, profiles/r04_multi_head_pipelined_bench_engine_5b.log).  Here a tile is 8 waves
// and a wave walks its <= NF * NB k-tiles through TWO register stages of NF fragments: the fragments of batches 0 and 1 are
// requested up front, batch b + 2 is requested into a stage as soon as batch b has been multiplied out of it -- the same bytes
// in flight per wave as before, half the waves per tile, two tiles per compute unit, all 300 resident at once.  MT = 1 (<= 16 rows),
// whole k-tiles, aligned rows (the callers' fast-path conditions).  LNF: folded LayerNorm (statistics on the matrix cores, as
// gemv_lnf_kernel); else the plain projection (bias / activation / residual / second output, as gemv_kernel).  The partial tiles of
// 8 waves are summed in wave order: results differ from the 16-wave kernels' in the last bits (another association), not in accuracy.
template <typename T, int NW, int NF, int NB, bool LNF, bool PIPE>
__global__ __launch_bounds__(NW * 64) void gemv_long_kernel(GemvParams p) {
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E, KT = Frag<T>::KT;
    static_assert(NB >= 2, "two register stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    f32x4* s_acc = reinterpret_cast<f32x4*>(s_dyn);                               // [NW][64]
    float* s_sum = reinterpret_cast<float*>(s_dyn + NW * 64 * sizeof(f32x4));     // [NW][16]   (LNF)
    float* s_sq = s_sum + NW * 16;                                                // [NW][16]   (LNF)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int jt = blockIdx.x;
    const T* x = (const T*)p.x;
    const int kt0 = (wave * p.nkt) / NW, kt1 = ((wave + 1) * p.nkt) / NW;
    const T* wbase = (const T*)p.W + ((int64_t)jt * p.nkt) * (64 * E) + (int64_t)lane * E;
    const int64_t xrow = (int64_t)min(c, p.n_rows - 1) * p.ldx;

    V wf[2][NF], xf[2][NF];
    auto issue_w = [&](int st, int b) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
            wf[st][i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)min(kt0 + b * NF + i, p.nkt - 1) * (64 * E)));
    };
    auto issue_x = [&](int st, int b) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int k0 = min(kt0 + b * NF + i, p.nkt - 1) * KT + g * E;
            if constexpr (PIPE) xf[st][i] = jb_ld_frag_sc1<T>(x, xrow + k0);      // the producer launch's rows
            else xf[st][i] = ld_frag<T>(x + xrow + k0);
        }
    };
    unsigned pipe_own = 0;
    if constexpr (PIPE) pipe_own = jb_pipe_own(p.pipe);
    // the position (k / v append, the last layer's cond row) is asked for FIRST: the second output's operand needs it for its address,
    // and a request made behind the weights would make that address wait for all of them.  (PIPE: it is the producer's, after the wait.)
    int t = 0;
    if constexpr (!PIPE) { if (p.epi.qkv_split || p.epi.add2) t = *p.t_dev; }
    // requests complete in the order they were made: ask in the order of use (batch 0's weights and rows, then batch 1's), so that
    // waiting for batch 0 leaves batch 1 in flight.  PIPE: the rows can only be asked for after the wait; the weights go first.
    if constexpr (PIPE) { issue_w(0, 0); issue_w(1, 1); }
    else { issue_w(0, 0); issue_x(0, 0); jb_issue_fence(); issue_w(1, 1); issue_x(1, 1); }
    EpiOperands<T, 1, NW * 64, PIPE> eo;        // 256 output elements: one per thread of the first four waves (EPT = 1 for NW >= 4)
    eo.request(p, jt, 1);
    jb_issue_fence();
    if constexpr (PIPE) {
        jb_pipe_wait(p.pipe, pipe_own);
        issue_x(0, 0);
        issue_x(1, 1);
    }
    if constexpr (PIPE) { if (p.epi.qkv_split || p.epi.add2) t = (int)jb_ld_word(reinterpret_cast<const unsigned*>(p.t_dev)); }
    float e_add2 = 0.f;
    if constexpr (!LNF) {
        int emt, er, el;
        epi_coords<PIPE>(threadIdx.x, emt, er, el);
        const int row = el & 15, j = jt * 16 + (el >> 4) * 4 + er;
        const int jc = min(j, p.epi.J - 1), rc = min(row, p.n_rows - 1);
        e_add2 = p.epi.add2 ? p.epi.add2[(int64_t)rc * p.epi.add2_n + (int64_t)t * p.epi.add2_t + jc] : 0.f;
    }

    V ones;
#pragma unroll
    for (int e = 0; e < E; ++e) ones[e] = (T)1.0f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, a1 = acc, a2 = acc;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int st = b & 1;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const bool valid = kt0 + b * NF + i < kt1;                        // tiles past the wave's range: loaded (clamped), zeroed
            if constexpr (LNF) {
                const V xm = keep_frag<T>(valid, xf[st][i]);
                acc = jb_mfma(wf[st][i], xm, acc);
                a1 = jb_mfma(ones, xm, a1);
                a2 = jb_mfma(xm, xm, a2);
            } else {
                acc = jb_mfma(keep_frag<T>(valid, wf[st][i]), xf[st][i], acc);
            }
        }
        // the stage is free: request batch b + 2 NOW, before the other stage is multiplied (scheduling barriers pin the requests between
        // the two batches' MFMAs; without them the scheduler sank every reload behind the multiplication of both first batches)
        __builtin_amdgcn_sched_barrier(0);
        if (b + 2 < NB) { issue_w(st, b + 2); issue_x(st, b + 2); }
        __builtin_amdgcn_sched_barrier(0);
    }
    s_acc[wave * 64 + (PIPE ? jb_pipe_tile_slot(lane) : lane)] = acc;
    if constexpr (LNF) {
        if (g == 0) s_sum[wave * 16 + c] = a1[0];
        if (g == (c >> 2)) {                            // this lane holds the Gram diagonal of row c in register c & 3
            const int r = c & 3;
            s_sq[wave * 16 + c] = r == 0 ? a2[0] : (r == 1 ? a2[1] : (r == 2 ? a2[2] : a2[3]));
        }
    }
    __syncthreads();
    eo.finish(p);
    const float* sa = reinterpret_cast<const float*>(s_acc);
    if (wave < 4) {
        int mt, r, l;
        epi_coords<PIPE>(threadIdx.x, mt, r, l);
        const int row = l & 15, j = jt * 16 + (l >> 4) * 4 + r;
        float v = 0.f, sm = 0.f, sq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            v += PIPE ? sa[w * 256 + threadIdx.x] : sa[(w * 64 + l) * 4 + r];      // (PIPE: parked as [row][column], jb_pipe_tile_slot)
            if constexpr (LNF) { sm += s_sum[w * 16 + row]; sq += s_sq[w * 16 + row]; }
        }
        if constexpr (LNF) {
            const float mean = sm / (float)p.K;
            const float var = fmaxf(sq / (float)p.K - mean * mean, 0.f);
            v = (v - mean * eo.c1[0]) / sqrtf(var + p.ln_eps);
        }
        const int64_t cache_row = (p.epi.qkv_split && t < p.epi.cache_cap) ? (int64_t)row * p.epi.cache_cap + t : -1;
        if constexpr (PIPE) {
            const float xo = epilogue_value<T>(p.epi, v, eo.bias[0], eo.res[0]);
            pipe_store8<T>(p.epi, xo, xo + e_add2, row, j, jt * 16, cache_row, row < p.n_rows && j < p.epi.J, false);
        } else if (row < p.n_rows && j < p.epi.J) {
            epilogue_store1<T>(p.epi, v, row, j, cache_row, eo.bias[0], eo.res[0], e_add2);
        }
    }
    if constexpr (PIPE) jb_pipe_publish(p.pipe, pipe_own);
}

// 1 (default): rows of 129 .. 160 k-tiles (fp16, <= 16 rows) take gemv_long_kernel (8 waves, two register stages) instead of the
// 16-wave kernels -- measured on the 5b_lyrics top prior's decode step at N = 3: 4.11 -> 3.84 ms (profiles/r05_bench_engine_5b.log);
// 0: the 16-wave kernels (jb_tune_gemv_long)
static int g_gemv_long = 1;
extern "C" void jb_tune_gemv_long(int on) { g_gemv_long = on; }

// attn.c_proj of the decode step fed by the KEY-SPLIT decode attention (jb_attn_decode_split): the B operand rows are
// formed on the fly from the n_parts (<= 4) partial softmax states of each (sample, head) --
//     x[n][k] = sum_s w_s * parts[n][s][k],   w_s = l_s * exp(m_s - m) / sum_s' l_s' * exp(m_s' - m),  m = max_s m_s
// (parts are normalised by their own sum l_s, f16; (m_s, l_s) fp32 in x_ml[n][head][s]) -- and rounded to half once, the
// point where the reference materialises the attention output (factored_attention.py:107-108).  A head owns whole
// k-tiles (d_head % 32 == 0), so a lane's weights depend on (row, k-tile) only.  Everything is requested up front;
// the log-sum-exp merge costs n_parts fused multiply-adds per operand element while the weight stream is in flight.
// This removes the merge from the attention kernel, which can then spread one (sample, head) over several CUs.
template <int MT, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_merge_kernel(GemvParams p) {
    using V = f16x8;
    constexpr int E = 8, KT = 32, WB = 4, PMAX = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    f32x4* s_acc = reinterpret_cast<f32x4*>(s_dyn);                 // [NW][MT][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int jt = blockIdx.x;
    const f16* parts = (const f16*)p.x_parts;
    const int P = p.n_parts, H = p.n_head;
    const int kt0 = (wave * p.nkt) / NW, kt1 = ((wave + 1) * p.nkt) / NW;
    const f16* wbase = (const f16*)p.W + ((int64_t)jt * p.nkt) * (64 * E) + (int64_t)lane * E;

    constexpr int EPT = (MT * 256 + NW * 64 - 1) / (NW * 64);
    EpiOperands<f16, EPT, NW * 64> eo;
    eo.request(p, jt, MT);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = kt0; kb < kt1; kb += WB) {
        V wf[WB], ph[WB][MT][PMAX];
        float mm[WB][MT][PMAX], ll[WB][MT][PMAX];
#pragma unroll
        for (int i = 0; i < WB; ++i) {
            const int kt = min(kb + i, p.nkt - 1);
            wf[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(wbase + (int64_t)kt * (64 * E)));
            const int k0 = kt * KT + g * E;
            const int h = (kt * KT) / p.d_head;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = min(mt * 16 + c, p.n_rows - 1);
#pragma unroll
                for (int s = 0; s < PMAX; ++s) {
                    const int sc = min(s, P - 1);                       // clamped: parts past P are loaded again, weight 0
                    ph[i][mt][s] = ld_frag<f16>(parts + ((int64_t)row * P + sc) * p.K + k0);
                    const float2 v = *reinterpret_cast<const float2*>(p.x_ml + (((int64_t)row * H + h) * P + sc) * 2);
                    mm[i][mt][s] = v.x;
                    ll[i][mt][s] = s < P ? v.y : 0.f;
                }
            }
        }
        jb_issue_fence();
#pragma unroll
        for (int i = 0; i < WB; ++i) {
            const V w = keep_frag<f16>(kb + i < kt1, wf[i]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // log-sum-exp weights of the parts (hardware exp2 / rcp: this sits on the critical path of every workgroup;
                // an empty part has l = 0 and a zero row, so no select is needed below)
                float m = mm[i][mt][0];
#pragma unroll
                for (int s = 1; s < PMAX; ++s) m = fmaxf(m, mm[i][mt][s]);
                float ws[PMAX], tot = 0.f;
#pragma unroll
                for (int s = 0; s < PMAX; ++s) {
                    const float ex = __builtin_amdgcn_exp2f((mm[i][mt][s] - m) * 1.4426950408889634f);
                    ws[s] = ll[i][mt][s] > 0.f ? ll[i][mt][s] * ex : 0.f;
                    tot += ws[s];
                }
                const float inv = tot > 0.f ? __builtin_amdgcn_rcpf(tot) : 0.f;   // no key at all (prev_block in block 0): a = 0
#pragma unroll
                for (int s = 0; s < PMAX; ++s) ws[s] *= inv;
                V xv;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float a = 0.f;
#pragma unroll
                    for (int s = 0; s < PMAX; ++s) a = fmaf(ws[s], (float)ph[i][mt][s][e], a);
                    xv[e] = (f16)a;
                }
                acc[mt] = jb_mfma(w, xv, acc[mt]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s_acc[(wave * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    eo.finish(p);
    const float* sa = reinterpret_cast<const float*>(s_acc);
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = threadIdx.x + u * NW * 64;
        const int mt = i >> 8, r = (i >> 6) & 3, l = i & 63;
        const int row = mt * 16 + (l & 15), j = jt * 16 + (l >> 4) * 4 + r;
        if (i >= MT * 256 || row >= p.n_rows || j >= p.epi.J) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += sa[((w * MT + mt) * 64 + l) * 4 + r];
        epilogue_store1<f16>(p.epi, v, row, j, -1, eo.bias[u], eo.res[u]);
    }
}

template <int NW>
static int launch_gemv_merge(const GemvParams& p, int njt, hipStream_t s) {
    const int mt = (p.n_rows + 15) / 16;
    const size_t lds = (size_t)NW * mt * 64 * sizeof(f32x4);
    switch (mt) {
        case 1: gemv_merge_kernel<1, NW><<<njt, NW * 64, lds, s>>>(p); break;
        case 2: gemv_merge_kernel<2, NW><<<njt, NW * 64, lds, s>>>(p); break;
        default: jb_set_error("jb_gemv: the split-attention operand supports n_rows <= 32"); return JB_ERR_UNSUPPORTED;
    }
    return JB_OK;
}

// k-tiles a wave must hold for the folded-LayerNorm kernel, or 0 when the problem is outside its envelope.
static int lnf_shape(int dtype, int K, int J, int n_rows, int* nw_out) {
    const int KT = dtype == JB_F16 ? 32 : 16, E = dtype == JB_F16 ? 8 : 4;
    if (K <= 0 || J < 4 || K % KT != 0 || J % 4 != 0 || K < E || n_rows < 1 || n_rows > 32) return 0;
    const int nkt = K / KT;
    int nw = nkt >= 32 ? 8 : 4;
    int per_wave = (nkt + nw - 1) / nw;
    if (per_wave > 16) {         // long rows (5b_lyrics: K = 4800 = 150 k-tiles): 16 waves, <= 10 fragments each (128 VGPRs)
        nw = 16;
        per_wave = (nkt + nw - 1) / nw;
        *nw_out = nw;
        if (n_rows > 16) return 0;
        return per_wave <= 8 ? 8 : (per_wave <= 10 ? 10 : 0);
    }
    *nw_out = nw;
    return per_wave <= 8 ? 8 : 16;
}

extern "C" int jb_gemv_ln_fold_supported(int dtype, int K, int J, int n_rows) {
    int nw;
    return (dtype == JB_F32 || dtype == JB_F16) && lnf_shape(dtype, K, J, n_rows, &nw) != 0;
}

template <typename T, int MT, int NW>
static int launch_gemv_lnf_nf(const GemvParams& p, int njt, int nf, hipStream_t s) {
    const size_t lds = (size_t)NW * MT * 64 * sizeof(f32x4) + (size_t)2 * NW * MT * 16 * sizeof(float);
    if (p.pipe.slot != -1) {
        const bool sync = p.pipe.slot >= 0;          // (else JB_PIPE_NO_SYNC: the same kernel form as a launch of a plain chain)
        if constexpr (MT == 1 && NW == 8 && sizeof(T) == 2) {
            if (nf == 8) {
                if (sync) gemv_lnf_kernel<T, 1, 8, 8, 1><<<njt, NW * 64, lds, s>>>(p);
                else gemv_lnf_kernel<T, 1, 8, 8, 2><<<njt, NW * 64, lds, s>>>(p);
                return JB_OK;
            }
        }
        if constexpr (MT == 1 && NW == 16 && sizeof(T) == 2) {       // 129 .. 160 k-tiles (5b_lyrics: K = 4800)
            if (nf == 10 && sync) { gemv_lnf_kernel<T, 1, 16, 10, 1><<<njt, NW * 64, lds, s>>>(p); return JB_OK; }
        }
        jb_set_error("jb_gemv: a pipelined launch of the folded-LayerNorm projection takes fp16, <= 16 rows, 33..64 or 129..160 k-tiles "
                     "(the form without a hand-shake: 33..64)");
        return JB_ERR_UNSUPPORTED;
    }
    if (nf == 8) gemv_lnf_kernel<T, MT, NW, 8><<<njt, NW * 64, lds, s>>>(p);
    else if constexpr (NW == 16) gemv_lnf_kernel<T, MT, NW, 10><<<njt, NW * 64, lds, s>>>(p);
    else gemv_lnf_kernel<T, MT, NW, 16><<<njt, NW * 64, lds, s>>>(p);
    return JB_OK;
}

template <typename T>
static int launch_gemv_lnf(const GemvParams& p, int njt, int nf, int nw, hipStream_t s) {
    const int mt = (p.n_rows + 15) / 16;
    if (nw == 16) {
        if (mt != 1) { jb_set_error("jb_gemv: folded LayerNorm over > 16 k-tiles per wave takes n_rows <= 16"); return JB_ERR_UNSUPPORTED; }
        if constexpr (sizeof(T) == 2) {
            if (g_gemv_long && p.nkt <= 160) {          // 8 waves x 4 batches of 5 k-tiles
                const size_t lds = (size_t)8 * 64 * sizeof(f32x4) + (size_t)2 * 8 * 16 * sizeof(float);
                if (p.pipe.slot >= 0) gemv_long_kernel<T, 8, 5, 4, true, true><<<njt, 8 * 64, lds, s>>>(p);
                else gemv_long_kernel<T, 8, 5, 4, true, false><<<njt, 8 * 64, lds, s>>>(p);
                return JB_OK;
            }
        }
        return launch_gemv_lnf_nf<T, 1, 16>(p, njt, nf, s);
    }
    if (nw == 8) return mt == 1 ? launch_gemv_lnf_nf<T, 1, 8>(p, njt, nf, s) : launch_gemv_lnf_nf<T, 2, 8>(p, njt, nf, s);
    return mt == 1 ? launch_gemv_lnf_nf<T, 1, 4>(p, njt, nf, s) : launch_gemv_lnf_nf<T, 2, 4>(p, njt, nf, s);
}

template <typename T, int MT, int NW, bool LNS, bool FAST, int NV>
static int launch_gemv_fast(const GemvParams& p, int njt, size_t lds, hipStream_t s) {
    // the opt-in for > 64 KiB of dynamic LDS is a property of (function, device): remember it per device
    static bool configured[64] = {};
    int dev = 0;
    if (lds > 64 * 1024) {
        JB_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !configured[dev]) {
            JB_HIP(hipFuncSetAttribute((const void*)gemv_kernel<T, MT, NW, LNS, FAST, NV>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) configured[dev] = true;
        }
    }
    gemv_kernel<T, MT, NW, LNS, FAST, NV><<<njt, NW * 64, lds, s>>>(p);
    return JB_OK;
}

template <typename T, int MT, int NW, bool LNS>
static int launch_gemv_inst(const GemvParams& p, int njt, size_t lds, hipStream_t s) {
    if (p.pipe.slot != -1) {
        if constexpr (MT == 1 && (NW == 8 || NW == 4) && !LNS) {
            if (p.fast && p.pipe.slot >= 0) { gemv_kernel<T, 1, NW, false, true, 0, 1><<<njt, NW * 64, lds, s>>>(p); return JB_OK; }
            if constexpr (NW == 8) {      // JB_PIPE_NO_SYNC: the same kernel form as a launch of a plain chain
                if (p.fast) { gemv_kernel<T, 1, 8, false, true, 0, 2><<<njt, NW * 64, lds, s>>>(p); return JB_OK; }
            }
        }
        jb_set_error("jb_gemv: a pipelined launch of the plain projection takes <= 16 rows, >= 32 whole k-tiles, aligned operands");
        return JB_ERR_UNSUPPORTED;
    }
    if (!p.fast) return launch_gemv_fast<T, MT, NW, LNS, false, 0>(p, njt, lds, s);
    if (LNS) {
        // vectors per lane per row on the register LayerNorm path (0 = staged-through-LDS variant)
        const int nv = (p.K / Frag<T>::E + 63) / 64;
        if (nv <= 2) return launch_gemv_fast<T, MT, NW, LNS, true, LNS ? 2 : 0>(p, njt, lds, s);
        if (nv <= 4) return launch_gemv_fast<T, MT, NW, LNS, true, LNS ? 4 : 0>(p, njt, lds, s);
        if (nv <= 8) return launch_gemv_fast<T, MT, NW, LNS, true, LNS ? 8 : 0>(p, njt, lds, s);
    }
    return launch_gemv_fast<T, MT, NW, LNS, true, 0>(p, njt, lds, s);
}

template <typename T, int NW, bool LNS>
static int launch_gemv_nw(const GemvParams& p, int njt, size_t lds, hipStream_t s) {
    int mt = (p.n_rows + 15) / 16;
    switch (mt) {
        case 1: return launch_gemv_inst<T, 1, NW, LNS>(p, njt, lds, s);
        case 2: return launch_gemv_inst<T, 2, NW, LNS>(p, njt, lds, s);
        case 3: return launch_gemv_inst<T, 3, NW, LNS>(p, njt, lds, s);
        default: return launch_gemv_inst<T, 4, NW, LNS>(p, njt, lds, s);
    }
}

template <typename T>
static int launch_gemv(GemvParams& p, int njt, bool ln, hipStream_t s) {
    constexpr int E = Frag<T>::E;
    const int mt = (p.n_rows + 15) / 16;
    // 8 waves when every wave still gets >= 4 k-tiles (more bytes in flight per CU), else 4; long rows (5b_lyrics:
    // K = 4800, 150 k-tiles) 16 waves of <= 10 k-tiles, all of them requested up front
    int nw = p.nkt >= 32 ? 8 : 4;
    if (!ln && mt == 1 && p.fast && p.nkt > 128 && p.nkt <= 160) {
        p.lds_pitch = 0;
        if constexpr (sizeof(T) == 2) {
            if (g_gemv_long) {                          // 8 waves x 4 batches of 5 k-tiles
                const size_t lds = (size_t)8 * 64 * sizeof(f32x4);
                if (p.pipe.slot >= 0) gemv_long_kernel<T, 8, 5, 4, false, true><<<njt, 8 * 64, lds, s>>>(p);
                else gemv_long_kernel<T, 8, 5, 4, false, false><<<njt, 8 * 64, lds, s>>>(p);
                return JB_OK;
            }
        }
        if (p.pipe.slot != -1) {
            if constexpr (sizeof(T) == 2) {
                if (p.pipe.slot >= 0) {
                    gemv_kernel<T, 1, 16, false, true, 0, 1><<<njt, 16 * 64, (size_t)16 * 64 * sizeof(f32x4), s>>>(p);
                    return JB_OK;
                }
            }
            jb_set_error("jb_gemv: a pipelined launch of the 16-wave projection takes fp16");
            return JB_ERR_UNSUPPORTED;
        }
        return launch_gemv_fast<T, 1, 16, false, true, 0>(p, njt, (size_t)16 * 64 * sizeof(f32x4), s);
    }
    if (ln && nw == 8) {   // keep the staged rows + partial tiles within the 160 KiB of LDS
        size_t pit = ((p.K + E - 1) / E) * E + E;
        size_t need = (size_t)8 * mt * 64 * sizeof(f32x4) + (size_t)p.n_rows * pit * sizeof(T) + 2 * pit * sizeof(float);
        if (need > 160 * 1024) nw = 4;
    }
    size_t lds = (size_t)nw * mt * 64 * sizeof(f32x4);
    if (ln) {
        // row pitch: K rounded up to E, plus one 16-byte slot so that the 16 rows of a fragment read
        // fall into different LDS slots (cdna_hip_programming.md Guideline 4)
        p.lds_pitch = ((p.K + E - 1) / E) * E + E;
        lds += (size_t)p.n_rows * p.lds_pitch * sizeof(T) + (size_t)2 * p.lds_pitch * sizeof(float);   // rows, then gamma | beta
        if (lds > 160 * 1024) {
            jb_set_error("jb_gemv: LayerNorm rows do not fit in LDS (n_rows x K too large)");
            return JB_ERR_UNSUPPORTED;
        }
        return nw == 8 ? launch_gemv_nw<T, 8, true>(p, njt, lds, s) : launch_gemv_nw<T, 4, true>(p, njt, lds, s);
    }
    p.lds_pitch = 0;
    return nw == 8 ? launch_gemv_nw<T, 8, false>(p, njt, lds, s) : launch_gemv_nw<T, 4, false>(p, njt, lds, s);
}

#ifdef JB_TIMING
long long* jb_dbg_ptr = nullptr;
extern "C" void jb_set_dbg(long long* p) { jb_dbg_ptr = p; }
#endif

extern "C" int jb_gemv(const jb_gemv_args* a, void* stream) { return jb_gemv_impl(a, nullptr, stream); }

// pipe != NULL: one launch of a software-pipelined chain (common.h, JbPipe)
int jb_gemv_impl(const jb_gemv_args* a, const JbPipe* pipe, void* stream) {
    JB_REQUIRE(a && (a->x || a->x_parts) && a->W && a->out, "null pointer");
    JB_REQUIRE(a->dtype == JB_F32 || a->dtype == JB_F16, "bad dtype");
    JB_REQUIRE(a->n_rows >= 1 && a->n_rows <= 64, "n_rows must be 1..64");
    JB_REQUIRE(a->K > 0 && a->J > 0, "empty problem");
    JB_REQUIRE(a->wide >= 0 && (a->wide == 0 || (a->qkv_split && a->vcache_wide)), "wide v' columns need qkv_split and vcache_wide");
    JB_REQUIRE(!a->qkv_split || (a->S > 0 && a->kcache && a->t_dev && (a->vcache || a->wide > 0) &&
                                 a->J == 2 * a->S + (a->vcache ? a->S : 0) + a->wide), "bad qkv split");
    JB_REQUIRE(!a->out2 || (a->ldo2 >= a->J && (!a->add2 || a->t_dev)), "bad second output (out2 / add2 need ldo2 >= J and t_dev)");
    JB_REQUIRE(!a->out2 || (!a->ln_gamma && !a->ln_fold_c1 && !a->x_parts), "out2 is available on the plain projection only");
    if (a->x_parts) {
        JB_REQUIRE(!a->x && a->x_ml && a->dtype == JB_F16 && !a->ln_gamma && !a->ln_fold_c1 && !a->qkv_split,
                   "split-attention operand: fp16, no LayerNorm, no q/k/v split, x must be NULL");
        JB_REQUIRE(a->n_parts >= 1 && a->n_parts <= 4 && a->n_head >= 1 && a->d_head > 0 && a->d_head % 32 == 0 &&
                       a->n_head * a->d_head == a->K && a->n_rows <= 32 && aligned_to(a->x_parts, 16),
                   "split-attention operand: 1..4 parts, d_head a multiple of 32, K = n_head * d_head, n_rows <= 32");
    }
    const int esz = a->dtype == JB_F16 ? 2 : 4;
    const int E = a->dtype == JB_F16 ? 8 : 4;
    GemvParams p;
    int njt, Edummy;
    packed_dims(a->K, a->J, a->dtype, &p.nkt, &njt, &Edummy);
    p.x = a->x; p.ldx = a->ldx; p.n_rows = a->n_rows;
    p.ln_gamma = a->ln_gamma; p.ln_beta = a->ln_beta; p.ln_eps = a->ln_eps; p.ln_c1 = a->ln_fold_c1;
    p.W = a->W; p.K = a->K;
    p.vec_x = a->x && (a->ldx % E == 0) && aligned_to(a->x, 16);
    p.t_dev = a->t_dev;
    p.x_parts = a->x_parts; p.x_ml = a->x_ml; p.n_parts = a->n_parts; p.n_head = a->n_head; p.d_head = a->d_head;
    p.pipe = pipe ? *pipe : JbPipe{nullptr, nullptr, nullptr, -1, -1, 0, nullptr};
    JB_REQUIRE(!pipe || (!a->x_parts && !a->ln_gamma && a->J % 16 == 0 && a->n_rows <= 16 && a->ldo % 8 == 0 && aligned_to(a->out, 16) &&
                         (!a->qkv_split || (a->S % 16 == 0 && a->wide % 8 == 0 && aligned_to(a->kcache, 16) &&
                                            (!a->vcache || aligned_to(a->vcache, 16)) && (!a->vcache_wide || aligned_to(a->vcache_wide, 16)))) &&
                         (!a->out2 || (a->ldo2 % 4 == 0 && aligned_to(a->out2, 16)))),
               "a pipelined launch takes the plain or the folded-LayerNorm projection, <= 16 rows, whole 16-column tiles, 16-byte "
               "aligned rows (its outputs leave as 16-byte pieces)");
    // completion protocol 1 has a flag word per ticket shard (workgroup index mod 8): a launch of fewer than 8 workgroups would
    // leave flags that nobody writes, and every consumer would sit out its time-out on them
    JB_REQUIRE(!pipe || pipe->slot < 0 || pipe->proto < 1 || njt >= 8, "completion protocol 1 needs launches of >= 8 workgroups (J >= 128)");
    JB_REQUIRE(!pipe || pipe->slot >= 0 || (pipe->slot == JB_PIPE_NO_SYNC && p.nkt >= 32 && p.nkt <= 128 && a->n_rows <= 16),
               "the pipelined kernel form without its hand-shake (JB_PIPE_NO_SYNC) exists for the 8-wave projections: 32..128 k-tiles");
    p.dbg = nullptr;
#ifdef JB_TIMING
    p.dbg = jb_dbg_ptr;
#endif
    {   // branch-free fast path: whole k-tiles, vector-aligned activations / bias / residual / output
        const int KT = a->dtype == JB_F16 ? 32 : 16;
        bool f = p.vec_x && (a->K % KT == 0) && (a->J % 4 == 0) && (a->J >= 4);
        if (a->bias) f = f && aligned_to(a->bias, 16);
        if (a->res) f = f && (a->ldr % 4 == 0) && aligned_to(a->res, 4 * esz);
        if (a->ln_gamma) f = f && (a->K >= E);
        p.fast = f;
    }
    p.epi.bias = a->bias; p.epi.out = a->out; p.epi.ldo = a->ldo; p.epi.res = a->res; p.epi.ldr = a->ldr;
    p.epi.J = a->J; p.epi.act = a->act; p.epi.res_scale = 1.0f;
    p.epi.qkv_split = a->qkv_split; p.epi.S = a->S; p.epi.kcache = a->kcache; p.epi.vcache = a->vcache;
    p.epi.cache_cap = a->cache_cap;
    p.epi.vcache2 = a->vcache_wide; p.epi.v2w = a->wide; p.epi.v_cols = a->vcache ? a->S : 0;
    p.epi.vec_out = !a->qkv_split && (a->ldo % 4 == 0) && aligned_to(a->out, 4 * esz);
    p.epi.out2 = a->out2; p.epi.ldo2 = a->ldo2; p.epi.add2 = a->out2 ? a->add2 : nullptr;
    p.epi.add2_n = a->add2_n_stride; p.epi.add2_t = a->add2_t_stride;
    int rc;
    if (a->x_parts) {
        rc = p.nkt >= 32 ? launch_gemv_merge<8>(p, njt, (hipStream_t)stream) : launch_gemv_merge<4>(p, njt, (hipStream_t)stream);
    } else if (a->ln_fold_c1) {
        JB_REQUIRE(!a->ln_gamma && !a->ln_beta, "ln_fold_c1 excludes ln_gamma / ln_beta (gamma and beta are folded into W and bias)");
        int nw = 0;
        const int nf = lnf_shape(a->dtype, a->K, a->J, a->n_rows, &nw);
        JB_REQUIRE(nf != 0 && p.fast, "folded LayerNorm is not available for this shape / alignment (jb_gemv_ln_fold_supported)");
        rc = a->dtype == JB_F16 ? launch_gemv_lnf<f16>(p, njt, nf, nw, (hipStream_t)stream)
                                : launch_gemv_lnf<float>(p, njt, nf, nw, (hipStream_t)stream);
    } else {
        rc = a->dtype == JB_F16 ? launch_gemv<f16>(p, njt, a->ln_gamma != nullptr, (hipStream_t)stream)
                                : launch_gemv<float>(p, njt, a->ln_gamma != nullptr, (hipStream_t)stream);
    }
    if (rc != JB_OK) return rc;
    JB_CHECK_LAUNCH();
    return JB_OK;
}
