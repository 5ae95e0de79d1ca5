// Pattern-aware cached attention for gfx950.
//
// Cache layout in HBM: per layer K and V are [sample][position][n_state] rows of the engine dtype,
// full length (cache_cap = sequence length; prime layers cache_cap = rounded prime length).  The
// reference trims its caches per pattern (factored_attention.py:328-353) to save V100 memory; with
// 288 GB the full-length cache lets every pattern be a closed-form index set over one array
// (SURVEY.md Appendix B): a contiguous tail (dense / block / prev / prime) or a stride-block_ctx
// gather (transpose).
#include <type_traits>
#include "common.h"

// Keeps the compiler from sinking the memory requests written above this point below it (nothing waits here): the decode
// kernels live on issuing every load of a phase back to back -- left alone, the scheduler parks the 16 value-row loads of
// a tile behind the QK^T MFMA chain (a third dependent round trip after position and keys).
__device__ __forceinline__ void jb_issue_fence() { asm volatile("" ::: "memory"); }
// The same, and additionally everything that consumes `x` (e.g. the MFMA chain fed by a query fragment) stays below it:
// register-only instructions are otherwise free to be hoisted above the requests the fence was meant to put first.
template <typename V> __device__ __forceinline__ void jb_issue_fence_before_use(V& x) { asm volatile("" : "+v"(x) :: "memory"); }

// key set of the single query at position t: positions start + i*stride, i < count
struct KeySet { int start, stride, count; };

__device__ __forceinline__ KeySet decode_key_set(int func, int t, int bc, int cap) {
    KeySet k{0, 1, 0};
    switch (func) {
        case JB_ATTN_DENSE: k.count = t + 1; break;
        case JB_ATTN_BLOCK: k.start = (t / bc) * bc; k.count = t - k.start + 1; break;
        case JB_ATTN_TRANSPOSE_BLOCK: k.start = t % bc; k.stride = bc; k.count = t / bc + 1; break;
        case JB_ATTN_PREV_BLOCK: {
            int blk = t / bc;
            if (blk > 0) { k.start = (blk - 1) * bc; k.count = bc; }   // block 0: zero rows -> output 0
            break;
        }
        case JB_ATTN_PRIME: k.count = min(t + 1, cap); break;
        case JB_ATTN_CROSS: k.count = cap; break;
    }
    return k;
}

// ------------------------------------------------------------------------------------------------
// Decode (q_l == 1).  One workgroup per (sample, head), one dependent memory round trip: wave w takes
// keys w, w+nw, ... and requests KB key rows AND their value rows together (a row is d_head * sizeof(T)
// <= 1-2 KiB, one or two fully coalesced wave loads), dots the keys with the query held in registers,
// and folds them into a per-wave online softmax (running max, sum and output slice).  The per-wave
// states are then merged in a fixed order through LDS.  HBM/L2-bound GEMV per sample: MFMA has no
// reuse to exploit with a single query row, so this runs on the vector ALU; the prefill kernel below
// is the MFMA one.  In half mode the probabilities are rounded to half before multiplying the values
// (factored_attention.py:98) relative to the running maximum.
template <typename T, int NCH, int KB>
__global__ void attn_decode_kernel(int func, const T* __restrict__ q, int64_t ldq, const T* __restrict__ kc,
                                   const T* __restrict__ vc, int cap, T* __restrict__ out, int64_t ldo, int n_head,
                                   int d, int bc, const int* __restrict__ t_dev) {
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* s_ml = smem;                      // [nw][2] running max / sum per wave
    float* s_o = smem + 2 * nw;              // [nw][d] partial outputs
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x, h = blockIdx.y;
    const int S = n_head * d;
    const int t = *t_dev;
    const KeySet ks = decode_key_set(func, t, bc, cap);
    T* o = out + (int64_t)n * ldo + h * d;
    if (ks.count == 0) {
        for (int i = threadIdx.x; i < d; i += blockDim.x) o[i] = (T)0;
        return;
    }
    const float scale = 1.0f / sqrtf(sqrtf((float)d));
    const float scale2 = scale * scale;
    const bool vec = (d % E == 0) && (S % E == 0) && (ldq % E == 0);

    float qf[NCH][E];
    const T* qrow = q + (int64_t)n * ldq + h * d;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = (ch * 64 + lane) * E;
        V qv = vec ? keep_frag<T>(c0 < d, ld_frag<T>(qrow + min(c0, d - E))) : load_row_frag<T>(qrow, true, c0, d, false);
#pragma unroll
        for (int e = 0; e < E; ++e) qf[ch][e] = (float)qv[e];
    }
    const T* kbase = kc + ((int64_t)n * cap) * S + h * d;
    const T* vbase = vc + ((int64_t)n * cap) * S + h * d;

    float m_w = -INFINITY, l_w = 0.f;
    float of[NCH][E];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int e = 0; e < E; ++e) of[ch][e] = 0.f;

    for (int i0 = wave; i0 < ks.count; i0 += nw * KB) {
        V kv[KB][NCH], vv[KB][NCH];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            const int i = i0 + b * nw;
            const bool ok = i < ks.count;
            const int64_t roff = (int64_t)(ks.start + (ok ? i : 0) * ks.stride) * S;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c0 = (ch * 64 + lane) * E;
                if (vec) {   // uniform: unconditional loads at clamped offsets (rows past count / lanes past d are unused)
                    const int cc = min(c0, d - E);
                    kv[b][ch] = ld_frag<T>(kbase + roff + cc);
                    vv[b][ch] = ld_frag<T>(vbase + roff + cc);
                } else {
                    kv[b][ch] = load_row_frag<T>(kbase + roff, ok, c0, d, false);
                    vv[b][ch] = load_row_frag<T>(vbase + roff, ok, c0, d, false);
                }
            }
        }
        jb_issue_fence();
        float part[KB];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            part[b] = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int e = 0; e < E; ++e) part[b] += qf[ch][e] * (float)kv[b][ch][e];
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1)
#pragma unroll
            for (int b = 0; b < KB; ++b) part[b] += __shfl_xor(part[b], o2, 64);
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            if (i0 + b * nw >= ks.count) break;
            // reference: w = matmul(q, k) (half result), w.mul_(scale*scale) (half), then .float()
            const float sc = jb_round<T>(jb_round<T>(part[b]) * scale2);
            const float m_new = fmaxf(m_w, sc);
            const float alpha = expf(m_w - m_new);          // exp(-inf) = 0 on the first key
            const float pe = expf(sc - m_new);
            l_w = l_w * alpha + pe;
            const float pr = jb_round<T>(pe);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int e = 0; e < E; ++e) of[ch][e] = of[ch][e] * alpha + pr * (float)vv[b][ch][e];
            m_w = m_new;
        }
    }
    if (lane == 0) { s_ml[2 * wave] = m_w; s_ml[2 * wave + 1] = l_w; }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int i = (ch * 64 + lane) * E + e;
            if (i < d) s_o[wave * d + i] = of[ch][e];
        }
    __syncthreads();
    float m = -INFINITY;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, s_ml[2 * w]);
    float lsum = 0.f;
    for (int w = 0; w < nw; ++w) lsum += s_ml[2 * w + 1] * expf(s_ml[2 * w] - m);
    const float inv = 1.0f / lsum;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += s_o[w * d + i] * expf(s_ml[2 * w] - m);
        o[i] = (T)(a * inv);
    }
}

// Decode attention, fp16 fast path: QK^T on MFMA.  Wave w owns 16-key tiles w, w+nw, ...; for a tile it requests the
// K fragments (lane = key l&15, channel slot l>>4: the A operand) and the 16 value rows (lane = 8-channel slice)
// together, multiplies the K tile with the query replicated in all 16 B columns -- so every lane ends up with the
// scores of keys (l>>4)*4 + r and the softmax statistics need only two cross-lane steps -- and accumulates p*V on the
// vector ALU from the coalesced value rows.  No per-key shuffle reduction (6 ds_bpermute per key in the generic kernel).
// RAGGED: d_head is any even number <= ND32 * 32 (5b_lyrics: 8 heads of 150 channels in 5 k-tiles).  The fragment that
// straddles d_head (and those past it) is read from channel d_head - 8 on, for the query and the keys alike, so every
// address stays inside the head; the query elements that an earlier fragment already covers, or that lie past d_head,
// are zeroed, which removes their products whatever the key registers hold.  Heads start at odd multiples of 4 bytes,
// so the 16-byte loads are only dword-aligned (global memory takes that).
typedef f16x8 __attribute__((aligned(4))) f16x8_a4;
// PIPE: one launch of a software-pipelined chain (common.h, JbPipe; multi-head engines).  As in the wide kernel below, what does
// not depend on the producer launch (c_attn) is asked for BEFORE the wait: the position (written by the previous step's sampler,
// which every launch but a step's first has seen complete through its stream predecessor), hence the key set, and the K fragments
// and value rows of a wave's FIRST tile with plain loads (rows of earlier positions: this launch's L1 is cold, L2 is coherent for
// write-through stores).  Behind the wait: the query, and position t's own row -- where it is a key it is the last one -- re-read
// write-through into the same registers; tiles beyond the first round (key sets of more than 16 keys per wave: the lyric
// cross-attention) are read write-through inside the loop.  The output leaves as 16-byte write-through pieces (eight neighbours'
// channels gathered in the vector ALU), then the workgroup publishes.  Round 6: the first form waited FIRST and read everything
// behind the wait with 2-byte stores at the end -- 9.0 us per phase of the 5b_lyrics step, inputs seen -> stores issued 6.6
// (profiles/r06c12_bench_engine_5b_stamps.log); same tiles per wave, same arithmetic: bit-identical.
template <int ND32, bool RAGGED, bool PIPE = false>
__global__ __launch_bounds__(512) void attn_decode_mfma_kernel(int func, const f16* __restrict__ q, int64_t ldq,
                                                               const f16* __restrict__ kc, const f16* __restrict__ vc, int cap,
                                                               f16* __restrict__ out, int64_t ldo, int n_head, int bc,
                                                               const int* __restrict__ t_dev, int d_head, JbPipe pipe) {
    const int d = RAGGED ? d_head : ND32 * 32;
    unsigned pipe_own = 0;
    if constexpr (PIPE) pipe_own = jb_pipe_own(pipe);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* s_ml = smem;                      // [nw][2]
    float* s_pw = smem + 2 * nw;             // [nw][16] probabilities of the wave's current tile
    float* s_o = s_pw + 16 * nw;             // [nw][d]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int n = blockIdx.x, h = blockIdx.y;
    const int S = n_head * d;
    // the query does not depend on the position: request it before the position is waited for
    const f16* qrow = q + (int64_t)n * ldq + h * d;
    f16x8 qf[ND32];
    int foff[ND32];                          // channel a fragment is read from (RAGGED: clamped into the head)
#pragma unroll
    for (int dt = 0; dt < ND32; ++dt) {
        const int cb = dt * 32 + g * 8;
        foff[dt] = RAGGED ? min(cb, d - 8) : cb;
        if constexpr (!PIPE) qf[dt] = RAGGED ? *reinterpret_cast<const f16x8_a4*>(qrow + foff[dt]) : ld_frag<f16>(qrow + cb);
    }
    jb_issue_fence();
    // (PIPE: read past the scalar cache like every other pipelined kernel's position -- not dependent on which launch's
    // dispatch-time invalidate happened to follow the sampler's write; BEFORE the wait: an attention launch is never a step's first)
    const int t = PIPE ? (int)jb_ld_word(reinterpret_cast<const unsigned*>(t_dev)) : *t_dev;
    const KeySet ks = decode_key_set(func, t, bc, cap);
    f16* o = out + (int64_t)n * ldo + h * d;
    const float scale = 1.0f / sqrtf(sqrtf((float)d));
    const float scale2 = scale * scale;
    const f16* kbase = kc + ((int64_t)n * cap) * S + h * d;
    const f16* vbase = vc + ((int64_t)n * cap) * S + h * d;
    const int64_t kv0 = ((int64_t)n * cap) * S + h * d;       // (PIPE: element offset of this (sample, head)'s rows in either cache)
    const int c0 = min(lane * 8, d - 8);     // this lane's value channels (lanes past d/8 compute unused duplicates)
    const int ntiles = (ks.count + 15) >> 4;

    // ---- requests of tile tt: K fragments of key (kbase_i + c), value rows kbase_i + 0..15 (plain, or write-through) ----
    auto request_tile = [&](f16x8 (&kf)[ND32], f16x8 (&vv)[16], int tt, auto sc1_tag) {
        constexpr bool SC1 = decltype(sc1_tag)::value;
        const int kbase_i = tt * 16;
        const int ki = min(kbase_i + c, max(ks.count - 1, 0));
        const f16* kr = kbase + (int64_t)(ks.start + ki * ks.stride) * S;
#pragma unroll
        for (int dt = 0; dt < ND32; ++dt) {
            if constexpr (SC1) kf[dt] = jb_ld_frag_sc1<f16>(kc, kv0 + (int64_t)(ks.start + ki * ks.stride) * S + foff[dt]);
            else kf[dt] = RAGGED ? *reinterpret_cast<const f16x8_a4*>(kr + foff[dt]) : ld_frag<f16>(kr + dt * 32 + g * 8);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int vi = min(kbase_i + k, max(ks.count - 1, 0));
            const f16* vr = vbase + (int64_t)(ks.start + vi * ks.stride) * S + c0;
            if constexpr (SC1) vv[k] = jb_ld_frag_sc1<f16>(vc, kv0 + (int64_t)(ks.start + vi * ks.stride) * S + c0);
            else vv[k] = RAGGED ? *reinterpret_cast<const f16x8_a4*>(vr) : ld_frag<f16>(vr);
        }
    };
    f16x8 kf0[PIPE ? ND32 : 1], vv0[PIPE ? 16 : 1];          // PIPE: the wave's first tile, in flight across the wait
    if constexpr (PIPE) {
        if (wave < ntiles) request_tile(kf0, vv0, wave, std::false_type{});
        jb_issue_fence();
        jb_pipe_wait(pipe, pipe_own, (int)(blockDim.x >> 6) - 1);      // (the last wave: it owns a tile only for the longest key sets)
#pragma unroll
        for (int dt = 0; dt < ND32; ++dt) qf[dt] = jb_ld_frag_sc1<f16>(q, (int64_t)n * ldq + h * d + foff[dt]);
        if (wave < ntiles) {
            const int last_pos = ks.start + (ks.count - 1) * ks.stride;      // == t when the query's own position is a key
            if (last_pos == t && wave == ntiles - 1) {                       // wave-uniform: the tile holds the producer's row
                const int kbase_i = wave * 16;
                const int64_t trow = kv0 + (int64_t)t * S;
                if (min(kbase_i + c, ks.count - 1) == ks.count - 1) {
#pragma unroll
                    for (int dt = 0; dt < ND32; ++dt) kf0[dt] = jb_ld_frag_sc1<f16>(kc, trow + foff[dt]);
                }
                const f16x8 fv = jb_ld_frag_sc1<f16>(vc, trow + c0);
#pragma unroll
                for (int k = 0; k < 16; ++k) vv0[k] = (kbase_i + k >= ks.count - 1) ? fv : vv0[k];
            }
        }
    }
    if constexpr (RAGGED) {
#pragma unroll
        for (int dt = 0; dt < ND32; ++dt) {
            const int cb = dt * 32 + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[dt][e] = (cb < d && foff[dt] + e >= cb) ? qf[dt][e] : (f16)0;
        }
    }
    if (ks.count == 0) {
        for (int i = threadIdx.x; i < d; i += blockDim.x) {
            if constexpr (PIPE) jb_st_sc1(out, (int64_t)n * ldo + h * d + i, (f16)0); else o[i] = (f16)0;
        }
        if constexpr (PIPE) jb_pipe_publish(pipe, pipe_own);
        return;
    }

    float m_w = -INFINITY, l_w = 0.f;
    float of[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) of[e] = 0.f;
    float* pw = s_pw + 16 * wave;

    auto tile_math = [&](const f16x8 (&kf)[ND32], const f16x8 (&vv)[16], int kbase_i) {
        jb_issue_fence_before_use(qf[0]);        // K fragments AND value rows are in flight before the QK^T chain starts
        // ---- scores of keys g*4 + r (identical in all 16 columns) ----
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < ND32; ++dt) sc = jb_mfma(kf[dt], qf[dt], sc);
        float pv[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = kbase_i + g * 4 + r < ks.count;
            // reference: w = matmul(q, k) (half result), w.mul_(scale*scale) (half), then .float()
            pv[r] = ok ? jb_round<f16>(jb_round<f16>(sc[r]) * scale2) : -INFINITY;
            mx = fmaxf(mx, pv[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_w, mx);                 // finite: every tile has at least one valid key
        const float alpha = expf(m_w - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pv[r] = (pv[r] == -INFINITY) ? 0.f : expf(pv[r] - m_new);
            ps += pv[r];
        }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l_w = l_w * alpha + ps;
        m_w = m_new;
        if (c == 0) *reinterpret_cast<f32x4*>(pw + g * 4) = f32x4{pv[0], pv[1], pv[2], pv[3]};
        // same wave: LDS accesses are issued in order, the compiler's lgkmcnt wait suffices
        f32x4 p4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p4[i] = *reinterpret_cast<const f32x4*>(pw + i * 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] *= alpha;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float pr = jb_round<f16>(p4[k >> 2][k & 3]);
#pragma unroll
            for (int e = 0; e < 8; ++e) of[e] += pr * (float)vv[k][e];
        }
    };
    if constexpr (PIPE) {
        if (wave < ntiles) tile_math(kf0, vv0, wave * 16);
        for (int tt = wave + nw; tt < ntiles; tt += nw) {      // (more than 16 keys per wave: the lyric cross-attention)
            f16x8 kf[ND32], vv[16];
            request_tile(kf, vv, tt, std::true_type{});
            tile_math(kf, vv, tt * 16);
        }
    } else {
        for (int tt = wave; tt < ntiles; tt += nw) {
            f16x8 kf[ND32], vv[16];
            request_tile(kf, vv, tt, std::false_type{});
            tile_math(kf, vv, tt * 16);
        }
    }
    if (lane == 0) { s_ml[2 * wave] = m_w; s_ml[2 * wave + 1] = l_w; }
    if (lane * 8 < d) {      // RAGGED: the last writing lane starts at d - 8 and re-writes channels its neighbour holds too (same values)
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[wave * d + c0 + e] = of[e];
    }
    __syncthreads();
    float m = -INFINITY;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, s_ml[2 * w]);
    float lsum = 0.f;
    for (int w = 0; w < nw; ++w) lsum += s_ml[2 * w + 1] * expf(s_ml[2 * w] - m);
    const float inv = 1.0f / lsum;
    if constexpr (PIPE) {
        // every thread takes a channel (blockDim.x >= d; those past d a duplicate); eight neighbours' channels leave as ONE
        // write-through store: lanes 1, 2, 3 of the thread's quad, then the next quad's (row_shl:4), all in the vector ALU; heads
        // start at multiples of 4 bytes and the last piece of a ragged head is 4, 8 or 12 bytes
        const int ch = min((int)threadIdx.x, d - 1);
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += s_o[w * d + ch] * expf(s_ml[2 * w] - m);
        const float v0 = (float)(f16)(a * inv);
        const float a1 = jb_dpp<0x55>(v0), a2 = jb_dpp<0xAA>(v0), a3 = jb_dpp<0xFF>(v0);
        const float a4 = jb_dpp<0x104>(v0), a5 = jb_dpp<0x104>(a1), a6 = jb_dpp<0x104>(a2), a7 = jb_dpp<0x104>(a3);
        if ((int)threadIdx.x < d && (threadIdx.x & 7) == 0) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const f16x8 o8 = {(f16)v0, (f16)a1, (f16)a2, (f16)a3, (f16)a4, (f16)a5, (f16)a6, (f16)a7};
            const u32x4 w4 = __builtin_bit_cast(u32x4, o8);
            const int off = (int)(((int64_t)n * ldo + h * d + (int)threadIdx.x) * 2), left = d - (int)threadIdx.x;
            if (left >= 8) __builtin_amdgcn_raw_buffer_store_b128(w4, jb_rsrc(out), off, 0, 16);
            else if (left >= 6) __builtin_amdgcn_raw_buffer_store_b96(u32x3{w4[0], w4[1], w4[2]}, jb_rsrc(out), off, 0, 16);
            else if (left >= 4) __builtin_amdgcn_raw_buffer_store_b64(u32x2{w4[0], w4[1]}, jb_rsrc(out), off, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b32(w4[0], jb_rsrc(out), off, 0, 16);
        }
        jb_pipe_publish(pipe, pipe_own);
        return;
    }
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += s_o[w * d + i] * expf(s_ml[2 * w] - m);
        o[i] = (f16)(a * inv);
    }
}

// Wide-value variant for single-head layers (the upsamplers): the cache row of a key holds v' = v·Wp (W channels), the
// value already carried through attn.c_proj, so sum_k p_k v'_k IS the projected attention output and this kernel writes
// the residual stream itself -- the attn.c_proj launch (4.7 us of the 28.6 us layer, a full kernel boundary plus a cold
// weight stream) is gone.  Grid (sample, slice): slice sl owns output channels [sl*d, (sl+1)*d) and reads the K rows
// (d channels) plus that slice of the v' rows, the same bytes per workgroup as the kernel above; the W/d slices of a
// sample recompute identical probabilities.
// PIPE: one launch of a software-pipelined chain (common.h, JbPipe): the query, the position and every cache row are read
// write-through after the producer launch (c_attn, which appends this position's k / v' rows) has completed; the residual
// rows are two launches old and the bias is constant: both are requested before the wait.
// QL ("lean"): the query fragments do not live in registers (15 x 4 of the 198-216 per lane) but in LDS -- every wave copies
// the 960-byte row into a region of its own (no workgroup barrier: a wave's LDS operations are ordered) and the QK^T chain
// reads its B operands from there.  At <= 168 registers per lane a workgroup shares a compute unit with a WAITING projection
// workgroup of a pipelined chain (8 waves at 88 registers): the fat form needs an EMPTY compute unit, which is why only one
// engine of a process could run pipelined launches, and why a plain chain next to a pipelined engine starved (HISTORY.md section 4.2).
template <int ND32, int PIPE = 0, bool QL = false>
__global__ __launch_bounds__(512) void attn_decode_wide_kernel(int func, const f16* __restrict__ q, int64_t ldq,
                                                               const f16* __restrict__ kc, const f16* __restrict__ vw, int cap,
                                                               const f16* __restrict__ res, int64_t ldr,
                                                               const float* __restrict__ bias, f16* __restrict__ out,
                                                               int64_t ldo, int W, int bc, const int* __restrict__ t_dev, JbPipe pipe) {
    constexpr int d = ND32 * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* s_ml = smem;                      // [nw][2]
    float* s_pw = smem + 2 * nw;             // [nw][16]
    float* s_o = s_pw + 16 * nw;             // [nw][d]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int n = blockIdx.x, sl = blockIdx.y;
    const f16* qrow = q + (int64_t)n * ldq;
    f16* s_q = reinterpret_cast<f16*>(s_o + nw * d) + wave * d;      // QL: [nw][d] halves, one row per wave
    unsigned pipe_own = 0;
    if constexpr (PIPE == 1) pipe_own = jb_pipe_own(pipe);
    f16x8 qf[QL ? 1 : ND32];
    if constexpr (!PIPE) {
        if constexpr (QL) {
            if (lane * 8 < d) *reinterpret_cast<f16x8*>(s_q + lane * 8) = ld_frag<f16>(qrow + lane * 8);
        } else {
#pragma unroll
            for (int dt = 0; dt < ND32; ++dt) qf[dt] = ld_frag<f16>(qrow + dt * 32 + g * 8);
        }
    }
    // epilogue operands of this thread's output channel (blockDim.x >= d): requested with the query, used at the end
    const int och = sl * d + min((int)threadIdx.x, d - 1);
    const float bias_e = bias[och];
    // (PIPE with operand-order hand-offs, common.h: the residual stream in and out is [k-tile][lane][8], 16 rows per block)
    const bool frag_res = PIPE != 0 && (pipe.frag & JB_FRAG_RES), frag_out = PIPE != 0 && (pipe.frag & JB_FRAG_OUT);
    const int64_t res_el = frag_res ? (int64_t)jb_frag_el(n, och) : (int64_t)n * ldr + och;
    // (a synchronised launch reads the residual rows -- two launches old -- behind the wait, write-through: in the three-stream
    // form of the step, engine.hip, their producer is not this stream's predecessor)
    f16 res_e = PIPE == 1 ? (f16)0 : res[res_el];
    jb_issue_fence();
    // PIPE: the position was written by the previous step's sampler.  An attention launch is never the first of a step, and in
    // the two-stream form its stream has seen that launch complete; the attention stream of the three-stream form runs AHEAD of
    // the sampler at the start of a step: its first launch waits for the previous step's last slot first (JB_PIPE_PRE_WAIT).
    // Read past the scalar cache either way; the key set is known before the wait for the producer.
    if constexpr (PIPE == 1) {
        if (pipe.frag & JB_PIPE_PRE_WAIT) jb_pipe_wait_previous_step(pipe, pipe_own, (int)(blockDim.x >> 6) - 1);
    }
    const int t = PIPE == 1 ? (int)jb_ld_word(reinterpret_cast<const unsigned*>(t_dev)) : *t_dev;
    const KeySet ks = decode_key_set(func, t, bc, cap);
    f16* o = out + (int64_t)n * ldo;
    const float scale = 1.0f / sqrtf(sqrtf((float)d));
    const float scale2 = scale * scale;
    const int64_t krow0 = (int64_t)n * cap;                      // first cache row of this sample
    const f16* kbase = kc + krow0 * d;
    const f16* vbase = vw + krow0 * W + sl * d;
    const int c0 = min(lane * 8, d - 8);
    const int ntiles = (ks.count + 15) >> 4;
    // One pass over 16-key tiles: requests (K fragments of key kbase_i + c, the 16 value rows), QK^T on MFMA, online softmax
    // of the tile, p.V on the vector ALU.
    float m_w = -INFINITY, l_w = 0.f;
    float of[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) of[e] = 0.f;
    float* pw = s_pw + 16 * wave;
    auto tile_math = [&](const f16x8 (&kf)[ND32], const f16x8 (&vv)[16], int kbase_i) {
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < ND32; ++dt) {
            if constexpr (QL) sc = jb_mfma(kf[dt], *reinterpret_cast<const f16x8*>(s_q + dt * 32 + g * 8), sc);
            else sc = jb_mfma(kf[dt], qf[dt], sc);
        }
        float pv[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = kbase_i + g * 4 + r < ks.count;
            pv[r] = ok ? jb_round<f16>(jb_round<f16>(sc[r]) * scale2) : -INFINITY;
            mx = fmaxf(mx, pv[r]);
        }
        mx = jb_rows_reduce(mx, [](float a, float b) { return fmaxf(a, b); });
        const float m_new = fmaxf(m_w, mx);
        const float alpha = expf(m_w - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pv[r] = (pv[r] == -INFINITY) ? 0.f : expf(pv[r] - m_new);
            ps += pv[r];
        }
        ps = jb_rows_reduce(ps, [](float a, float b) { return a + b; });
        l_w = l_w * alpha + ps;
        m_w = m_new;
        if (c == 0) *reinterpret_cast<f32x4*>(pw + g * 4) = f32x4{pv[0], pv[1], pv[2], pv[3]};
        f32x4 p4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p4[i] = *reinterpret_cast<const f32x4*>(pw + i * 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] *= alpha;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float pr = jb_round<f16>(p4[k >> 2][k & 3]);
#pragma unroll
            for (int e = 0; e < 8; ++e) of[e] += pr * (float)vv[k][e];
        }
    };

    if constexpr (PIPE) {
        // One tile per wave (key sets of <= 16 * waves keys; jb_engine_pipeline checks it).  The cache rows of EARLIER
        // positions were written in earlier steps: requested BEFORE the wait with plain loads (this launch's L1 is cold,
        // L2 is coherent for write-through stores).  Position t itself is being written by the producer launch: where it is
        // a key (the last one), its row is re-read write-through after the wait, straight into the same registers.
        f16x8 kf[ND32], vv[16];
        const int kbase_i = wave * 16;
        const int ki = min(kbase_i + c, max(ks.count - 1, 0));
        if (wave < ntiles) {
            const int64_t kpos = ks.start + ki * ks.stride;
#pragma unroll
            for (int dt = 0; dt < ND32; ++dt) kf[dt] = ld_frag<f16>(kbase + kpos * d + g * 8 + dt * 32);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t vpos = ks.start + min(kbase_i + k, ks.count - 1) * ks.stride;
                vv[k] = ld_frag<f16>(vbase + vpos * W + c0);
            }
        }
        jb_issue_fence();
        if constexpr (PIPE == 1) jb_pipe_wait(pipe, pipe_own, (int)(blockDim.x >> 6) - 1);      // (the last wave: usually nothing of its own in flight)
        if constexpr (PIPE == 1) res_e = jb_ld_sc1(res, res_el);
        if constexpr (QL) {
            if (lane * 8 < d) *reinterpret_cast<f16x8*>(s_q + lane * 8) = jb_ld_frag_sc1<f16>(q, (int64_t)n * ldq + lane * 8);
        } else {
#pragma unroll
            for (int dt = 0; dt < ND32; ++dt) qf[dt] = jb_ld_frag_sc1<f16>(q, (int64_t)n * ldq + dt * 32 + g * 8);
        }
        if (ks.count == 0) {      // zero rows -> attention output 0 -> c_proj gives its bias
            if (threadIdx.x < d) jb_st_sc1(out, frag_out ? (int64_t)jb_frag_el(n, och) : (int64_t)n * ldo + och,
                                           (f16)jb_round<f16>((float)res_e + jb_round<f16>(jb_round<f16>(bias_e))));
            if constexpr (PIPE == 1) jb_pipe_publish(pipe, pipe_own);
            return;
        }
        if (wave < ntiles) {
            const int last_pos = ks.start + (ks.count - 1) * ks.stride;      // == t when the query's own position is a key
            if (last_pos == t && wave == ntiles - 1) {                       // wave-uniform
                const int64_t trow = krow0 + t;
                if (ki == ks.count - 1) {
#pragma unroll
                    for (int dt = 0; dt < ND32; ++dt) kf[dt] = jb_ld_frag_sc1<f16>(kc, trow * d + g * 8 + dt * 32);
                }
                const f16x8 fv = jb_ld_frag_sc1<f16>(vw, trow * W + sl * d + c0);
#pragma unroll
                for (int k = 0; k < 16; ++k) vv[k] = (kbase_i + k >= ks.count - 1) ? fv : vv[k];
            }
            if constexpr (QL) jb_issue_fence_before_use(kf[0]);
            else jb_issue_fence_before_use(qf[0]);
            JB_SEG_VM(pipe, 5);
            tile_math(kf, vv, kbase_i);
        }
    } else {
        if (ks.count == 0) {      // zero rows -> attention output 0 -> c_proj gives its bias
            if (threadIdx.x < d) o[och] = (f16)jb_round<f16>((float)res_e + jb_round<f16>(jb_round<f16>(bias_e)));
            return;
        }
        for (int tt = wave; tt < ntiles; tt += nw) {
            const int kbase_i = tt * 16;
            const int ki = min(kbase_i + c, ks.count - 1);
            const int64_t kpos = ks.start + ki * ks.stride;
            f16x8 kf[ND32];
#pragma unroll
            for (int dt = 0; dt < ND32; ++dt) kf[dt] = ld_frag<f16>(kbase + kpos * d + g * 8 + dt * 32);
            f16x8 vv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t vpos = ks.start + min(kbase_i + k, ks.count - 1) * ks.stride;
                vv[k] = ld_frag<f16>(vbase + vpos * W + c0);
            }
            if constexpr (QL) jb_issue_fence_before_use(kf[0]);
            else jb_issue_fence_before_use(qf[0]);
            tile_math(kf, vv, kbase_i);
        }
    }
    if (lane == 0) { s_ml[2 * wave] = m_w; s_ml[2 * wave + 1] = l_w; }
    if (lane * 8 < d) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[wave * d + lane * 8 + e] = of[e];
    }
    if constexpr (PIPE) JB_SEG_LGKM(pipe, 6);
    __syncthreads();
    if constexpr (PIPE) JB_SEG(pipe, 7);
    float m = -INFINITY;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, s_ml[2 * w]);
    // the waves' weights exp(m_w - m), once (they scale the sums AND the channel partials; launches have <= 8 waves): lane l
    // evaluates wave (l & 7)'s, the eight values are read back as scalars -- one expf per thread on the epilogue's dependent chain
    // instead of eight (the same function of the same operands: bit-identical)
    float ew[8];
    float lsum = 0.f;
    const int wl = lane & 7;
    const float e_l = wl < nw ? expf(s_ml[2 * wl] - m) : 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        ew[w] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_l), w));
        if (w < nw) lsum += s_ml[2 * w + 1] * ew[w];
    }
    const float inv = 1.0f / lsum;
    if constexpr (PIPE) {
        // every thread takes a channel (those past d a duplicate)
        const int ch = min((int)threadIdx.x, d - 1);
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w)
            if (w < nw) a += s_o[w * d + ch] * ew[w];
        const float cp = jb_round<f16>(a * inv + jb_round<f16>(bias_e));
        const float v1 = jb_round<f16>((float)res_e + cp);
        // eight neighbours leave as ONE 16-byte write-through store: lanes 1, 2, 3 of the thread's quad, then the next quad's
        // (row_shl:4), all in the vector ALU; eight consecutive channels are contiguous in either layout (d, ldo multiples of 8)
        const float a1 = jb_dpp<0x55>(v1), a2 = jb_dpp<0xAA>(v1), a3 = jb_dpp<0xFF>(v1);
        const float a4 = jb_dpp<0x104>(v1), a5 = jb_dpp<0x104>(a1), a6 = jb_dpp<0x104>(a2), a7 = jb_dpp<0x104>(a3);
        if ((int)threadIdx.x < d && (threadIdx.x & 7) == 0) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const f16x8 o8 = {(f16)v1, (f16)a1, (f16)a2, (f16)a3, (f16)a4, (f16)a5, (f16)a6, (f16)a7};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), jb_rsrc(out),
                                                   (int)((frag_out ? (int64_t)jb_frag_el(n, och) : (int64_t)n * ldo + och) * 2), 0, 16);
        }
        if constexpr (PIPE == 1) jb_pipe_publish(pipe, pipe_own);
        return;
    }
    if (threadIdx.x < d) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w)
            if (w < nw) a += s_o[w * d + threadIdx.x] * ew[w];
        // attn.c_proj's epilogue (bias, round) and the residual add of the block, as jb_gemv does them
        const float cp = jb_round<f16>(a * inv + jb_round<f16>(bias_e));
        o[och] = (f16)jb_round<f16>((float)res_e + cp);
    }
}

// Key-split variant of the kernel above (the decode step's default in fp16): grid (sample, head, split).  One CU pulls
// ~100 GB/s, so a single workgroup reading a whole 128-key K and V set (245 KB at d = 480) spends 2-3 us on its own L1
// fill -- longer than the rest of the chip needs for everything.  Here split s of n_parts takes the 16-key tiles
// tt = s*nw + w + i*n_parts*nw (w = wave), keeps its own online-softmax state and writes it out UNMERGED:
//     parts[n][s][h*d + ch] = sum_k p_k v_k[ch] / l_s   (f16; zeros when the split saw no key)
//     ml[n][h][s]           = (m_s, l_s)                (running max and sum of exp(score - m_s); (-inf, 0) when empty)
// The log-sum-exp merge over the splits happens in the operand load of attn.c_proj (gemv_merge_kernel, gemm.hip), so
// the splits never wait for each other and there is no extra launch.
template <int ND32>
__global__ __launch_bounds__(512) void attn_decode_split_kernel(int func, const f16* __restrict__ q, int64_t ldq,
                                                                const f16* __restrict__ kc, const f16* __restrict__ vc, int cap,
                                                                f16* __restrict__ parts, float* __restrict__ ml, int n_head,
                                                                int bc, const int* __restrict__ t_dev, int n_parts) {
    constexpr int d = ND32 * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* s_ml = smem;                      // [nw][2]
    float* s_pw = smem + 2 * nw;             // [nw][16] probabilities of the wave's current tile
    float* s_o = s_pw + 16 * nw;             // [nw][d]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int n = blockIdx.x, h = blockIdx.y, split = blockIdx.z;
    const int S = n_head * d;
    const f16* qrow = q + (int64_t)n * ldq + h * d;
    f16x8 qf[ND32];
#pragma unroll
    for (int dt = 0; dt < ND32; ++dt) qf[dt] = ld_frag<f16>(qrow + dt * 32 + g * 8);
    jb_issue_fence();
    const int t = *t_dev;
    const KeySet ks = decode_key_set(func, t, bc, cap);
    f16* o = parts + ((int64_t)n * n_parts + split) * S + h * d;
    float* oml = ml + (((int64_t)n * n_head + h) * n_parts + split) * 2;
    const int ntiles = (ks.count + 15) >> 4;
    if (split * nw >= ntiles) {              // nothing for this split (early positions, prev_block in block 0)
        for (int i = threadIdx.x; i < d; i += blockDim.x) o[i] = (f16)0;
        if (threadIdx.x == 0) { oml[0] = -INFINITY; oml[1] = 0.f; }
        return;
    }
    const float scale = 1.0f / sqrtf(sqrtf((float)d));
    const float scale2 = scale * scale;
    const f16* kbase = kc + ((int64_t)n * cap) * S + h * d;
    const f16* vbase = vc + ((int64_t)n * cap) * S + h * d;
    const int c0 = min(lane * 8, d - 8);     // this lane's value channels (lanes past d/8 compute unused duplicates)

    float m_w = -INFINITY, l_w = 0.f;
    float of[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) of[e] = 0.f;
    float* pw = s_pw + 16 * wave;

    for (int tt = split * nw + wave; tt < ntiles; tt += n_parts * nw) {
        const int kbase_i = tt * 16;
        const int ki = min(kbase_i + c, ks.count - 1);
        const f16* kr = kbase + (int64_t)(ks.start + ki * ks.stride) * S + g * 8;
        f16x8 kf[ND32];
#pragma unroll
        for (int dt = 0; dt < ND32; ++dt) kf[dt] = ld_frag<f16>(kr + dt * 32);
        f16x8 vv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int vi = min(kbase_i + k, ks.count - 1);
            vv[k] = ld_frag<f16>(vbase + (int64_t)(ks.start + vi * ks.stride) * S + c0);
        }
        jb_issue_fence_before_use(qf[0]);        // K fragments AND value rows are in flight before the QK^T chain starts
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < ND32; ++dt) sc = jb_mfma(kf[dt], qf[dt], sc);
        float pv[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = kbase_i + g * 4 + r < ks.count;
            // reference: w = matmul(q, k) (half result), w.mul_(scale*scale) (half), then .float()
            pv[r] = ok ? jb_round<f16>(jb_round<f16>(sc[r]) * scale2) : -INFINITY;
            mx = fmaxf(mx, pv[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_w, mx);                 // finite: every tile has at least one valid key
        const float alpha = expf(m_w - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pv[r] = (pv[r] == -INFINITY) ? 0.f : expf(pv[r] - m_new);
            ps += pv[r];
        }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l_w = l_w * alpha + ps;
        m_w = m_new;
        if (c == 0) *reinterpret_cast<f32x4*>(pw + g * 4) = f32x4{pv[0], pv[1], pv[2], pv[3]};
        f32x4 p4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p4[i] = *reinterpret_cast<const f32x4*>(pw + i * 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] *= alpha;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float pr = jb_round<f16>(p4[k >> 2][k & 3]);
#pragma unroll
            for (int e = 0; e < 8; ++e) of[e] += pr * (float)vv[k][e];
        }
    }
    if (lane == 0) { s_ml[2 * wave] = m_w; s_ml[2 * wave + 1] = l_w; }
    if (lane * 8 < d) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[wave * d + lane * 8 + e] = of[e];
    }
    __syncthreads();
    float m = -INFINITY;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, s_ml[2 * w]);          // finite: wave 0 of this split had a tile
    float lsum = 0.f;
    for (int w = 0; w < nw; ++w) lsum += s_ml[2 * w + 1] > 0.f ? s_ml[2 * w + 1] * expf(s_ml[2 * w] - m) : 0.f;
    const float inv = 1.0f / lsum;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += s_ml[2 * w + 1] > 0.f ? s_o[w * d + i] * expf(s_ml[2 * w] - m) : 0.f;
        o[i] = (f16)(a * inv);
    }
    if (threadIdx.x == 0) { oml[0] = m; oml[1] = lsum; }
}

// launch shape of the decode attention: threads per (sample, head) workgroup and key/value row pairs in flight per wave
static int g_dec_threads = 512, g_dec_kb = 4, g_dec_mfma = 1;
extern "C" void jb_tune_attn_decode(int threads, int kb) {
    if (threads > 0) g_dec_threads = threads;
    if (kb > 0) g_dec_kb = kb;
    g_dec_mfma = kb >= 0;          // kb < 0 selects the generic (vector-ALU QK^T) kernel for every dtype
}
// key-split decode attention: most splits per (sample, head), waves per split workgroup for short key sets, and the
// key-set size from which a layer is split at all.  Measured on MI355X (upsampler, N = 16, key sets of <= 128 keys):
// 2.11 ms per step unsplit vs 2.37-2.41 ms split 2..4 ways -- up to 128 keys are ONE pass of the 8-wave kernel (a wave
// per 16-key tile), so a split shortens nothing on the critical path and the merge in attn.c_proj costs ~1 us of VALU.
// Long key sets (dense 6528 keys: 51 tile passes per wave; prime 448 keys: 4) are where the split pays.
static int g_split_max = 4, g_split_waves = 2, g_split_min_keys = 129;
extern "C" void jb_tune_attn_decode_split(int max_parts, int waves) {
    if (max_parts >= 1 && max_parts <= 4) g_split_max = max_parts;
    if (waves >= 1 && waves <= 8) g_split_waves = waves;
}
extern "C" void jb_tune_attn_decode_split_min_keys(int min_keys) { if (min_keys >= 1) g_split_min_keys = min_keys; }
static inline bool split_d_ok(int d_head) {
    const int nd = d_head / 32;
    return d_head % 32 == 0 && (nd == 1 || nd == 2 || nd == 4 || nd == 8 || nd == 15 || nd == 16);
}
// Splits per (sample, head) for a layer whose key set never exceeds max_keys; 0 = shape not supported (use jb_attn_decode).
extern "C" int jb_attn_decode_split_parts(int dtype, int d_head, int max_keys) {
    if (dtype != JB_F16 || !split_d_ok(d_head) || max_keys < g_split_min_keys) return 0;
    const int tiles = (max_keys + 15) / 16;
    int parts = (tiles + g_split_waves - 1) / g_split_waves;
    return parts < 1 ? 1 : (parts > g_split_max ? g_split_max : parts);
}

extern "C" int jb_attn_decode_split(int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache,
                                    int cache_cap, void* parts, float* ml, int n_batch, int n_head, int d_head,
                                    int block_ctx, const int* t_dev, int max_keys, int n_parts, void* stream) {
    JB_REQUIRE(q && kcache && vcache && parts && ml && t_dev, "null pointer");
    JB_REQUIRE(n_batch > 0 && n_head > 0 && max_keys > 0 && n_parts >= 1 && n_parts <= 4, "bad dims");
    JB_REQUIRE(split_d_ok(d_head) && ldq % 8 == 0, "d_head must be 32 x {1,2,4,8,15,16} and q rows 16-byte aligned");
    JB_REQUIRE(attn_func == 0 || attn_func == 7 || attn_func == 6 || block_ctx > 0, "block_ctx required");
    const int tiles = (max_keys + 15) / 16;
    int nw = (tiles + n_parts - 1) / n_parts;              // waves per split so that one pass covers the longest key set
    nw = nw < 1 ? 1 : (nw > 8 ? 8 : nw);
    if (nw < g_split_waves && tiles >= g_split_waves) nw = g_split_waves;
    dim3 grid(n_batch, n_head, n_parts);
    const size_t lds = (size_t)(2 * nw + 16 * nw + nw * d_head) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define JB_LAUNCH_DECS(ND)                                                                                          \
    attn_decode_split_kernel<ND><<<grid, nw * 64, lds, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,        \
                                                          (const f16*)vcache, cache_cap, (f16*)parts, ml, n_head,    \
                                                          block_ctx, t_dev, n_parts)
    switch (d_head / 32) {
        case 1: JB_LAUNCH_DECS(1); break;
        case 2: JB_LAUNCH_DECS(2); break;
        case 4: JB_LAUNCH_DECS(4); break;
        case 8: JB_LAUNCH_DECS(8); break;
        case 15: JB_LAUNCH_DECS(15); break;
        default: JB_LAUNCH_DECS(16); break;
    }
#undef JB_LAUNCH_DECS
    JB_CHECK_LAUNCH();
    return JB_OK;
}

extern "C" int jb_attn_decode_wide_supported(int attn_func, int d_head, int width, int block_ctx, int max_len) {
    (void)max_len;
    const bool func_ok = attn_func == 0 || attn_func == 7 || ((attn_func == 1 || attn_func == 2 || attn_func == 3) && block_ctx > 0);
    return func_ok && split_d_ok(d_head) && d_head <= 512 && width > 0 && width % d_head == 0 && width % 8 == 0;
}

extern "C" int jb_attn_decode_wide(int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache_w,
                                   int cache_cap, const void* res, int64_t ldr, const float* bias, void* x_out, int64_t ldo,
                                   int n_batch, int d_head, int width, int block_ctx, const int* t_dev, int max_len,
                                   void* stream) {
    return jb_attn_decode_wide_impl(attn_func, q, ldq, kcache, vcache_w, cache_cap, res, ldr, bias, x_out, ldo, n_batch, d_head,
                                    width, block_ctx, t_dev, max_len, nullptr, stream);
}

// 480-channel heads (the 1b upsamplers): 1 = the lean form of the kernel (query through LDS, <= 168 registers per lane: shares a
// compute unit with a waiting projection workgroup of a pipelined chain; round 5 ran two pipelined engines side by side on it:
// 2.02 against 2.09 ms per step, no gain for the job, the two-engine admission is gone since round 6), 0 (default) = the fat
// form, 0.4 us per launch faster for an engine that has the GPU to itself (1.557 against 1.587 ms per upsampler step,
// profiles/r05_bench_engine_lean_vs_fat_attention.log).  jb_tune_attn_decode_wide_lean: kept as a measured variant, bit-equal
// to the fat form (tests/test_hip_kernels.py).
static int g_wide_lean = 0;
extern "C" void jb_tune_attn_decode_wide_lean(int on) { g_wide_lean = on ? 1 : 0; }

int jb_attn_decode_wide_impl(int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache_w, int cache_cap,
                             const void* res, int64_t ldr, const float* bias, void* x_out, int64_t ldo, int n_batch, int d_head,
                             int width, int block_ctx, const int* t_dev, int max_len, const JbPipe* pipe, void* stream) {
    JB_REQUIRE(q && kcache && vcache_w && res && bias && x_out && t_dev, "null pointer");
    JB_REQUIRE(n_batch > 0 && max_len > 0, "bad dims");
    JB_REQUIRE(jb_attn_decode_wide_supported(attn_func, d_head, width, block_ctx, max_len) && ldq % 8 == 0,
               "wide-value attention: self-attention pattern, d_head = 32 x {1,2,4,8,15,16}, width a multiple of d_head");
    const int nw = 8;
    const bool lean = g_wide_lean && d_head == 480;
    const size_t lds = (size_t)(2 * nw + 16 * nw + nw * d_head) * sizeof(float) + (lean ? (size_t)nw * d_head * sizeof(f16) : 0);
    dim3 grid(n_batch, width / d_head);
    hipStream_t s = (hipStream_t)stream;
    const JbPipe nopipe{nullptr, nullptr, nullptr, -1, -1, 0, nullptr};
    if (pipe) {
        JB_REQUIRE((d_head == 480 || d_head == 256) && (int64_t)n_batch * cache_cap * width < (1ll << 30) && ldo % 8 == 0 &&
                       ((uintptr_t)x_out % 16) == 0,
                   "a pipelined launch of the wide-value attention takes d_head = 480 or 256, caches below 2 GiB and 16-byte aligned output rows");
        JB_REQUIRE(pipe->slot < 0 || pipe->proto < 1 || (int64_t)grid.x * grid.y >= 8, "completion protocol 1 needs launches of >= 8 workgroups");
#define JB_LAUNCH_DECW_PIPE(FORM, LEAN)                                                                                \
    attn_decode_wide_kernel<15, FORM, LEAN><<<grid, nw * 64, lds, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,   \
                                                                     (const f16*)vcache_w, cache_cap, (const f16*)res, ldr, \
                                                                     bias, (f16*)x_out, ldo, width, block_ctx, t_dev, *pipe)
        // (FORM 1: synchronised through the completion words; 2 = JB_PIPE_NO_SYNC: the same kernel form as a launch of a plain chain)
        if (d_head == 256) {      // (small_prior / small_upsampler: one head of 256 channels on 1024)
            if (pipe->slot >= 0)
                attn_decode_wide_kernel<8, 1, false><<<grid, nw * 64, lds, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache, (const f16*)vcache_w,
                                                                                cache_cap, (const f16*)res, ldr, bias, (f16*)x_out, ldo, width, block_ctx, t_dev, *pipe);
            else
                attn_decode_wide_kernel<8, 2, false><<<grid, nw * 64, lds, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache, (const f16*)vcache_w,
                                                                                cache_cap, (const f16*)res, ldr, bias, (f16*)x_out, ldo, width, block_ctx, t_dev, *pipe);
        } else if (pipe->slot >= 0) { if (lean) JB_LAUNCH_DECW_PIPE(1, true); else JB_LAUNCH_DECW_PIPE(1, false); }
        else { if (lean) JB_LAUNCH_DECW_PIPE(2, true); else JB_LAUNCH_DECW_PIPE(2, false); }
#undef JB_LAUNCH_DECW_PIPE
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
    if (lean) {
        attn_decode_wide_kernel<15, 0, true><<<grid, nw * 64, lds, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,
                                                                          (const f16*)vcache_w, cache_cap, (const f16*)res, ldr, bias,
                                                                          (f16*)x_out, ldo, width, block_ctx, t_dev, nopipe);
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
#define JB_LAUNCH_DECW(ND)                                                                                          \
    attn_decode_wide_kernel<ND><<<grid, nw * 64, lds, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,          \
                                                         (const f16*)vcache_w, cache_cap, (const f16*)res, ldr, bias, \
                                                         (f16*)x_out, ldo, width, block_ctx, t_dev, nopipe)
    switch (d_head / 32) {
        case 1: JB_LAUNCH_DECW(1); break;
        case 2: JB_LAUNCH_DECW(2); break;
        case 4: JB_LAUNCH_DECW(4); break;
        case 8: JB_LAUNCH_DECW(8); break;
        case 15: JB_LAUNCH_DECW(15); break;
        default: JB_LAUNCH_DECW(16); break;
    }
#undef JB_LAUNCH_DECW
    JB_CHECK_LAUNCH();
    return JB_OK;
}

extern "C" int jb_attn_decode(int dtype, int attn_func, const void* q, int64_t ldq, const void* kcache,
                              const void* vcache, int cache_cap, void* out, int64_t ldo, int n_batch, int n_head,
                              int d_head, int block_ctx, const int* t_dev, int max_len, void* stream) {
    return jb_attn_decode_impl(dtype, attn_func, q, ldq, kcache, vcache, cache_cap, out, ldo, n_batch, n_head, d_head, block_ctx,
                               t_dev, max_len, nullptr, stream);
}

// 1 if a pipelined launch of the decode attention exists for this head size (the MFMA kernels' instantiations below)
int jb_attn_decode_pipe_supported(int dtype, int d_head, int ldq, int ldo, int S) {
    if (dtype != JB_F16 || !g_dec_mfma) return 0;
    if (d_head % 32 != 0) return d_head % 2 == 0 && d_head > 128 && d_head < 160 && ldq % 2 == 0 && ldo % 2 == 0;
    return (d_head == 256 || d_head == 512) && ldq % 8 == 0 && S % 8 == 0;
}

int jb_attn_decode_impl(int dtype, int attn_func, const void* q, int64_t ldq, const void* kcache,
                        const void* vcache, int cache_cap, void* out, int64_t ldo, int n_batch, int n_head,
                        int d_head, int block_ctx, const int* t_dev, int max_len, const JbPipe* pipe, void* stream) {
    JB_REQUIRE(q && kcache && vcache && out && t_dev, "null pointer");
    JB_REQUIRE(dtype == JB_F32 || dtype == JB_F16, "bad dtype");
    JB_REQUIRE(n_batch > 0 && n_head > 0 && d_head > 0 && max_len > 0, "bad dims");
    JB_REQUIRE(attn_func == 0 || attn_func == 7 || attn_func == 6 || block_ctx > 0, "block_ctx required");
    const int E = dtype == JB_F16 ? 8 : 4;
    const int nch = (d_head + 64 * E - 1) / (64 * E);
    if (nch > 2) JB_UNSUPPORTED("d_head too large for the decode attention kernel");
    // long (dense) rows get 16 waves to keep more row loads in flight; short patterns 4
    const int threads = g_dec_threads;
    const int nw = threads / 64;
    size_t lds = (size_t)(2 * nw + nw * d_head) * sizeof(float);
    (void)max_len;
    dim3 grid(n_batch, n_head);
    hipStream_t s = (hipStream_t)stream;
    const JbPipe nopipe{nullptr, nullptr, nullptr, -1, -1, 0, nullptr};
    JB_REQUIRE(!pipe || pipe->proto < 1 || (int64_t)n_batch * n_head >= 8, "completion protocol 1 needs launches of >= 8 workgroups");
    JB_REQUIRE(!pipe || (jb_attn_decode_pipe_supported(dtype, d_head, (int)ldq, (int)ldo, n_head * d_head) &&
                         (int64_t)n_batch * cache_cap * n_head * d_head < (1ll << 30)),
               "a pipelined launch of the decode attention takes fp16 heads of 150 (ragged), 256 or 512 channels and caches below 2 GiB");
    if (dtype == JB_F16 && g_dec_mfma && d_head % 32 != 0 && d_head % 2 == 0 && d_head > 128 && d_head < 160 && ldq % 2 == 0 &&
        ldo % 2 == 0) {
        // ragged head size in 5 k-tiles (5b_lyrics: 150 channels per head)
        const int nwm = 8;
        size_t ldsm = (size_t)(2 * nwm + 16 * nwm + nwm * d_head) * sizeof(float);
        if (pipe)
            attn_decode_mfma_kernel<5, true, true><<<grid, nwm * 64, ldsm, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,
                                                                              (const f16*)vcache, cache_cap, (f16*)out, ldo, n_head,
                                                                              block_ctx, t_dev, d_head, *pipe);
        else
            attn_decode_mfma_kernel<5, true><<<grid, nwm * 64, ldsm, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,
                                                                        (const f16*)vcache, cache_cap, (f16*)out, ldo, n_head,
                                                                        block_ctx, t_dev, d_head, nopipe);
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
    if (dtype == JB_F16 && g_dec_mfma && d_head % 32 == 0 && d_head <= 512 && ldq % 8 == 0 && (n_head * d_head) % 8 == 0) {
        const int nwm = 8;
        size_t ldsm = (size_t)(2 * nwm + 16 * nwm + nwm * d_head) * sizeof(float);
        if (pipe) {
            if (d_head == 256)
                attn_decode_mfma_kernel<8, false, true><<<grid, nwm * 64, ldsm, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,
                                                                                   (const f16*)vcache, cache_cap, (f16*)out, ldo,
                                                                                   n_head, block_ctx, t_dev, d_head, *pipe);
            else if (d_head == 512)
                attn_decode_mfma_kernel<16, false, true><<<grid, nwm * 64, ldsm, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache,
                                                                                    (const f16*)vcache, cache_cap, (f16*)out, ldo,
                                                                                    n_head, block_ctx, t_dev, d_head, *pipe);
            else JB_UNSUPPORTED("a pipelined launch of the decode attention takes heads of 150 (ragged), 256 or 512 channels");
            JB_CHECK_LAUNCH();
            return JB_OK;
        }
#define JB_LAUNCH_DECM(ND)                                                                                      \
    attn_decode_mfma_kernel<ND, false><<<grid, nwm * 64, ldsm, s>>>(attn_func, (const f16*)q, ldq, (const f16*)kcache, \
                                                           (const f16*)vcache, cache_cap, (f16*)out, ldo, n_head, \
                                                           block_ctx, t_dev, d_head, nopipe)
        switch (d_head / 32) {
            case 1: JB_LAUNCH_DECM(1); break;
            case 2: JB_LAUNCH_DECM(2); break;
            case 4: JB_LAUNCH_DECM(4); break;
            case 8: JB_LAUNCH_DECM(8); break;
            case 15: JB_LAUNCH_DECM(15); break;
            case 16: JB_LAUNCH_DECM(16); break;
            default: goto generic;
        }
#undef JB_LAUNCH_DECM
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
generic:
#define JB_LAUNCH_DEC(T, NCH)                                                                                   \
    do {                                                                                                        \
        if (g_dec_kb == 8 && NCH == 1)                                                                          \
            attn_decode_kernel<T, NCH, 8><<<grid, threads, lds, s>>>(attn_func, (const T*)q, ldq, (const T*)kcache, \
                                                                 (const T*)vcache, cache_cap, (T*)out, ldo, n_head, \
                                                                 d_head, block_ctx, t_dev);                     \
        else if (g_dec_kb == 2)                                                                                 \
            attn_decode_kernel<T, NCH, 2><<<grid, threads, lds, s>>>(attn_func, (const T*)q, ldq, (const T*)kcache, \
                                                                 (const T*)vcache, cache_cap, (T*)out, ldo, n_head, \
                                                                 d_head, block_ctx, t_dev);                     \
        else                                                                                                    \
            attn_decode_kernel<T, NCH, 4><<<grid, threads, lds, s>>>(attn_func, (const T*)q, ldq, (const T*)kcache, \
                                                                 (const T*)vcache, cache_cap, (T*)out, ldo, n_head, \
                                                                 d_head, block_ctx, t_dev);                     \
    } while (0)
    if (dtype == JB_F16) { if (nch == 1) JB_LAUNCH_DEC(f16, 1); else JB_LAUNCH_DEC(f16, 2); }
    else { if (nch == 1) JB_LAUNCH_DEC(float, 1); else JB_LAUNCH_DEC(float, 2); }
#undef JB_LAUNCH_DEC
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// ------------------------------------------------------------------------------------------------
// Prefill (q_l > 1): one wave per tile of 16 queries, flash-style online softmax over KT-key steps.
//
//   S^T = K Q^T   on MFMA with A = K tile [16 keys x d] (LDS-staged), B = Q^T (LDS-staged):
//                 lane l then holds, for query col l&15, the scores of keys (l>>4)*4 + r -- so the
//                 row reductions of the softmax are 3 in-lane ops + 2 cross-lane shuffles.
//   O^T += V^T P^T on MFMA with A = V^T [16 channels x KT keys] read from the LDS-staged V tile and
//                 B = P^T, whose fragment is exactly the registers the lane already holds once the
//                 MFMA k-slot <-> key assignment is chosen as slot (g, e) <-> key g*4 + (e&3) + 16*(e>>2).
//
// Query tiles: 16 consecutive positions (dense / block / prev / prime) or 16 consecutive members of one
// residue class mod block_ctx (transpose); the per-element mask is the closed-form rule of the pattern.
template <typename T, int ND16>
__global__ __launch_bounds__(64) void attn_prefill_kernel(int func, const T* __restrict__ q, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, int cap, T* __restrict__ out,
                                                          int n_head, int d, int bc, int t0, int nq, int tiles_per_class) {
    using V = typename Frag<T>::vec;
    constexpr int E = Frag<T>::E, KT = Frag<T>::KT;
    constexpr int DP = ND16 * 16;            // padded head dim
    constexpr int LDR = DP + E;              // LDS row stride (elements), breaks the power-of-two stride
    constexpr int NG = KT / 16;              // 16-key score groups per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sQ = reinterpret_cast<T*>(smem_raw);  // [16][LDR]
    T* sK = sQ + 16 * LDR;                   // [KT][LDR]
    T* sV = sK + KT * LDR;                   // [KT][LDR]
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int tile = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int S = n_head * d;

    // ---- which queries / keys does this tile cover -------------------------------------------
    int qpos0, qstep, nvalid;                // query r sits at position qpos0 + r*qstep, r < nvalid
    int kstart, kstep, nkeys;                // candidate key u sits at position kstart + u*kstep
    if (func == JB_ATTN_TRANSPOSE_BLOCK) {
        int cls = tile / tiles_per_class, it = tile % tiles_per_class;
        int first = t0 + ((cls - t0 % bc) + bc) % bc;        // first position >= t0 in class cls
        qpos0 = first + it * 16 * bc; qstep = bc;
        int last = t0 + nq - 1;
        nvalid = qpos0 > last ? 0 : min(16, (last - qpos0) / bc + 1);
        kstart = cls; kstep = bc;
        nkeys = nvalid > 0 ? (qpos0 + (nvalid - 1) * bc) / bc + 1 : 0;
    } else {
        qpos0 = t0 + tile * 16; qstep = 1;
        nvalid = min(16, t0 + nq - qpos0);
        int pmax = qpos0 + nvalid - 1;
        kstep = 1;
        if (func == JB_ATTN_DENSE) { kstart = 0; nkeys = pmax + 1; }
        else if (func == JB_ATTN_BLOCK) { kstart = (qpos0 / bc) * bc; nkeys = pmax - kstart + 1; }
        else if (func == JB_ATTN_PREV_BLOCK) {
            int b0 = qpos0 / bc, b1 = pmax / bc;
            kstart = max(b0 - 1, 0) * bc;
            nkeys = b1 * bc - kstart;       // keys up to the end of block b1-1; <= 0 when b1 == 0
        } else if (func == JB_ATTN_PRIME) { kstart = 0; nkeys = min(pmax + 1, cap); }
        else { kstart = 0; nkeys = cap; }   // cross attention: every encoder position
    }
    if (nvalid <= 0) return;
    const int my_q = qpos0 + c * qstep;      // position of this lane's query column (may be invalid)
    const bool q_ok = c < nvalid;

    // ---- stage the query tile -----------------------------------------------------------------
    const int dvec = d / E;                  // 16-byte chunks per row (vector path requires d % E == 0)
    const bool vec = (d % E == 0) && (S % E == 0);
    for (int r = 0; r < 16; ++r) {
        const bool rok = r < nvalid;
        const T* src = q + ((int64_t)n * nq + (qpos0 + r * qstep - t0)) * S + h * d;
        for (int i = lane; i < DP; i += 64) sQ[r * LDR + i] = (rok && i < d) ? src[i] : (T)0;
    }
    (void)dvec; (void)vec;

    f32x4 oacc[ND16];
#pragma unroll
    for (int i = 0; i < ND16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float scale = 1.0f / sqrtf(sqrtf((float)d));
    const float scale2 = scale * scale;
    const T* kbase = kc + ((int64_t)n * cap) * S + h * d;
    const T* vbase = vc + ((int64_t)n * cap) * S + h * d;

    for (int u0 = 0; u0 < nkeys; u0 += KT) {
        __syncthreads();                      // previous step's LDS reads done (single wave: cheap)
        // ---- stage K and V tiles: KT rows of d channels, coalesced along the row --------------
        for (int r = 0; r < KT; ++r) {
            const int u = u0 + r;
            const bool rok = u < nkeys;
            const int64_t pos = kstart + (int64_t)u * kstep;
            const T* ks = kbase + pos * S;
            const T* vs = vbase + pos * S;
            for (int i = lane; i < DP; i += 64) {
                const bool ok = rok && i < d;
                sK[r * LDR + i] = ok ? ks[i] : (T)0;
                sV[r * LDR + i] = ok ? vs[i] : (T)0;
            }
        }
        __syncthreads();

        // ---- scores: S^T[key][q] for NG groups of 16 keys --------------------------------------
        f32x4 sc[NG];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) sc[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int dk = 0; dk < DP; dk += KT) {
            if (dk >= d) break;
            V qfrag;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                int ch = dk + g * E + e;
                qfrag[e] = ch < DP ? sQ[c * LDR + ch] : (T)0;
            }
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                V kfrag;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    int ch = dk + g * E + e;
                    kfrag[e] = ch < DP ? sK[(gi * 16 + c) * LDR + ch] : (T)0;
                }
                sc[gi] = jb_mfma(kfrag, qfrag, sc[gi]);
            }
        }
        // ---- mask + online softmax (lane: query column c, keys gi*16 + g*4 + r) ------------------
        float pv[NG][4];
        float mx = -INFINITY;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = u0 + gi * 16 + g * 4 + r;
                const int j = kstart + u * kstep;
                bool ok = q_ok && u < nkeys;
                if (ok) {
                    switch (func) {
                        case JB_ATTN_DENSE: case JB_ATTN_PRIME: case JB_ATTN_TRANSPOSE_BLOCK: ok = j <= my_q; break;
                        case JB_ATTN_BLOCK: ok = j <= my_q && (j / bc) == (my_q / bc); break;
                        case JB_ATTN_PREV_BLOCK: ok = (j / bc) == (my_q / bc) - 1; break;
                        default: break;
                    }
                }
                float sv = jb_round<T>(jb_round<T>(sc[gi][r]) * scale2);
                pv[gi][r] = ok ? sv : -INFINITY;
                mx = fmaxf(mx, pv[gi][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.0f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pe = (pv[gi][r] == -INFINITY) ? 0.f : expf(pv[gi][r] - m_new);
                pv[gi][r] = pe;
                psum += pe;
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        V pfrag;                              // B operand: slot (g, e) <-> key g*4 + (e&3) + 16*(e>>2)
#pragma unroll
        for (int e = 0; e < E; ++e) pfrag[e] = (T)pv[e >> 2][e & 3];
#pragma unroll
        for (int i = 0; i < ND16; ++i) {
            oacc[i] *= alpha;
            V vfrag;                          // A operand: channel row i*16 + c, same slot <-> key map
#pragma unroll
            for (int e = 0; e < E; ++e) vfrag[e] = sV[(g * 4 + (e & 3) + 16 * (e >> 2)) * LDR + i * 16 + c];
            oacc[i] = jb_mfma(vfrag, pfrag, oacc[i]);
        }
    }

    // ---- normalise and store: lane holds query c, channels i*16 + g*4 + r ---------------------------
    if (!q_ok) return;
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;   // no admissible key (prev-block in block 0) -> 0
    T* orow = out + ((int64_t)n * nq + (my_q - t0)) * S + h * d;
#pragma unroll
    for (int i = 0; i < ND16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int ch = i * 16 + g * 4 + r;
            if (ch < d) orow[ch] = (T)(oacc[i][r] * inv);
        }
}

// Default for fp16 (jb_tune_attn_prefill_v2(0) restores the kernel above): the prefill attention with the staging fixed.
// attn_prefill_kernel runs one wave per 16-query tile and copies every K / V tile into its private LDS with 2-byte
// loads; at the upsamplers' sizes that is 4 TFLOP/s.  Here a workgroup of 4 waves owns 4 query tiles of one (sample, head)
// that share their keys -- 64 consecutive positions under the dense / block / prev / prime / cross patterns, 64 consecutive
// members of one residue class mod block_ctx under the transpose pattern (round 4: its one-wave launches were 1.18 ms against
// 0.11 ms for the other patterns) -- and stages each 32-key K and V tile ONCE, with 16-byte loads by all 256 threads; every
// wave then runs the MFMA / online-softmax body on its own tile.  Round 4: operands leave LDS as vectors -- the Q and K
// fragments are one ds_read_b128 each (rows padded by 16 bytes: conflict free), and V is staged TRANSPOSED ([channel][key],
// 80-byte rows) so that the A operand of O^T += V^T P^T (8 keys of one channel) is two ds_read_b64 instead of eight 2-byte
// reads.  fp16, d_head and n_state multiples of 8.
template <int ND16>
__global__ __launch_bounds__(256) void attn_prefill_v2_kernel(int func, const f16* __restrict__ q, const f16* __restrict__ kc,
                                                              const f16* __restrict__ vc, int cap, f16* __restrict__ out,
                                                              int n_head, int d, int bc, int t0, int nq, int wgs_per_class) {
    using V = f16x8;
    constexpr int E = 8, KT = 32, QW = 4;
    constexpr int DP = ND16 * 16;
    constexpr int LDR = DP + E;
    constexpr int VP = KT + 8;                       // pitch of a transposed V row (keys of one channel), elements
    constexpr int NG = KT / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f16* sQ = reinterpret_cast<f16*>(smem_raw);      // [QW][16][LDR]
    f16* sK = sQ + QW * 16 * LDR;                    // [KT][LDR]
    f16* sVt = sK + KT * LDR;                        // [DP][VP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.y, n = blockIdx.z;
    const int S = n_head * d;
    const int dv = d / E;                            // 16-byte vectors per row

    // ---- queries of this workgroup (64 of them, `qstep` positions apart) and of this wave (16 of them); candidate keys
    // kstart + u * kstep, u < nkeys, cover all 64 queries ----
    const int q_end = t0 + nq;                       // one past the last query position of the chunk
    int wg_q0, qstep, kstart, kstep, nkeys;
    if (func == JB_ATTN_TRANSPOSE_BLOCK) {
        const int cls = blockIdx.x / wgs_per_class, it = blockIdx.x % wgs_per_class;
        const int first = t0 + ((cls - t0 % bc) + bc) % bc;          // first position >= t0 in residue class cls
        wg_q0 = first + it * (QW * 16) * bc; qstep = bc;
        kstart = cls; kstep = bc;
    } else {
        wg_q0 = t0 + blockIdx.x * (QW * 16); qstep = 1;
        kstart = 0; kstep = 1;
    }
    if (wg_q0 >= q_end) return;                      // (transpose: a class with fewer members than the grid allows for)
    const int wg_nq = min(QW * 16, (q_end - 1 - wg_q0) / qstep + 1);
    const int wg_qlast = wg_q0 + (wg_nq - 1) * qstep;
    const int qpos0 = wg_q0 + wave * 16 * qstep;
    const int nvalid = max(0, min(16, wg_nq - wave * 16));
    const int my_q = qpos0 + c * qstep;
    const bool q_ok = c < nvalid;
    if (func == JB_ATTN_TRANSPOSE_BLOCK) nkeys = wg_qlast / bc + 1;
    else if (func == JB_ATTN_DENSE) nkeys = wg_qlast + 1;
    else if (func == JB_ATTN_BLOCK) { kstart = (wg_q0 / bc) * bc; nkeys = wg_qlast - kstart + 1; }
    else if (func == JB_ATTN_PREV_BLOCK) { kstart = max(wg_q0 / bc - 1, 0) * bc; nkeys = (wg_qlast / bc) * bc - kstart; }
    else if (func == JB_ATTN_PRIME) nkeys = min(wg_qlast + 1, cap);
    else nkeys = cap;

    // ---- stage the 4 query tiles (zero rows / padding channels where there is no query) ----
    for (int idx = tid; idx < QW * 16 * (LDR / E); idx += 256) {
        const int r = idx / (LDR / E), i = idx - r * (LDR / E);
        const int pos = wg_q0 + r * qstep;
        V v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < wg_nq && i < dv) v = *reinterpret_cast<const V*>(q + ((int64_t)n * nq + (pos - t0)) * S + h * d + i * E);
        *reinterpret_cast<V*>(sQ + r * LDR + i * E) = v;
    }

    f32x4 oacc[ND16];
#pragma unroll
    for (int i = 0; i < ND16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float scale = 1.0f / sqrtf(sqrtf((float)d));
    const float scale2 = scale * scale;
    const f16* kbase = kc + ((int64_t)n * cap) * S + h * d;
    const f16* vbase = vc + ((int64_t)n * cap) * S + h * d;
    const f16* sQw = sQ + wave * 16 * LDR;

    for (int u0 = 0; u0 < nkeys; u0 += KT) {
        __syncthreads();                              // every wave is done with the previous K / V tile (and sQ is written)
        // K rows as they are (one 16-byte vector per thread and pass, pad vector zeroed); V transposed: a thread takes 8
        // channels of one key and scatters them to 8 channel rows -- key index fastest over the lanes, so that a wave's
        // 2-byte stores fall into 16 consecutive dwords per channel row
        for (int idx = tid; idx < KT * (LDR / E); idx += 256) {
            const int r = idx / (LDR / E), i = idx - r * (LDR / E);
            const int u = u0 + r;
            V kv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (u < nkeys && i < dv) kv = *reinterpret_cast<const V*>(kbase + (int64_t)(kstart + (int64_t)u * kstep) * S + i * E);
            *reinterpret_cast<V*>(sK + r * LDR + i * E) = kv;
        }
        for (int idx = tid; idx < KT * (DP / E); idx += 256) {
            const int r = idx & (KT - 1), i = idx / KT;
            const int u = u0 + r;
            V vv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (u < nkeys && i < dv) vv = *reinterpret_cast<const V*>(vbase + (int64_t)(kstart + (int64_t)u * kstep) * S + i * E);
#pragma unroll
            for (int e = 0; e < E; ++e) sVt[(i * E + e) * VP + r] = vv[e];
        }
        __syncthreads();
        if (nvalid <= 0) continue;                    // wave-uniform: this wave has no queries (tail of the chunk)

        f32x4 sc[NG];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) sc[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dk = 0; dk < DP; dk += KT) {
            // channels dk + g*8 .. + 7: inside the row (padding channels are staged as zeros, the 16-byte pad behind the row
            // too) unless the whole vector lies past it (DP not a multiple of 32: the upper half of the last k-tile)
            const bool in_row = dk + g * E < DP;
            const int off = in_row ? dk + g * E : 0;
            const V qfrag = keep_frag<f16>(in_row, *reinterpret_cast<const V*>(sQw + c * LDR + off));
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const V kfrag = keep_frag<f16>(in_row, *reinterpret_cast<const V*>(sK + (gi * 16 + c) * LDR + off));
                sc[gi] = jb_mfma(kfrag, qfrag, sc[gi]);
            }
        }
        float pv[NG][4];
        float mx = -INFINITY;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = u0 + gi * 16 + g * 4 + r;
                const int j = kstart + u * kstep;
                bool ok = q_ok && u < nkeys;
                if (ok) {
                    switch (func) {
                        case JB_ATTN_DENSE: case JB_ATTN_PRIME: case JB_ATTN_TRANSPOSE_BLOCK: ok = j <= my_q; break;
                        case JB_ATTN_BLOCK: ok = j <= my_q && (j / bc) == (my_q / bc); break;
                        case JB_ATTN_PREV_BLOCK: ok = (j / bc) == (my_q / bc) - 1; break;
                        default: break;
                    }
                }
                const float sv = jb_round<f16>(jb_round<f16>(sc[gi][r]) * scale2);
                pv[gi][r] = ok ? sv : -INFINITY;
                mx = fmaxf(mx, pv[gi][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.0f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = (pv[gi][r] == -INFINITY) ? 0.f : expf(pv[gi][r] - m_new);
                pv[gi][r] = pe;
                psum += pe;
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        V pfrag;                                      // B operand: slot (g, e) <-> key g*4 + (e&3) + 16*(e>>2)
#pragma unroll
        for (int e = 0; e < E; ++e) pfrag[e] = (f16)pv[e >> 2][e & 3];
#pragma unroll
        for (int i = 0; i < ND16; ++i) {
            oacc[i] *= alpha;
            // A operand: channel row i*16 + c, the same slot <-> key map: keys g*4 .. +3 and 16 + g*4 .. +3 of the row
            const f16* vrow = sVt + (i * 16 + c) * VP + g * 4;
            const f16x4 lo = *reinterpret_cast<const f16x4*>(vrow), hi = *reinterpret_cast<const f16x4*>(vrow + 16);
            const V vfrag = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            oacc[i] = jb_mfma(vfrag, pfrag, oacc[i]);
        }
    }

    if (!q_ok) return;
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    f16* orow = out + ((int64_t)n * nq + (my_q - t0)) * S + h * d;
#pragma unroll
    for (int i = 0; i < ND16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = i * 16 + g * 4 + r;
            if (ch < d) orow[ch] = (f16)(oacc[i][r] * inv);
        }
}

static int g_prefill_v2 = 1;   // measured on MI355X (upsampler, 4096 x 16 tokens per window): 586 ms (one-wave kernel) -> 459 ms
extern "C" void jb_tune_attn_prefill_v2(int enable) { g_prefill_v2 = enable != 0; }

extern "C" int jb_attn_prefill(int dtype, int attn_func, const void* q, const void* kcache, const void* vcache,
                               int cache_cap, void* out, int n_batch, int n_head, int d_head, int block_ctx, int t0,
                               int n_q, void* stream) {
    JB_REQUIRE(q && kcache && vcache && out, "null pointer");
    JB_REQUIRE(dtype == JB_F32 || dtype == JB_F16, "bad dtype");
    JB_REQUIRE(n_batch > 0 && n_head > 0 && d_head > 0 && n_q > 0 && t0 >= 0, "bad dims");
    JB_REQUIRE(attn_func == 0 || attn_func == 7 || attn_func == 6 || block_ctx > 0, "block_ctx required");
    int tiles, tpc = 1;
    if (attn_func == JB_ATTN_TRANSPOSE_BLOCK) {
        int per_class = (n_q + block_ctx - 1) / block_ctx;     // max members of one residue class in the chunk
        tpc = (per_class + 15) / 16;
        tiles = block_ctx * tpc;
    } else {
        tiles = (n_q + 15) / 16;
    }
    const int nd16 = (d_head + 15) / 16;
    const int esz = dtype == JB_F16 ? 2 : 4, E = dtype == JB_F16 ? 8 : 4, KT = dtype == JB_F16 ? 32 : 16;
    dim3 grid(tiles, n_head, n_batch);
    hipStream_t s = (hipStream_t)stream;
    if (g_prefill_v2 && dtype == JB_F16 && d_head % 8 == 0 && (n_head * d_head) % 8 == 0) {
        // workgroups of 64 queries: consecutive positions, or (transpose) consecutive members of one residue class
        int wpc = 1;
        dim3 g2((n_q + 63) / 64, n_head, n_batch);
        if (attn_func == JB_ATTN_TRANSPOSE_BLOCK) {
            const int per_class = (n_q + block_ctx - 1) / block_ctx;
            wpc = (per_class + 63) / 64;
            g2.x = block_ctx * wpc;
        }
#define JB_LAUNCH_PF2(ND)                                                                                            \
    do {                                                                                                            \
        size_t lds2 = ((size_t)(4 * 16 + 32) * (ND * 16 + 8) + (size_t)(ND * 16) * 40) * 2;                         \
        if (lds2 > 64 * 1024)                                                                                       \
            JB_HIP(hipFuncSetAttribute((const void*)attn_prefill_v2_kernel<ND>,                                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                    \
        attn_prefill_v2_kernel<ND><<<g2, 256, lds2, s>>>(attn_func, (const f16*)q, (const f16*)kcache, (const f16*)vcache, \
                                                        cache_cap, (f16*)out, n_head, d_head, block_ctx, t0, n_q, wpc); \
    } while (0)
        if (nd16 <= 1) JB_LAUNCH_PF2(1);
        else if (nd16 <= 2) JB_LAUNCH_PF2(2);
        else if (nd16 <= 4) JB_LAUNCH_PF2(4);
        else if (nd16 <= 8) JB_LAUNCH_PF2(8);
        else if (nd16 <= 10) JB_LAUNCH_PF2(10);
        else if (nd16 <= 16) JB_LAUNCH_PF2(16);
        else if (nd16 <= 30) JB_LAUNCH_PF2(30);
        else JB_UNSUPPORTED("d_head > 480");
#undef JB_LAUNCH_PF2
        JB_CHECK_LAUNCH();
        return JB_OK;
    }
#define JB_LAUNCH_PF(T, ND)                                                                                         \
    do {                                                                                                            \
        size_t lds = (size_t)(16 + 2 * KT) * (ND * 16 + E) * esz;                                                   \
        if (lds > 64 * 1024)                                                                                        \
            JB_HIP(hipFuncSetAttribute((const void*)attn_prefill_kernel<T, ND>,                                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
        attn_prefill_kernel<T, ND><<<grid, 64, lds, s>>>(attn_func, (const T*)q, (const T*)kcache, (const T*)vcache, \
                                                         cache_cap, (T*)out, n_head, d_head, block_ctx, t0, n_q, tpc); \
    } while (0)
#define JB_DISPATCH_PF(T)                                         \
    do {                                                          \
        if (nd16 <= 1) JB_LAUNCH_PF(T, 1);                        \
        else if (nd16 <= 2) JB_LAUNCH_PF(T, 2);                   \
        else if (nd16 <= 4) JB_LAUNCH_PF(T, 4);                   \
        else if (nd16 <= 8) JB_LAUNCH_PF(T, 8);                   \
        else if (nd16 <= 10) JB_LAUNCH_PF(T, 10);                 \
        else if (nd16 <= 16) JB_LAUNCH_PF(T, 16);                 \
        else if (nd16 <= 30) JB_LAUNCH_PF(T, 30);                 \
        else JB_UNSUPPORTED("d_head > 480");                      \
    } while (0)
    if (dtype == JB_F16) JB_DISPATCH_PF(f16); else JB_DISPATCH_PF(float);
#undef JB_DISPATCH_PF
#undef JB_LAUNCH_PF
    JB_CHECK_LAUNCH();
    return JB_OK;
}


// ------------------------------------------------------------------------------------------------
// Attention probabilities of ONE head for a chunk of queries (alignment recording, jukebox/align.py:44-49 via
// FactoredAttention.record_attn, factored_attention.py:101-105): softmax over the pattern's key set of each query
// position, written as fp32 rows out[n][out_row0 + i][key position] for key positions < n_keys_out (zeros elsewhere).
// One wave per query; lanes stride over the keys.  Not on the timed path.
template <typename T>
__global__ __launch_bounds__(64) void attn_probs_kernel(int func, const T* __restrict__ q, const T* __restrict__ kc, int cap,
                                                        float* __restrict__ out, int64_t out_n_stride, int out_row0,
                                                        int n_keys_out, int n_head, int d, int head, int bc, int t0, int nq) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, i = blockIdx.x, n = blockIdx.y;
    const int p = t0 + i, S = n_head * d;
    const KeySet ks = decode_key_set(func, p, bc, cap);
    float* orow = out + (int64_t)n * out_n_stride + (int64_t)(out_row0 + i) * n_keys_out;
    for (int j = lane; j < n_keys_out; j += 64) orow[j] = 0.f;
    if (ks.count == 0) return;
    const T* qrow = q + ((int64_t)n * nq + i) * S + head * d;
    const T* kbase = kc + ((int64_t)n * cap) * S + head * d;
    const float scale = 1.0f / sqrtf(sqrtf((float)d));
    const float scale2 = scale * scale;
    float mx = -INFINITY;
    for (int k = lane; k < ks.count; k += 64) {
        const T* kr = kbase + (int64_t)(ks.start + k * ks.stride) * S;
        float acc = 0.f;
        for (int c = 0; c < d; ++c) acc += (float)qrow[c] * (float)kr[c];
        const float sc = jb_round<T>(jb_round<T>(acc) * scale2);
        sm[k] = sc;
        mx = fmaxf(mx, sc);
    }
    mx = jb_wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < ks.count; k += 64) sum += expf(sm[k] - mx);
    sum = jb_wave_sum(sum);
    __syncthreads();
    for (int k = lane; k < ks.count; k += 64) {
        const int pos = ks.start + k * ks.stride;
        if (pos < n_keys_out) orow[pos] = jb_round<T>(expf(sm[k] - mx) / sum);
    }
}

extern "C" int jb_attn_probs(int dtype, int attn_func, const void* q, const void* kcache, int cache_cap, float* out,
                             int64_t out_n_stride, int out_row0, int n_keys_out, int n_batch, int n_head, int d_head,
                             int head, int block_ctx, int t0, int n_q, void* stream) {
    JB_REQUIRE(q && kcache && out, "null pointer");
    JB_REQUIRE(dtype == JB_F32 || dtype == JB_F16, "bad dtype");
    JB_REQUIRE(n_batch > 0 && n_q > 0 && head >= 0 && head < n_head && n_keys_out > 0, "bad dims");
    size_t lds = (size_t)(t0 + n_q > cache_cap ? t0 + n_q : cache_cap) * sizeof(float);
    if (lds > 160 * 1024) JB_UNSUPPORTED("key set too long for the LDS score row");
    dim3 grid(n_q, n_batch);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == JB_F16) {
        if (lds > 64 * 1024) JB_HIP(hipFuncSetAttribute((const void*)attn_probs_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attn_probs_kernel<f16><<<grid, 64, lds, s>>>(attn_func, (const f16*)q, (const f16*)kcache, cache_cap, out, out_n_stride,
                                                     out_row0, n_keys_out, n_head, d_head, head, block_ctx, t0, n_q);
    } else {
        if (lds > 64 * 1024) JB_HIP(hipFuncSetAttribute((const void*)attn_probs_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attn_probs_kernel<float><<<grid, 64, lds, s>>>(attn_func, (const float*)q, (const float*)kcache, cache_cap, out,
                                                       out_n_stride, out_row0, n_keys_out, n_head, d_head, head, block_ctx, t0, n_q);
    }
    JB_CHECK_LAUNCH();
    return JB_OK;
}
