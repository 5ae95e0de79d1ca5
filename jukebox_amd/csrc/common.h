// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA fragments, half rounding).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/jukebox_hip.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- host-side error plumbing --------------------------------------------------------------
void jb_set_error(const std::string& msg);
#define JB_REQUIRE(cond, msg)                                              \
    do {                                                                   \
        if (!(cond)) {                                                     \
            jb_set_error(std::string(__func__) + ": " + (msg));            \
            return JB_ERR_ARG;                                             \
        }                                                                  \
    } while (0)
#define JB_UNSUPPORTED(msg)                                                \
    do {                                                                   \
        jb_set_error(std::string(__func__) + ": " + (msg));                \
        return JB_ERR_UNSUPPORTED;                                         \
    } while (0)
#define JB_CHECK_LAUNCH()                                                  \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            jb_set_error(std::string(__func__) + ": " + hipGetErrorString(e__)); \
            return JB_ERR_HIP;                                             \
        }                                                                  \
    } while (0)
#define JB_HIP(call)                                                       \
    do {                                                                   \
        hipError_t e__ = (call);                                           \
        if (e__ != hipSuccess) {                                           \
            jb_set_error(std::string(__func__) + ": " #call ": " + hipGetErrorString(e__)); \
            return JB_ERR_HIP;                                             \
        }                                                                  \
    } while (0)

// ---- per-dtype MFMA fragment description ---------------------------------------------------
// One MFMA "k-tile" holds KT contraction elements; lane l owns E consecutive ones starting at
// (l >> 4) * E, for row/col (l & 15) of its operand.  f16: v_mfma_f32_16x16x32_f16 (one
// instruction per k-tile).  f32: four v_mfma_f32_16x16x4_f32, element e of the fragment feeding
// the e-th instruction -- exact fp32 (an fmaf chain), used for the parity mode and the conv stacks.
template <typename T> struct Frag;
template <> struct Frag<f16> {
    typedef f16x8 vec;
    static constexpr int E = 8, KT = 32;
};
template <> struct Frag<float> {
    typedef f32x4 vec;
    static constexpr int E = 4, KT = 16;
};

__device__ __forceinline__ f32x4 jb_mfma(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 jb_mfma(f32x4 a, f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

template <typename T> __device__ __forceinline__ typename Frag<T>::vec jb_zero_frag() {
    typename Frag<T>::vec v;
#pragma unroll
    for (int e = 0; e < Frag<T>::E; ++e) v[e] = (T)0;
    return v;
}

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

// Branch-free variants for the fast paths: unconditional 16-byte load (caller clamps the address) and a
// per-element select.  A guarded load compiles to a branch plus s_waitcnt, which serialises cold loads.
template <typename T>
__device__ __forceinline__ typename Frag<T>::vec ld_frag(const T* __restrict__ p) {
    return *reinterpret_cast<const typename Frag<T>::vec*>(p);
}
template <typename T>
__device__ __forceinline__ typename Frag<T>::vec keep_frag(bool keep, typename Frag<T>::vec v) {
#pragma unroll
    for (int e = 0; e < Frag<T>::E; ++e) v[e] = keep ? v[e] : (T)0;
    return v;
}

// Round an fp32 value to the storage type and back (models "this tensor is half in the reference").
template <typename T> __device__ __forceinline__ float jb_round(float x);
template <> __device__ __forceinline__ float jb_round<float>(float x) { return x; }
template <> __device__ __forceinline__ float jb_round<f16>(float x) { return (float)(f16)x; }

// quick_gelu (jukebox/transformer/ops.py:33-35): x * sigmoid(1.702 x); each op rounds in half mode.
template <typename T> __device__ __forceinline__ float jb_quick_gelu(float x) {
    float u = jb_round<T>(1.702f * x);
    float s = jb_round<T>(1.0f / (1.0f + expf(-u)));
    return jb_round<T>(x * s);
}

template <typename T> __device__ __forceinline__ float jb_apply_act(float v, int act) {
    if (act == JB_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == JB_ACT_QUICK_GELU) return jb_quick_gelu<T>(v);
    return v;
}

// op over the four 16-lane rows of a wave, per column -- what `x = op(x, __shfl_xor(x, 16)); x = op(x, __shfl_xor(x, 32))`
// computes for a commutative op, bit for bit -- with gfx950's v_permlane16_swap / v_permlane32_swap (rows 1 / 3 of one copy against
// rows 0 / 2 of another, then the upper half against the lower) instead of two ds_bpermute round trips through the LDS crossbar.
template <typename Op> __device__ __forceinline__ float jb_rows_reduce(float x, Op op) {
    // (inline assembly: ROCm 7.2's clang lowers BOTH results of __builtin_amdgcn_permlane16_swap / 32_swap to the first operand's
    // register -- op(a, a), a silently wrong softmax; the s_nops are the wait states a VALU write needs before a permlane reads it)
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));      // {r0 r0 r2 r2}, {r1 r1 r3 r3}
    const float y = op(a, b);
    float c = y, d = y;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));      // {lo lo}, {hi hi}
    return op(c, d);
}

// Wave-wide (64-lane) reductions without the LDS crossbar: four DPP steps reduce each 16-lane row in the VALU
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row results are read with
// v_readlane and combined.  A __shfl_xor butterfly is six dependent ds_bpermute round trips (~100+ cycles each).
template <int CTRL> __device__ __forceinline__ float jb_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float jb_readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float jb_wave_sum(float v) {
    v += jb_dpp<0xB1>(v);
    v += jb_dpp<0x4E>(v);
    v += jb_dpp<0x141>(v);
    v += jb_dpp<0x140>(v);
    return (jb_readlane(v, 0) + jb_readlane(v, 16)) + (jb_readlane(v, 32) + jb_readlane(v, 48));
}
__device__ __forceinline__ float jb_wave_max(float v) {
    v = fmaxf(v, jb_dpp<0xB1>(v));
    v = fmaxf(v, jb_dpp<0x4E>(v));
    v = fmaxf(v, jb_dpp<0x141>(v));
    v = fmaxf(v, jb_dpp<0x140>(v));
    return fmaxf(fmaxf(jb_readlane(v, 0), jb_readlane(v, 16)), fmaxf(jb_readlane(v, 32), jb_readlane(v, 48)));
}

__device__ __forceinline__ float jb_load_any(const void* p, int dtype, int64_t i) {
    return dtype == JB_F16 ? (float)((const f16*)p)[i] : ((const float*)p)[i];
}

// ---- software-pipelined launches (engine.hip; measurements: HISTORY.md section 4.2) ------------------------------------------------
// The kernels of a decode step alternate between two streams, so launch j+1 is dispatched -- and requests its weights --
// while launch j still runs; the dependency itself is a completion word per launch slot:
//   runs[slot]    how often the slot has completed since the words were last zeroed (written by its last workgroup),
//   tickets[slot] arrivals of the current run: 8 shard counters (workgroup index mod 8; 120-180 arrivals on ONE word
//                 serialise at ~12 ns each) + one counter of finished shards, each in its own 128-byte line (+ 8 shard flag
//                 words for protocol 1, below).
// A launch reads its own count i, then thread 0 of every workgroup polls the producer slot until it has completed i + 1
// times (slot 0 follows the last slot of the previous step: i times), and only then reads what the producer wrote -- with
// sc1 loads (L1 bypass; the producer stored write-through and drained before it took its ticket), no cache-wide
// invalidate or write-back anywhere.  Data of launches j-2 and older is ordinary: the same stream ordered it.
// Polls are bounded (2 s of the 100 MHz clock): a timeout records slot + 1 in *err and goes on; every later wait gives up as
// soon as it sees the error word set (the caller clears it), so a failed call ends in about one bound, not one per launch.
// Measured alternatives (round 3, upsampler step): ONE ticket word for all 120-180 workgroups -- its returning atomics
// serialise; the last workgroup of a shard ADDING to the polled word with a fire-and-forget atomic instead of the
// second ticket + flag store -- 1.71 vs 1.59 ms per step: a word that is being polled must be written once, by a store.
struct JbPipe {
    unsigned* runs; unsigned* tickets; unsigned* err;
    int slot, prev;                                  // slot >= 0: a synchronised launch of a pipelined step (kernels' PIPE = 1);
                                                     // JB_PIPE_NO_SYNC (-2): the pipelined KERNEL FORM (operand-order blocks, 16-byte
                                                     // stores) as a launch of a plain chain -- the kernel boundary orders it, no poll,
                                                     // no ticket (PIPE = 2: own / wait / publish compiled out); -1: the plain kernels
    long long timeout;                               // poll bound in ticks of the 100 MHz clock (engine: 2 s, JB_PIPE_TIMEOUT_MS)
    long long* dbg;                                  // optional [slot][4] stamps of the 100 MHz clock (JB_PIPE_DEBUG): poll
                                                     // entered, producer seen, own completion published
    int proto;                                       // completion protocol: 0 = two-level ticket + one flag, 1 = a flag word per shard
    int frag;                                        // JB_FRAG_*: which activation blocks of this launch are in OPERAND ORDER (below)
    int last;                                        // the step's last slot (JB_PIPE_PRE_WAIT)
};
// Activation blocks of <= 16 rows handed from one pipelined launch to the next may be laid out in the order the consumer's MFMA
// B operands want them -- [k-tile][lane = (channel / 8 mod 4) * 16 + row][8 channels], 1 KiB per k-tile -- instead of [row][channel]:
// a consumer wave's 16-byte-per-lane fetch of a k-tile is then ONE contiguous KiB (8 full 128-byte lines) instead of 16 half
// lines 2 * width bytes apart, and a producer workgroup's 16 x 16 tile leaves as 2 runs of 256 bytes instead of 16 runs of 32.
// The block always holds 16 rows (engines of fewer samples leave the others untouched).
constexpr int JB_FRAG_X = 1, JB_FRAG_OUT = 2, JB_FRAG_RES = 4;
// JbPipe.frag also carries: this launch is the first of its STREAM in a step and that stream's previous launch ran before the
// previous step's sampler (the attention stream of the three-stream form, engine.hip): wait for the previous step's last slot
// (JbPipe.timeout's neighbour `last`) before the position is read.
constexpr int JB_PIPE_PRE_WAIT = 8;
__host__ __device__ __forceinline__ int jb_frag_el(int row, int col) {      // element index of (row, channel) in operand order
    return (col >> 5) * 512 + ((((col >> 3) & 3) << 4) + row) * 8 + (col & 7);
}
constexpr int JB_PIPE_STAMPS = 16;                   // clock stamps per launch slot (JB_PIPE_DEBUG; 4 in the product build, 10 with JB_PIPE_SEGMENTS)
// Build with -DJB_PIPE_SEGMENTS (python -m jukebox_amd.csrc.build --segments -> libjukebox_hip_segments.so, tools/phase_segments.py)
// for the per-segment account of a pipelined phase: workgroup 0 of every launch stamps, and WAITS where the product build does
// not have to (operands landed, partial tiles in LDS), so its own timeline is the account's; the other workgroups run as ever.
//   0 poll entered   1 producer's flags seen   4 barrier behind the poll passed   5 operand fragments landed (vmcnt 0)
//   6 last MFMA / tile arithmetic retired, partials written to LDS   7 LDS exchange barrier passed   3 stores issued
//   8 stores drained (vmcnt 0)   9 own ticket returned   2 the launch's last workgroup has published
#ifdef JB_PIPE_SEGMENTS
#define JB_SEG_ON(P) ((P).dbg && blockIdx.x == 0 && blockIdx.y == 0)
#define JB_SEG(P, k) do { if (JB_SEG_ON(P) && threadIdx.x == 0) (P).dbg[(P).slot * JB_PIPE_STAMPS + (k)] = wall_clock64(); } while (0)
#define JB_SEG_VM(P, k) do { if (JB_SEG_ON(P)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); JB_SEG(P, k); } } while (0)
#define JB_SEG_LGKM(P, k) do { if (JB_SEG_ON(P)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); JB_SEG(P, k); } } while (0)
#else
#define JB_SEG(P, k) do { } while (0)
#define JB_SEG_VM(P, k) do { } while (0)
#define JB_SEG_LGKM(P, k) do { } while (0)
#endif
constexpr int JB_PIPE_NO_SYNC = -2;
constexpr int JB_PIPE_PAD = 32;                      // words between two slots' completion words (128 bytes)
constexpr int JB_PIPE_TICKET_WORDS = 17 * JB_PIPE_PAD;  // per slot: 8 shard tickets + the shard count + (protocol 1) 8 shard flags
// (Round 6 measured 16 shards -- half the arrivals per ticket word, 16 flag words to poll -- on the upsampler step: 1.412 against
// 1.407 ms, no gain; profiles/r06c3_bench_engine_16B_stores_and_shards.log.)
// words the caller provides: completion counts, tickets, one error word
__host__ __device__ constexpr size_t jb_pipe_words(int n_slots) { return (size_t)n_slots * (JB_PIPE_PAD + JB_PIPE_TICKET_WORDS) + JB_PIPE_PAD; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t jb_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ unsigned jb_ld_word(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void jb_st_word(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte operand fragment / scalar through the write-through path: `base` must be wave-uniform (a kernel argument), the
// element offset may differ per lane (< 2^30 elements).
template <typename T> __device__ __forceinline__ typename Frag<T>::vec jb_ld_frag_sc1(const T* base, int64_t el) {
    return __builtin_bit_cast(typename Frag<T>::vec, __builtin_amdgcn_raw_buffer_load_b128(jb_rsrc(base), (int)(el * (int64_t)sizeof(T)), 0, 16));
}
__device__ __forceinline__ float jb_ld_sc1(const float* base, int64_t el) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(jb_rsrc(base), (int)(el * 4), 0, 16));
}
__device__ __forceinline__ f16 jb_ld_sc1(const f16* base, int64_t el) {
    return __builtin_bit_cast(f16, __builtin_amdgcn_raw_buffer_load_b16(jb_rsrc(base), (int)(el * 2), 0, 16));
}
__device__ __forceinline__ void jb_st_sc1(float* base, int64_t el, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), jb_rsrc(base), (int)(el * 4), 0, 16);
}
__device__ __forceinline__ void jb_st_sc1(f16* base, int64_t el, f16 v) {
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), jb_rsrc(base), (int)(el * 2), 0, 16);
}

// Own completion count: every thread asks for it next to its first requests (a broadcast load).
// (Unconditional and early.  Round 6 first gave the no-sync form of the kernels runtime guards: `slot >= 0 ? load : 0` compiled
// to a branch whose body ends in s_waitcnt vmcnt(0) -- a dependent round trip in FRONT of the weight requests, 1.425 against
// 1.397 ms per pipelined step, profiles/r06c8_ab_two_builds.log; the forms are template instantiations now: PIPE = 1 / 2.)
__device__ __forceinline__ unsigned jb_pipe_own(const JbPipe& P) { return jb_ld_word(P.runs + P.slot * JB_PIPE_PAD); }
// Wait for the producer launch; ends in a workgroup barrier.
// poll_wave: the wave that polls (the others wait at the barrier).  A poll is a vector load and returns IN ORDER behind
// everything its wave has requested before: a wave with a long prefetch outstanding (the attention's K / v' rows: 245 KB per
// workgroup, longer than its head start) learns of the flags only when that has landed, so the attention polls from its last
// wave, which owns a 16-key tile only at the latest positions.
__device__ __forceinline__ void jb_pipe_wait(const JbPipe& P, unsigned own, int poll_wave = 0) {
    const int pt = (int)threadIdx.x - poll_wave * 64;              // lane of the polling wave (other waves: outside 0..63)
    if (P.proto >= 1) {
        // Protocol 1: the last arriver of each of the 8 ticket shards stores the run's number into the shard's own flag word
        // (its own 128-byte line, written once per run); lanes 0..7 of the polling wave poll one flag each.  No second-level
        // ticket: one atomic round trip less per launch.  Every launch of the step has >= 8 workgroups (engines of >= 8 samples).
        // (Round 6 measured FOUR staggered polls in flight instead of the serial loop -- a flag would be seen a round trip and a
        // quarter after it was written instead of one to two: probe 4.04 against 4.08 us per phase, engine 1.607 against 1.572 ms
        // per step, 1.484 against 1.438 with operand-order hand-offs: no gain, removed; profiles/r06c1_*.)
        if (pt >= 0 && pt < 64) {
            const unsigned need = P.slot == 0 ? own : own + 1;
            const unsigned* w = P.tickets + (size_t)P.prev * JB_PIPE_TICKET_WORDS + (9 + (pt & 7)) * JB_PIPE_PAD;
            const bool stamp = P.dbg && pt == 0 && blockIdx.x == 0 && blockIdx.y == 0;
            if (stamp) P.dbg[P.slot * JB_PIPE_STAMPS] = wall_clock64();
            const long long t0 = wall_clock64();
            unsigned spins = 0;
            while (!__all(pt >= 8 || jb_ld_word(w) >= need)) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    if (jb_ld_word(P.err) != 0u) break;
                    if (wall_clock64() - t0 > P.timeout) { if (pt == 0) jb_st_word(P.err, (unsigned)P.slot + 1u); break; }
                }
            }
            if (stamp) P.dbg[P.slot * JB_PIPE_STAMPS + 1] = wall_clock64();
        }
        __syncthreads();
        JB_SEG(P, 4);
        return;
    }
    if (pt == 0) {
        const unsigned need = P.slot == 0 ? own : own + 1;
        const unsigned* w = P.runs + P.prev * JB_PIPE_PAD;
        const bool stamp = P.dbg && blockIdx.x == 0 && blockIdx.y == 0;
        if (stamp) P.dbg[P.slot * JB_PIPE_STAMPS] = wall_clock64();
        if (jb_ld_word(w) < need) {
            const long long t0 = wall_clock64();
            unsigned spins = 0;
            while (jb_ld_word(w) < need) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    // the error word is sticky: once ONE wait of this engine has timed out, the tokens are void and the
                    // remaining launches of the call must not each sit out their own bound (290 waits per step)
                    if (jb_ld_word(P.err) != 0u) break;
                    if (wall_clock64() - t0 > P.timeout) { jb_st_word(P.err, (unsigned)P.slot + 1u); break; }
                }
            }
        }
        if (stamp) P.dbg[P.slot * JB_PIPE_STAMPS + 1] = wall_clock64();
    }
    __syncthreads();
    JB_SEG(P, 4);
}
// JB_PIPE_PRE_WAIT: until the PREVIOUS step's last launch has completed (as slot 0 waits for it: `own` completions).
__device__ __forceinline__ void jb_pipe_wait_previous_step(const JbPipe& P, unsigned own, int poll_wave = 0) {
    JbPipe Q = P;
    Q.slot = 0; Q.prev = P.last; Q.dbg = nullptr;
    jb_pipe_wait(Q, own, poll_wave);
}
// After the last store of every thread: drain the write-through stores, count the workgroup in, the last one publishes.
__device__ __forceinline__ void jb_pipe_publish(const JbPipe& P, unsigned own) {
    if (P.dbg && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) P.dbg[P.slot * JB_PIPE_STAMPS + 3] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    JB_SEG(P, 8);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* tk = P.tickets + (size_t)P.slot * JB_PIPE_TICKET_WORDS;
        const unsigned n_wg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), shard = b & 7u;
        const unsigned members = (n_wg - shard + 7u) >> 3, n_shards = n_wg < 8u ? n_wg : 8u;
        const unsigned arrived = __hip_atomic_fetch_add(tk + shard * JB_PIPE_PAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef JB_PIPE_SEGMENTS
        if (JB_SEG_ON(P)) {
            P.dbg[P.slot * JB_PIPE_STAMPS + 9] = wall_clock64() + (long long)(arrived & 0u);      // (waits for the returned value)
            P.dbg[P.slot * JB_PIPE_STAMPS + 10] = wall_clock64();      // 10, 11: two stamps back to back -- what a stamp itself costs
            P.dbg[P.slot * JB_PIPE_STAMPS + 11] = wall_clock64();
        }
#endif
        if (arrived == members - 1) {
            jb_st_word(tk + shard * JB_PIPE_PAD, 0u);
            // protocol 1: the shard's flag is what the consumer waits for -- published first; the count of finished shards comes
            // after it, off the consumer's path, and only guards the slot's OWN count: that word is read by every workgroup of
            // the slot's next run at its start AND by late starters of THIS run (an attention workgroup waits for an empty
            // compute unit), so it may move only once every workgroup of this run has arrived.  (Round 4's first form let
            // shard 0's last arriver write it: a workgroup that started after that read the next run's number, waited for a
            // completion that belongs to the next step, and timed out -- once in a 20-second job.)
            if (P.proto >= 1) jb_st_word(tk + (9 + shard) * JB_PIPE_PAD, own + 1);
            if (__hip_atomic_fetch_add(tk + 8 * JB_PIPE_PAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_shards - 1) {
                jb_st_word(tk + 8 * JB_PIPE_PAD, 0u);
                jb_st_word(P.runs + P.slot * JB_PIPE_PAD, own + 1);
                if (P.dbg) P.dbg[P.slot * JB_PIPE_STAMPS + 2] = wall_clock64();
            }
        }
    }
}

// library-internal forms of the decode step's launches that take a pipeline slot (pipe == NULL: the exported behaviour)
int jb_gemv_impl(const jb_gemv_args* a, const JbPipe* pipe, void* stream);
int jb_attn_decode_impl(int dtype, int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache, int cache_cap,
                        void* out, int64_t ldo, int n_batch, int n_head, int d_head, int block_ctx, const int* t_dev, int max_len,
                        const JbPipe* pipe, void* stream);
int jb_attn_decode_pipe_supported(int dtype, int d_head, int ldq, int ldo, int S);
int jb_attn_decode_wide_impl(int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache_w, int cache_cap,
                             const void* res, int64_t ldr, const float* bias, void* x_out, int64_t ldo, int n_batch, int d_head,
                             int width, int block_ctx, const int* t_dev, int max_len, const JbPipe* pipe, void* stream);
int jb_sample_step_impl(const float* logits, int n_batch, int bins, const jb_sample_params* params, int64_t* tokens,
                        int64_t tok_stride, int* t_dev, float* preds, int64_t preds_n_stride, int x_dtype, void* x_next,
                        const float* x_emb, const float* pos_emb, const float* x_cond, int64_t xc_n_stride, int64_t xc_t_stride,
                        int width, int seq_len, unsigned* ticket, const JbPipe* pipe, void* stream);
