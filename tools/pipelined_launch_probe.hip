// Micro-benchmark for SOFTWARE-PIPELINED LAUNCHES of a dependent kernel chain (HISTORY.md sections 4.2 / 4.4).
//
// The decode step is a chain of ~290 kernels; each costs ~6 us of which ~3.5 us do not depend on the producer (launch
// boundary, wave start, the cold round trip of a weight stream whose addresses are known at capture time).  Here the
// kernels of the chain alternate between NS streams of one captured graph, so kernel j+1 is dispatched while kernel j runs:
// it requests its weights, then ONE lane per workgroup polls a flag that kernel j's last workgroup (atomic ticket)
// publishes with release semantics, and only then reads the activation block.  Every phase models one decode projection:
// each of G workgroups streams its own 60 KB of a weight matrix that is cold (K distinct matrices, 2 GB in total), reads
// the whole 16 x 1920 f16 activation block written by ALL workgroups of the previous phase, and writes its slice of the
// next one.  Visibility across the 8 XCD L2s is checked every phase with a checksum (stale data changes the sum).
//
//   NS = 1, no wait : one kernel per phase in stream order -- today's structure (the flag is still published: + ~0.2 us)
//   NS = 1, wait    : the same with the poll (always satisfied): cost of the flag machinery alone
//   NS = 2 / 3, wait: pipelined, kernel j+1 (and j+2) resident while j runs
//   V0: plain loads/stores, agent-scope release (buffer_wbl2 sc1) before the ticket, acquire (buffer_inv sc1) after the poll
//   V1: activations moved with relaxed agent-scope 8-byte atomics (sc1), no L2 write-back / invalidate
//   V2: V1 with a flag ROW per slot (one word per producer workgroup, no ticket atomics), polled by wave 0 with one
//       16-byte sc1 load per lane (round 3)
//   V3: V1 with the completion count sharded per XCD: every workgroup adds 1 (non-returning atomic) to the counter of the XCD
//       it runs on, the consumer's lanes 0..7 read the 8 counters with one sc1 load and compare their sum with (i + 1) * G
//   V8 / V9 / V10 (round 4, the candidate for round 5): NO completion word at all -- the activations carry their own validity.  V8: every
//       8-byte word = two halves + a 32-bit tag (the producing slot's number), written and read with relaxed agent-scope 8-byte
//       atomics; V9 / V10: 16-byte words = four / six halves + the tag (buffer_store / buffer_load b128 sc1; a torn word would show as a checksum
//       error; a producer's slice ends in a zero-padded word).  The consumer's first wave polls the last word of 64 producers until they carry the producer's tag, then every
//       thread reads its share of the block and re-reads whatever is not there yet.  Drops the producer's drain (s_waitcnt vmcnt(0)), the ticket atomics, the flag store and
//       the consumer's flag poll; costs 2x (V8, V9) / 1.35x (V10) the activation bytes.  The tag of a slot never changes: the word a
//       slot overwrites was written two slots earlier (another tag), so no run counter is needed.
//   V11: V10's words behind protocol 1's flags (V5) with the producer's drain removed: the flags say "stores issued", the tags catch
//       the words still in flight.
//   V12: protocol 1 (V5) with contiguous shards -- the producers of one eighth of the block -- and consumer waves that wait for
//       their own eighth only.
//   V13 / V14 (round 6): protocol 1 (V5) with the ENGINE's access pattern on the activation block instead of the probe's coalesced
//       8-byte words -- 16-byte sc1 loads of MFMA operand fragments, 8 per lane, and 64 8-byte stores per workgroup -- V13 with the
//       block as [row][channel] (a wave request = 16 half lines 3840 bytes apart, a workgroup's tile = 16 runs of 32 bytes), V14
//       with the block in operand order [k-tile][lane][8] (a wave request = one contiguous KiB, a tile = 2 runs of 256 bytes).
//       G = 120 only.
//   V15 / V16: V5 / V14 with FOUR polls of the consumer in flight, a quarter of a round trip apart (straight-line code).
//   "2 graphs": the even and the odd kernels of the chain as two single-stream graphs replayed on two streams (a 2-stream
//       capture in ONE graph replays at 22 us per kernel: round 3, profiles/r03_pipelined_launch_probe.log)
// Spin loops are bounded: after 2^22 polls a workgroup raises the abort flag, every later poll returns at once and the
// run is reported as aborted (no hung GPU).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools -o tools/pipelined_launch_probe tools/pipelined_launch_probe.hip -lhsa-runtime64
// Run:   tools/pipelined_launch_probe [graph launches, default 40] [first protocol] [last protocol] [1: the AQL modes too]
//        [1: sweep of the workgroup count instead of the three shapes of the step]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#include "aql_queue.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int N_EL = 16 * 1920;            // halfs per activation block
constexpr int N_W8 = N_EL / 4;             // 8-byte words
constexpr int THREADS = 512;
constexpr int K = 288;                     // kernels per graph (even: the double buffer of phase j is static)
constexpr int PAD = 32;                    // flags / tickets 128 B apart
constexpr size_t TAGGED_BLOCK = 131072;    // bytes per block of tagged words (V8 / V9: 122 880, V10: <= 82 944)
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    __half* act; const u32x4* wts; size_t w_phase_u4; unsigned* flags; unsigned* tickets; unsigned* err; unsigned* abort_flag;
    float* sink; int G; unsigned* rows;      // rows: [K][256] per-producer flags (V2)
    void* tagged;                             // V8 / V9: two blocks of tagged words
};
__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, 0x7fffffff, 0x00020000);
}

__device__ inline float expected_sum(unsigned p) { return 107520.0f + 30720.0f * (float)(p & 3); }
__device__ inline __half value_at(int i, unsigned p) { return __float2half((float)(((i + p) & 7) + (p & 3))); }

template <int V> __device__ inline u64 ld8(const u64* p) {
    if constexpr (V >= 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <int V> __device__ inline void st8(u64* p, u64 v) {
    if constexpr (V >= 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ inline unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int V, bool WAIT> __global__ __launch_bounds__(THREADS) void phase_kernel(Args a, int j) {
    __shared__ float red[THREADS / 64];
    __shared__ unsigned s_i;
    const int tid = threadIdx.x, wg = blockIdx.x;
    // 1. the weight stream does not depend on the producer: request it first (up to 16 x 16 B per thread = 128 KB per workgroup;
    //    8 at G = 120 / 192, 15 at G = 60)
    const size_t per_wg = a.w_phase_u4 / a.G;
    const u32x4* w = a.wts + (size_t)j * a.w_phase_u4 + (size_t)wg * per_wg;
    u32x4 wv[16];
    const int n_ld = min(16, (int)((per_wg + THREADS - 1) / THREADS));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const size_t k = (size_t)tid + (size_t)u * THREADS;
        wv[u] = u32x4{0u, 0u, 0u, 0u};
        if (u < n_ld) wv[u] = __builtin_nontemporal_load(w + (k < per_wg ? k : per_wg - 1));
    }
    asm volatile("" ::: "memory");
    if constexpr (V >= 8 && V <= 11) {
        // ---- tagged activations: the data is the flag ----
        constexpr int HPW = V == 8 ? 2 : (V == 9 ? 4 : 6);        // halves per word (V11: as V10)
        constexpr int WB = V == 8 ? 8 : 16;                       // bytes per word (V9: 4 bytes of padding)
        constexpr int PER_T = V == 8 ? 30 : (V == 9 ? 15 : 11);   // words per thread (upper bound)
        const int per = N_EL / a.G;                               // halves per producer workgroup (256 / 160)
        const int spw = (per + HPW - 1) / HPW;                    // words per producer workgroup (the last one padded with zeros)
        const int nwt = a.G * spw;                                // words per block
        const unsigned want = j == 0 ? (unsigned)K : (unsigned)j; // the producer slot's tag (slot j writes tag j + 1)
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.tagged) + (size_t)(j & 1) * TAGGED_BLOCK;
        unsigned char* dst = reinterpret_cast<unsigned char*>(a.tagged) + (size_t)((j + 1) & 1) * TAGGED_BLOCK;
        auto ld_word = [&](int w, float& sum) -> bool {           // true if the word carries the wanted tag; sum = its halves
            if (w >= nwt) { sum = 0.f; return true; }
            if constexpr (V == 8) {
                const u64 q = __hip_atomic_load(reinterpret_cast<const u64*>(src) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                union { unsigned u; __half h[2]; } cv; cv.u = (unsigned)q;
                sum = __half2float(cv.h[0]) + __half2float(cv.h[1]);
                return (unsigned)(q >> 32) == want;
            } else {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc(src), w * 16, 0, 16);
                union { unsigned u[3]; __half h[6]; } cv; cv.u[0] = q.x; cv.u[1] = q.y; cv.u[2] = q.z;
                sum = 0.f;
                for (int k = 0; k < HPW; ++k) sum += __half2float(cv.h[k]);
                return q.w == want;
            }
        };
        float part[PER_T];
        unsigned own_runs = 0;
        if constexpr (V == 11) {
            // V11: protocol 1's flags say "every producer has ISSUED its stores" (no drain before the ticket); the tags catch the words
            // that are still in flight
            if (tid < 64) {
                const unsigned i = ld_flag(a.flags + j * PAD + 1);
                if constexpr (WAIT) {
                    const int prev = j == 0 ? K - 1 : j - 1;
                    const unsigned n_shards = a.G < 8 ? (unsigned)a.G : 8u, need = j == 0 ? i : i + 1;
                    const unsigned* w = a.rows + (size_t)prev * 1024 + 256 + (tid & 7) * 32;
                    for (unsigned spins = 0;; ++spins) {
                        const bool ok = (unsigned)tid >= n_shards || ld_flag(w) >= need;
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (spins > (1u << 22) || ((spins & 255) == 255 && ld_flag(a.abort_flag))) {
                            __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                if (tid == 0) s_i = i;
            }
            __syncthreads();
            own_runs = s_i;
        } else if constexpr (WAIT) {
            // a. the cheap poll: the lanes of the first wave watch the LAST word of 64 producers spread over the launch (64 lines per
            //    workgroup and round) until all of them carry the producer's tag; the other waves sleep at the barrier
            if (tid < 64) {
                const int w = ((tid * a.G) >> 6) * spw + spw - 1;
                for (unsigned spins = 0;; ++spins) {
                    if (__all(ld_word(w, part[0]))) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > (1u << 22) || ((spins & 255) == 255 && ld_flag(a.abort_flag))) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
        }
        // b. the thread's share of the block, all requests in flight; c. whatever is not there yet, again
        bool ok[PER_T];
#pragma unroll
        for (int u = 0; u < PER_T; ++u) ok[u] = ld_word(tid + u * THREADS, part[u]);
        if constexpr (WAIT) {
            for (unsigned spins = 0;; ++spins) {
                bool all_ok = true;
#pragma unroll
                for (int u = 0; u < PER_T; ++u) all_ok = all_ok && ok[u];
                if (__all(all_ok)) break;
#pragma unroll
                for (int u = 0; u < PER_T; ++u) if (!ok[u]) ok[u] = ld_word(tid + u * THREADS, part[u]);
                if (spins > (1u << 22) || ((spins & 255) == 255 && ld_flag(a.abort_flag))) {
                    __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < PER_T; ++u) s += part[u];
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        float tot = 0.f;
        for (int k = 0; k < THREADS / 64; ++k) tot += red[k];
        const unsigned p = (unsigned)j;                           // K is a multiple of 8: the values depend on j only
        if (tid == 0 && tot != expected_sum(p)) atomicAdd(a.err, 1u);
        // my slice of the next block, tagged with this slot's number; nothing to drain, nothing to publish
        if (tid < spw) {
            __half hv[6];
            for (int k = 0; k < HPW; ++k) {
                const int o = tid * HPW + k;                      // position inside this workgroup's slice
                hv[k] = o < per ? value_at(wg * per + o, p + 1) : __float2half(0.f);
            }
            if constexpr (V == 8) {
                union { unsigned u; __half h[2]; } cv; cv.h[0] = hv[0]; cv.h[1] = hv[1];
                __hip_atomic_store(reinterpret_cast<u64*>(dst) + wg * spw + tid, ((u64)(unsigned)(j + 1) << 32) | cv.u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            } else {
                union { unsigned u[3]; __half h[6]; } cv; cv.u[2] = 0u;
                for (int k = 0; k < HPW; ++k) cv.h[k] = hv[k];
                u32x4 q; q.x = cv.u[0]; q.y = cv.u[1]; q.z = cv.u[2]; q.w = (unsigned)(j + 1);
                __builtin_amdgcn_raw_buffer_store_b128(q, rsrc(dst), (wg * spw + tid) * 16, 0, 16);
            }
        }
        float wacc = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) wacc += __uint_as_float(wv[u].x ^ wv[u].y ^ wv[u].z ^ wv[u].w);
        if (wacc + tot * 1e-30f == 123.456f) a.sink[0] = wacc;
        if constexpr (V == 11) {
            // publish as V5, but WITHOUT waiting for the stores to be acknowledged: the barrier only orders their issue
            __syncthreads();
            if (tid == 0) {
                const unsigned shard = (unsigned)wg & 7u, members = ((unsigned)a.G - shard + 7u) >> 3;
                unsigned* tk = a.rows + (size_t)j * 1024 + shard * 32;
                if (__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                    __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (shard == 0) __hip_atomic_store(a.flags + j * PAD + 1, own_runs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.rows + (size_t)j * 1024 + 256 + shard * 32, own_runs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        return;
    }
    // 2. how often has this slot run (its own flag), then wait for the producer slot to have run once more
    u64 r12[N_W8 / THREADS];                 // V12: the wave's eighth of the block, requested as soon as its shard is complete
    if constexpr (V == 3) {
        if (tid < 64) {
            // counters never reset: slot j has run i times when its 8 counters sum to i * G
            auto total = [&](int slot) {
                unsigned c = tid < 8 ? ld_flag(a.rows + (size_t)slot * 256 + tid * 16) : 0u;
                for (int o = 4; o; o >>= 1) c += __shfl_xor(c, o);
                return __shfl(c, 0);
            };
            const unsigned i = total(j) / (unsigned)a.G;
            if constexpr (WAIT) {
                const int prev = j == 0 ? K - 1 : j - 1;
                const unsigned need = (j == 0 ? i : i + 1) * (unsigned)a.G;
                for (unsigned spins = 0;; ++spins) {
                    if (total(prev) >= need) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > (1u << 20) || ((spins & 255) == 255 && ld_flag(a.abort_flag))) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            if (tid == 0) s_i = i;
        }
    } else if constexpr (V == 2) {
        if (tid < 64) {
            // how often has this slot run: its own row entry of workgroup 0; the producer slot must have run once more
            const unsigned i = ld_flag(a.rows + (size_t)j * 256 + wg);
            if constexpr (WAIT) {
                const int prev = j == 0 ? K - 1 : j - 1;
                const unsigned need = j == 0 ? i : i + 1;
                for (unsigned spins = 0;; ++spins) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.rows + (size_t)prev * 256), tid * 16, 0, 16);
                    bool ok = true;
                    for (int r = 0; r < 4; ++r) ok = ok && (tid * 4 + r >= a.G || v[r] >= need);
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > (1u << 20) || ((spins & 255) == 255 && ld_flag(a.abort_flag))) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            if (tid == 0) s_i = i;
        }
    } else if constexpr (V == 4) {
        // V4: 8 shard tickets; the last arriver of a shard stores ONE BYTE of the slot's 8-byte flag word (write-through, each byte
        // written once per run); the consumer polls that one word and wants every shard's byte at the run's number mod 256.
        // Against V1 this drops the serialisation of 120-192 returning atomics on one word, against the engine's two-level
        // ticket the second atomic round trip.
        if (tid == 0) {
            const unsigned i = ld_flag(a.flags + j * PAD + 1);               // full count of this slot's runs (same-stream order)
            if constexpr (WAIT) {
                const int prev = j == 0 ? K - 1 : j - 1;
                const unsigned n_shards = a.G < 8 ? (unsigned)a.G : 8u;
                const u64 keep = n_shards == 8 ? ~0ull : ((1ull << (8 * n_shards)) - 1ull);
                const u64 want = (0x0101010101010101ull * (u64)((j == 0 ? i : i + 1) & 0xffu)) & keep;
                const u64* word = reinterpret_cast<const u64*>(a.flags + prev * PAD + 2);
                unsigned spins = 0;
                while ((ld8<1>(word) & keep) != want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22) || ld_flag(a.abort_flag)) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            s_i = i;
        }
    } else if constexpr (V == 12) {
        // V12: protocol 1 with CONTIGUOUS shards (workgroups g / (G / 8): the producers of one eighth of the activation block) and a
        // consumer whose wave w needs only shard w: it polls that one flag and requests its eighth of the block at once -- the
        // fetch of early shards overlaps the producers' tail.  (The barrier below then only joins the waves for the reduction.)
        {
            const int wv_id = tid >> 6, ln = tid & 63;
            const unsigned i = ld_flag(a.flags + j * PAD + 1);
            if constexpr (WAIT) {
                const int prev = j == 0 ? K - 1 : j - 1;
                const unsigned need = j == 0 ? i : i + 1;
                const unsigned* w = a.rows + (size_t)prev * 1024 + 256 + wv_id * 32;
                for (unsigned spins = 0;; ++spins) {
                    if (ld_flag(w) >= need) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > (1u << 22) || ((spins & 255) == 255 && ld_flag(a.abort_flag))) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            const u64* s8 = reinterpret_cast<const u64*>(a.act + (size_t)(j & 1) * N_EL) + wv_id * (N_W8 / 8);
#pragma unroll
            for (int u = 0; u < N_W8 / THREADS; ++u) r12[u] = ld8<1>(s8 + ln + u * 64);
            if (tid == 0) s_i = i;
        }
    } else if constexpr (V == 5 || V == 7 || V >= 13) {
        // V5: 8 shard tickets; the last arriver of a shard stores the run's number into the shard's own FLAG WORD (write-through,
        // once per run); lanes 0..7 of wave 0 poll one flag word each.  V5: the 8 words in 8 different 128-byte lines; V7: in
        // one line.  Against the engine's protocol (V6) this drops the second-level ticket: one atomic round trip per launch.
        if (tid < 64) {
            const unsigned i = ld_flag(a.flags + j * PAD + 1);
            if constexpr (WAIT) {
                const int prev = j == 0 ? K - 1 : j - 1;
                const unsigned n_shards = a.G < 8 ? (unsigned)a.G : 8u, need = j == 0 ? i : i + 1;
                const unsigned* w = a.rows + (size_t)prev * 1024 + (V != 7 ? 256 + (tid & 7) * 32 : 512 + (tid & 7));
                bool seen = false;
                if constexpr (V == 15 || V == 16) {
                    unsigned q0 = ld_flag(w);
                    __builtin_amdgcn_s_sleep(6);
                    unsigned q1 = ld_flag(w);
                    __builtin_amdgcn_s_sleep(6);
                    unsigned q2 = ld_flag(w);
                    __builtin_amdgcn_s_sleep(6);
                    unsigned q3 = ld_flag(w);
#pragma unroll
                    for (int it = 0; it < 12; ++it) {
                        if (__all((unsigned)tid >= n_shards || q0 >= need)) { seen = true; break; }
                        q0 = ld_flag(w);
                        if (__all((unsigned)tid >= n_shards || q1 >= need)) { seen = true; break; }
                        q1 = ld_flag(w);
                        if (__all((unsigned)tid >= n_shards || q2 >= need)) { seen = true; break; }
                        q2 = ld_flag(w);
                        if (__all((unsigned)tid >= n_shards || q3 >= need)) { seen = true; break; }
                        q3 = ld_flag(w);
                    }
                }
                for (unsigned spins = 0; !seen; ++spins) {
                    const bool ok = (unsigned)tid >= n_shards || ld_flag(w) >= need;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > (1u << 22) || ((spins & 255) == 255 && ld_flag(a.abort_flag))) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            if (tid == 0) s_i = i;
        }
    } else if (tid == 0) {
        const unsigned i = ld_flag(a.flags + j * PAD);
        if constexpr (WAIT) {
            const int prev = j == 0 ? K - 1 : j - 1;
            const unsigned need = j == 0 ? i : i + 1;
            unsigned spins = 0;
            while (ld_flag(a.flags + prev * PAD) < need) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || ld_flag(a.abort_flag)) {
                    __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        s_i = i;
    }
    __syncthreads();
    if constexpr (V == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // buffer_inv sc1
    const unsigned i = s_i, p = i * (unsigned)K + (unsigned)j;
    const __half* src = a.act + (size_t)(j & 1) * N_EL;
    __half* dst = a.act + (size_t)((j + 1) & 1) * N_EL;
    // 3. the whole activation block, checksummed
    float s = 0.f;
    if constexpr (V == 13 || V == 14 || V == 16) {
        // the engine's fetch: wave w takes k-tiles 8 w .. 8 w + 7 (60 in all), lane l = (g, row) 16 bytes of each
        const int wv_id = tid >> 6, ln = tid & 63, row = ln & 15, g = ln >> 4;
        u32x4 f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kt = min(wv_id * 8 + u, 59);
            const int off = V == 13 ? row * 3840 + kt * 64 + g * 16 : kt * 1024 + ln * 16;
            f[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(src), off, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            union { u32x4 q; __half h[8]; } cv; cv.q = f[u];
            float t8 = 0.f;
            for (int k = 0; k < 8; ++k) t8 += __half2float(cv.h[k]);
            s += (wv_id * 8 + u < 60) ? t8 : 0.f;
        }
    } else {
    const u64* s8 = reinterpret_cast<const u64*>(src);
    u64 r[N_W8 / THREADS];
#pragma unroll
    for (int u = 0; u < N_W8 / THREADS; ++u) r[u] = V == 12 ? r12[u] : ld8<(V >= 12 ? 1 : V)>(s8 + tid + u * THREADS);
#pragma unroll
    for (int u = 0; u < N_W8 / THREADS; ++u) {
        union { u64 q; __half h[4]; } cv; cv.q = r[u];
        s += __half2float(cv.h[0]) + __half2float(cv.h[1]) + __half2float(cv.h[2]) + __half2float(cv.h[3]);
    }
    }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    float tot = 0.f;
    for (int k = 0; k < THREADS / 64; ++k) tot += red[k];
    if (tid == 0 && tot != expected_sum(p)) atomicAdd(a.err, 1u);
    // 4. my slice of the next block
    const int per = N_EL / a.G, w8 = per / 4;
    if constexpr (V == 13 || V == 14 || V == 16) {
        // the engine's store: workgroup wg owns columns 16 wg .. 16 wg + 15 of all 16 rows; 64 threads, 4 columns (8 bytes) each
        if (tid < 64) {
            const int row = tid & 15, col = wg * 16 + (tid >> 4) * 4;
            union { u64 q; __half h[4]; } cv;
            for (int k = 0; k < 4; ++k) cv.h[k] = value_at(row * 1920 + col + k, p + 1);
            const int el = V == 13 ? row * 1920 + col : (col >> 5) * 512 + ((((col >> 3) & 3) << 4) + row) * 8 + (col & 7);
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            union { u64 q; u32x2 v; } o; o.q = cv.q;
            __builtin_amdgcn_raw_buffer_store_b64(o.v, rsrc(dst), el * 2, 0, 16);
        }
    } else if (tid < w8) {
        union { u64 q; __half h[4]; } cv;
        const int i0 = wg * per + tid * 4;
        for (int k = 0; k < 4; ++k) cv.h[k] = value_at(i0 + k, p + 1);
        st8<(V >= 12 ? 1 : V)>(reinterpret_cast<u64*>(dst) + (i0 >> 2), cv.q);
    }
    float wacc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) wacc += __uint_as_float(wv[u].x ^ wv[u].y ^ wv[u].z ^ wv[u].w);
    if (wacc + tot * 1e-30f == 123.456f) a.sink[0] = wacc;
    // 5. publish: every workgroup's stores are visible device-wide before its ticket; the last ticket raises the flag
    if constexpr (V == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");           // s_waitcnt + buffer_wbl2 sc1
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
    if constexpr (V == 3) {
        if (tid == 0) {
            const unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;
            __hip_atomic_fetch_add(a.rows + (size_t)j * 256 + x * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if constexpr (V == 2) {
        if (tid == 0) __hip_atomic_store(a.rows + (size_t)j * 256 + wg, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (V == 4) {
        if (tid == 0) {
            const unsigned shard = (unsigned)wg & 7u, members = ((unsigned)a.G - shard + 7u) >> 3;
            unsigned* tk = a.rows + (size_t)j * 256 + shard * 16;            // the shard's ticket, 64 bytes from its neighbours
            if (__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (shard == 0) __hip_atomic_store(a.flags + j * PAD + 1, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<unsigned char*>(a.flags + j * PAD + 2) + shard, (unsigned char)((i + 1) & 0xffu),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if constexpr (V == 12) {
        if (tid == 0) {
            const unsigned spb = ((unsigned)a.G + 7u) >> 3, shard = (unsigned)wg / spb;
            const unsigned members = min(spb, (unsigned)a.G - shard * spb);
            unsigned* tk = a.rows + (size_t)j * 1024 + shard * 32;
            if (__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the slot's own count moves with the LAST shard's flag: in this probe every workgroup of a run is resident before any
                // can finish (G <= 256), so no late starter reads it (the engine guards it with a count of finished shards)
                if (shard == 0) __hip_atomic_store(a.flags + j * PAD + 1, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.rows + (size_t)j * 1024 + 256 + shard * 32, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if constexpr (V == 5 || V == 7 || V >= 13) {
        if (tid == 0) {
            const unsigned shard = (unsigned)wg & 7u, members = ((unsigned)a.G - shard + 7u) >> 3;
            unsigned* tk = a.rows + (size_t)j * 1024 + shard * 32;
            if (__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (shard == 0) __hip_atomic_store(a.flags + j * PAD + 1, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.rows + (size_t)j * 1024 + (V != 7 ? 256 + shard * 32 : 512 + shard), i + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if constexpr (V == 6) {
        // V6: the engine's protocol (common.h, jb_pipe_publish): 8 shard tickets + a count of finished shards, the last shard's last
        // workgroup stores the slot's completion count, which the consumer polls (as V1)
        if (tid == 0) {
            const unsigned shard = (unsigned)wg & 7u, members = ((unsigned)a.G - shard + 7u) >> 3, n_shards = a.G < 8 ? (unsigned)a.G : 8u;
            unsigned* tk = a.rows + (size_t)j * 1024;
            if (__hip_atomic_fetch_add(tk + shard * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(tk + shard * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(tk + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_shards - 1) {
                    __hip_atomic_store(tk + 256, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.flags + j * PAD, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    } else if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.tickets + j * PAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)a.G - 1) {
            __hip_atomic_store(a.tickets + j * PAD, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.flags + j * PAD, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int V, bool WAIT> static void launch(const Args& a, int j, hipStream_t s) {
    phase_kernel<V, WAIT><<<a.G, THREADS, 0, s>>>(a, j);
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 40;
    std::vector<__half> h(N_EL);
    for (int i = 0; i < N_EL; ++i) h[i] = __float2half((float)((i & 7)));
    __half* act; unsigned *flags, *tickets, *err, *abortf; float* sink; u32x4* wts;
    const bool layers = argc > 6 && atoi(argv[6]) != 0;          // argv[6] = 1: cycles of phases that model a whole layer (below)
    const size_t W_MAX_PHASE = layers ? 25804800 : 11059200;     // 1920 x 2880 f16 (1920 x 6720 for the 3-phase layer's first phase)
    CK(hipMalloc(&act, 2 * N_EL * 2)); CK(hipMalloc(&flags, K * PAD * 4)); CK(hipMalloc(&tickets, K * PAD * 4));
    CK(hipMalloc(&err, 4)); CK(hipMalloc(&abortf, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&wts, W_MAX_PHASE * K)); CK(hipMemset(wts, 1, W_MAX_PHASE * K));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t st[3];
    for (int k = 0; k < 3; ++k) {                                // distinct priority classes: distinct hardware queues
        int pr = hi + k; if (pr > lo) pr = lo;
        CK(hipStreamCreateWithPriority(&st[k], hipStreamNonBlocking, pr));
    }
    hipEvent_t e0, e1, fork, join[3];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int k = 0; k < 3; ++k) CK(hipEventCreateWithFlags(&join[k], hipEventDisableTiming));
    struct Shape { int G; size_t wbytes; const char* what; };
    // argv[5] = 1: the same 1920 x 1920 matrix on 60 / 120 / 240 workgroups (32 / 16 / 8 columns each) and the weightless phase on as
    // many: does the phase get cheaper when fewer workgroups fetch the 61-KB activation block?
    const bool sweep = argc > 5 && atoi(argv[5]) != 0;
    // argv[6] = 1 (round 5): a LAYER as a cycle of phases with their own workgroup counts and cold bytes per slot, protocol 1,
    // two graphs on two streams -- the cost model of the engine's pipelined launches.  What a phase streams before its poll:
    //   today, 4 phases:   wide c_attn W x (2S + W) on 180 workgroups | attention: K + the v' slice of ~53 keys (the mean over the
    //                      three patterns and a window) per (sample, slice): 64 workgroups x 100 KB | c_fc | mlp.c_proj
    //   3 phases: c_attn + u = W'fc x + the v'' = W'fc (Wp Wv) columns: W x (2S + 3W) on 420 workgroups |
    //                      attention producing x1 AND h_raw = u + sum p v'': 128 workgroups x 100 KB | mlp.c_proj with the LN /
    //                      gelu prologue (same bytes)
    // Every phase still all-gathers the 61-KB activation block of its producer and writes its slice of the next one.
    std::vector<std::vector<Shape>> cycles;
    if (layers) {
        // (workgroup counts rounded to divisors of the block with whole 8-byte words per slice: 192 for 180, 384 / 480 for 420)
        cycles.push_back({{192, 11059200, "c_attn"}, {64, 6553600, "attention"}, {120, 7372800, "c_fc"}, {120, 7372800, "c_proj"}});
        cycles.push_back({{384, 25804800, "c_attn+u+v''"}, {128, 13107200, "attention x1+h_raw"}, {120, 7372800, "c_proj"}});
        cycles.push_back({{480, 25804800, "c_attn+u+v''"}, {128, 13107200, "attention x1+h_raw"}, {120, 7372800, "c_proj"}});
        cycles.push_back({{240, 25804800, "c_attn+u+v'' on 32-column tiles"}, {128, 13107200, "attention x1+h_raw"}, {120, 7372800, "c_proj"}});
    }
    const std::vector<Shape> shapes = layers ? std::vector<Shape>{{-1, 0, "today: 4 phases per layer"}, {-2, 0, "3 phases, first on 384 wgs"},
                                                                  {-3, 0, "3 phases, first on 480 wgs"},
                                                                  {-4, 0, "3 phases, first on 240 wgs"}} : sweep
        ? std::vector<Shape>{{60, 7372800, "1920x1920 on 60 workgroups"}, {120, 7372800, "1920x1920 on 120 workgroups"},
                             {240, 7372800, "1920x1920 on 240 workgroups"}, {60, 0, "no weights, 60 workgroups"},
                             {120, 0, "no weights, 120 workgroups"}, {240, 0, "no weights, 240 workgroups"}}
        : std::vector<Shape>{{120, 7372800, "c_fc / mlp.c_proj (1920x1920)"}, {192, 11059200, "wide c_attn (1920x2880)"},
                             {120, 0, "no weights"}};
    unsigned* rows; CK(hipMalloc(&rows, (size_t)K * 1024 * 4));
    void* tagged; CK(hipMalloc(&tagged, 2 * TAGGED_BLOCK));
    // kind 0: one graph (all streams captured into it)   1: eager   2: one single-stream graph per stream, replayed side by side
    struct Mode { int ns; bool wait; int kind; const char* name; };
    const Mode modes[] = {{1, false, 0, "graph 1 stream, no wait"}, {1, true, 0, "graph 1 stream, wait"},
                          {2, true, 1, "eager 2 streams, wait"},
                          {2, true, 2, "2 graphs on 2 streams, wait"}, {3, true, 2, "3 graphs on 3 streams, wait"}};
    const int v_lo = argc > 2 ? atoi(argv[2]) : 1, v_hi = argc > 3 ? atoi(argv[3]) : 4;     // protocol variants to run
    for (const Shape& sh : shapes) for (int v = v_lo; v <= v_hi; ++v) for (const Mode& m : modes) {
        if ((v == 13 || v == 14 || v == 16) && sh.G != 120) continue;        // the engine's tile ownership: 120 workgroups of 16 columns
        Args a{act, wts, sh.wbytes / 16, flags, tickets, err, abortf, sink, sh.G, rows, tagged};
        if (a.w_phase_u4 == 0) a.w_phase_u4 = (size_t)sh.G;      // one dummy vector per workgroup
        CK(hipMemcpy(act, h.data(), N_EL * 2, hipMemcpyHostToDevice));
        CK(hipMemset(flags, 0, K * PAD * 4)); CK(hipMemset(tickets, 0, K * PAD * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(abortf, 0, 4));
        CK(hipMemset(rows, 0, (size_t)K * 1024 * 4));
        if (v >= 8 && v <= 11) {
            // block 0 as the last slot of a previous replay would have left it (tag K), block 1 with a tag no slot waits for
            const int hpw = v == 8 ? 2 : (v == 9 ? 4 : 6), wu = v == 8 ? 2 : 4;          // (V11: as V10)        // halves / 32-bit units per word
            const int per = N_EL / sh.G, spw = (per + hpw - 1) / hpw;
            std::vector<unsigned> t(2 * TAGGED_BLOCK / 4, 0xffffffffu);
            for (int g = 0; g < sh.G; ++g) for (int w = 0; w < spw; ++w) {
                unsigned* word = &t[((size_t)g * spw + w) * wu];
                for (int u = 0; u < wu; ++u) word[u] = 0u;
                for (int k = 0; k < hpw; ++k) if (w * hpw + k < per) reinterpret_cast<__half*>(word)[k] = h[(size_t)g * per + w * hpw + k];
                word[wu - 1] = (unsigned)K;
            }
            CK(hipMemcpy(tagged, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        }
        CK(hipDeviceSynchronize());
        const std::vector<Shape>* cyc = sh.G < 0 ? &cycles[-sh.G - 1] : nullptr;
        const Args a_uniform = a;
        auto launch_j = [&](int j, hipStream_t s) {
            Args a = a_uniform;
            if (cyc) { const Shape& c = (*cyc)[j % cyc->size()]; a.G = c.G; a.w_phase_u4 = c.wbytes / 16; }
            if (v == 1) { if (m.wait) launch<1, true>(a, j, s); else launch<1, false>(a, j, s); }
            else if (v == 2) { if (m.wait) launch<2, true>(a, j, s); else launch<2, false>(a, j, s); }
            else if (v == 3) { if (m.wait) launch<3, true>(a, j, s); else launch<3, false>(a, j, s); }
            else if (v == 4) { if (m.wait) launch<4, true>(a, j, s); else launch<4, false>(a, j, s); }
            else if (v == 5) { if (m.wait) launch<5, true>(a, j, s); else launch<5, false>(a, j, s); }
            else if (v == 6) { if (m.wait) launch<6, true>(a, j, s); else launch<6, false>(a, j, s); }
            else if (v == 7) { if (m.wait) launch<7, true>(a, j, s); else launch<7, false>(a, j, s); }
            else if (v == 8) { if (m.wait) launch<8, true>(a, j, s); else launch<8, false>(a, j, s); }
            else if (v == 9) { if (m.wait) launch<9, true>(a, j, s); else launch<9, false>(a, j, s); }
            else if (v == 10) { if (m.wait) launch<10, true>(a, j, s); else launch<10, false>(a, j, s); }
            else if (v == 11) { if (m.wait) launch<11, true>(a, j, s); else launch<11, false>(a, j, s); }
            else if (v == 12) { if (m.wait) launch<12, true>(a, j, s); else launch<12, false>(a, j, s); }
            else if (v == 13) { if (m.wait) launch<13, true>(a, j, s); else launch<13, false>(a, j, s); }
            else if (v == 14) { if (m.wait) launch<14, true>(a, j, s); else launch<14, false>(a, j, s); }
            else if (v == 15) { if (m.wait) launch<15, true>(a, j, s); else launch<15, false>(a, j, s); }
            else { if (m.wait) launch<16, true>(a, j, s); else launch<16, false>(a, j, s); }
        };
        float ms = 0.f;
        if (m.kind == 0) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
            if (m.ns > 1) {
                CK(hipEventRecord(fork, st[0]));
                for (int k = 1; k < m.ns; ++k) CK(hipStreamWaitEvent(st[k], fork, 0));
            }
            for (int j = 0; j < K; ++j) launch_j(j, st[j % m.ns]);
            for (int k = 1; k < m.ns; ++k) { CK(hipEventRecord(join[k], st[k])); CK(hipStreamWaitEvent(st[0], join[k], 0)); }
            CK(hipStreamEndCapture(st[0], &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st[0])); CK(hipStreamSynchronize(st[0]));          // warm-up
            CK(hipEventRecord(e0, st[0]));
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st[0]));
            CK(hipEventRecord(e1, st[0])); CK(hipStreamSynchronize(st[0]));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        } else if (m.kind == 1) {
            // eager: the flags are the ONLY dependency between streams
            for (int j = 0; j < K; ++j) launch_j(j, st[j % m.ns]);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st[0]));
            for (int r = 0; r < R; ++r) for (int j = 0; j < K; ++j) launch_j(j, st[j % m.ns]);
            for (int k = 1; k < m.ns; ++k) { CK(hipEventRecord(join[k], st[k])); CK(hipStreamWaitEvent(st[0], join[k], 0)); }
            CK(hipEventRecord(e1, st[0])); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        } else {
            hipGraph_t g[3]; hipGraphExec_t ge[3];
            for (int k = 0; k < m.ns; ++k) {
                CK(hipStreamBeginCapture(st[k], hipStreamCaptureModeThreadLocal));
                for (int j = k; j < K; j += m.ns) launch_j(j, st[k]);
                CK(hipStreamEndCapture(st[k], &g[k])); CK(hipGraphInstantiate(&ge[k], g[k], nullptr, nullptr, 0));
            }
            for (int k = 0; k < m.ns; ++k) CK(hipGraphLaunch(ge[k], st[k]));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st[0]));
            for (int r = 0; r < R; ++r) for (int k = 0; k < m.ns; ++k) CK(hipGraphLaunch(ge[k], st[k]));
            for (int k = 1; k < m.ns; ++k) { CK(hipEventRecord(join[k], st[k])); CK(hipStreamWaitEvent(st[0], join[k], 0)); }
            CK(hipEventRecord(e1, st[0])); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
            for (int k = 0; k < m.ns; ++k) { CK(hipGraphExecDestroy(ge[k])); CK(hipGraphDestroy(g[k])); }
        }
        unsigned errs = 0, ab = 0;
        CK(hipMemcpy(&errs, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost));
        if (cyc) printf("%-34s V%d %-28s %6.2f us/phase = %6.2f us per LAYER of %zu phases  (checksum errors %u, aborted %u)\n", sh.what, v,
                        m.name, ms * 1e3 / ((double)R * K), ms * 1e3 / ((double)R * K) * cyc->size(), cyc->size(), errs, ab);
        else
        printf("%-30s G=%3d V%d %-28s %6.2f us/phase  (checksum errors %u, aborted %u)\n", sh.what, sh.G, v, m.name,
               ms * 1e3 / ((double)R * K), errs, ab);
        fflush(stdout);
        if (ab) { printf("aborted: a poll timed out -- stopping\n"); return 2; }
    }
    // ---- the same chain through a hardware queue of our own (tools/aql_queue.h): pre-built AQL packets, one memcpy into the
    // ring + one doorbell per replay.  barrier = 1: the packet processor keeps stream order (today's chain without HIP);
    // barrier = 0: packet j+1 is dispatched as soon as packet j's workgroups are placed -- pipelined launches in ONE queue.
    // acquire / release: the cache actions the packet processor performs around each dispatch (2 = agent scope as HIP's launches,
    // 0 = none: V1 / V4 move their activations with sc1 accesses and need none).  `lds`: extra LDS per workgroup, i.e. how many
    // launches can be resident at once (the depth of the run-ahead).
    if (argc > 4 && atoi(argv[4]) != 0) {
        jb_aql::Queue Q;
        if (!Q.create(0, 16384)) { printf("aql: %s\n", Q.error.c_str()); return 0; }
        struct AqlMode { int v; bool wait; bool barrier; int acq, rel; unsigned lds; const char* name; };
        const AqlMode am[] = {
            {1, false, true, 2, 2, 0, "aql barrier=1 acq/rel agent, no wait"},
            {1, true, true, 2, 2, 0, "aql barrier=1 acq/rel agent, wait"},
            {1, true, false, 2, 2, 0, "aql barrier=0 acq/rel agent"},
            {1, true, false, 2, 0, 0, "aql barrier=0 acq agent rel none"},
            {1, true, false, 0, 0, 0, "aql barrier=0 no fences"},
            {1, true, false, 0, 0, 40960, "aql barrier=0 no fences lds 40K"},
            {1, true, false, 0, 0, 81920, "aql barrier=0 no fences lds 80K"},
            {1, true, false, 2, 2, 81920, "aql barrier=0 agent lds 80K"},
            {4, true, false, 2, 2, 0, "aql barrier=0 acq/rel agent"},
            {4, true, false, 0, 0, 0, "aql barrier=0 no fences"},
            {4, true, false, 0, 0, 81920, "aql barrier=0 no fences lds 80K"},
        };
        for (const Shape& sh : shapes) for (const AqlMode& m : am) {
            Args a{act, wts, sh.wbytes / 16, flags, tickets, err, abortf, sink, sh.G, rows, tagged};
            if (a.w_phase_u4 == 0) a.w_phase_u4 = (size_t)sh.G;
            CK(hipMemcpy(act, h.data(), N_EL * 2, hipMemcpyHostToDevice));
            CK(hipMemset(flags, 0, K * PAD * 4)); CK(hipMemset(tickets, 0, K * PAD * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(abortf, 0, 4));
            CK(hipMemset(rows, 0, (size_t)K * 1024 * 4));
            CK(hipDeviceSynchronize());
            const void* fn = m.v == 1 ? (m.wait ? (const void*)&phase_kernel<1, true> : (const void*)&phase_kernel<1, false>)
                                      : (const void*)&phase_kernel<4, true>;
            jb_aql::Kernel kd;
            if (!Q.lookup(fn, &kd)) { printf("aql: %s\n", Q.error.c_str()); return 0; }
            const size_t ka = (kd.kernarg_size + 63) & ~(size_t)63;
            std::vector<uint8_t> img(ka * K, 0);
            uint8_t* dargs; CK(hipMalloc(&dargs, ka * K));
            std::vector<hsa_kernel_dispatch_packet_t> pk(K);
            std::vector<uint16_t> hd(K);
            for (int j = 0; j < K; ++j) {
                uint8_t* im = img.data() + (size_t)j * ka;
                memcpy(im, &a, sizeof(Args)); memcpy(im + sizeof(Args), &j, sizeof(int));
                jb_aql::Queue::fill_implicit(im, kd, sizeof(Args) + sizeof(int), dim3(sh.G), dim3(THREADS));
                pk[j] = jb_aql::Queue::packet(kd, dim3(sh.G), dim3(THREADS), m.lds, dargs + (size_t)j * ka);
                hd[j] = jb_aql::header(m.barrier, m.acq, m.rel);
            }
            CK(hipMemcpy(dargs, img.data(), ka * K, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            std::vector<uint16_t> hd_first = hd;
            hd_first[0] = jb_aql::header(true, 2, 2);
            Q.arm(); Q.submit(pk.data(), hd_first.data(), K, true);        // warm-up replay
            if (!Q.wait(5000000000ull)) { printf("aql: warm-up replay timed out (%s)\n", m.name); return 2; }
            Q.arm();
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < R; ++r) Q.submit(pk.data(), r == 0 ? hd_first.data() : hd.data(), K, r == R - 1);
            const auto t_enq = std::chrono::steady_clock::now();
            const bool fin = Q.wait(20000000000ull);
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            unsigned errs = 0, ab = 0;
            CK(hipMemcpy(&errs, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost));
            const double us = std::chrono::duration<double, std::micro>(t1 - t0).count(), ue = std::chrono::duration<double, std::micro>(t_enq - t0).count();
            printf("%-30s G=%3d V%d %-36s %6.2f us/phase  host enqueue %5.2f us/phase (kernarg %u B, lds %u; checksum errors %u, aborted %u%s)\n",
                   sh.what, sh.G, m.v, m.name, us / ((double)R * K), ue / ((double)R * K), kd.kernarg_size, kd.group_size + m.lds, errs, ab,
                   fin ? "" : ", TIMED OUT");
            fflush(stdout);
            CK(hipFree(dargs));
            if (ab || !fin) { printf("aborted: a poll timed out -- stopping\n"); return 2; }
        }
    }
    return 0;
}
