// The K-loop of gemm_8phase_kernel (gemm.hip; stand-alone form: tools/gemm_glds_probe.hip): the prefill's big projections
// (Conv1D.forward at q > 1, jukebox/transformer/ops.py:97-101) on a 256 x 256 output tile, 8 waves as 2 (rows) x 4 (columns),
// each 128 rows x 64 columns = 32 accumulator tiles of v_mfma_f32_16x16x32_f16, 64 channels per K-tile.
//
// What gemm_glds_kernel (128 x 128, one LDS stage, two barriers per K-step, three workgroups per CU) cannot do is keep the matrix
// cores busy while a workgroup loads: all of its waves load, wait, multiply in lock step, and only the OTHER resident workgroups
// cover the gaps (716-745 TFLOP/s = 29 % of the dense fp16 peak on the prefill's shapes).  Here ONE workgroup owns the CU
// (128 KiB of LDS: two stages of four 16-KiB half-tiles) and its two wave groups (the two wave rows: one wave of each per SIMD)
// run half a phase apart -- group 1 takes one extra barrier before the loop -- so that while one group issues its operand reads
// and LDS-DMA requests the other is inside a cluster of 16 MFMAs (raised priority), and vice versa (CDNA4 guide, section 5:
// "the 256^2 8-phase template").  A K-tile is consumed in four phases, one 64 x 32 quadrant of the wave's sub-tile each, in the
// order (A0,B0) (A0,B1) (A1,B1) (A1,B0): a phase reads only what is new -- 12 / 4 / 8 / 0 operand fragments -- and every
// half-tile of a stage is read in exactly ONE phase, which is what lets the next K-tile's half-tiles be requested into a stage
// that is still being consumed:
//
//   phase        1        2        3        4        5        6        7        8
//   ds_read   E.B0 E.A0  E.B1     E.A1      -     O.B0 O.A0  O.B1     O.A1      -          E / O: this iteration's even / odd K-tile
//   LDS-DMA    O.A1     E'.B0    E'.A0    E'.B1    E'.A1    O'.B0    O'.A0    O'.B1        ' : the next iteration's
//   wait                                 vmcnt(6)                            vmcnt(6)      (2 requests per half-tile and wave)
//
// Orderings this schedule relies on (MI355X_MICROARCH.md, wave co-residence item 7: nothing orders a ds_read behind a pending
// LDS-DMA except the issuing wave's covering vmcnt, plus a barrier for the other waves):
//   RAW  phase 4's vmcnt(6) leaves the three youngest half-tiles in flight, i.e. retires O.A1 and everything older: every wave
//        has the odd stage's four half-tiles landed BEFORE its phase-4 first barrier; the first reads of the odd stage are in
//        phase 5, behind that barrier for both groups (group 0's phase-4 second barrier is group 1's phase-4 first).  Same for
//        phase 8 and the even stage.
//   WAR  a phase's operand reads are waited for (lgkmcnt(0)) right behind its first barrier, before its MFMAs: a half-tile read
//        in phase p by the trailing group is retired before the barrier that closes ITS phase p, which the leading group passes
//        inside its phase p + 1 -- a request in phase p + 2 or later cannot overtake a read (A0, B1, A1 are re-requested two
//        phases after their reads).  B0 is re-requested ONE phase after its read: phases 1 and 5 read the weight fragments
//        first and retire them (lgkmcnt(8) of 12 in-order LDS operations) BEFORE their first barrier.
// Same MFMA, same operand roles and the same ascending k order per output element as gemm_lds_kernel / gemm_glds_kernel: the
// three kernels' outputs are equal bit for bit (tools/gemm_glds_probe, tests/test_hip_kernels.py).
//
// The index arithmetic is host-compilable so that the CPU suite can check it exhaustively (tests/test_gemm_glds_index.py).
#pragma once
#include "gemm_glds_index.h"

namespace g8 {
constexpr int BM = 256;                        // rows of the output tile
constexpr int BJT = 16;                        // 16-column tiles of the output tile (256 columns)
constexpr int KTILE = 64;                      // channels per K-tile = two 32-channel k-tiles of the packed weights
constexpr int HALF_BYTES = 16384;              // one half-tile: 128 rows (or columns) x 64 channels of f16
constexpr int STAGE_BYTES = 4 * HALF_BYTES;    // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // 128 KiB

// Byte offset of half-tile (kind 0 = activations / 1 = weights, h) of `stage`.
GI_FN int half_base(int stage, int kind, int h) { return stage * STAGE_BYTES + (kind * 2 + h) * HALF_BYTES; }
// Activation half h holds, for either wave row, the 64 rows of its quadrant row h: local row rl (0..127) -> row of the tile.
GI_FN int a_tile_row(int h, int rl) { return (rl >> 6) * 128 + h * 64 + (rl & 63); }
// ... its image is gi's 128-row image: byte of (local row, 16-byte segment of the 128-byte line)
GI_FN int a_half_byte(int rl, int seg) { return gi::a_byte(rl, seg); }
// Weight half h holds, for each wave column wn, the two 16-column tiles of its quadrant column h, two k-tiles each:
// 1-KiB tile ti (0..15) = ((wn * 2 + jq) << 1) | ks  ->  16-column tile of the 256-column block, k-tile of the K-tile.
GI_FN int b_tile_jt(int h, int ti) { return (ti >> 2) * 4 + h * 2 + ((ti >> 1) & 1); }
GI_FN int b_tile_ks(int ti) { return ti & 1; }
GI_FN int b_half_byte(int wn, int jq, int ks, int lane) { return ((((wn * 2 + jq) << 1) | ks) << 10) + (lane << 4); }
// The wave's sub-tile: accumulator [jt 0..3][mt 0..7]; quadrant (qa, qb) = mt 4 qa .. + 3, jt 2 qb .. + 1.
// Operand read of activation fragment (quadrant row h, mt_l 0..3, ks) by lane `lane` of a wave of wave row wm:
GI_FN int a_frag_byte(int wm, int mt_l, int ks, int lane) { return a_half_byte(wm * 64 + mt_l * 16 + (lane & 15), ks * 4 + (lane >> 4)); }
}  // namespace g8

#if defined(__HIPCC__)
namespace g8 {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Per-lane sources of the wave's LDS-DMA requests, at K-tile 0: a[h][u] the 8-row piece 2 * wave + u of activation half h (the
// lane's row and XOR-ed segment: gi::a_src_row / a_src_seg), w[h][u] the 1-KiB weight tile 2 * wave + u of weight half h (the
// lane's own fragment of the packed image).  K-tile s is + s * 64 halves for a, + s * 1024 halves (two k-tiles) for w.
struct Sources {
    const _Float16* a[2][2];
    const _Float16* w[2][2];
};

__device__ __forceinline__ void dma16(const void* src, unsigned char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// acc[jt][mt] += the tile's product over n_ktiles K-tiles of 64 channels (n_ktiles even, >= 2).  All 512 threads of the
// workgroup call this together; `lds` is the workgroup's one shared array of LDS_BYTES, aligned to 1 KiB.
// ABL (probe only; wrong results): 1 = no LDS-DMA requests inside the loop, 2 = no operand reads inside the loop, 4 = no MFMAs.
template <int ABL = 0>
__device__ __forceinline__ void mainloop(const Sources& src, int n_ktiles, unsigned char* lds, f4 (&acc)[4][8], bool stagger = true) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    // the lane's operand-read bytes relative to a half-tile: activations per ks (+ mt_l * 2048), weights (+ (jq * 2 + ks) * 1024)
    const int a_off0 = a_frag_byte(wm, 0, 0, lane), a_off1 = a_frag_byte(wm, 0, 1, lane);
    const int b_off = b_half_byte(wn, 0, 0, lane);
    const int dma_off = wave * 2048;

    h8 a0[4][2], a1[4][2], b0[2][2], b1[2][2];
    bool in_loop = false;
    auto request = [&](int kind, int h, int s, int stage) {
        if ((ABL & 1) && in_loop) return;
        unsigned char* dst = lds + half_base(stage, kind, h) + dma_off;
        if (kind == 0) {
            dma16(src.a[h][0] + (long long)s * KTILE, dst);
            dma16(src.a[h][1] + (long long)s * KTILE, dst + 1024);
        } else {
            dma16(src.w[h][0] + (long long)s * 1024, dst);
            dma16(src.w[h][1] + (long long)s * 1024, dst + 1024);
        }
    };
    auto read_a = [&](h8 (&a)[4][2], int stage, int h) {
        if ((ABL & 2) && in_loop) return;
        const unsigned char* base = lds + half_base(stage, 0, h);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            a[mt][0] = *reinterpret_cast<const h8*>(base + a_off0 + mt * 2048);
            a[mt][1] = *reinterpret_cast<const h8*>(base + a_off1 + mt * 2048);
        }
    };
    auto read_b = [&](h8 (&b)[2][2], int stage, int h) {
        if ((ABL & 2) && in_loop) return;
        const unsigned char* base = lds + half_base(stage, 1, h) + b_off;
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            b[jq][0] = *reinterpret_cast<const h8*>(base + (jq * 2 + 0) * 1024);
            b[jq][1] = *reinterpret_cast<const h8*>(base + (jq * 2 + 1) * 1024);
        }
    };
    // the phase's second half: everybody's requests of the previous wait are visible behind the barrier, this wave's operand
    // reads have returned; 16 MFMAs on 8 different accumulators per k-tile (k ascending per accumulator)
    // the phase's second half, behind its first barrier: everybody's requests of the previous wait are visible, this wave's
    // operand reads are waited for; 16 MFMAs on 8 different accumulators per k-tile (k ascending per accumulator)
    auto multiply = [&](h8 (&a)[4][2], h8 (&b)[2][2], int qa, int qb) {
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!(ABL & 4))
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jq = 0; jq < 2; ++jq)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[qb * 2 + jq][qa * 4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[jq][ks], a[mt][ks], acc[qb * 2 + jq][qa * 4 + mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // phases 1 and 5: the 4 weight fragments first, then the 8 activation fragments; LDS operations return in order, so
    // lgkmcnt(8) BEFORE the phase's first barrier retires the weight reads -- B0 is the one half-tile that is re-requested in the
    // very next phase
    auto read_ba = [&](int stage) {
        read_b(b0, stage, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(a0, stage, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto retire_b = [&] {
        if (!(ABL & 2)) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    };

    // prologue: K-tile 0 whole, K-tile 1 but for its last half-tile
    request(1, 0, 0, 0); request(0, 0, 0, 0); request(1, 1, 0, 0); request(0, 1, 0, 0);
    request(1, 0, 1, 1); request(0, 0, 1, 1); request(1, 1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1 && stagger) __builtin_amdgcn_s_barrier();          // the trailing group: half a phase behind from here on
    __builtin_amdgcn_sched_barrier(0);

    if (ABL & 2) { read_b(b0, 0, 0); read_a(a0, 0, 0); read_b(b1, 0, 1); read_a(a1, 0, 1); }
    in_loop = true;
    auto iteration = [&](int s, auto last_tag) {          // K-tiles s (even stage) and s + 1 (odd stage)
        constexpr bool LAST = decltype(last_tag)::value;
        read_ba(0);
        request(0, 1, s + 1, 1);
        retire_b();
        multiply(a0, b0, 0, 0);
        read_b(b1, 0, 1);
        if (!LAST) request(1, 0, s + 2, 0);
        multiply(a0, b1, 0, 1);
        read_a(a1, 0, 1);
        if (!LAST) request(0, 0, s + 2, 0);
        multiply(a1, b1, 1, 1);
        if (!LAST) {
            request(1, 1, s + 2, 0);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        multiply(a1, b0, 1, 0);
        read_ba(1);
        if (!LAST) request(0, 1, s + 2, 0);
        retire_b();
        multiply(a0, b0, 0, 0);
        read_b(b1, 1, 1);
        if (!LAST) request(1, 0, s + 3, 1);
        multiply(a0, b1, 0, 1);
        read_a(a1, 1, 1);
        if (!LAST) request(0, 0, s + 3, 1);
        multiply(a1, b1, 1, 1);
        if (!LAST) {
            request(1, 1, s + 3, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        multiply(a1, b0, 1, 0);
    };
    int s = 0;
    for (; s + 2 < n_ktiles; s += 2) iteration(s, std::false_type{});
    iteration(s, std::true_type{});
    if (wm == 0 && stagger) __builtin_amdgcn_s_barrier();          // the leading group takes the barrier the trailing one took first
}
}  // namespace g8
#endif
