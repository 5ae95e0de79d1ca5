// Probe: a prefill-GEMM candidate for gfx950 against the library's gemm_lds_kernel, on the prefill's own shapes.
//
// Why: the prefill projections (16 samples x 2048 positions = 32768 rows, K = 1920, J = 1440 .. 2880) run at 220-270 TFLOP/s
// through gemm_lds_kernel (profiles/r03_full_job_kernel_stats.csv: 135.7 us per call) -- a 256 x 128 tile whose operands go
// global -> registers -> ds_write_b128 -> LDS, one barrier per 32 channels, two workgroups per CU.  What the CDNA4 guide
// measures for this problem class (cdna_hip_programming.md section 5, "optimization ladder"): the same MFMA with a 128 x 128
// tile and 16-byte LDS-DMA staging reaches 874-912 TFLOP/s at 4096^3.  This file is that structure on OUR data layout:
//   * 128 x 128 output tile, 4 waves as 2 x 2, each 64 rows x 64 columns = acc[4][4] of 16x16x32 MFMA tiles (64 VGPRs);
//   * 64 channels per K-step; both operands by `global_load_lds_dwordx4`: activations as 8-row x 128-byte pieces into an
//     XOR-swizzled image (jukebox_amd/csrc/gemm_glds_index.h; operand reads conflict free, checked on the CPU by
//     tests/test_gemm_glds_index.py), weights straight from the packed MFMA-order image (one 1-KiB tile per instruction);
//   * NBUF = 1: load, barrier, multiply, barrier (32 KiB of LDS: the other resident workgroups of the CU cover the load);
//     NBUF = 2: the next K-step's loads are issued before the multiply (64 KiB), one barrier per K-step;
//   * XCD-aware tile order (a 128-row panel stays on one XCD).
// Same MFMA instruction, same k order per output element as the library kernel: the two outputs must be EQUAL bit for bit
// (checked), and both are checked against an fp64 host sum on sampled elements.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I tools tools/gemm_glds_probe.hip \
//             -L jukebox_amd/csrc -ljukebox_hip -Wl,-rpath,$PWD/jukebox_amd/csrc -o tools/gemm_glds_probe
// Run:    tools/gemm_glds_probe            (prints one line per shape and variant)
// History: written at the end of round 3, first run in round 4 (gemm_glds_kernel went into the library: profiles/r04_gemm_glds_probe.log);
// round 5 added the 8-phase 256 x 256 kernel (jukebox_amd/csrc/gemm_8phase.h: the K-loop the library's gemm_8phase_kernel shares) with
// the wave groups in lock step or half a phase apart, the library's own gemm_8phase_kernel (line-wise epilogue) timed through jb_gemm,
// and ablations of the loop (no LDS-DMA / no operand reads / neither / no MFMAs / barriers only): profiles/r05_gemm_8phase_probe*.log.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#include <type_traits>

#include "../jukebox_amd/csrc/gemm_8phase.h"
#include "../jukebox_amd/csrc/gemm_glds_index.h"
#include "jukebox_hip.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HIP_OK(x)                                                                        \
    do {                                                                                 \
        hipError_t e__ = (x);                                                            \
        if (e__ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

struct Params {
    const f16* A; long long lda; long long m_total;
    const f16* W; int nkt, njt;                 // packed image P[jt][kt][lane][8]
    const float* bias;
    f16* out; long long ldo; int J;
    int MB, NB;                                 // 128-row panels, 128-column tiles
    int stagger;                                // 8-phase kernel: the two wave groups half a phase apart (1) or in lock step (0)
};

extern __shared__ __attribute__((aligned(1024))) unsigned char s_raw[];

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

template <int NBUF>
__global__ __launch_bounds__(256, NBUF == 1 ? 3 : 2) void gemm_glds_kernel(Params p) {
    using namespace gi;
    int mp, nt;
    if (!tile_of_block((int)blockIdx.x, p.MB, p.NB, &mp, &nt)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wave_m = wave >> 1, wave_n = wave & 1;
    const long long m0 = (long long)mp * BM;
    const int jt0 = nt * BJT;

    // sources of this wave's 4 activation pieces and 4 weight tiles (per K-step: + 64 channels / + 2 k-tiles)
    const f16* a_src[4];
    const f16* w_src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int piece = wave * 4 + u;
        long long row = m0 + piece * 8 + a_src_row(lane);
        row = row < p.m_total ? row : p.m_total - 1;              // rows past the end are never stored
        a_src[u] = p.A + row * p.lda + a_src_seg(lane) * 8;
        const int t = wave * 4 + u, jt = t >> 1;
        const int jtg = jt0 + jt < p.njt ? jt0 + jt : p.njt - 1;
        w_src[u] = p.W + ((long long)jtg * p.nkt) * 512 + lane * 8;
    }
    const int ksteps = (p.nkt + 1) >> 1;
    auto issue = [&](int s, int buf) {
        unsigned char* st = s_raw + buf * STAGE_BYTES;
        const bool tail = 2 * s + 1 >= p.nkt;                    // odd number of k-tiles: the step's second half does not exist
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // the segment a lane fetches is (lane & 7) ^ (lane >> 3); in a tail step segments 4..7 are folded onto 0..3 (their
            // slots are never read)
            const f16* src = a_src[u] + (long long)s * KSTEP;
            if (tail && a_src_seg(lane) >= 4) src -= 32;
            glds16(src, st + (wave * 4 + u) * 1024);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = u & 1;                                 // t = wave*4 + u  ->  ks = t & 1 = u & 1
            const int kt = (tail && ks) ? 2 * s : 2 * s + ks;
            glds16(w_src[u] + (long long)kt * 512, st + A_BYTES + (wave * 4 + u) * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto multiply = [&](int s, int buf) {
        const unsigned char* st = s_raw + buf * STAGE_BYTES;
        const int n_ks = 2 * s + 1 < p.nkt ? 2 : 1;
        for (int ks = 0; ks < n_ks; ++ks) {
            f16x8 af[4], wf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                af[mt] = *reinterpret_cast<const f16x8*>(st + a_byte(frag_row(wave_m, mt, lane) , frag_seg(ks, lane)));
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const f16x8*>(st + w_byte(wave_n * 4 + j, ks, lane));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[mt], acc[j][mt], 0, 0, 0);
        }
    };

    if (NBUF == 1) {
        for (int s = 0; s < ksteps; ++s) {
            issue(s, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            multiply(s, 0);
            __syncthreads();
        }
    } else {
        issue(0, 0);
        for (int s = 0; s < ksteps; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of step s has landed ...
            __syncthreads();                                        // ... everybody's has, and nobody still reads the other buffer
            if (s + 1 < ksteps) issue(s + 1, (s + 1) & 1);
            multiply(s, s & 1);
        }
    }

    // D[j][m]: lane l holds row (l & 15) of the 16-row tile and columns (l >> 4) * 4 .. + 3 of the 16-column tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const long long row = m0 + wave_m * 64 + mt * 16 + (lane & 15);
        if (row >= p.m_total) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jb = (jt0 + wave_n * 4 + j) * 16 + (lane >> 4) * 4;
            if (jb >= p.J) continue;
            f16 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[j][mt][r];
                if (p.bias && jb + r < p.J) x += (float)(f16)p.bias[jb + r];      // the library's epilogue: bias rounded to the output type
                v[r] = (f16)x;
            }
            f16* dst = p.out + row * p.ldo + jb;
            if (jb + 3 < p.J) {
                *reinterpret_cast<f16x4*>(dst) = f16x4{v[0], v[1], v[2], v[3]};
            } else {
                for (int r = 0; r < 4 && jb + r < p.J; ++r) dst[r] = v[r];
            }
        }
    }
}

// The same structure with the tile as a parameter: WM x WN waves, each MT x 4 MFMA tiles (16 MT rows x 64 columns); two LDS
// buffers.  <4,2,2> is the 128 x 128 kernel above; <8,2,4> is a 256 x 256 tile on 8 waves (128 KiB of LDS, one workgroup per
// CU: 12 fragment reads per 32 MFMAs instead of 8 per 16, half the LDS-DMA bytes per flop); <4,4,2> a 256 x 128 tile on 8 waves.
template <int MT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 1) void gemm_glds_tile_kernel(Params p) {
    using namespace gi;
    constexpr int NWV = WM * WN, TBM = WM * MT * 16, TBJT = WN * 4;
    constexpr int AP = TBM / 8, WT = TBJT * 2, A_B = AP * 1024, STAGE = A_B + WT * 1024;
    constexpr int APW = AP / NWV, WTW = WT / NWV;
    static_assert(AP % NWV == 0 && WT % NWV == 0, "pieces must divide among the waves");
    const int MBt = (int)((p.m_total + TBM - 1) / TBM), NBt = (p.njt + TBJT - 1) / TBJT;
    int mp, nt;
    if (!tile_of_block((int)blockIdx.x, MBt, NBt, &mp, &nt)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wave_m = wave / WN, wave_n = wave % WN;
    const long long m0 = (long long)mp * TBM;
    const int jt0 = nt * TBJT;
    const f16* a_src[APW];
    const f16* w_src[WTW];
#pragma unroll
    for (int u = 0; u < APW; ++u) {
        long long row = m0 + (wave * APW + u) * 8 + a_src_row(lane);
        row = row < p.m_total ? row : p.m_total - 1;
        a_src[u] = p.A + row * p.lda + a_src_seg(lane) * 8;
    }
#pragma unroll
    for (int u = 0; u < WTW; ++u) {
        const int jt = (wave * WTW + u) >> 1;
        const int jtg = jt0 + jt < p.njt ? jt0 + jt : p.njt - 1;
        w_src[u] = p.W + ((long long)jtg * p.nkt) * 512 + lane * 8;
    }
    const int ksteps = (p.nkt + 1) >> 1;
    auto issue = [&](int s, int buf) {
        unsigned char* st = s_raw + buf * STAGE;
        const bool tail = 2 * s + 1 >= p.nkt;
#pragma unroll
        for (int u = 0; u < APW; ++u) {
            const f16* src = a_src[u] + (long long)s * KSTEP;
            if (tail && a_src_seg(lane) >= 4) src -= 32;
            glds16(src, st + (wave * APW + u) * 1024);
        }
#pragma unroll
        for (int u = 0; u < WTW; ++u) {
            const int ks = (wave * WTW + u) & 1;
            const int kt = (tail && ks) ? 2 * s : 2 * s + ks;
            glds16(w_src[u] + (long long)kt * 512, st + A_B + (wave * WTW + u) * 1024);
        }
    };
    f32x4 acc[4][MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0, 0);
    for (int s = 0; s < ksteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < ksteps) issue(s + 1, (s + 1) & 1);
        const unsigned char* st = s_raw + (s & 1) * STAGE;
        const int n_ks = 2 * s + 1 < p.nkt ? 2 : 1;
        for (int ks = 0; ks < n_ks; ++ks) {
            f16x8 af[MT], wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const f16x8*>(st + A_B + (((wave_n * 4 + j) * 2 + ks) << 10) + (lane << 4));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[mt] = *reinterpret_cast<const f16x8*>(st + a_byte(wave_m * (MT * 16) + mt * 16 + (lane & 15), frag_seg(ks, lane)));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[mt], acc[j][mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long row = m0 + wave_m * (MT * 16) + mt * 16 + (lane & 15);
        if (row >= p.m_total) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jb = (jt0 + wave_n * 4 + j) * 16 + (lane >> 4) * 4;
            if (jb >= p.J) continue;
            f16 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[j][mt][r];
                if (p.bias && jb + r < p.J) x += (float)(f16)p.bias[jb + r];
                v[r] = (f16)x;
            }
            f16* dst = p.out + row * p.ldo + jb;
            if (jb + 3 < p.J) {
                *reinterpret_cast<f16x4*>(dst) = f16x4{v[0], v[1], v[2], v[3]};
            } else {
                for (int r = 0; r < 4 && jb + r < p.J; ++r) dst[r] = v[r];
            }
        }
    }
}

template <int MT, int WM, int WN>
static void launch_tile(const Params& p, hipStream_t s) {
    static bool configured = false;
    constexpr int TBM = WM * MT * 16, TBJT = WN * 4;
    const size_t lds = 2 * (size_t)((TBM / 8) * 1024 + TBJT * 2 * 1024);
    if (!configured) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_tile_kernel<MT, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const int MBt = (int)((p.m_total + TBM - 1) / TBM), NBt = (p.njt + TBJT - 1) / TBJT;
    gemm_glds_tile_kernel<MT, WM, WN><<<(MBt + 7) / 8 * 8 * NBt, WM * WN * 64, lds, s>>>(p);
    HIP_OK(hipGetLastError());
}

// The 8-phase 256 x 256 kernel (jukebox_amd/csrc/gemm_8phase.h holds the K-loop the library kernel shares): K a multiple of 128.
template <int ABL>
__global__ __launch_bounds__(512, 1) void gemm_8phase_probe_kernel(Params p) {
    using namespace g8;
    const int MBt = (int)((p.m_total + BM - 1) / BM), NBt = (p.njt + BJT - 1) / BJT;
    int mp, nt;
    if (!gi::tile_of_block((int)blockIdx.x, MBt, NBt, &mp, &nt)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    const long long m0 = (long long)mp * BM;
    const int jt0 = nt * BJT;
    Sources src;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            long long row = m0 + a_tile_row(h, (wave * 2 + u) * 8 + gi::a_src_row(lane));
            row = row < p.m_total ? row : p.m_total - 1;
            src.a[h][u] = p.A + row * p.lda + gi::a_src_seg(lane) * 8;
            const int ti = wave * 2 + u;
            const int jt = jt0 + b_tile_jt(h, ti);
            src.w[h][u] = p.W + ((long long)(jt < p.njt ? jt : p.njt - 1) * p.nkt + b_tile_ks(ti)) * 512 + lane * 8;
        }
    f4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[j][mt] = f4{0.f, 0.f, 0.f, 0.f};
    mainloop<ABL>(src, p.nkt >> 1, s_raw, acc, p.stagger != 0);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const long long row = m0 + wm * 128 + mt * 16 + (lane & 15);
        if (row >= p.m_total) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jb = (jt0 + wn * 4 + j) * 16 + (lane >> 4) * 4;
            if (jb >= p.J) continue;
            f16 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[j][mt][r];
                if (p.bias && jb + r < p.J) x += (float)(f16)p.bias[jb + r];
                v[r] = (f16)x;
            }
            f16* dst = p.out + row * p.ldo + jb;
            if (jb + 3 < p.J) {
                *reinterpret_cast<f16x4*>(dst) = f16x4{v[0], v[1], v[2], v[3]};
            } else {
                for (int r = 0; r < 4 && jb + r < p.J; ++r) dst[r] = v[r];
            }
        }
    }
}

template <int ABL = 0>
static void launch_8phase(const Params& p, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_8phase_probe_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, g8::LDS_BYTES));
        configured = true;
    }
    const int MBt = (int)((p.m_total + g8::BM - 1) / g8::BM), NBt = (p.njt + g8::BJT - 1) / g8::BJT;
    gemm_8phase_probe_kernel<ABL><<<(MBt + 7) / 8 * 8 * NBt, 512, g8::LDS_BYTES, s>>>(p);
    HIP_OK(hipGetLastError());
}

template <int NBUF>
static void launch(const Params& p, hipStream_t s) {
    static bool configured = false;
    const size_t lds = (size_t)NBUF * gi::STAGE_BYTES;
    if (!configured) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const int grid = (p.MB + 7) / 8 * 8 * p.NB;
    gemm_glds_kernel<NBUF><<<grid, 256, lds, s>>>(p);
    HIP_OK(hipGetLastError());
}

static double time_ms(const std::function<void()>& f, int iters) {
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    HIP_OK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) f();
    HIP_OK(hipEventRecord(b, nullptr));
    HIP_OK(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    HIP_OK(hipEventDestroy(a));
    HIP_OK(hipEventDestroy(b));
    return ms / iters;
}

static int run_shape(long long M, int K, int J) {
    std::mt19937 rng(1234u + (unsigned)K * 7u + (unsigned)J);
    std::uniform_real_distribution<float> ua(-1.f, 1.f);
    std::vector<f16> hA((size_t)M * K), hW((size_t)K * J);
    std::vector<float> hb(J);
    for (auto& v : hA) v = (f16)ua(rng);
    for (auto& v : hW) v = (f16)(ua(rng) * 0.05f);
    for (auto& v : hb) v = ua(rng) * 0.1f;
    f16 *dA, *dW, *dP, *o_lib, *o_new;
    float* db;
    const long long packed = jb_packed_weight_bytes(K, J, JB_F16);
    HIP_OK(hipMalloc(&dA, hA.size() * 2));
    HIP_OK(hipMalloc(&dW, hW.size() * 2));
    HIP_OK(hipMalloc(&dP, packed));
    HIP_OK(hipMalloc(&db, J * 4));
    HIP_OK(hipMalloc(&o_lib, (size_t)M * J * 2));
    HIP_OK(hipMalloc(&o_new, (size_t)M * J * 2));
    HIP_OK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, hb.data(), J * 4, hipMemcpyHostToDevice));
    if (jb_pack_weight(dW, JB_F16, J, 1, K, J, dP, JB_F16, nullptr) != 0) { std::fprintf(stderr, "pack: %s\n", jb_last_error()); return 1; }

    jb_gemm_args g = {};
    g.dtype = JB_F16; g.A = dA; g.lda = K; g.W = dP; g.bias = db; g.out = o_lib; g.ldo = J;
    g.n_seq = 1; g.t_in = (int)M; g.t_out = (int)M; g.in_seq_stride = M; g.out_seq_stride = M; g.K = K; g.J = J;
    g.n_taps = 1; g.in_stride = 1; g.out_stride = 1; g.res_scale = 1.f;
    Params p;
    p.A = dA; p.lda = K; p.m_total = M; p.W = dP; p.nkt = (K + 31) / 32; p.njt = (J + 15) / 16; p.bias = db; p.out = o_new; p.ldo = J; p.J = J;
    p.MB = (int)((M + 127) / 128); p.NB = (p.njt + 7) / 8;
    if (K % 32) { std::fprintf(stderr, "K must be a multiple of 32\n"); return 1; }

    const double flop = 2.0 * (double)M * K * J;
    int bad = 0;
    jb_tune_gemm_8phase(-1);
    jb_tune_gemm_glds(-1);
    const double t_lib = time_ms([&] { if (jb_gemm(&g, nullptr) != 0) { std::fprintf(stderr, "jb_gemm: %s\n", jb_last_error()); std::exit(1); } }, 20);
    std::vector<f16> r_lib((size_t)M * J), r_new((size_t)M * J);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(r_lib.data(), o_lib, r_lib.size() * 2, hipMemcpyDeviceToHost));
    std::printf("M=%lld K=%d J=%d  library gemm_lds_kernel (256x128 tile, register staging)   %8.1f us  %7.1f TFLOP/s\n", M, K, J, t_lib * 1e3,
                flop / (t_lib * 1e-3) / 1e12);
    if (K % 128 == 0) {          // the library's own 8-phase kernel (line-wise epilogue through LDS), forced on at any size
        jb_tune_gemm_glds(1);
        jb_tune_gemm_8phase(1);
        g.out = o_new;
        HIP_OK(hipMemset(o_new, 0xff, (size_t)M * J * 2));
        const double t = time_ms([&] { if (jb_gemm(&g, nullptr) != 0) { std::fprintf(stderr, "jb_gemm: %s\n", jb_last_error()); std::exit(1); } }, 20);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(r_new.data(), o_new, r_new.size() * 2, hipMemcpyDeviceToHost));
        long long differ = 0;
        for (size_t i = 0; i < r_new.size(); ++i) differ += (__builtin_bit_cast(unsigned short, r_new[i]) != __builtin_bit_cast(unsigned short, r_lib[i]));
        bad += differ != 0;
        std::printf("M=%lld K=%d J=%d  library gemm_8phase_kernel (256x256, 8 phases, line-wise epilogue)         %8.1f us  %7.1f TFLOP/s   %s (elements differing "
                    "from the library: %lld)\n", M, K, J, t * 1e3, flop / (t * 1e-3) / 1e12, differ == 0 ? "EQUAL" : "MISMATCH", differ);
        g.out = o_lib;
        jb_tune_gemm_8phase(-1);
        jb_tune_gemm_glds(-1);
    }
    for (int nbuf = 1; nbuf <= 2; ++nbuf) {
        HIP_OK(hipMemset(o_new, 0xff, (size_t)M * J * 2));
        const double t = time_ms([&] { if (nbuf == 1) launch<1>(p, nullptr); else launch<2>(p, nullptr); }, 20);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(r_new.data(), o_new, r_new.size() * 2, hipMemcpyDeviceToHost));
        long long differ = 0;
        for (size_t i = 0; i < r_new.size(); ++i) differ += (__builtin_bit_cast(unsigned short, r_new[i]) != __builtin_bit_cast(unsigned short, r_lib[i]));
        // sampled elements against an fp64 host sum (asymmetric operands: a transposed or shifted store cannot pass)
        double worst = 0.0;
        std::mt19937 pick(99);
        for (int sidx = 0; sidx < 2048; ++sidx) {
            const long long r = (long long)(pick() % (unsigned long long)M);
            const int c = (int)(pick() % (unsigned)J);
            double ref = (double)(float)(f16)hb[c];
            for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)r * K + k] * (double)(float)hW[(size_t)k * J + c];
            const double err = std::fabs((double)(float)r_new[(size_t)r * J + c] - ref) / (std::fabs(ref) + 1.0);
            worst = err > worst ? err : worst;
        }
        const bool ok = differ == 0 && worst < 4e-3;
        bad += !ok;
        std::printf("M=%lld K=%d J=%d  128x128 tile, LDS-DMA staging, %d LDS buffer%s                    %8.1f us  %7.1f TFLOP/s   %s (elements differing "
                    "from the library: %lld, worst relative error vs fp64 on 2048 samples: %.2e)\n",
                    M, K, J, nbuf, nbuf == 1 ? " " : "s", t * 1e3, flop / (t * 1e-3) / 1e12, ok ? "EQUAL" : "MISMATCH", differ, worst);
    }
    for (int var = 0; var < 3; ++var) {
        const char* names[3] = {"128x128 / 4 waves (template)", "256x256 / 8 waves", "256x128 / 8 waves"};
        auto go = [&] { if (var == 0) launch_tile<4, 2, 2>(p, nullptr); else if (var == 1) launch_tile<8, 2, 4>(p, nullptr); else launch_tile<4, 4, 2>(p, nullptr); };
        HIP_OK(hipMemset(o_new, 0xff, (size_t)M * J * 2));
        const double t = time_ms(go, 20);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(r_new.data(), o_new, r_new.size() * 2, hipMemcpyDeviceToHost));
        long long differ = 0;
        for (size_t i = 0; i < r_new.size(); ++i) differ += (__builtin_bit_cast(unsigned short, r_new[i]) != __builtin_bit_cast(unsigned short, r_lib[i]));
        bad += differ != 0;
        std::printf("M=%lld K=%d J=%d  %-30s LDS-DMA, 2 LDS buffers              %8.1f us  %7.1f TFLOP/s   %s (elements differing from the library: %lld)\n",
                    M, K, J, names[var], t * 1e3, flop / (t * 1e-3) / 1e12, differ == 0 ? "EQUAL" : "MISMATCH", differ);
    }
    for (int stagger = 1; stagger >= 0 && K % 128 == 0; --stagger) {
        p.stagger = stagger;
        HIP_OK(hipMemset(o_new, 0xff, (size_t)M * J * 2));
        const double t = time_ms([&] { launch_8phase(p, nullptr); }, 20);
        HIP_OK(hipDeviceSynchronize());
        long long differ = 0;
        for (int rep = 0; rep < 3; ++rep) {          // a race shows as a rare wrong tile: compare several runs
            HIP_OK(hipMemcpy(r_new.data(), o_new, r_new.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < r_new.size(); ++i) differ += (__builtin_bit_cast(unsigned short, r_new[i]) != __builtin_bit_cast(unsigned short, r_lib[i]));
            HIP_OK(hipMemset(o_new, 0xff, (size_t)M * J * 2));
            for (int i = 0; i < 5; ++i) launch_8phase(p, nullptr);
            HIP_OK(hipDeviceSynchronize());
        }
        bad += differ != 0;
        std::printf("M=%lld K=%d J=%d  %-30s LDS-DMA, 8 phases, 2 stages         %8.1f us  %7.1f TFLOP/s   %s (elements differing from the library: %lld)\n",
                    M, K, J, stagger ? "256x256 / 8 waves, 2 groups" : "256x256 / 8 waves, lock step", t * 1e3, flop / (t * 1e-3) / 1e12, differ == 0 ? "EQUAL" : "MISMATCH", differ);
    }
    if (K % 128 == 0 && M >= 8192) {        // ablations of the 8-phase loop (results are wrong by construction): what bounds it?
        p.stagger = 1;
        const double t1 = time_ms([&] { launch_8phase<1>(p, nullptr); }, 20);
        const double t2 = time_ms([&] { launch_8phase<2>(p, nullptr); }, 20);
        const double t3 = time_ms([&] { launch_8phase<3>(p, nullptr); }, 20);
        const double t4 = time_ms([&] { launch_8phase<4>(p, nullptr); }, 20);
        const double t7 = time_ms([&] { launch_8phase<7>(p, nullptr); }, 20);
        std::printf("M=%lld K=%d J=%d  8-phase ablations, us: no LDS-DMA in the loop %.1f | no operand reads %.1f | neither %.1f | no MFMAs %.1f | barriers only %.1f\n",
                    M, K, J, t1 * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3, t7 * 1e3);
    }
    HIP_OK(hipFree(dA)); HIP_OK(hipFree(dW)); HIP_OK(hipFree(dP)); HIP_OK(hipFree(db)); HIP_OK(hipFree(o_lib)); HIP_OK(hipFree(o_new));
    return bad;
}

int main() {
    int bad = 0;
    // the prefill's GEMMs: 16 samples x 2048 positions; c_attn (q, k, v), attn.c_proj / v.Wp, mlp.c_fc and mlp.c_proj of the
    // upsamplers, then ragged edges (rows and columns that are not multiples of the tile, an odd number of k-tiles)
    const long long shapes[][3] = {{32768, 1920, 1440}, {32768, 1920, 1920}, {32768, 480, 1920}, {32768, 1920, 2880},
                                   {65536, 1920, 1440}, {65536, 1920, 1920}, {65536, 1920, 3360}, {8192, 2048, 2048}, {8192, 8192, 8192},
                                   {6144, 4800, 4800}, {1000, 480, 200}, {1000, 384, 200}, {130, 128, 72}, {130, 96, 72}};
    for (auto& s : shapes) bad += run_shape(s[0], (int)s[1], (int)s[2]);
    std::printf(bad ? "FAILED: %d variant(s) differ\n" : "all variants equal the library kernel bit for bit\n", bad);
    return bad ? 1 : 0;
}
