// Measurement tool (not product): what a COLD streaming launch of B bytes costs inside a chain of dependent launches --
// the ceiling of the 5b_lyrics decode step's projections (11.5 / 34.5 / 46 MB of f16 weights per launch at 3 rows,
// jukebox_amd/csrc/gemm.hip: gemv_long_kernel).  Every launch reads its own buffer (a ring of buffers larger than L2 + MALL),
// 1 KiB per wave request, non-temporal, the workgroup's chunk contiguous as a packed weight tile is; a wave keeps U requests
// in flight in two register stages (as gemv_long_kernel does) or asks for everything up front.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/stream_probe tools/stream_probe.hip      Run: tools/stream_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid = tiles, block = NW * 64; tile jt reads nkt KiB-pieces: wave w the pieces [w * nkt / NW, (w + 1) * nkt / NW)
template <int NW, int NF>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const f32x4* __restrict__ W, int nkt, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kt0 = (wave * nkt) / NW, kt1 = ((wave + 1) * nkt) / NW;
    const f32x4* base = W + ((int64_t)blockIdx.x * nkt) * 64 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 st[2][NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) st[0][i] = __builtin_nontemporal_load(base + (int64_t)min(kt0 + i, nkt - 1) * 64);
#pragma unroll
    for (int i = 0; i < NF; ++i) st[1][i] = __builtin_nontemporal_load(base + (int64_t)min(kt0 + NF + i, nkt - 1) * 64);
    for (int b = 0; kt0 + b * NF < kt1; ++b) {
        const int s = b & 1;
        if (s == 0) {
#pragma unroll
            for (int i = 0; i < NF; ++i) acc += st[0][i];
            __builtin_amdgcn_sched_barrier(0);
            if (kt0 + (b + 2) * NF < kt1) {
#pragma unroll
                for (int i = 0; i < NF; ++i) st[0][i] = __builtin_nontemporal_load(base + (int64_t)min(kt0 + (b + 2) * NF + i, nkt - 1) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int i = 0; i < NF; ++i) acc += st[1][i];
            __builtin_amdgcn_sched_barrier(0);
            if (kt0 + (b + 2) * NF < kt1) {
#pragma unroll
                for (int i = 0; i < NF; ++i) st[1][i] = __builtin_nontemporal_load(base + (int64_t)min(kt0 + (b + 2) * NF + i, nkt - 1) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float v = acc[0] + acc[1] + acc[2] + acc[3];
    if (v == 12345.678f) out[blockIdx.x] = v;      // (never: keeps the loads)
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int NW, int NF>
static double run(const char* buf, size_t ring_bytes, size_t launch_bytes, int tiles, float* out, hipStream_t s, int chain) {
    const int nkt = (int)(launch_bytes / 1024 / tiles);
    const size_t real = (size_t)nkt * 1024 * tiles;
    const int nbuf = (int)(ring_bytes / real);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chain; ++i)
        stream_kernel<NW, NF><<<tiles, NW * 64, 0, s>>>((const f32x4*)(buf + (size_t)(i % nbuf) * real), nkt, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    const double t0 = now();
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = (now() - t0) / (3 * chain) * 1e6;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    printf("  NW=%2d NF=%2d tiles=%5d pieces/tile=%4d  %6.2f MB  %7.2f us  %5.2f TB/s\n", NW, NF, tiles, nkt, real / 1e6, us, real / us * 1e-6);
    return us;
}

int main() {
    const size_t ring = (size_t)3 << 30;
    char* buf; float* out;
    CK(hipMalloc(&buf, ring)); CK(hipMemset(buf, 1, ring)); CK(hipMalloc(&out, 1 << 16));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int chain = 200;
    const double mb[] = {11.52, 34.56, 46.08};
    for (double m : mb) {
        const size_t bytes = (size_t)(m * 1e6);
        printf("launch of %.2f MB (cold, chained):\n", m);
        // the projections' own shape: one tile per 16 output columns -> 300 / 225 tiles; and finer / coarser cuts of the same bytes
        for (int tiles : {225, 300, 512, 600, 1024, 1200, 2400}) {
            run<8, 10>(buf, ring, bytes, tiles, out, s, chain);
            run<8, 5>(buf, ring, bytes, tiles, out, s, chain);
            run<4, 10>(buf, ring, bytes, tiles, out, s, chain);
            run<16, 5>(buf, ring, bytes, tiles, out, s, chain);
        }
    }
    return 0;
}
