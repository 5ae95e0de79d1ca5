// Error plumbing of the C ABI.
#include "common.h"

static thread_local std::string g_last_error;

void jb_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" const char* jb_last_error(void) { return g_last_error.c_str(); }

extern "C" int jb_version(void) { return 1; }

// Streams in distinct priority classes.  The ROCm runtime multiplexes streams of one priority class onto a small pool
// of hardware queues (two equal-priority decode chains were measured to serialise completely), whereas each priority
// class has its own queues -- so the level pipeline asks for one stream per class (tools/bench_concurrent.py).
extern "C" int jb_stream_priority_range(int* least, int* greatest) {
    JB_REQUIRE(least && greatest, "null pointer");
    JB_HIP(hipDeviceGetStreamPriorityRange(least, greatest));
    return JB_OK;
}

extern "C" int jb_stream_create(int priority, void** stream) {
    JB_REQUIRE(stream, "null pointer");
    hipStream_t s = nullptr;
    JB_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
    *stream = (void*)s;
    return JB_OK;
}

extern "C" int jb_stream_destroy(void* stream) {
    if (stream) JB_HIP(hipStreamDestroy((hipStream_t)stream));
    return JB_OK;
}

// A stream whose kernels may only run on the compute units whose bit is set in `cu_mask` (n_words x 32 bits).  The decode
// chain is latency-bound and occupies at most ~120 of the 256 CUs; throughput work that shares the GPU with it (another
// level's chain, a look-ahead prefill) is confined to its own CUs so that its workgroups never sit where the chain's next
// kernel wants to start.  How mask bits map to (XCD, CU) is not documented: jb_cu_census reports where a launch landed.
extern "C" int jb_stream_create_cu_mask(const uint32_t* cu_mask, int n_words, void** stream) {
    JB_REQUIRE(cu_mask && stream && n_words > 0, "null pointer / empty mask");
    hipStream_t s = nullptr;
    JB_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask));
    *stream = (void*)s;
    return JB_OK;
}

__global__ void cu_census_kernel(uint32_t* out) {
    if (threadIdx.x == 0) {
        const uint32_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw_id;
    }
    // stay resident for a moment so that the grid spreads over every CU the queue may use
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

// Diagnostic: launches n_blocks workgroups of 64 threads on `stream`; out[2 b] = XCC id, out[2 b + 1] = HW_ID of block b.
extern "C" int jb_cu_census(int n_blocks, uint32_t* out, void* stream) {
    JB_REQUIRE(out && n_blocks > 0, "null pointer / empty grid");
    cu_census_kernel<<<n_blocks, 64, 0, (hipStream_t)stream>>>(out);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

__global__ void clock_probe_kernel(long long* out, int spin_ticks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long w0 = wall_clock64(), c0 = clock64();
        while (wall_clock64() - w0 < spin_ticks) {}
        out[0] = clock64() - c0;          // shader-clock cycles
        out[1] = wall_clock64() - w0;     // constant 100 MHz ticks
    }
}

// Diagnostic: shader clock actually running while `stream` executes -- out[0] shader cycles and out[1] 100-MHz ticks over a
// spin of `spin_ticks` ticks (device array of 2 int64).  A latency-bound kernel chain lives or dies by this clock.
extern "C" int jb_clock_probe(long long* out, int spin_ticks, void* stream) {
    JB_REQUIRE(out && spin_ticks > 0, "null pointer / empty spin");
    clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out, spin_ticks);
    JB_CHECK_LAUNCH();
    return JB_OK;
}
