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
