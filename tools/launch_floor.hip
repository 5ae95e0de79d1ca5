// Micro-benchmark: cost of a dependent kernel boundary on this box (eager stream vs hipGraph replay).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_trivial(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1; }
__global__ void k_touch(float* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* d; float* a; const int n = 1 << 16;
    CK(hipMalloc(&d, 4)); CK(hipMalloc(&a, n * 4)); CK(hipMemset(d, 0, 4)); CK(hipMemset(a, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int K = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        auto launch = [&](hipStream_t st) {
            if (variant == 0) k_trivial<<<1, 64, 0, st>>>(d);
            else if (variant == 1) k_trivial<<<256, 256, 0, st>>>(d);
            else k_touch<<<n / 256, 256, 0, st>>>(a, n);
        };
        for (int i = 0; i < 100; ++i) launch(s);
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int i = 0; i < K; ++i) launch(s);
        CK(hipStreamSynchronize(s));
        double eager = (now() - t0) / K * 1e6;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 400; ++i) launch(s);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        t0 = now();
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double graph = (now() - t0) / (5 * 400) * 1e6;
        // null (legacy) stream replay, as torch's default stream
        t0 = now();
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, 0));
        CK(hipDeviceSynchronize());
        double graph0 = (now() - t0) / (5 * 400) * 1e6;
        printf("variant %d: eager %.2f us/kernel, graph %.2f us/kernel, graph on null stream %.2f us/kernel\n", variant, eager, graph, graph0);
    }
    return 0;
}
