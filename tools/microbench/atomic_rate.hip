// How fast does an MI355X retire 64-bit atomic adds WITHOUT return value (device scope) issued by all CUs?
// N atomics from 2048 persistent waves to `span` distinct 8-byte addresses, uniformly random per lane.
// usage: atomic_rate [millions of atomics] -> time and rate for span = 8192, 65536, 2^21; and for 32-bit adds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <typename T>
__global__ __launch_bounds__(512) void k(T *p, unsigned span, unsigned iters) {
    const unsigned tid = blockIdx.x * 512 + threadIdx.x;
    for (unsigned i = 0; i < iters; i++) {
        const unsigned a = mix(tid * 2654435761u + i * 40503u) % span;
        __hip_atomic_fetch_add(p + a, (T)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: no return
    }
}
template <typename T>
static void run(const char *name, unsigned span, double millions) {
    T *p; CHECK(hipMalloc(&p, (size_t)span * sizeof(T))); CHECK(hipMemset(p, 0, (size_t)span * sizeof(T)));
    const unsigned threads = 256 * 512;
    const unsigned iters = (unsigned)(millions * 1e6 / threads);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(512), 0, 0, p, span, 16u);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(512), 0, 0, p, span, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%s span %8u: %.1f M atomics in %.3f ms = %.1f G/s\n", name, span, iters * (double)threads / 1e6, ms, iters * (double)threads / ms / 1e6);
    CHECK(hipFree(p));
}
int main(int argc, char **argv) {
    const double m = argc > 1 ? std::atof(argv[1]) : 48.0;
    for (unsigned span : {8192u, 65536u, 1u << 21}) run<unsigned long long>("u64 add", span, m);
    for (unsigned span : {8192u, 1u << 21}) run<unsigned>("u32 add", span, m);
    return 0;
}
