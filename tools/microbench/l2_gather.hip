// What the L2s deliver to a gather like the DFIRE kernel's: 8-byte entries at random 128-byte lines of a 5.5 MB
// table (L2 resident), `share` lanes of a wave per line, W waves per CU, two loads in flight per wave.
// Build: hipcc -O2 --offload-arch=gfx950 l2_gather.hip -o l2_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(64) void gather(const double *table, unsigned n_lines, int iters, int share, double *out) {
    const unsigned lane = threadIdx.x, group = lane / share;
    unsigned seed = (blockIdx.x * 64u + group) * 2654435761u + 12345u;
    double acc = 0.0, p0 = 0.0, p1 = 0.0;
    for (int i = 0; i < iters; i++) {
        seed = mix(seed + i);
        const unsigned l0 = seed % n_lines, l1 = mix(seed) % n_lines;
        acc += p0;
        acc += p1;
        p0 = table[(size_t)l0 * 16 + (lane & 15)];
        p1 = table[(size_t)l1 * 16 + ((lane + 5) & 15)];
    }
    acc += p0 + p1;
    if (acc == 1.2345) out[0] = acc;
}

int main(int argc, char **argv) {
    const unsigned n_lines = (argc > 1 ? (unsigned)(std::atof(argv[1]) * 1e6) : 5500000u) / 128;   // table size in MB
    std::printf("table %.2f MB\n", n_lines * 128 / 1e6);
    std::vector<double> h((size_t)n_lines * 16, 1.0);
    double *table, *out;
    CHECK(hipMalloc(&table, h.size() * 8)); CHECK(hipMalloc(&out, 8));
    CHECK(hipMemcpy(table, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int waves_per_cu : {argc > 2 ? std::atoi(argv[2]) : 8, 24}) {
        for (int share : {1, 2, 4}) {
            const int blocks = 256 * waves_per_cu;
            hipLaunchKernelGGL(gather, dim3(blocks), dim3(64), 0, 0, table, n_lines, 50, share, out);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather, dim3(blocks), dim3(64), 0, 0, table, n_lines, iters, share, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double lines = (double)blocks * iters * 2.0 * ((64 + share - 1) / share);   // distinct lines asked for, at most
            const double lanes = (double)blocks * iters * 2.0 * 64;
            std::printf("waves/CU %2d  lanes per line %d: %7.3f ms  %6.2f TB/s of 128-byte lines  %6.1f G entries/s\n", waves_per_cu, share, ms,
                        lines * 128 / (ms * 1e-3) / 1e12, lanes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
