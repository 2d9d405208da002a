// Like l2_gather.hip, through a raw buffer like the DFIRE kernel's: only `pct` per cent of the lanes hold an offset inside
// the table, the others one past its end (reads 0.0 without a memory request).  What does an instruction cost when most
// of its lanes ask for nothing?   usage: l2_gather_sparse <table MB> <waves per CU>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(64) void gather(const double *table, unsigned n_lines, int iters, int pct, double *out, int skew) {
    const unsigned lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(table), 0, (int)(n_lines * 128u), 0x00020000);
    unsigned seed = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
    double acc = 0.0, p0 = 0.0, p1 = 0.0;
    for (int i = 0; i < iters; i++) {
        seed = mix(seed + i);
        const unsigned s1 = mix(seed);
        // skew: 88 % of the reads go to the first third of the table (the far distance bins of the DFIRE table)
        const unsigned hot = n_lines / 3u;
        const unsigned l0 = !skew ? seed % n_lines : (seed >> 20) % 100u < 88u ? seed % hot : hot + seed % (n_lines - hot);
        const unsigned l1 = !skew ? s1 % n_lines : (s1 >> 20) % 100u < 88u ? s1 % hot : hot + s1 % (n_lines - hot);
        const unsigned o0 = (seed >> 8) % 100u < (unsigned)pct ? l0 * 128u + (lane & 15) * 8u : 0x40000000u;
        const unsigned o1 = (s1 >> 8) % 100u < (unsigned)pct ? l1 * 128u + ((lane + 5) & 15) * 8u : 0x40000000u;
        acc += p0;
        acc += p1;
        v2u a = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)o0, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)o1, 0, 0);
        p0 = __longlong_as_double((long long)(((unsigned long long)a.y << 32) | a.x));
        p1 = __longlong_as_double((long long)(((unsigned long long)b.y << 32) | b.x));
    }
    acc += p0 + p1;
    if (acc == 1.2345) out[0] = acc;
}

int main(int argc, char **argv) {
    const unsigned n_lines = (argc > 1 ? (unsigned)(std::atof(argv[1]) * 1e6) : 5500000u) / 128;
    const int waves_per_cu = argc > 2 ? std::atoi(argv[2]) : 24;
    const int skew = argc > 3 ? std::atoi(argv[3]) : 0;
    std::vector<double> h((size_t)n_lines * 16, 1.0);
    double *table, *out;
    CHECK(hipMalloc(&table, h.size() * 8)); CHECK(hipMalloc(&out, 8));
    CHECK(hipMemcpy(table, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    std::printf("table %.2f MB, %d waves per CU%s\n", n_lines * 128 / 1e6, waves_per_cu, skew ? ", 88 % of the reads in the first third" : "");
    for (int pct : {100, 30, 15, 0}) {
        const int blocks = 256 * waves_per_cu;
        hipLaunchKernelGGL(gather, dim3(blocks), dim3(64), 0, 0, table, n_lines, 50, pct, out, skew);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather, dim3(blocks), dim3(64), 0, 0, table, n_lines, iters, pct, out, skew);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instrs = (double)blocks * iters * 2.0;
        const double cyc = ms * 1e-3 * 2.4e9 / (instrs / 256.0);   // CU cycles per gather instruction
        std::printf("  %3d %% of the lanes inside: %7.3f ms  %6.1f cycles per instruction per CU  %6.2f TB/s of lines\n", pct, ms, cyc,
                    instrs * 64 * pct / 100.0 * 128 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
