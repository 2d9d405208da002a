// What does an LDS read beyond the workgroup's allocation return on gfx950?  (The block-major DFIRE kernel could drop a clamp
// per pair if such reads return 0.)   usage: lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BYTES>
__global__ void probe(const unsigned *offsets, unsigned *out_u8, unsigned *out_b32, unsigned long long *out_b64) {
    __shared__ unsigned char lds[BYTES];
    for (int i = threadIdx.x; i < BYTES; i += blockDim.x) lds[i] = 0xA5;
    __syncthreads();
    const unsigned off = offsets[threadIdx.x];
    unsigned a, b;
    unsigned long long c;
    // address = base of `lds` + off, computed so the compiler cannot reason about the bound
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + off;
    asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr));
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr & ~3u));
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(addr & ~7u));
    out_u8[blockIdx.x * 64 + threadIdx.x] = a;
    out_b32[blockIdx.x * 64 + threadIdx.x] = b;
    out_b64[blockIdx.x * 64 + threadIdx.x] = c;
}

int main() {
    unsigned h_off[64];
    const unsigned list[16] = {0, 100, 4095, 4096, 4100, 8192, 16384, 40000, 65535, 65536, 100000, 163839, 163840, 200000, 1u << 20, 0xfffffff0u};
    for (int i = 0; i < 64; i++) h_off[i] = list[i % 16];
    unsigned *d_off, *d_a, *d_b; unsigned long long *d_c;
    CHECK(hipMalloc(&d_off, 256)); CHECK(hipMalloc(&d_a, 64 * 4 * 4)); CHECK(hipMalloc(&d_b, 64 * 4 * 4)); CHECK(hipMalloc(&d_c, 64 * 8 * 4));
    CHECK(hipMemcpy(d_off, h_off, 256, hipMemcpyHostToDevice));
    // four workgroups so that several allocations sit next to each other on a CU
    hipLaunchKernelGGL((probe<4096>), dim3(4), dim3(64), 0, 0, d_off, d_a, d_b, d_c);
    CHECK(hipDeviceSynchronize());
    unsigned a[256], b[256]; unsigned long long c[256];
    CHECK(hipMemcpy(a, d_a, sizeof a, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b, d_b, sizeof b, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(c, d_c, sizeof c, hipMemcpyDeviceToHost));
    std::printf("workgroup allocation 4096 bytes filled with 0xA5; reads at base + offset:\n");
    for (int wg = 0; wg < 4; wg += 3)
        for (int i = 0; i < 16; i++)
            std::printf("  wg %d offset %10u: u8 %02x  b32 %08x  b64 %016llx\n", wg, list[i], a[wg * 64 + i], b[wg * 64 + i], c[wg * 64 + i]);
    return 0;
}
