// What do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the block-major DFIRE kernels?
// (MI355X_MICROARCH.md: FETCH_SIZE is half the bytes of a 16-byte-per-lane streaming read; other widths are uncalibrated.)
// Each kernel moves a KNOWN number of bytes over arrays far larger than the 256 MiB Infinity Cache; run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -- ./fetch_calib      and again with --pmc WRITE_SIZE
// and divide the counter (KiB) by the printed byte counts.   usage: fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// A: 16 bytes per lane, coalesced
__global__ void stream16(const uint4 *in, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i].x;
    if (acc == 0x12345678u) *sink = acc;
}
// B: 8 bytes per lane, coalesced, read-modify-write (the pair kernel's partial sums: consecutive lanes, consecutive entries)
__global__ void rmw8(unsigned long long *a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += 3ull;
}
// C: 8 bytes per lane at RANDOM places of a big array (the gather's reads of partial sums)
__global__ void gather8(const unsigned long long *a, size_t n, size_t reads, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < reads; i += (size_t)gridDim.x * blockDim.x)
        acc += a[((size_t)mix((unsigned)i) * 2654435761ull) % n];
    if (acc == 0x12345678ull) *sink = acc;
}
// D: 48 bytes per lane (three 16-byte loads) at random 48-byte rows of a big table (an entry's affine map when it came from HBM)
__global__ void rows48(const uint4 *t, size_t rows, size_t reads, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < reads; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 *r = t + (((size_t)mix((unsigned)i) * 2654435761ull) % rows) * 3;
        acc += r[0].x + r[1].y + r[2].z;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// E: 12-byte entries written by scattered lanes: 4 + 8 bytes at random places (the culling kernel's entries)
__global__ void scatter12(unsigned *rows_out, unsigned long long *mask_out, size_t n, size_t writes) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < writes; i += (size_t)gridDim.x * blockDim.x) {
        const size_t at = ((size_t)mix((unsigned)i) * 2654435761ull) % n;
        rows_out[at] = (unsigned)i;
        mask_out[at] = i;
    }
}

int main() {
    const size_t GiB = (size_t)1 << 30;
    void *buf; CHECK(hipMalloc(&buf, 4 * GiB)); CHECK(hipMemset(buf, 1, 4 * GiB));
    void *sink; CHECK(hipMalloc(&sink, 64));
    const dim3 grid(256 * 16), block(256);
    hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const uint4 *)buf, (2 * GiB) / 16, (unsigned *)sink);
    std::printf("stream16   reads %zu bytes\n", 2 * GiB);
    hipLaunchKernelGGL(rmw8, grid, block, 0, 0, (unsigned long long *)buf, GiB / 8);
    std::printf("rmw8       reads %zu bytes, writes %zu bytes\n", GiB, GiB);
    const size_t reads = (size_t)1 << 26;
    hipLaunchKernelGGL(gather8, grid, block, 0, 0, (const unsigned long long *)buf, (4 * GiB) / 8, reads, (unsigned long long *)sink);
    std::printf("gather8    %zu reads of 8 bytes = %zu useful bytes (x4 = %zu in 32-byte sectors, x8 in 64-byte, x16 in 128-byte lines)\n", reads, reads * 8, reads * 32);
    hipLaunchKernelGGL(rows48, grid, block, 0, 0, (const uint4 *)buf, (4 * GiB) / 48, reads / 4, (unsigned *)sink);
    std::printf("rows48     %zu reads of 48 bytes = %zu useful bytes (64-byte granules: x1.33 .. x2.67)\n", reads / 4, reads / 4 * 48);
    hipLaunchKernelGGL(scatter12, grid, block, 0, 0, (unsigned *)buf, (unsigned long long *)((char *)buf + 2 * GiB), (2 * GiB) / 8, reads / 4);
    std::printf("scatter12  %zu entries of 4 + 8 bytes = %zu useful bytes written\n", reads / 4, reads / 4 * 12);
    CHECK(hipDeviceSynchronize());
    return 0;
}
