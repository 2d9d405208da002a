// LDS address arithmetic on gfx950: what does `ds_read_b64 v, vaddr offset:K` return when vaddr = 0xFFFFFFF8 (= -8)?
// (a) the sum wraps mod 2^32 -> the 8 bytes in front of K; (b) out of range -> 0.  The block-major DFIRE pair kernel
// turns a LUT code c into a table address by `v_sub_co_u32 x, mask, c, 8` (borrow = "flagged cell") and reads
// `ds_read_b64 tv, x offset:ROW`: a flagged pair must read 0.0 either way.  Also: ds_read_u16 at an odd address.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(unsigned long long *out, unsigned *out16, unsigned long long *masks) {
    __shared__ unsigned long long lds[2048];   // 16 KB at LDS address 0
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 0x1111000000000000ull + i;
    __syncthreads();
    unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned long long *)lds;   // 0: the only LDS variable
    unsigned code = (threadIdx.x & 7) * 8;   // 0, 8, 16, ...: code 0 = "flagged"
    unsigned x; unsigned long long m, a, b, c;
    asm volatile("v_sub_co_u32 %0, %1, %2, 8" : "=v"(x), "=s"(m) : "v"(code));
    asm volatile("ds_read_b64 %0, %1 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(x), "v"(base) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(x), "v"(base) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:65528\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(x), "v"(base) : "memory");
    out[threadIdx.x * 3] = a; out[threadIdx.x * 3 + 1] = b; out[threadIdx.x * 3 + 2] = c;
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<unsigned char *>(lds)[i] = (unsigned char)i;
    __syncthreads();
    unsigned odd = 2 * threadIdx.x + 1, r16, r16b;
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r16) : "v"(odd), "v"(base) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r16b) : "v"(odd), "v"(base) : "memory");
    out16[threadIdx.x * 2] = r16; out16[threadIdx.x * 2 + 1] = r16b;
    if (threadIdx.x == 0) { masks[0] = m; masks[1] = base; }
}

int main() {
    unsigned long long *d_out, *d_m; unsigned *d_16;
    CHECK(hipMalloc(&d_out, 64 * 3 * 8)); CHECK(hipMalloc(&d_16, 64 * 2 * 4)); CHECK(hipMalloc(&d_m, 16));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_16, d_m);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[192], m; unsigned h16[128];
    CHECK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h16, d_16, sizeof h16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&m, d_m, 8, hipMemcpyDeviceToHost));
    std::printf("LDS qword i holds 0x1111000000000000 + i; allocation 16384 bytes\n");
    std::printf("borrow mask of v_sub_co_u32 x, m, code, 8 (code = (lane & 7) * 8): %016llx (expect 0101010101010101)\n", m);
    for (int l = 0; l < 3; l++)
        std::printf("lane %d: x = code - 8 = %d: offset 4096 -> %016llx (wrap: qword %d)   offset 0 -> %016llx   offset 65528 -> %016llx\n", l, l * 8 - 8, h[l * 3], 512 + l - 1, h[l * 3 + 1], h[l * 3 + 2]);
    std::printf("ds_read_u16 at byte 1: %04x, at byte 3: %04x (byte i holds i: unaligned = 0201 / 0403, low bit dropped = 0100 / 0302)\n", h16[0], h16[2]);
    std::printf("ds_read_u16 at byte 1 offset:1: %04x, lane 1 (byte 3+1): %04x\n", h16[1], h16[3]);
    return 0;
}
