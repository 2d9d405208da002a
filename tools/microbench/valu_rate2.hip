// Issue cost of more gfx950 vector / LDS instructions (companion of valu_rate.hip): every SIMD runs WPS waves, every wave a
// loop of 16 copies of the instruction.  Build: hipcc -O2 --offload-arch=gfx950 valu_rate2.hip -o valu_rate2
// usage: valu_rate2 [waves per SIMD = 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define REP16(S) S S S S S S S S S S S S S S S S
typedef float v2f __attribute__((ext_vector_type(2)));

struct Op { const char *name; };

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float *out) {
    __shared__ unsigned char lds[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<unsigned *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    float a = threadIdx.x * 1.0f, b = 1.5f, c = 2.5f;
    double d = threadIdx.x * 1.0, e = 1.25;
    unsigned u = threadIdx.x, w = 3, x = 5;
    unsigned long long q = threadIdx.x * 77ull;
    v2f p = {a, b}, r = {c, a};
    unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + ((threadIdx.x * 2654435761u) >> 18 & 0x3ff8u);
    unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + ((threadIdx.x & 63) * 8u);
    unsigned lu = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;   // uniform (broadcast)
    unsigned t0 = 1, t1; unsigned long long t2 = 12345;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 2) { REP16(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 3) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 6) { REP16(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 7) { REP16(asm volatile("v_or_b32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 8) { REP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 9) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u));) }
        if (OP == 10) { REP16(asm volatile("v_max_u32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 11) { REP16(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 12) { REP16(asm volatile("v_min_i32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 13) { REP16(asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(x));) }
        if (OP == 14) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(x));) }
        if (OP == 15) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u) : "v"(w));) }
        if (OP == 16) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(x));) }
        if (OP == 17) { REP16(asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u));) }
        if (OP == 18) { REP16(asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a) : "v"(u));) }
        if (OP == 19) { REP16(asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(w));) }
        if (OP == 20) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u) : "v"(w) : "vcc");) }
        if (OP == 21) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(r));) }
        if (OP == 22) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(x));) }
        if (OP == 23) { REP16(asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a) : "v"(b));) }
        if (OP == 24) { REP16(asm volatile("v_max_i32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 25) { REP16(asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(u));) }
        if (OP == 26) { REP16(asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(x));) }
        if (OP == 27) { REP16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 28) { REP16(asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(a));) }
        if (OP == 29) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (OP == 30) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p) : "v"(r));) }
        if (OP == 31) { REP16(asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 32) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "s"(c));) }
        if (OP == 33) { REP16(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 34) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u) : "v"(w));) }
        if (OP == 35) { REP16(asm volatile("v_cmp_gt_i32 vcc, 0, %0" : : "v"(u) : "vcc");) }
        if (OP == 36) { REP16(asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a) : "v"(b));) }
        if (OP == 37) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "s"(b));) }
        if (OP == 38) { REP16(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 40) { REP16(asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(q) : "v"(t2));) }
        if (OP == 41) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(u), "+v"(w) : "v"(x) : "vcc");) }
        if (OP == 42) { REP16(asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(q));) }
        if (OP == 39) { REP16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a));) }
        // LDS: 16 reads then one wait
        if (OP == 50) { REP16(asm volatile("ds_read_u8 %0, %1" : "=v"(t0) : "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); u += t0; }
        if (OP == 51) { REP16(asm volatile("ds_read_b64 %0, %1" : "=v"(t2) : "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); q += t2; }
        if (OP == 52) { REP16(asm volatile("ds_read_b64 %0, %1" : "=v"(t2) : "v"(lb));) asm volatile("s_waitcnt lgkmcnt(0)"); q += t2; }
        if (OP == 53) { REP16(asm volatile("ds_read_b64 %0, %1" : "=v"(t2) : "v"(lu));) asm volatile("s_waitcnt lgkmcnt(0)"); q += t2; }
        if (OP == 54) { REP16(asm volatile("ds_read_b32 %0, %1" : "=v"(t0) : "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); u += t0; }
        if (OP == 55) { REP16(asm volatile("ds_read_u16 %0, %1" : "=v"(t0) : "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); u += t0; }
        if (OP == 56) { REP16(asm volatile("ds_read_i8 %0, %1" : "=v"(t0) : "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); u += t0; }
        if (OP == 57) { REP16(asm volatile("ds_read_b128 %0, %1" : "=v"(*(uint4*)&lds[0]) : "v"(lb));) asm volatile("s_waitcnt lgkmcnt(0)"); }
        // a mix like the block-major pair step: ds_read_u8 + v_add_u32 + ds_read_b64 + v_add_f64 (dependent through LDS)
        if (OP == 60) { REP16(asm volatile("ds_read_u8 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, %3\n\tds_read_b64 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f64 %4, %4, %1" : "=&v"(t0), "=&v"(t2), "+v"(la), "+v"(lu), "+v"(d));) }
    }
    if (a + p.x + p.y + (float)d + (float)u + (float)q + (float)t0 + (float)t2 == 12345.678f) out[0] = a;
}

static int g_wps = 8;
template <int OP>
double run(const char *name, double base) {
    const int iters = 2000;
    float *out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * g_wps;
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, 100, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms * 1e-3 / ((double)iters * 16 * g_wps);  // seconds per instruction per SIMD
    std::printf("%-28s %8.3f ms  %6.2f ns/instr/SIMD  x%.2f of v_add_f32\n", name, ms, per * 1e9, base > 0 ? per / base : 1.0);
    CHECK(hipFree(out));
    return per;
}

int main(int argc, char **argv) {
    if (argc > 1) g_wps = std::atoi(argv[1]);
    std::printf("# %d waves per SIMD (LDS 32 KB per workgroup of 4 waves limits this to 5 workgroups per CU)\n", g_wps);
    const double b = run<0>("v_add_f32", 0);
#define R(n, s) run<n>(s, b)
    R(1, "v_fma_f32"); R(2, "v_fmac_f32"); R(3, "v_mul_f32"); R(4, "v_sub_f32"); R(36, "v_subrev_f32"); R(23, "v_add_f32 |a|,-b (VOP3)"); R(32, "v_fma_f32 (sgpr)"); R(37, "v_mul_f32 (sgpr)");
    R(11, "v_max_f32"); R(31, "v_min_f32"); R(38, "v_med3_f32");
    R(5, "v_add_u32"); R(6, "v_sub_u32"); R(20, "v_add_co_u32"); R(7, "v_or_b32"); R(27, "v_and_b32"); R(8, "v_xor_b32"); R(9, "v_lshlrev_b32"); R(25, "v_ashrrev_i32");
    R(10, "v_max_u32"); R(24, "v_max_i32"); R(12, "v_min_i32"); R(26, "v_max3_u32"); R(13, "v_or3_b32"); R(14, "v_and_or_b32"); R(34, "v_lshl_or_b32");
    R(15, "v_alignbit_b32"); R(16, "v_perm_b32"); R(17, "v_bfe_u32"); R(22, "v_mad_u32_u24"); R(33, "v_pk_add_u16");
    R(18, "v_cvt_f32_u32"); R(28, "v_cvt_u32_f32"); R(39, "v_cvt_f64_f32"); R(19, "v_mov_b32"); R(35, "v_cmp_gt_i32");
    R(40, "v_lshl_add_u64"); R(41, "v_add_co_u32 + v_addc_co_u32"); R(42, "v_ashrrev_i64");
    R(21, "v_pk_mul_f32"); R(30, "v_pk_fma_f32"); R(29, "v_add_f64");
    R(50, "ds_read_u8 random"); R(56, "ds_read_i8 random"); R(55, "ds_read_u16 random"); R(54, "ds_read_b32 random"); R(51, "ds_read_b64 random");
    R(52, "ds_read_b64 lane-linear"); R(53, "ds_read_b64 broadcast"); R(60, "lut+add+table+add_f64 chain");
    return 0;
}
