// Issue cost of single gfx950 vector instructions, relative to v_add_f32: every SIMD runs 8 waves, every wave a
// loop of 16 independent copies of the instruction.  Build: hipcc -O2 --offload-arch=gfx950 valu_rate.hip -o valu_rate
// Every run also stamps s_memtime (the shader-clock counter) and s_memrealtime (the constant 100 MHz counter) around the
// loop of one wave: their ratio is the clock the SIMDs HELD during that loop, so the figure "cycles per instruction" is
// measured, not derived from a nominal 2.4 GHz.  The last line is a JSON summary (profiles/valu_issue.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

#define REP16(S) S S S S S S S S S S S S S S S S

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float *out, const double *table, unsigned long long *stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1.0f, b = 1.5f, c = 2.5f;
    double d = threadIdx.x * 1.0, e = 1.25;
    unsigned u = threadIdx.x, w = 3;
    unsigned long long q = threadIdx.x * 77ull;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p = {a, b}, r = {c, a};
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p) : "v"(r));) }
        if (OP == 2) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (OP == 3) { REP16(asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q) : "v"(w));) }
        if (OP == 4) { REP16(asm volatile("v_cvt_u32_f32 %0, %1" : "+v"(u) : "v"(a));) }
        if (OP == 5) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u) : "v"(w));) }
        if (OP == 6) { REP16(asm volatile("v_fract_f32 %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 7) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(u) : "v"(w));) }
        if (OP == 8) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p) : "v"(r));) }
        if (OP == 9) { REP16(asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d) : "v"(e));) }
        if (OP == 10) { REP16(asm volatile("v_min_f32 %0, 0x45000000, %0" : "+v"(a));) }
        if (OP == 11) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u) : "v"(w));) }
        if (OP == 12) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u), "v"(w) : "vcc");) }
        if (OP == 13) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 14) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(d));) }
        if (OP == 15) { REP16(asm volatile("v_and_b32 %0, 0x78, %0" : "+v"(u));) }
        if (OP == 16) { REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u) : "v"(w));) }
        if (OP == 17) { REP16(asm volatile("v_max3_f32 %0, %0, %1, 0" : "+v"(a) : "v"(b));) }
        if (OP == 18) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(w) : "vcc");) }
        if (OP == 19) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
    }
    if (a + p.x + p.y + (float)d + (float)u + (float)q == 12345.678f) out[0] = a;
    if (stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        stamps[0] = __builtin_amdgcn_s_memtime() - t0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

struct Result { double ns, cycles, ghz; };
static Result g_last;
template <int OP>
double run(const char *name, double base) {
    const int iters = 4000;
    float *out; CHECK(hipMalloc(&out, 4));
    unsigned long long *stamps; CHECK(hipMalloc(&stamps, 16)); CHECK(hipMemset(stamps, 0, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8;  // 8 waves per SIMD
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, 100, out, nullptr, nullptr);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, iters, out, nullptr, stamps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms * 1e-3 / ((double)iters * 16 * 8);  // seconds per instruction per SIMD
    unsigned long long h[2]; CHECK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
    // one wave's loop: `h[0]` shader-clock ticks in `h[1]` ticks of 10 ns -- the clock that wave's SIMD held; the cost of an
    // instruction in cycles = its time per SIMD (events around the whole launch: independent of how many waves were resident
    // at once) x that clock
    const double ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;
    const double cycles = per * 1e9 * ghz;
    std::printf("%-28s %8.3f ms  %6.2f ns/instr/SIMD  x%.2f of v_add_f32   shader clock %.3f GHz  %5.2f cycles/instr/SIMD (s_memtime)\n", name, ms, per * 1e9,
                base > 0 ? per / base : 1.0, ghz, cycles);
    g_last = Result{per * 1e9, cycles, ghz};
    CHECK(hipFree(out)); CHECK(hipFree(stamps));
    return per;
}

int main() {
    const double b = run<0>("v_add_f32", 0);
    const Result add = g_last;
    run<1>("v_pk_fma_f32", b);
    const Result pkfma = g_last; run<8>("v_pk_add_f32 (op_sel)", b); run<2>("v_add_f64", b); run<9>("v_fma_f64", b); run<19>("v_mul_f64", b);
    run<3>("v_lshrrev_b64", b); run<4>("v_cvt_u32_f32", b);
    const Result cvt = g_last; run<5>("v_add3_u32", b); run<6>("v_fract_f32", b); run<7>("v_mbcnt_lo_u32_b32", b);
    run<10>("v_min_f32 (literal)", b); run<11>("v_lshl_add_u32", b); run<12>("v_cmp_lt_u32", b); run<13>("v_mul_hi_u32", b); run<14>("v_rcp_f64", b);
    run<15>("v_and_b32 (literal)", b); run<16>("v_mov_b32_dpp", b); run<17>("v_max3_f32", b); run<18>("v_cndmask_b32", b);
    std::printf("{\"v_add_f32\": {\"ns\": %.4f, \"cycles\": %.3f, \"shader_clock_ghz\": %.4f}, \"v_pk_fma_f32\": {\"ns\": %.4f, \"cycles\": %.3f, \"shader_clock_ghz\": %.4f}, "
                "\"v_cvt_u32_f32\": {\"ns\": %.4f, \"cycles\": %.3f, \"shader_clock_ghz\": %.4f}}\n",
                add.ns, add.cycles, add.ghz, pkfma.ns, pkfma.cycles, pkfma.ghz, cvt.ns, cvt.cycles, cvt.ghz);
    return 0;
}
