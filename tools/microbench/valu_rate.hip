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
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f m4 = {a, b, c, a};
    unsigned long long q2 = threadIdx.x * 13ull;
    unsigned u1 = u + 1, u2 = u + 2, u3 = u + 3, w2 = 5, sres = 7;
    unsigned long long smask = 0x5555aaaa3333ccccull;
    asm volatile("" : "+s"(smask), "+s"(sres));
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
        if (OP == 20) { REP16(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 21) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 22) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 23) { REP16(asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(q) : "v"(q2));) }
        if (OP == 24) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 25) { REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(m4) : "v"(b), "v"(c));) }
        if (OP == 26) { REP16(asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m4) : "v"(b), "v"(c));) }
        if (OP == 27) { REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n\tv_cvt_u32_f32 %1, %2" : "+v"(m4), "+v"(u) : "v"(b), "v"(c));) }
        if (OP == 28) { REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n\tv_cvt_u32_f32 %1, %2\n\tv_cvt_u32_f32 %1, %3" : "+v"(m4), "+v"(u) : "v"(b), "v"(c));) }
        if (OP == 33) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u) : "v"(w), "s"(smask));) }
        if (OP == 34) { asm volatile(".rept 4\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %4, %5, vcc\n\tv_cndmask_b32 %2, %4, %5, vcc\n\tv_cndmask_b32 %3, %4, %5, vcc\n\t.endr"
                                      : "+v"(u), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(w2) : "vcc"); }
        if (OP == 35) { REP16(asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "+v"(u) : "s"(smask));) }
        if (OP == 36) { REP16(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u) : "v"(w), "v"(w2));) }
        if (OP == 37) { REP16(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x30" : "+v"(u) : "v"(w), "v"(w2));) }
        if (OP == 38) { REP16(asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(u));) }
        if (OP == 39) { REP16(asm volatile("v_sub_u32 %0, %1, %0" : "+v"(u) : "v"(w));) }
        if (OP == 40) { REP16(asm volatile("v_mov_b64 %0, %1" : "+v"(q) : "v"(q2));) }
        if (OP == 41) { REP16(asm volatile("v_mov_b32 %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 42) { REP16(asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sres) : "v"(w));) }
        if (OP == 43) { REP16(asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sres) : "v"(w));) }
        if (OP == 44) { REP16(asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(u) : "s"(sres));) }
        if (OP == 45) { REP16(asm volatile("v_cmp_ne_u32 %0, 0, %1" : "=s"(smask) : "v"(w));) }
        if (OP == 46) { REP16(asm volatile("v_min_u32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 47) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(w2));) }
        if (OP == 48) { REP16(asm volatile("v_bfe_i32 %0, %0, 11, 1" : "+v"(u));) }
        if (OP == 49) { REP16(asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(u) : "s"(sres));) }
        if (OP == 29) { REP16(asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u));) }
        if (OP == 30) { REP16(asm volatile("v_or_b32 %0, %0, %1" : "+v"(u) : "v"(w));) }
        if (OP == 31) { REP16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 32) { REP16(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    }
    if (a + p.x + p.y + (float)d + (float)u + (float)u1 + (float)u2 + (float)u3 + (float)sres + (float)smask + (float)q + m4.x + m4.y + m4.z + m4.w == 12345.678f) out[0] = a;
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
    run<20>("v_fmac_f32", b); run<21>("v_fma_f32", b); run<24>("v_mul_f32", b); run<31>("v_sub_f32", b); run<32>("v_max_f32", b); run<22>("v_add_u32", b);
    run<29>("v_lshlrev_b32", b); run<30>("v_or_b32", b); run<23>("v_lshl_add_u64", b);
    run<25>("v_mfma_f32_4x4x1_16b_f32 (dependent chain)", b); run<26>("v_mfma_f32_16x16x4_f32 (dependent chain)", b);
    run<27>("mfma_4x4x1 + 1 v_cvt (per pair)", b); run<28>("mfma_4x4x1 + 2 v_cvt (per triple)", b);
    run<33>("v_cndmask_b32_e64 (sgpr mask)", b); run<34>("v_cndmask_b32 vcc, 4 independent dsts", b); run<35>("v_cndmask_b32_e64 0, 1, sgpr", b);
    run<36>("v_bfi_b32", b); run<37>("v_bitop3_b32", b); run<38>("v_ashrrev_i32", b); run<39>("v_sub_u32", b); run<40>("v_mov_b64", b); run<41>("v_mov_b32", b);
    run<42>("v_readlane_b32", b); run<43>("v_readfirstlane_b32", b); run<44>("v_mbcnt_hi (sgpr)", b); run<45>("v_cmp_ne_u32 -> sgpr pair", b);
    run<46>("v_min_u32", b); run<47>("v_and_or_b32", b); run<48>("v_bfe_i32", b); run<49>("v_writelane_b32", b);
    run<15>("v_and_b32 (literal)", b); run<16>("v_mov_b32_dpp", b); run<17>("v_max3_f32", b); run<18>("v_cndmask_b32", b);
    std::printf("{\"v_add_f32\": {\"ns\": %.4f, \"cycles\": %.3f, \"shader_clock_ghz\": %.4f}, \"v_pk_fma_f32\": {\"ns\": %.4f, \"cycles\": %.3f, \"shader_clock_ghz\": %.4f}, "
                "\"v_cvt_u32_f32\": {\"ns\": %.4f, \"cycles\": %.3f, \"shader_clock_ghz\": %.4f}}\n",
                add.ns, add.cycles, add.ghz, pkfma.ns, pkfma.cycles, pkfma.ghz, cvt.ns, cvt.cycles, cvt.ghz);
    return 0;
}
