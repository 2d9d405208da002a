// What does it cost a gfx950 wave to read, with a scalar instruction, an SGPR pair that a vector instruction has just written
// (carry-out of v_sub_co_u32, v_cmp)?  The block-major DFIRE pair kernel collects "flagged cell" lanes as borrow masks.
// Every wave runs a loop of 16 x { v_sub_co_u32 } followed by scalar ORs of the masks, with GAP independent vector
// instructions between the last write and the first read; also a taken branch per iteration.   usage: sgpr_dep [waves per SIMD = 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define REP4(S) S S S S
#define REP16(S) REP4(S) REP4(S) REP4(S) REP4(S)

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out) {
    unsigned c = threadIdx.x + 8, x0 = 0, acc = 0;
    float a = threadIdx.x, b = 1.5f;
    unsigned long long m = 0, q = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 16 plain subtractions + 16 independent f32 adds
            REP16(asm volatile("v_sub_u32 %0, %1, 8" : "=v"(x0) : "v"(c));)
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 1) {   // 16 carry-outs, never read
            REP16(asm volatile("v_sub_co_u32 %0, %1, %2, 8" : "=v"(x0), "=s"(m) : "v"(c));)
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 2) {   // 16 carry-outs, ONE scalar read right behind the last
            REP16(asm volatile("v_sub_co_u32 %0, %1, %2, 8" : "=v"(x0), "=s"(m) : "v"(c));)
            asm volatile("s_or_b64 %0, %0, %1" : "+s"(q) : "s"(m) : "scc");
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 3) {   // 16 carry-outs, one scalar read after 16 independent vector instructions
            REP16(asm volatile("v_sub_co_u32 %0, %1, %2, 8" : "=v"(x0), "=s"(m) : "v"(c));)
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
            asm volatile("s_or_b64 %0, %0, %1" : "+s"(q) : "s"(m) : "scc");
        }
        if (MODE == 4) {   // every carry-out read at once (16 round trips)
            REP16(asm volatile("v_sub_co_u32 %0, %1, %2, 8\n\ts_or_b64 %3, %3, %1" : "=v"(x0), "=&s"(m), "+v"(c), "+s"(q) : : "scc");)
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 5) {   // MODE 0 + one taken branch
            REP16(asm volatile("v_sub_u32 %0, %1, 8" : "=v"(x0) : "v"(c));)
            asm volatile("s_branch 1f\n\ts_nop 0\n\ts_nop 0\n1:");
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 6) {   // MODE 0 + 16 taken branches
            REP16(asm volatile("v_sub_u32 %0, %1, 8\n\ts_branch 1f\n\ts_nop 0\n1:" : "=v"(x0) : "v"(c));)
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 7) {   // v_cmp -> vcc -> s_or (the classic way)
            REP16(asm volatile("v_cmp_eq_u32 vcc, 0, %0" : : "v"(c) : "vcc");)
            asm volatile("s_or_b64 %0, %0, vcc" : "+s"(q) : : "scc");
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 8) {   // v_readfirstlane -> s_add (a VGPR -> SGPR round trip of another kind)
            unsigned s;
            REP16(asm volatile("v_sub_u32 %0, %1, 8" : "=v"(x0) : "v"(c));)
            asm volatile("v_readfirstlane_b32 %0, %1\n\ts_add_u32 %0, %0, 1" : "=s"(s) : "v"(x0));
            acc += s;
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        if (MODE == 9) {   // 16 carry-outs, 16 scalar reads after 16 independent vector instructions
            unsigned long long mm[4];
            REP4(asm volatile("v_sub_co_u32 %0, %1, %5, 8\n\tv_sub_co_u32 %0, %2, %5, 8\n\tv_sub_co_u32 %0, %3, %5, 8\n\tv_sub_co_u32 %0, %4, %5, 8" : "=&v"(x0), "=&s"(mm[0]), "=&s"(mm[1]), "=&s"(mm[2]), "=&s"(mm[3]) : "v"(c));)
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
            REP4(asm volatile("s_or_b64 %0, %0, %1\n\ts_or_b64 %0, %0, %2\n\ts_or_b64 %0, %0, %3\n\ts_or_b64 %0, %0, %4" : "+s"(q) : "s"(mm[0]), "s"(mm[1]), "s"(mm[2]), "s"(mm[3]) : "scc");)
        }
    }
    if (a + (float)x0 + (float)q + (float)m + (float)acc == 12345.678f) out[0] = x0;
}

static int g_wps = 2;
template <int MODE> void run(const char *name) {
    const int iters = 4000;
    unsigned *out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * g_wps), dim3(256), 0, 0, 100, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * g_wps), dim3(256), 0, 0, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%-78s %7.3f ms  %7.1f ns per iteration per wave\n", name, ms, ms * 1e6 / iters);
    CHECK(hipFree(out));
}
int main(int argc, char **argv) {
    if (argc > 1) g_wps = std::atoi(argv[1]);
    const int only = argc > 2 ? std::atoi(argv[2]) : -1;
    if (only < 0) std::printf("# %d waves per SIMD; an iteration = 16 subtractions + 16 v_add_f32 (dependent) + what the line says\n", g_wps);
#define RUN(M, S) if (only < 0 || only == M) { run<M>(S); std::fflush(stdout); }
    RUN(0, "v_sub_u32 x 16");
    RUN(1, "v_sub_co_u32 (SGPR pair written) x 16, never read");
    RUN(2, "... + one s_or_b64 of the last mask right behind it");
    RUN(3, "... + one s_or_b64 after 16 independent vector instructions");
    RUN(4, "every v_sub_co_u32 followed by an s_or_b64 of its mask");
    RUN(9, "16 v_sub_co_u32 to 4 pairs, 16 s_or_b64 after 16 vector instructions");
    RUN(7, "v_cmp_eq_u32 vcc x 16 + one s_or_b64 of vcc");
    RUN(8, "v_sub_u32 x 16 + v_readfirstlane_b32 + s_add_u32");
    RUN(5, "v_sub_u32 x 16 + one taken s_branch");
    RUN(6, "(v_sub_u32 + taken s_branch) x 16");
    return 0;
}
