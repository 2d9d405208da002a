// would the distance form of a dfire_bm_pairs batch be cheaper on the idle matrix pipe?
// One batch = 64 pair slots per lane (lane = pose): E = (Rs - l2) + Rz lz + Ry ly + Rx lx, cell = (u32)E, code = lut[cell],
// value = cube[row][code], 64-bit add.  Two forms of the 64 slots (tools/microbench/gen_mfma_batch.py):
//   valu   the product's hand-scheduled block: 128 packed f32 + 64 v_cvt + 64 v_lshl_add_u64 + 128 LDS reads
//   mfma   E by 64 v_mfma_f32_4x4x1_16b_f32 (block = 4 lanes; register r of lane j = receptor atom r x lane j's ligand atom: lane = pose
//          survives and the cube row stays an instruction constant), the rest the same
// plus the two halves of the mfma form alone.  Every form runs the kernel's occupancy: 2 workgroups of 4 waves per CU (79 KB of LDS
// each), 256 VGPRs.  Between two batches a lane "poses" 8 new atoms (60 packed FMAs, as the kernel does), so the vector port carries
// what it carries there.  The sums of the valu and mfma forms must be EQUAL (same operations in the same order: v_mfma_f32 is a
// bitwise fmaf chain) -- checked on the host.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_batch.hip -o mfma_batch     usage: mfma_batch [batches per wave]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mfma_batch_gen.inc"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kLut = 14592, kRow = 176, kCube = 64 * kRow, kWaves = 4;
struct Shared {
    unsigned char lut[kLut];
    unsigned char cube[kWaves][kCube];
    unsigned char pad[81216 - kLut - kWaves * kCube];   // the kernel's 79.3 KB: two workgroups per CU
};

struct Args {
    const unsigned char *lut;
    const long long *rows;       // [64][22]
    const float *rec;            // [36]: Rs[8] Rz[8] Ry[8] Rx[8] (atom order) + pad
    const float *maps;           // [poses][12]
    unsigned n_maps;
    int batches;
    unsigned long long *out;     // [threads]
    unsigned long long *stamps;  // [2]: s_memtime, s_memrealtime of wave 0
};

template <int FORM, int WAVE>
__device__ __forceinline__ void batch(unsigned long long &acc0, unsigned long long &acc1, const v2f (&Rs)[4], const v2f (&Rz)[4], const v2f (&Ry)[4],
                                      const v2f (&Rx)[4], const v2f (&L2)[4], const v2f (&LZ)[4], const v2f (&LY)[4], const v2f (&LX)[4], float ONES,
                                      const v4f (&RsT)[2], const float (&AZ)[2], const float (&AY)[2], const float (&AX)[2]) {
    constexpr unsigned CUBE = kLut + WAVE * kCube;
    if (FORM == 0) {
        MB_BATCH_VALU;
    } else if (FORM == 4) {
        MB_BATCH_VALU_B32;
    } else if (FORM == 5) {
        MB_BATCH_VALU_NOLUT;
    } else if (FORM == 6) {
        MB_BATCH_VALU_NOLUT_B32;
    } else if (FORM == 7) {
        MB_BATCH_VALU_NOLDS;
    } else {
        float NL2[8], LZ1[8], LY1[8], LX1[8];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            NL2[2 * p] = -L2[p].x; NL2[2 * p + 1] = -L2[p].y;
            LZ1[2 * p] = LZ[p].x; LZ1[2 * p + 1] = LZ[p].y;
            LY1[2 * p] = LY[p].x; LY1[2 * p + 1] = LY[p].y;
            LX1[2 * p] = LX[p].x; LX1[2 * p + 1] = LX[p].y;
        }
        if (FORM == 1) { MB_BATCH_MFMA; }
        if (FORM == 2) { MB_BATCH_MFMA_ONLY; }
        if (FORM == 3) { MB_BATCH_REST_ONLY; }
    }
}

template <int FORM>
__global__ __launch_bounds__(kWaves * 64, 2) void run(const Args A) {
    __shared__ __attribute__((aligned(16))) Shared S;
    if ((unsigned)(uintptr_t)&S != 0u) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kLut / 16; i += kWaves * 64) reinterpret_cast<uint4 *>(S.lut)[i] = reinterpret_cast<const uint4 *>(A.lut)[i];
    for (int i = lane; i < kCube / 16; i += 64) reinterpret_cast<uint4 *>(S.cube[wave])[i] = reinterpret_cast<const uint4 *>(A.rows)[i];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    typedef const __attribute__((address_space(4))) float const_f32;
    const_f32 *ops = (const_f32 *)(uintptr_t)A.rec;
    v2f Rs[4], Rz[4], Ry[4], Rx[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        Rs[q] = v2f{ops[2 * q], ops[2 * q + 1]};
        Rz[q] = v2f{ops[8 + 2 * q], ops[9 + 2 * q]};
        Ry[q] = v2f{ops[16 + 2 * q], ops[17 + 2 * q]};
        Rx[q] = v2f{ops[24 + 2 * q], ops[25 + 2 * q]};
    }
    // the MFMA form's receptor operands: lane-patterned (atom 4 G + lane % 4) and, for the chain's start, broadcast
    float AZ[2], AY[2], AX[2];
    v4f RsT[2];
#pragma unroll
    for (int G = 0; G < 2; G++) {
        AZ[G] = A.rec[8 + 4 * G + (lane & 3)];
        AY[G] = A.rec[16 + 4 * G + (lane & 3)];
        AX[G] = A.rec[24 + 4 * G + (lane & 3)];
        RsT[G] = v4f{A.rec[4 * G], A.rec[4 * G + 1], A.rec[4 * G + 2], A.rec[4 * G + 3]};
    }
    float ONES = 1.0f;
    asm volatile("" : "+v"(ONES), "+v"(AZ[0]), "+v"(AZ[1]), "+v"(AY[0]), "+v"(AY[1]), "+v"(AX[0]), "+v"(AX[1]), "+v"(RsT[0]), "+v"(RsT[1]));
    // a subtile's local coordinates (uniform), 8 atoms in a 4 A cube
    const v2f LocX[4] = {{-1.5f, 0.5f}, {1.25f, -0.75f}, {0.25f, 1.75f}, {-0.5f, 0.0f}}, LocY[4] = {{0.5f, -1.0f}, {1.5f, 0.25f}, {-1.75f, 0.75f}, {1.0f, -0.25f}};
    const v2f LocZ[4] = {{1.0f, 1.5f}, {-0.5f, -1.25f}, {0.75f, 0.0f}, {-1.5f, 0.5f}};
    unsigned long long total = 0ull;
    const unsigned gthread = blockIdx.x * (kWaves * 64) + tid;
    for (int b = 0; b < A.batches; b++) {
        const unsigned m = (gthread * 2654435761u + (unsigned)b * 40503u) % A.n_maps;
        const float4 *ap = reinterpret_cast<const float4 *>(A.maps) + (size_t)m * 3;
        const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
        v2f LX[4], LY[4], LZ[4], L2[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {   // bm_apply's nesting, two atoms at a time
            const v2f X = LocX[p], Y = LocY[p], Z = LocZ[p];
            LX[p] = __builtin_elementwise_fma(v2f{a0.x, a0.x}, X, __builtin_elementwise_fma(v2f{a0.y, a0.y}, Y, __builtin_elementwise_fma(v2f{a0.z, a0.z}, Z, v2f{a0.w, a0.w})));
            LY[p] = __builtin_elementwise_fma(v2f{a1.x, a1.x}, X, __builtin_elementwise_fma(v2f{a1.y, a1.y}, Y, __builtin_elementwise_fma(v2f{a1.z, a1.z}, Z, v2f{a1.w, a1.w})));
            LZ[p] = __builtin_elementwise_fma(v2f{a2.x, a2.x}, X, __builtin_elementwise_fma(v2f{a2.y, a2.y}, Y, __builtin_elementwise_fma(v2f{a2.z, a2.z}, Z, v2f{a2.w, a2.w})));
            L2[p] = __builtin_elementwise_fma(LX[p], LX[p], __builtin_elementwise_fma(LY[p], LY[p], LZ[p] * LZ[p]));
        }
        unsigned long long acc0 = 0ull, acc1 = 0ull;
        switch (wave) {
            case 0: batch<FORM, 0>(acc0, acc1, Rs, Rz, Ry, Rx, L2, LZ, LY, LX, ONES, RsT, AZ, AY, AX); break;
            case 1: batch<FORM, 1>(acc0, acc1, Rs, Rz, Ry, Rx, L2, LZ, LY, LX, ONES, RsT, AZ, AY, AX); break;
            case 2: batch<FORM, 2>(acc0, acc1, Rs, Rz, Ry, Rx, L2, LZ, LY, LX, ONES, RsT, AZ, AY, AX); break;
            default: batch<FORM, 3>(acc0, acc1, Rs, Rz, Ry, Rx, L2, LZ, LY, LX, ONES, RsT, AZ, AY, AX); break;
        }
        total += acc0 + acc1;
    }
    A.out[gthread] = total;
    if (A.stamps != nullptr && blockIdx.x == 0 && tid == 0) {
        A.stamps[0] = __builtin_amdgcn_s_memtime() - t0;
        A.stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// ---- three waves per SIMD (timing only): the same batch in 168 registers (temporaries at v132..v167), workgroups of
// 53 KB -- three per CU --, two waves sharing a cube (what a kernel with room for twelve cubes per CU would not need to do)
struct Shared3 {
    unsigned char lut[kLut];
    unsigned char cube[2][kCube];
    unsigned char pad[54000 - kLut - 2 * kCube];
};
template <int CUBE>
__device__ __forceinline__ void batch3(unsigned long long &acc0, unsigned long long &acc1, const v2f (&Rs)[4], const v2f (&Rz)[4], const v2f (&Ry)[4],
                                       const v2f (&Rx)[4], const v2f (&L2)[4], const v2f (&LZ)[4], const v2f (&LY)[4], const v2f (&LX)[4]) {
    MB_BATCH_VALU_T132;
}
__global__ __launch_bounds__(kWaves * 64, 3) void run3(const Args A) {
    __shared__ __attribute__((aligned(16))) Shared3 S;
    if ((unsigned)(uintptr_t)&S != 0u) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kLut / 16; i += kWaves * 64) reinterpret_cast<uint4 *>(S.lut)[i] = reinterpret_cast<const uint4 *>(A.lut)[i];
    for (int i = tid; i < 2 * kCube / 16; i += kWaves * 64) reinterpret_cast<uint4 *>(S.cube[0])[i] = reinterpret_cast<const uint4 *>(A.rows)[i % (kCube / 16)];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    typedef const __attribute__((address_space(4))) float const_f32;
    const_f32 *ops = (const_f32 *)(uintptr_t)A.rec;
    v2f Rs[4], Rz[4], Ry[4], Rx[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        Rs[q] = v2f{ops[2 * q], ops[2 * q + 1]};
        Rz[q] = v2f{ops[8 + 2 * q], ops[9 + 2 * q]};
        Ry[q] = v2f{ops[16 + 2 * q], ops[17 + 2 * q]};
        Rx[q] = v2f{ops[24 + 2 * q], ops[25 + 2 * q]};
    }
    const v2f LocX[4] = {{-1.5f, 0.5f}, {1.25f, -0.75f}, {0.25f, 1.75f}, {-0.5f, 0.0f}}, LocY[4] = {{0.5f, -1.0f}, {1.5f, 0.25f}, {-1.75f, 0.75f}, {1.0f, -0.25f}};
    const v2f LocZ[4] = {{1.0f, 1.5f}, {-0.5f, -1.25f}, {0.75f, 0.0f}, {-1.5f, 0.5f}};
    unsigned long long total = 0ull;
    const unsigned gthread = blockIdx.x * (kWaves * 64) + tid;
    for (int b = 0; b < A.batches; b++) {
        const unsigned m = (gthread * 2654435761u + (unsigned)b * 40503u) % A.n_maps;
        const float4 *ap = reinterpret_cast<const float4 *>(A.maps) + (size_t)m * 3;
        const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
        v2f LX[4], LY[4], LZ[4], L2[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const v2f X = LocX[p], Y = LocY[p], Z = LocZ[p];
            LX[p] = __builtin_elementwise_fma(v2f{a0.x, a0.x}, X, __builtin_elementwise_fma(v2f{a0.y, a0.y}, Y, __builtin_elementwise_fma(v2f{a0.z, a0.z}, Z, v2f{a0.w, a0.w})));
            LY[p] = __builtin_elementwise_fma(v2f{a1.x, a1.x}, X, __builtin_elementwise_fma(v2f{a1.y, a1.y}, Y, __builtin_elementwise_fma(v2f{a1.z, a1.z}, Z, v2f{a1.w, a1.w})));
            LZ[p] = __builtin_elementwise_fma(v2f{a2.x, a2.x}, X, __builtin_elementwise_fma(v2f{a2.y, a2.y}, Y, __builtin_elementwise_fma(v2f{a2.z, a2.z}, Z, v2f{a2.w, a2.w})));
            L2[p] = __builtin_elementwise_fma(LX[p], LX[p], __builtin_elementwise_fma(LY[p], LY[p], LZ[p] * LZ[p]));
        }
        unsigned long long acc0 = 0ull, acc1 = 0ull;
        if (wave < 2) batch3<kLut>(acc0, acc1, Rs, Rz, Ry, Rx, L2, LZ, LY, LX);
        else batch3<kLut + kCube>(acc0, acc1, Rs, Rz, Ry, Rx, L2, LZ, LY, LX);
        total += acc0 + acc1;
    }
    A.out[gthread] = total;
    if (A.stamps != nullptr && blockIdx.x == 0 && tid == 0) {
        A.stamps[0] = __builtin_amdgcn_s_memtime() - t0;
        A.stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
static double time_three(const char *name, Args A, int groups) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, run3, kWaves * 64, 0));
    Args W = A; W.batches = 20; W.stamps = nullptr;
    hipLaunchKernelGGL(run3, dim3(groups), dim3(kWaves * 64), 0, 0, W);
    CHECK(hipMemset(A.stamps, 0, 16));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(run3, dim3(groups), dim3(kWaves * 64), 0, 0, A);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, A.stamps, 16, hipMemcpyDeviceToHost));
    const double ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;
    const double us_batch = ms * 1e3 / A.batches;
    std::printf("%-28s %8.3f ms   %6.3f us per batch of a wave, %d workgroups resident per CU (%d waves per SIMD) = %.3f us per batch at the SIMD's rate of two waves   shader clock %.3f GHz\n",
                name, ms, us_batch, per_cu, per_cu, us_batch * 2.0 / per_cu, ghz);
    return us_batch;
}

template <int FORM>
static double time_form(const char *name, Args A, int groups, std::vector<unsigned long long> *sums) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    Args W = A; W.batches = 20; W.stamps = nullptr;
    hipLaunchKernelGGL(run<FORM>, dim3(groups), dim3(kWaves * 64), 0, 0, W);
    CHECK(hipMemset(A.stamps, 0, 16));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(run<FORM>, dim3(groups), dim3(kWaves * 64), 0, 0, A);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, A.stamps, 16, hipMemcpyDeviceToHost));
    const double ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;
    const double us_batch = ms * 1e3 / A.batches;   // per batch of one wave, two waves a SIMD running together
    std::printf("%-28s %8.3f ms   %6.3f us per batch (2 waves/SIMD)   shader clock %.3f GHz   %6.0f cycles per batch\n", name, ms, us_batch, ghz, us_batch * 1e3 * ghz);
    if (sums) {
        sums->resize((size_t)groups * kWaves * 64);
        CHECK(hipMemcpy(sums->data(), A.out, sums->size() * 8, hipMemcpyDeviceToHost));
    }
    return us_batch;
}

int main(int argc, char **argv) {
    const int batches = argc > 1 ? std::atoi(argv[1]) : 2000;
    const double near_a = argc > 2 ? std::atof(argv[2]) : 10.0, far_a = argc > 3 ? std::atof(argv[3]) : 26.0;   // the subtiles' distance range, A
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int groups = prop.multiProcessorCount * 2;
    // a LUT like the kernel's: cell' = floor(14583.5 - 64 d2): the far end "miss" (160), then bins by r = sqrt(d2): (2 r - 1) truncated
    std::vector<unsigned char> lut(kLut);
    for (int c = 0; c < kLut; c++) {
        const double d2 = (14583.5 - c) / 64.0;
        if (d2 > 225.0) { lut[c] = 160; continue; }
        const double r = std::sqrt(std::max(0.0, d2));
        int bin = (int)(2.0 * r - 1.0);
        bin = bin < 0 ? 0 : bin > 29 ? 29 : bin;
        static const int d2b[30] = {1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20};
        lut[c] = (unsigned char)(8 * (d2b[bin] - 1 > 19 ? 19 : d2b[bin] - 1));
    }
    std::vector<long long> rows(64 * 22);
    for (size_t i = 0; i < rows.size(); i++) rows[i] = (i % 22 == 20) ? 0 : (long long)((i * 2654435761ull) % 2000003ull) - 1000001ll;
    // a receptor subtile around its box centre, record units (8 per A): atoms within +-2.5 A
    float rec[36] = {0};
    for (int j = 0; j < 8; j++) {
        const float x = 8.f * (-2.5f + 0.7f * j), y = 8.f * (1.5f - 0.45f * j), z = 8.f * (((j * 5) % 8) * 0.6f - 2.0f);
        rec[j] = std::fmaf(-x, x, std::fmaf(-y, y, std::fmaf(-z, z, 14583.5f)));
        rec[8 + j] = 2.f * z; rec[16 + j] = 2.f * y; rec[24 + j] = 2.f * x;
    }
    // poses: rotations about z by an angle, the subtile's centre near_a .. far_a from the block's (10 .. 26 A: ~30 % of the slots inside the
    // 15 A cutoff, like 1k4c -- the lanes of a slot that miss all read LUT cell 0 and the row's "miss" slot: broadcasts, no bank conflict)
    const unsigned n_maps = 8192;
    std::vector<float> maps((size_t)n_maps * 12);
    for (unsigned p = 0; p < n_maps; p++) {
        const double a = 0.37 * p, dist = 8.0 * (near_a + (far_a - near_a) * ((p * 7919u) % 1000u) / 1000.0), el = 0.011 * p;
        float *o = &maps[(size_t)p * 12];
        o[0] = 8.f * (float)std::cos(a); o[1] = -8.f * (float)std::sin(a); o[2] = 0.f; o[3] = (float)(dist * std::cos(el));
        o[4] = 8.f * (float)std::sin(a); o[5] = 8.f * (float)std::cos(a); o[6] = 0.f; o[7] = (float)(dist * std::sin(el) * 0.8);
        o[8] = 0.f; o[9] = 0.f; o[10] = 8.f; o[11] = (float)(dist * std::sin(el) * 0.6);
    }
    Args A;
    unsigned char *d_lut; long long *d_rows; float *d_rec, *d_maps; unsigned long long *d_out, *d_st;
    CHECK(hipMalloc(&d_lut, kLut)); CHECK(hipMemcpy(d_lut, lut.data(), kLut, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_rows, rows.size() * 8)); CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_rec, sizeof(rec))); CHECK(hipMemcpy(d_rec, rec, sizeof(rec), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_maps, maps.size() * 4)); CHECK(hipMemcpy(d_maps, maps.data(), maps.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)groups * kWaves * 64 * 8)); CHECK(hipMalloc(&d_st, 16));
    A.lut = d_lut; A.rows = d_rows; A.rec = d_rec; A.maps = d_maps; A.n_maps = n_maps; A.batches = batches; A.out = d_out; A.stamps = d_st;
    std::vector<unsigned long long> s_valu, s_mfma;
    {   // the share of pair slots inside the cutoff (host replay of a sample of the poses)
        size_t in = 0, all = 0;
        const float lx[8] = {-1.5f, 0.5f, 1.25f, -0.75f, 0.25f, 1.75f, -0.5f, 0.0f}, ly[8] = {0.5f, -1.0f, 1.5f, 0.25f, -1.75f, 0.75f, 1.0f, -0.25f};
        const float lz[8] = {1.0f, 1.5f, -0.5f, -1.25f, 0.75f, 0.0f, -1.5f, 0.5f};
        for (unsigned p = 0; p < n_maps; p += 7) {
            const float *o = &maps[(size_t)p * 12];
            for (int i = 0; i < 8; i++) {
                const float x = o[0] * lx[i] + o[1] * ly[i] + o[2] * lz[i] + o[3], y = o[4] * lx[i] + o[5] * ly[i] + o[6] * lz[i] + o[7], z = o[8] * lx[i] + o[9] * ly[i] + o[10] * lz[i] + o[11];
                for (int j = 0; j < 8; j++) {
                    const float E = rec[j] - (x * x + y * y + z * z) + rec[8 + j] * z + rec[16 + j] * y + rec[24 + j] * x;
                    in += E >= 183.0f;   // cell of 64 d2 = 14400
                    all++;
                }
            }
        }
        std::printf("# subtile distance %.1f .. %.1f A: %.1f %% of the pair slots inside the cutoff\n", near_a, far_a, 100.0 * in / all);
    }
    std::printf("# %d workgroups of %d waves (2 per CU: 2 waves per SIMD), %d batches per wave\n", groups, kWaves, batches);
    const double tv = time_form<0>("valu (the product's block)", A, groups, &s_valu);
    const double tm = time_form<1>("mfma 4x4x1 distance form", A, groups, &s_mfma);
    time_form<2>("  its 64 MFMAs alone", A, groups, nullptr);
    time_form<3>("  its cvt/LUT/table/add alone", A, groups, nullptr);
    std::printf("# LDS sensitivity of the product's block (timing only, wrong sums):\n");
    time_form<4>("valu, table reads 32 bits wide", A, groups, nullptr);
    time_form<5>("valu, no LUT read", A, groups, nullptr);
    time_form<6>("valu, no LUT read, 32-bit table", A, groups, nullptr);
    time_form<7>("valu, no LDS read at all", A, groups, nullptr);
    std::printf("# three waves per SIMD (the product's block in 168 registers, 3 workgroups of 53 KB per CU; every wave runs %d batches):\n", batches);
    time_three("valu, 3 waves per SIMD", A, prop.multiProcessorCount * 3);
    size_t differ = 0;
    for (size_t i = 0; i < s_valu.size(); i++) differ += s_valu[i] != s_mfma[i];
    std::printf("sums of the two forms differ in %zu of %zu lanes%s\n", differ, s_valu.size(), differ ? "  <-- NOT the same cells" : " (bitwise the same E: same cells, same sums)");
    std::printf("{\"us_per_batch_valu\": %.4f, \"us_per_batch_mfma\": %.4f, \"ratio\": %.4f, \"lanes_differ\": %zu}\n", tv, tm, tm / tv, differ);
    return 0;
}
