// Inner loop of a BLOCK-MAJOR DFIRE pair kernel, as a microbenchmark: lane = pose, the atom pair (i, j) of an
// 8 x 8 block is wave-uniform.  Per batch of 64 poses a lane loads its pose's f32 rotation + translation, poses the 8
// ligand atoms of the block's ligand subtile (uniform local coordinates) and runs 32 packed steps (ligand atom i x
// receptor atoms 2q, 2q+1 from scalar registers): D'' = 64 d2, cell = (u32)D'', code = lut[cell] (u8, LDS, 16 cells per
// unit of 4 d2), value = cube[(i, j)][code] (f64, LDS, the block's 64 table rows of 22 slots), f64 add.
// What does a packed step (128 pair slots) cost per CU?   usage: bm_step <workgroups per CU> <waves per workgroup> [mode]
// mode bits: 1 no LUT read, 2 no table read, 4 no flagged-cell test, 8 reads masked to the lanes in range, 16 no posing
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kLutCells = 14464;       // 904 * 16
constexpr int kRowBytes = 176;         // 22 slots: slot 0 = 0.0 (miss), 1..21 = bins 0..20
constexpr int kCubeBytes = 64 * kRowBytes + 16;
constexpr unsigned kFlagged = 176;     // = slot 0 of the next row: reads 0.0, and is larger than every bin code
constexpr float kCellMax = 14463.0f;
constexpr int kQueue = 256;

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned cvt_u32_sat(float f) {
    unsigned r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

struct Args {
    const float *poses;          // [n_poses][12]: rows of 8 R, then 8 (t - c)
    unsigned n_poses;
    const float *lig_local;      // [8][4] uniform local coordinates of the ligand subtile
    const float *rec;            // [n_blocks][4 records][8]: x0 x1 y0 y1 z0 z1 . .
    unsigned n_blocks;
    const unsigned char *lut;    // kLutCells
    const double *cube;          // [64][22]
    int batches;
    double *out;
};

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bm(const Args A) {
    __shared__ __attribute__((aligned(16))) unsigned char s_lut[kLutCells];
    __shared__ __attribute__((aligned(16))) unsigned char s_cube[kCubeBytes];
    __shared__ unsigned short s_queue[WAVES][kQueue];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kLutCells / 16; i += WAVES * 64) reinterpret_cast<uint4 *>(s_lut)[i] = reinterpret_cast<const uint4 *>(A.lut)[i];
    for (int i = tid; i < kCubeBytes / 16; i += WAVES * 64) reinterpret_cast<uint4 *>(s_cube)[i] = reinterpret_cast<const uint4 *>(A.cube)[i];
    __syncthreads();
    double total = 0.0;
    unsigned queued = 0;
    float lx[8], ly[8], lz[8];
    for (int b = 0; b < A.batches; b++) {
        const unsigned blk = (blockIdx.x * WAVES + wave + b) % A.n_blocks;
        const float *rr = A.rec + (size_t)blk * 32;   // uniform: scalar loads
        if (!(MODE & 16) || b == 0) {
            const unsigned pose = mix((blockIdx.x * WAVES + wave) * 7919u + b * 64u + lane) % A.n_poses;
            const float4 r0 = reinterpret_cast<const float4 *>(A.poses + (size_t)pose * 12)[0];
            const float4 r1 = reinterpret_cast<const float4 *>(A.poses + (size_t)pose * 12)[1];
            const float4 r2 = reinterpret_cast<const float4 *>(A.poses + (size_t)pose * 12)[2];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float x = A.lig_local[4 * i], y = A.lig_local[4 * i + 1], z = A.lig_local[4 * i + 2];
                lx[i] = __builtin_fmaf(r0.x, x, __builtin_fmaf(r0.y, y, __builtin_fmaf(r0.z, z, r0.w)));
                ly[i] = __builtin_fmaf(r1.x, x, __builtin_fmaf(r1.y, y, __builtin_fmaf(r1.z, z, r1.w)));
                lz[i] = __builtin_fmaf(r2.x, x, __builtin_fmaf(r2.y, y, __builtin_fmaf(r2.z, z, r2.w)));
            }
        }
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const v2f rx = {rr[8 * q], rr[8 * q + 1]}, ry = {rr[8 * q + 2], rr[8 * q + 3]}, rz = {rr[8 * q + 4], rr[8 * q + 5]};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const v2f dx = rx - v2f{lx[i], lx[i]}, dy = ry - v2f{ly[i], ly[i]}, dz = rz - v2f{lz[i], lz[i]};
                v2f D = dz * dz;
                D = __builtin_elementwise_fma(dy, dy, D);
                D = __builtin_elementwise_fma(dx, dx, D);
                const bool in0 = D.x < 14400.5f, in1 = D.y < 14400.5f;
                const unsigned c0 = cvt_u32_sat(fminf(D.x, kCellMax)), c1 = cvt_u32_sat(fminf(D.y, kCellMax));
                unsigned w0, w1;
                if (MODE & 1) {
                    w0 = (c0 & 15u) * 8u; w1 = (c1 & 15u) * 8u;
                } else if (MODE & 8) {
                    w0 = in0 ? s_lut[c0] : 0u; w1 = in1 ? s_lut[c1] : 0u;
                } else {
                    w0 = s_lut[c0]; w1 = s_lut[c1];
                }
                if (MODE & 2) {
                    acc += (double)w0; acc += (double)w1;
                } else if (MODE & 8) {
                    if (in0) acc += *reinterpret_cast<const double *>(s_cube + (i * 8 + 2 * q) * kRowBytes + w0);
                    if (in1) acc += *reinterpret_cast<const double *>(s_cube + (i * 8 + 2 * q + 1) * kRowBytes + w1);
                } else {
                    acc += *reinterpret_cast<const double *>(s_cube + (i * 8 + 2 * q) * kRowBytes + w0);
                    acc += *reinterpret_cast<const double *>(s_cube + (i * 8 + 2 * q + 1) * kRowBytes + w1);
                }
                if (!(MODE & 4)) {
                    const unsigned m = w0 > w1 ? w0 : w1;
                    if (__builtin_expect(__ballot(m >= kFlagged) != 0ull, 0)) {
                        const bool f0 = w0 >= kFlagged, f1 = w1 >= kFlagged;
                        const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
                        const unsigned n0 = (unsigned)__popcll(m0);
                        const unsigned i0 = queued + __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
                        const unsigned i1 = queued + n0 + __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
                        const unsigned short item = (unsigned short)(lane | (i * 8 + 2 * q) << 6);
                        if (f0) s_queue[wave][i0 % kQueue] = item;
                        if (f1) s_queue[wave][i1 % kQueue] = item + 64;
                        queued += n0 + (unsigned)__popcll(m1);
                    }
                }
            }
        }
        total += acc;
    }
    if (total == 1.2345 || queued == 0x7fffffffu) A.out[0] = total + s_queue[wave][lane];
    if (blockIdx.x == 0 && tid == 0) A.out[1] = (double)queued;
}

template <int MODE, int WAVES>
static double run(const Args &a, int wg_per_cu, int batches, bool print) {
    Args A = a;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * wg_per_cu;
    A.batches = 4;
    hipLaunchKernelGGL((bm<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, A);
    A.batches = batches;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((bm<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, A);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double steps = (double)blocks * WAVES * batches * 32.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (steps / 256.0);
    double q[2]; CHECK(hipMemcpy(q, A.out, 16, hipMemcpyDeviceToHost));
    if (print)
        std::printf("  mode %2d, %d x %d waves per CU: %8.3f ms  %6.2f CU-cycles per packed step (128 pair slots)  %.3g pair slots/s  [queued by wave 0: %.0f of %d pairs]\n",
                    MODE, wg_per_cu, WAVES, ms, cyc, steps * 128 / (ms * 1e-3), q[1], batches * 64 * 64);
    return cyc;
}

int main(int argc, char **argv) {
    const int wg_per_cu = argc > 1 ? std::atoi(argv[1]) : 3;
    const int batches = argc > 2 ? std::atoi(argv[2]) : 200;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    const double kappa = 8.0;
    // ligand subtile: 8 atoms within +-2 A of a local centre; receptor subtiles: 8 atoms within +-2 A of the origin
    double L0[3] = {20.0, 5.0, -3.0}, lig[8][3];
    std::vector<float> lig_local(32, 0.f);
    for (int i = 0; i < 8; i++)
        for (int c = 0; c < 3; c++) { lig[i][c] = L0[c] + 4.0 * U(rng) - 2.0; lig_local[4 * i + c] = (float)lig[i][c]; }
    const unsigned n_blocks = 64;
    std::vector<float> rec(n_blocks * 32, 0.f);
    for (unsigned b = 0; b < n_blocks; b++)
        for (int q = 0; q < 4; q++)
            for (int c = 0; c < 3; c++)
                for (int h = 0; h < 2; h++) rec[b * 32 + q * 8 + 2 * c + h] = (float)(kappa * (4.0 * U(rng) - 2.0));
    const unsigned n_poses = 8192;
    std::vector<float> poses(n_poses * 12);
    for (unsigned p = 0; p < n_poses; p++) {
        double q[4], n = 0;
        for (double &v : q) { v = N(rng); n += v * v; }
        n = std::sqrt(n);
        const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
        const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                                {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                                {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
        // the subtile's centre lands at distance 21 u^(1/3) A from the receptor subtile: ~36 % of the pairs within 15 A
        double dir[3], dn = 0;
        for (double &v : dir) { v = N(rng); dn += v * v; }
        dn = std::sqrt(dn);
        const double rad = 21.0 * std::cbrt(U(rng));
        for (int r = 0; r < 3; r++) {
            double rl = 0;
            for (int c = 0; c < 3; c++) { poses[p * 12 + 4 * r + c] = (float)(kappa * R[r][c]); rl += R[r][c] * L0[c]; }
            poses[p * 12 + 4 * r + 3] = (float)(kappa * (dir[r] / dn * rad - rl));
        }
    }
    std::vector<unsigned char> lut(kLutCells, 0);
    auto bin_of = [](double d2) { double d = std::sqrt(d2) * 2.0 - 1.0; int idx = d > 0 ? (int)d : 0; return idx < 3 ? 0 : idx < 15 ? idx - 2 : 13 + (idx - 15) / 2; };
    int flagged = 0;
    for (int c = 0; c < 14400; c++) {
        const int b0 = bin_of(c / 64.0), b1 = bin_of((c + 1) / 64.0 - 1e-9);
        const int bm1 = c > 0 ? bin_of((c - 1) / 64.0) : b0, bp1 = bin_of((c + 2) / 64.0 - 1e-9);
        const bool near_step = b0 != bm1 || bp1 != b1 || b0 != b1 || c < 400 || c >= 14399;
        lut[c] = near_step ? (unsigned char)kFlagged : (unsigned char)((b0 + 1) * 8);
        flagged += near_step;
    }
    std::vector<double> cube(kCubeBytes / 8, 0.0);
    for (int r = 0; r < 64; r++)
        for (int s = 1; s < 22; s++) cube[r * 22 + s] = 4.0 * U(rng) - 2.0;
    Args A{};
    float *d_poses, *d_lig, *d_rec; unsigned char *d_lut; double *d_cube, *d_out;
    CHECK(hipMalloc(&d_poses, poses.size() * 4)); CHECK(hipMemcpy(d_poses, poses.data(), poses.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_lig, 128)); CHECK(hipMemcpy(d_lig, lig_local.data(), 128, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_rec, rec.size() * 4)); CHECK(hipMemcpy(d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_lut, kLutCells)); CHECK(hipMemcpy(d_lut, lut.data(), kLutCells, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_cube, kCubeBytes)); CHECK(hipMemcpy(d_cube, cube.data(), kCubeBytes, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, 16));
    A.poses = d_poses; A.n_poses = n_poses; A.lig_local = d_lig; A.rec = d_rec; A.n_blocks = n_blocks; A.lut = d_lut; A.cube = d_cube; A.out = d_out;
    std::printf("block-major step microbenchmark: %d flagged cells of 14400, %d workgroups per CU, %d batches per wave\n", flagged, wg_per_cu, batches);
    run<0, 8>(A, wg_per_cu, batches, true);
    run<4, 8>(A, wg_per_cu, batches, true);
    run<4 | 1, 8>(A, wg_per_cu, batches, true);
    run<4 | 2, 8>(A, wg_per_cu, batches, true);
    run<4 | 1 | 2, 8>(A, wg_per_cu, batches, true);
    run<8, 8>(A, wg_per_cu, batches, true);
    run<8 | 4, 8>(A, wg_per_cu, batches, true);
    run<16, 8>(A, wg_per_cu, batches, true);
    run<0, 4>(A, wg_per_cu, batches, true);
    run<0, 4>(A, 2 * wg_per_cu, batches, true);
    return 0;
}
