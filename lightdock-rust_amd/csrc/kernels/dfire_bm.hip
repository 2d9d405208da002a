// dfire_bm.hip -- K1 for DFIRE, block-major (gfx950 / MI355X).  Interface, data flow and numerics: dfire_bm.hpp.
//
// The pair work of a launch is ordered by 8 x 8 atom-pair BLOCK, not by pose: a block's 64 table rows
// T[type_i][type_j][bin] sit in LDS while every pose in which the block is within the cutoff walks its 64 pairs, one pose
// per lane, the atom pair wave-uniform.  The potential never travels through the vector L1 as a gather.
// Compiled with -ffp-contract=off; every f64 operation is the reference's (src/dfire.rs:325-345); the f32 filter uses
// explicit fmas.  No MFMA (lookup/reduction).
#include "dfire_bm.hpp"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "dfire_device.hpp"
#include "dfire_bm_batch.inc"

// The LD_BM_DIAG_* blocks below are TIMING EXPERIMENTS (wrong sums by construction).  They compile only in a diagnostic build
// (-DLD_DIAG_BUILD, which tools/build_variant.sh passes): the shipped library cannot be built with one of them by accident, and
// tests/test_host_cpu.py checks that it carries no diagnostic switch.
#if !defined(LD_DIAG_BUILD) && (defined(LD_BM_DIAG_ANM_COST) || defined(LD_BM_DIAG_ANM_LDS) || defined(LD_BM_DIAG_FIRST) || defined(LD_BM_DIAG_NO_ATOMIC) || defined(LD_BM_DIAG_NO_DMA) || defined(LD_BM_DIAG_NO_EXACT) || defined(LD_BM_DIAG_NO_EXACT_ATOMIC) || defined(LD_BM_DIAG_NO_PAIRS) || defined(LD_BM_DIAG_NO_PARTIAL) || defined(LD_BM_DIAG_NO_POSE) || defined(LD_BM_DIAG_NO_TRACKED) || defined(LD_BM_DIAG_ROW_OF_LANE) || defined(LD_BM_DIAG_WAIT) || defined(LD_BM_DIAG_CULL_TIMES) || defined(LD_BM_DIAG_NO_TP_ATOMIC))
#error "LD_BM_DIAG_* needs -DLD_DIAG_BUILD (tools/build_variant.sh)"
#endif

namespace ld {

namespace {

// The launch arguments stay where the dispatch put them, in the kernarg segment (constant address space): every field
// is a scalar load at its point of use, nothing is copied to registers up front or to scratch when a non-inlined function
// wants the whole block.
typedef const __attribute__((address_space(4))) BmLaunch BmArgs;
#define LD_BM_ARGS ((BmArgs *)__builtin_amdgcn_kernarg_segment_ptr())

#ifndef LD_BM_CULL_POSES
#define LD_BM_CULL_POSES 8
#endif
constexpr int kBmCullPoses = LD_BM_CULL_POSES;   // poses a wave of dfire_bm_cull walks with its ligand tile
constexpr int kBmCullQueues = kBmCullQueueWords;   // (1k4c: 8 queues 695 us, 16 581, 32 388, 64 320, 128 ~310, 256 296, 512 316; static 337)   // counters the waves of dfire_bm_cull draw their items from
constexpr float kBmBoxCut = kBmBoxCutUnits2;  // (8 * 15 A)^2 in record units, padded for the rounding of the box test
// A receptor subtile's box as the culling kernel keeps it in LDS: per axis the pair {lo, -hi}.  With the ligand subtile's box as
// {-hi, lo} the two differences of an axis' gap -- lo_r - hi_l and lo_l - hi_r, the values axis_gap forms -- are ONE packed add.
struct alignas(16) BmCullBox {
    v2f x, y, z;
    float cut, unused;   // the subtile's reach, squared (kBmBoxCut unless its atoms' rows of the potential are zero: scorer.cpp, build_bm)
};
static_assert(sizeof(BmCullBox) == sizeof(TiledBox), "same room in LDS");
__device__ __forceinline__ float bm_cull_gap2(v2f lx, v2f ly, v2f lz, const BmCullBox &r) {   // = box_gap2, bit for bit
    const v2f dx = r.x + lx, dy = r.y + ly, dz = r.z + lz;
    const float gx = fmaxf(0.0f, fmaxf(dx.x, dx.y)), gy = fmaxf(0.0f, fmaxf(dy.x, dy.y)), gz = fmaxf(0.0f, fmaxf(dz.x, dz.y));
    return gx * gx + gy * gy + gz * gz;
}
// select by a mask held in a scalar register pair (v_cndmask_b32_e64): 4.4 cycles of the vector port; the compiler's usual form, with the
// mask in VCC (v_cndmask_b32_e32), 23 (measured, tools/microbench/valu_rate.hip)
__device__ __forceinline__ float bm_select(unsigned long long mask, float if_set, float if_clear) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}
// v_writelane_b32 (value, lane: wave-uniform; the other lanes keep `old`): this compiler has the intrinsic but no builtin for it
extern "C" __device__ int bm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// row of the pass -> pose row, or -1 beyond the list of this launch / inactive
__device__ __forceinline__ long long bm_pose_of(BmArgs *T, size_t listed) {
    if (T->pose_count != nullptr && T->first + listed >= (size_t)*T->pose_count) return -1;
    const size_t pose = T->pose_list ? (size_t)T->pose_list[T->first + listed] : T->first + listed;
    if (T->active != nullptr && T->active[pose] == 0) return -1;
    return (long long)pose;
}

// rows of this launch that exist: all n_poses of a plain batch; with a GSO list, what the device-side count leaves of them
__device__ __forceinline__ size_t bm_rows(BmArgs *T) {
    if (T->pose_count == nullptr) return T->n_poses;
    const size_t count = (size_t)*T->pose_count;
    return count <= T->first ? 0 : (count - T->first < T->n_poses ? count - T->first : T->n_poses);
}

// The f32 affine map of a pose, applied in ONE operation order wherever a ligand atom is posed in f32 (culling
// boxes and pair batches see the same bits).
struct Affine {
    float r00, r01, r02, tx, r10, r11, r12, ty, r20, r21, r22, tz;
};
__device__ __forceinline__ void bm_apply(const Affine &A, float x, float y, float z, float &ux, float &uy, float &uz) {
    ux = __builtin_fmaf(A.r00, x, __builtin_fmaf(A.r01, y, __builtin_fmaf(A.r02, z, A.tx)));
    uy = __builtin_fmaf(A.r10, x, __builtin_fmaf(A.r11, y, __builtin_fmaf(A.r12, z, A.ty)));
    uz = __builtin_fmaf(A.r20, x, __builtin_fmaf(A.r21, y, __builtin_fmaf(A.r22, z, A.tz)));
}
__device__ __forceinline__ ExactCtx bm_exact_ctx(BmArgs *T, size_t pose) {
    ExactCtx ex;
    ex.rx = T->m.rec_x;
    ex.ry = T->m.rec_y;
    ex.rz = T->m.rec_z;
    ex.modes = nullptr;
    ex.rec_nm = nullptr;
    ex.pad = 0;
    ex.num_anm = 0;
    ex.rec_tindex = T->m.rec_tindex;
    ex.rec_slot = T->m.rec_slot;
    ex.lig_slot = T->m.lig.slot;
    ex.step4 = T->m.bin_step;  // already 4 * step (scorer.cpp)
    ex.table = T->m.table;
    ex.iface_scaled = T->m.iface_scaled;
    ex.pose_flags = T->flags + pose * (size_t)(T->m.rec_flag_words + T->m.lig.flag_words);   // (by pose: pose_energy_finish reads them)
    ex.rec_flag_words = T->m.rec_flag_words;
    return ex;
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_pose: pose row -> f32 affine map into the record frame, [row of the pass][12].  v' = q v q^-1 + t (src/qt.rs:48-61) is the
// rotation matrix of q / |q|; computed in f64, rounded once.  Its error is part of eps (dfire_bm_error_bound).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dfire_bm_pose(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    const size_t rows = bm_rows(T);   // (a launch sized for every glowworm of a GSO costs what the glowworms that moved cost)
    {   // the sequence's counters: entries per tile pair, jobs, the culling kernel's item counters (one memset launch less)
        const size_t words = (size_t)T->m.lig.n_tiles * T->m.rec_n_tiles + kBmCounters + kBmCullQueueWords;
        for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < words; k += (size_t)gridDim.x * 256) T->tp_count[k] = 0u;
    }
    {   // the (ligand tile, row) sums of the pass, [ligand tile][cap rows]: per ligand tile the first `rows` words, cleared by consecutive
        // threads
        // (thread = row, a loop over the ligand tiles: consecutive threads clear consecutive words, and no division per word)
        const int n_lt = T->m.lig.n_tiles;
        for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256)
            for (int lt = 0; lt < n_lt; lt++) T->tile_sum[(size_t)lt * T->cap + r] = 0;
    }
    for (size_t listed = (size_t)blockIdx.x * 256 + threadIdx.x; listed < rows; listed += (size_t)gridDim.x * 256) {
        const long long p = bm_pose_of(T, listed);
        if (p < 0) continue;
        const size_t pose = (size_t)p;
        const double *row = T->poses + pose * T->stride;
        const double tx = row[0], ty = row[1], tz = row[2], w = row[3], x = row[4], y = row[5], z = row[6];
        const double n2 = w * w + x * x + y * y + z * z;
        const double k = kBmKappa / n2;
        float *o = T->rt + listed * 12;
        o[0] = (float)(k * (w * w + x * x - y * y - z * z));
        o[1] = (float)(k * 2.0 * (x * y - w * z));
        o[2] = (float)(k * 2.0 * (x * z + w * y));
        o[3] = (float)(kBmKappa * (tx - T->m.cx));
        o[4] = (float)(k * 2.0 * (x * y + w * z));
        o[5] = (float)(k * (w * w - x * x + y * y - z * z));
        o[6] = (float)(k * 2.0 * (y * z - w * x));
        o[7] = (float)(kBmKappa * (ty - T->m.cy));
        o[8] = (float)(k * 2.0 * (x * z - w * y));
        o[9] = (float)(k * 2.0 * (y * z + w * x));
        o[10] = (float)(k * (w * w - x * x - y * y + z * z));
        o[11] = (float)(kBmKappa * (tz - T->m.cz));
        {   // the exact path's row: the pose's own numbers and its index, 64 bytes
            double *e = T->rt_exact + listed * 8;
            e[0] = tx; e[1] = ty; e[2] = tz; e[3] = w; e[4] = x; e[5] = y; e[6] = z;
            e[7] = __longlong_as_double((long long)pose);
        }
        {   // the pose's interface-flag words (the exact path sets bits, pose_energy_finish reads them)
            const int words = T->m.rec_flag_words + T->m.lig.flag_words;
            uint32_t *f = T->flags + pose * (size_t)words;
            for (int k = 0; k < words; k++) f[k] = 0u;
        }
        if (T->exact_fix) T->exact_fix[listed] = 0;
        if (T->exact_pairs) T->exact_pairs[listed] = 0;
        if (T->amp != nullptr) {   // the pose's mode amplitudes as the pair kernel loads them (f32: their rounding is part of eps), and whether it is WILD
            float *am = T->amp + listed * kBmAmpFloats;
            float n2_rec = 0.f, n2_lig = 0.f;   // |amplitudes|^2 of either molecule (the f32 values the kernels multiply with)
#pragma unroll
            for (int k = 0; k < kBmMaxModes; k++) {
                const float ar = k < T->m.anm_rec ? (float)row[7 + k] : 0.f, al = k < T->m.anm_lig ? (float)row[7 + T->m.anm_rec + k] : 0.f;
                am[k] = ar;
                am[kBmMaxModes + k] = al;
                n2_rec = __builtin_fmaf(ar, ar, n2_rec);
                n2_lig = __builtin_fmaf(al, al, n2_lig);
            }
            // Cauchy-Schwarz: no coordinate of an atom, and no partial sum of its ten terms, moves further than |a|_2 x the largest 2-norm
            // of an atom's mode components of that coordinate (BmModel); 1.0001: the roundings of this sum, of the root and of the ten
            // terms.  (NaN or infinite amplitudes: not below the bound -> wild -> the exact path, where the reference's arithmetic decides)
            const float na_rec = sqrtf(n2_rec) * 1.0001f, na_lig = sqrtf(n2_lig) * 1.0001f;
            am[20] = na_rec * T->m.rec_mode_norm <= kBmWildUnits && na_lig * T->m.lig_mode_norm <= kBmWildUnits ? 0.f : 1.f;
            am[21] = na_lig * T->m.lig_mode_norm_vec;   // how far a ligand atom can move (the culling kernel's sphere test)
            am[22] = am[23] = 0.f;
            double *ax = T->amp_exact + listed * (2 * kBmMaxModes);
#pragma unroll
            for (int k = 0; k < kBmMaxModes; k++) {
                ax[k] = k < T->m.anm_rec ? row[7 + k] : 0.0;
                ax[kBmMaxModes + k] = k < T->m.anm_lig ? row[7 + T->m.anm_rec + k] : 0.0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_rec_boxes (ANM): the boxes of the FLEXED receptor, per row of the pass (src/dfire.rs:304-320 moves the receptor's atoms
// with the pose's amplitudes).  Wave = (receptor tile, 64 rows), LANE = ROW: the lane keeps its row's ten amplitudes, the tile's
// static records and modes are wave-uniform -- scalar loads, straight into the scalar operands of the packed multiply-adds
// (two atoms an instruction: 15 per atom pair and coordinate triple... 30 per pair of atoms) --, and the boxes are running
// minima in the lane's registers: no cross-lane reduction at all.  f32 throughout: the atom of a pose that is not WILD lies
// within rec_box_pad of the exactly flexed one (scorer.cpp), the boxes are widened by that; a wild row's boxes are not used
// (the culling kernel lists all of its blocks).
// (History: first the per-pose boxes came from dfire_packed_prepare -- f64, by pose of the launch whatever the list --, 159 us for
// 16 384 poses of 2uuy; then from a lane = atom form of this kernel with DPP box reductions, 103 us, bound by its 120 vector
// instructions per (row, tile).)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dfire_bm_rec_boxes(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const size_t rows = bm_rows(T);
    const int n_rt = T->m.rec_n_tiles;
    const size_t n_items = (rows + 63) / 64 * (size_t)n_rt;
    const float pad = T->m.rec_box_pad;
    typedef const __attribute__((address_space(4))) float const_f32;
    for (size_t item = (size_t)blockIdx.x * 4 + wave; item < n_items; item += (size_t)gridDim.x * 4) {
        const int RT = (int)(item % (unsigned)n_rt);
        const size_t row = item / (unsigned)n_rt * 64 + (size_t)lane;
        const bool have = row < rows;
        v2f amp[kBmMaxModes / 2];
        float wild = 1.f;
        {
            const float4 *am = reinterpret_cast<const float4 *>(T->amp + (have ? row : 0) * kBmAmpFloats);
            const float4 q0 = am[0], q1 = am[1], q2 = am[2];
            amp[0] = v2f{q0.x, q0.y}; amp[1] = v2f{q0.z, q0.w}; amp[2] = v2f{q1.x, q1.y}; amp[3] = v2f{q1.z, q1.w}; amp[4] = v2f{q2.x, q2.y};
            if (have) wild = reinterpret_cast<const float *>(am)[2 * kBmMaxModes];
        }
        float tlo[3] = {INFINITY, INFINITY, INFINITY}, thi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 1
        for (int sb = 0; sb < 8; sb++) {
            const_f32 *modes = (const_f32 *)(uintptr_t)(T->m.rec_modes_f32 + ((size_t)RT * 8 + (size_t)sb) * kBmModeFloats);
            const_f32 *recs = (const_f32 *)(uintptr_t)(T->m.rec_pairs + (size_t)RT * 32 + (size_t)sb * 4);   // 4 records: x0 x1 y0 y1 z0 z1 . .
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int first = RT * 64 + sb * 8 + 2 * q;   // (padding atoms -- the tail of the last tile -- are in no box)
                if (first >= T->m.rec_n_real) break;
                const bool both = first + 1 < T->m.rec_n_real;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    v2f f{recs[q * 8 + 2 * c], recs[q * 8 + 2 * c + 1]};
#pragma unroll
                    for (int k2 = 0; k2 < kBmMaxModes / 2; k2++) {
                        const_f32 *m = modes + ((q * 3 + c) * kBmMaxModes + 2 * k2) * 2;
                        f = __builtin_elementwise_fma(v2f{amp[k2].x, amp[k2].x}, v2f{m[0], m[1]}, f);
                        f = __builtin_elementwise_fma(v2f{amp[k2].y, amp[k2].y}, v2f{m[2], m[3]}, f);
                    }
                    const float second = both ? f.y : f.x;
                    lo[c] = fminf(lo[c], fminf(f.x, second));
                    hi[c] = fmaxf(hi[c], fmaxf(f.x, second));
                }
            }
            const float sub_cut = ((const_f32 *)(uintptr_t)(T->m.rec_sub + (size_t)RT * 8 + (size_t)sb))[3];   // (TiledBox::pad0: the static image's reach)
            if (have && wild == 0.f)
                reinterpret_cast<BmCullBox *>(T->anm_sub)[(row * (size_t)n_rt + RT) * 8 + sb] =
                    BmCullBox{v2f{lo[0] - pad, -(hi[0] + pad)}, v2f{lo[1] - pad, -(hi[1] + pad)}, v2f{lo[2] - pad, -(hi[2] + pad)}, sub_cut, 0.f};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                tlo[c] = fminf(tlo[c], lo[c]);
                thi[c] = fmaxf(thi[c], hi[c]);
            }
        }
        if (have && wild == 0.f)
            T->anm_tile[row * (size_t)n_rt + RT] = TiledBox{tlo[0] - pad, tlo[1] - pad, tlo[2] - pad, ((const_f32 *)(uintptr_t)(T->m.rec_tile + RT))[3], thi[0] + pad, thi[1] + pad, thi[2] + pad, 0.f};
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_cull: wave = (ligand tile, kBmCullPoses consecutive poses of the launch).  Phase 1, pose by pose: the tile's
// atoms posed in f32, boxes, the 64 x 64 and 8 x 8 box tests; the block masks go to LDS.  Phase 2, lane = receptor
// tile: ONE atomic per tile pair for all the poses of the wave (the lists of a small complex have few heads: one
// returning atomic per pose and tile pair serialises on them), then the entries.
// ---------------------------------------------------------------------------------------------
// ANM (DFIRE with normal modes, src/dfire.rs:288-320): the receptor's boxes differ per pose -- read from what dfire_bm_rec_boxes
// wrote for the row (BmLaunch::anm_sub / anm_tile) instead of the static ones in LDS -- and the ligand tile's atoms are
// flexed after the affine map: + sum_k amplitude_k x mode_k of the atom (kappa x, f32; the amplitudes from the [row] table).
template <bool COUNT, bool ANM>
__global__ __launch_bounds__(kBmCullWaves * 64) void dfire_bm_cull(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    // LDS: the receptor's subtile and tile boxes (read by every item; a global load per surviving tile was most of an
    // item's time), then [wave][pose of the wave][receptor tile]: block mask, 0 = not within reach
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_cull[];   // (16-byte aligned: a box is two ds_read_b128)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int n_lt = T->m.lig.n_tiles, n_rt = T->m.rec_n_tiles;
    BmCullBox *s_sub = reinterpret_cast<BmCullBox *>(s_cull);          // [n_rt * 8]
    TiledBox *s_tile = reinterpret_cast<TiledBox *>(s_sub + (size_t)n_rt * 8);   // [n_rt]
    // per wave: the HITS (pose of the wave, receptor tile, block mask) of the item so far, room for bm_cull_hit_tiles(n_rt) * n_rt of
    // them (a pose adds at most n_rt; the list is flushed when the next pose might not fit), and per receptor tile a counter
    // and the first entry of the wave in that tile pair's list
    const int hit_cap = bm_cull_hit_cap(n_rt);
    unsigned char *s_wave = reinterpret_cast<unsigned char *>(s_tile + n_rt) + (size_t)wave * bm_cull_wave_lds(n_rt);
    unsigned long long *s_hmask = reinterpret_cast<unsigned long long *>(s_wave);          // [hit_cap]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_hmask + hit_cap);                      // [n_rt]
    uint32_t *s_base = s_cnt + n_rt;                                                        // [n_rt]
    uint32_t *s_hkey = s_base + n_rt;                                                       // [hit_cap]: pose of the wave << 16 | receptor tile
    unsigned short *s_hrank = reinterpret_cast<unsigned short *>(s_hkey + hit_cap);         // [hit_cap]: place among the wave's hits of that tile pair
    if (bm_rows(T) == 0) return;   // (a quiet GSO step)
    if (!ANM) {
        static_assert(sizeof(TiledBox) == 32, "two 16-byte pieces");
        const uint4 *src_sub = reinterpret_cast<const uint4 *>(T->m.rec_sub), *src_tile = reinterpret_cast<const uint4 *>(T->m.rec_tile);
        for (int k = threadIdx.x; k < n_rt * 8; k += kBmCullWaves * 64) {
            const uint4 lo = src_sub[2 * k], hi = src_sub[2 * k + 1];   // a TiledBox: lo x y z ., hi x y z .
            reinterpret_cast<uint4 *>(s_sub)[2 * k] = uint4{lo.x, hi.x ^ 0x80000000u, lo.y, hi.y ^ 0x80000000u};
            reinterpret_cast<uint4 *>(s_sub)[2 * k + 1] = uint4{lo.z, hi.z ^ 0x80000000u, lo.w, 0u};   // (lo.w: TiledBox::pad0, the subtile's reach)
        }
        for (int k = threadIdx.x; k < n_rt * 2; k += kBmCullWaves * 64) reinterpret_cast<uint4 *>(s_tile)[k] = src_tile[k];
        __syncthreads();
    }
    const size_t rows = bm_rows(T);
    // poses per item: kBmCullPoses for a launch that fills the chip anyway; a small one (the late steps of a GSO: the count is
    // known on the device only) takes 2, so that four times the waves share its latency -- a wave walks its poses one after the other
    const int group_poses = rows * (size_t)n_lt >= (size_t)gridDim.x * kBmCullWaves * kBmCullPoses ? kBmCullPoses : 2;
    const size_t n_items = (rows + group_poses - 1) / group_poses * (size_t)n_lt;
    const float ubound = T->m.ubound, pad = T->m.box_pad;
    const int bj = lane & 7;
    // The waves of a workgroup are independent (no barrier).  Items differ several times over in length (a ligand tile at
    // the interface lists twenty receptor tiles, one on the far side none), and a fixed stride through the items gives a wave
    // only the tiles of one residue class: handed out statically the waves of a 1k4c launch lived 129 to 332 us.  So a wave
    // DRAWS its items, one ahead, from one of up to kBmCullQueues counters (few counters serialise: 53 k atomics over 8
    // addresses took twice the kernel's time): queue q holds a contiguous range of items (every ligand tile equally often) and
    // is served by the workgroups q, q + Q, ...
    if (n_items == 0) return;   // (a quiet GSO step)
    const uint32_t n_queues = gridDim.x < (unsigned)kBmCullQueues ? gridDim.x : (uint32_t)kBmCullQueues;
    const uint32_t queue = blockIdx.x % n_queues;
    const size_t per_queue = (n_items + n_queues - 1) / n_queues;
    const size_t queue_first = (size_t)queue * per_queue;
    const size_t queue_end = queue_first + per_queue < n_items ? queue_first + per_queue : n_items;
    uint32_t *queue_counter = T->job_count + kBmCounters + queue;
    auto draw = [&]() {
        uint32_t ticket = 0;
        if (lane == 0) ticket = atomicAdd(queue_counter, 1u);
        return ticket;   // lane 0, on its way
    };
    uint32_t next_ticket = draw();
#ifdef LD_BM_DIAG_CULL_TIMES   // (diagnostic builds: where a culling wave's time goes, in 10 ns ticks, into the pair kernel's debug buffer)
    unsigned long long ct_start = __builtin_amdgcn_s_memrealtime(), ct_draw = 0, ct_box = 0, ct_loop = 0, ct_flush = 0, ct_items = 0;
#define LD_CT(x) x
#else
#define LD_CT(x)
#endif
    for (;;) {
    LD_CT(const unsigned long long ct_a = __builtin_amdgcn_s_memrealtime();)
    const size_t item = queue_first + (uint32_t)__builtin_amdgcn_readfirstlane((int)next_ticket);
    LD_CT(const unsigned long long ct_b = __builtin_amdgcn_s_memrealtime(); ct_draw += ct_b - ct_a;)
    if (item >= queue_end) break;
    LD_CT(ct_items++;)
    const size_t group = item / (unsigned)n_lt;
    const int lt = (int)(item % (unsigned)n_lt);
    const size_t listed0 = group * (size_t)group_poses;

    // The rigid form boxes the item's poses all at once, lane = (pose of the item, ligand subtile): below.  The ANM form keeps lane = atom
    // (its atom's thirty mode components live in the lane's registers), pose by pose.
    const int la = lt * 64 + lane;
    float4 loc = float4{0.f, 0.f, 0.f, 0.f}, sphere = loc;
    if (ANM) {
        loc = reinterpret_cast<const float4 *>(T->m.lig_local)[la];
        sphere = reinterpret_cast<const float4 *>(T->m.lig_tile_sphere)[lt];
    }
    const bool valid = loc.w != 0.f;
    // this lane's receptor tile box (the first 64 tiles; larger receptors read the rest per pose)
    const TiledBox no_tile = TiledBox{INFINITY, INFINITY, INFINITY, 0.f, -INFINITY, -INFINITY, -INFINITY, 0.f};
    TiledBox my_tile = no_tile;
    if (!ANM && lane < n_rt) my_tile = s_tile[lane];
    // ANM: this lane's atom's modes (kappa x, f32: [mode][x y z], 32 floats an atom), kept for the item's poses
    float mode_x[kBmMaxModes], mode_y[kBmMaxModes], mode_z[kBmMaxModes];
    if (ANM) {
        const float4 *mp = reinterpret_cast<const float4 *>(T->m.lig_modes_atom) + (size_t)la * 8;
        float4 q[8];
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = mp[k];
        const float *f = reinterpret_cast<const float *>(q);
#pragma unroll
        for (int k = 0; k < kBmMaxModes; k++) {
            mode_x[k] = f[3 * k];
            mode_y[k] = f[3 * k + 1];
            mode_z[k] = f[3 * k + 2];
        }
    }

    // the item's poses and their affine maps: lane g loads those of pose g, all in flight together
    long long my_pose = -1;   // (only its sign is used: the workspace of a pass goes by row)
    const uint32_t my_row = (uint32_t)(listed0 + lane);
    float4 my_a0 = float4{0.f, 0.f, 0.f, 0.f}, my_a1 = my_a0, my_a2 = my_a0;
    if (lane < group_poses && listed0 + lane < rows) my_pose = bm_pose_of(T, listed0 + lane);
    float my_amp[kBmMaxModes];   // ANM: the pose's ligand amplitudes (lane g: pose g)
    float my_wild = 0.f, my_flex = 0.f;
#pragma unroll
    for (int k = 0; k < kBmMaxModes; k++) my_amp[k] = 0.f;
    if (ANM && my_pose >= 0) {
        const float4 *ap = reinterpret_cast<const float4 *>(T->rt + (size_t)my_row * 12);
        my_a0 = ap[0];
        my_a1 = ap[1];
        my_a2 = ap[2];
        {
            const float *am = T->amp + (size_t)my_row * kBmAmpFloats + kBmMaxModes;
#pragma unroll
            for (int k = 0; k < kBmMaxModes; k++) my_amp[k] = am[k];
            my_wild = am[kBmMaxModes];   // (the row's float 20)
            my_flex = am[kBmMaxModes + 1];
        }
    }
    auto pose_lane = [](float v, int g) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), g)); };
    auto affine_of = [&](int g) {
        return Affine{pose_lane(my_a0.x, g), pose_lane(my_a0.y, g), pose_lane(my_a0.z, g), pose_lane(my_a0.w, g),
                      pose_lane(my_a1.x, g), pose_lane(my_a1.y, g), pose_lane(my_a1.z, g), pose_lane(my_a1.w, g),
                      pose_lane(my_a2.x, g), pose_lane(my_a2.y, g), pose_lane(my_a2.z, g), pose_lane(my_a2.w, g)};
    };

    // the tile's bounding sphere posed by all the item's poses at once: lane g, with the map it holds
    float my_sx = 0.f, my_sy = 0.f, my_sz = 0.f;
    if (ANM) bm_apply(Affine{my_a0.x, my_a0.y, my_a0.z, my_a0.w, my_a1.x, my_a1.y, my_a1.z, my_a1.w, my_a2.x, my_a2.y, my_a2.z, my_a2.w}, sphere.x, sphere.y, sphere.z, my_sx, my_sy, my_sz);

    // ---- rigid form: the boxes of ALL the item's poses at once, lane = (pose g = lane / 8, ligand subtile s = lane % 8).  The lane poses
    // the 8 atoms of its subtile with its pose's map (bm_apply, the operations the ANM form's lane = atom code and the exact path use:
    // the same bits), keeps the running minima and maxima in its registers -- an atom outside the frame or a padding atom enters as
    // the empty box, as below -- and widens them; three DPP levels over the 8 lanes of a pose give the tile's box.  Until round 6
    // every pose of the item was boxed on its own with lane = atom: 12 v_readlane for the map, 9 multiply-adds, the frame test, 36
    // DPP min / max for the subtile and tile boxes and a bounding-sphere pre-test in front -- ~120 vector instructions per pose and
    // tile, over half of this kernel's; now ~45, and the poses' latencies overlap instead of following each other.
    BoxRegs my_sub{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, my_whole = my_sub;
    if constexpr (!ANM) {
        static_assert(kBmCullPoses == 8, "lane = (pose of the item, ligand subtile)");
        const int my_g = lane >> 3, my_s = lane & 7;
        const bool have = __shfl((int)(my_pose >= 0), my_g, 64) != 0;   // (lane g holds pose g of the item)
        const float4 *ap = reinterpret_cast<const float4 *>(T->rt + (have ? listed0 + (size_t)my_g : (size_t)0) * 12);
        const float4 q0 = ap[0], q1 = ap[1], q2 = ap[2];
        const float4 *atoms = reinterpret_cast<const float4 *>(T->m.lig_local) + ((size_t)lt * 64 + (size_t)my_s * 8);
        float4 at[8];
#pragma unroll
        for (int k = 0; k < 8; k++) at[k] = atoms[k];
        const Affine A{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float fx, fy, fz;
            bm_apply(A, at[k].x, at[k].y, at[k].z, fx, fy, fz);
            const bool inside = fabsf(fx) <= ubound && fabsf(fy) <= ubound && fabsf(fz) <= ubound;
            const float pen = bm_select(__builtin_amdgcn_ballot_w64(have && at[k].w != 0.f && inside), 0.0f, INFINITY);
            my_sub.lox = fminf(my_sub.lox, fx + pen); my_sub.loy = fminf(my_sub.loy, fy + pen); my_sub.loz = fminf(my_sub.loz, fz + pen);
            my_sub.hix = fmaxf(my_sub.hix, fx - pen); my_sub.hiy = fmaxf(my_sub.hiy, fy - pen); my_sub.hiz = fmaxf(my_sub.hiz, fz - pen);
        }
        const float wide = pad + 2.384185791015625e-07f * ubound;   // (as below)
        my_sub.lox -= wide; my_sub.loy -= wide; my_sub.loz -= wide;
        my_sub.hix += wide; my_sub.hiy += wide; my_sub.hiz += wide;
        my_whole = my_sub;
        box_reduce8(my_whole);   // (every lane of pose g: the tile's box in pose g)
    }

    uint32_t n_hits = 0;                 // wave-uniform: hits listed and not flushed yet
    // ---- the list -> entries.  One LDS atomic per hit (its place among the wave's hits of the tile pair), ONE global atomic per
    // tile pair for the whole wave (the lists of a small complex have few heads: one returning atomic per pose and tile pair
    // serialises on them), then lane = hit writes the entry: 12 bytes (the pair kernel reads the pose's affine map from the
    // [row][12] table, which stays in L2; a copy per entry was 48 more bytes written here and read per item from HBM there).
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = lane; k < n_rt; k += 64) s_cnt[k] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t h = (uint32_t)lane; h < n_hits; h += 64) s_hrank[h] = (unsigned short)atomicAdd(&s_cnt[s_hkey[h] & 0xffffu], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = lane; k < n_rt; k += 64) {
            const uint32_t total = s_cnt[k];
#ifdef LD_BM_DIAG_NO_TP_ATOMIC   // (diagnostic builds: timing only, wrong lists -- what the culling kernel takes without its returning atomics on the tile pairs' counters)
            if (total) { s_base[k] = (uint32_t)(listed0 % 1024u); T->tp_count[(size_t)lt * n_rt + k] = s_base[k] + total; }
#else
            if (total) s_base[k] = atomicAdd(&T->tp_count[(size_t)lt * n_rt + k], total);
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t h0 = 0; h0 < n_hits; h0 += 64) {
            const uint32_t h = h0 + (uint32_t)lane;
            const bool act = h < n_hits;
            const uint32_t key = act ? s_hkey[h] : 0u;
            const int g = (int)(key >> 16), RT = (int)(key & 0xffffu);
            if (act) {
                const unsigned long long mask = s_hmask[h];
                const uint32_t idx = s_base[RT] + s_hrank[h];
                const size_t at = ((size_t)lt * n_rt + RT) * T->cap + idx;
                const uint32_t row = (uint32_t)listed0 + (uint32_t)g;
                T->ent_row[at] = row;
                T->ent_mask[at] = mask;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the list is read before it is written again
        n_hits = 0;
    };

    LD_CT(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); asm volatile("" :: "v"(my_whole.lox), "v"(my_whole.hiz)); const unsigned long long ct_c = __builtin_amdgcn_s_memrealtime(); ct_box += ct_c - ct_b;)
    long long pose_of[kBmCullPoses];   // wave-uniform
#pragma unroll
    for (int g = 0; g < kBmCullPoses; g++) {
        pose_of[g] = (long long)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_pose, g) |
                                 (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)my_pose >> 32), g) << 32);
        if (pose_of[g] < 0) continue;
        if (n_hits + (uint32_t)n_rt > (uint32_t)hit_cap) flush();
        // the hits of one ballot of receptor tiles, lane-distributed, written to the list together (one LDS store per lane
        // instead of a scalar branch, five register moves and two stores per hit)
        uint32_t held = 0;                  // wave-uniform
        uint32_t held_lo = 0, held_hi = 0, held_key = 0;   // (lane k: hit k, written by v_writelane)
        auto put_held = [&]() {
            if ((uint32_t)lane < held) {
                s_hmask[n_hits + (uint32_t)lane] = (unsigned long long)held_hi << 32 | held_lo;
                s_hkey[n_hits + (uint32_t)lane] = held_key;
            }
            n_hits += held;
            held = 0;
        };
        if (ANM && pose_lane(my_wild, g) != 0.f) {
            // A WILD pose (amplitudes beyond what the f32 bounds cover, or not finite): no box is trusted -- every block of every
            // tile pair is listed, the pair kernel sends them whole to the exact path, and the reference's arithmetic decides.
            for (int base = 0; base < n_rt; base += 64) {
                held = (uint32_t)(n_rt - base < 64 ? n_rt - base : 64);
                held_lo = held_hi = 0xffffffffu;
                held_key = (uint32_t)(g << 16 | (base + lane));
                put_held();
            }
            if (COUNT && lane == 0) T->tile_tested[(listed0 + g) * (size_t)n_lt + lt] = (uint32_t)n_rt * 64u;
            continue;
        }
        // ANM: this pose's receptor boxes (global memory, the pose's image) and how far its amplitudes can move a ligand atom
        const TiledBox *tiles_g = nullptr;
        const BmCullBox *subs_g = nullptr;
        float flex_reach = 0.f;
        if (ANM) {
            tiles_g = T->anm_tile + (listed0 + (size_t)g) * n_rt;
            subs_g = reinterpret_cast<const BmCullBox *>(T->anm_sub) + (listed0 + (size_t)g) * n_rt * 8;
            my_tile = lane < n_rt ? tiles_g[lane] : no_tile;
            flex_reach = pose_lane(my_flex, g);   // (dfire_bm_pose: |amplitudes|_2 x the largest mode vector norm of a ligand atom)
        }
        auto tile_at = [&](int i) { return ANM ? tiles_g[i] : s_tile[i]; };
        auto sub_at = [&](int i) { return ANM ? subs_g[i] : s_sub[i]; };   // subtile i of the receptor as {lo, -hi} pairs
        BoxRegs sub, whole;   // lane: the box of ligand subtile lane / 8 in pose g; wave-uniform: the tile's
        if constexpr (ANM) {
            {   // A tile whose bounding sphere stays beyond the cutoff of every receptor tile's box has nothing to list (most tiles
                // of a large ligand, in most poses): one point posed and one test per receptor tile instead of 64 atoms posed,
                // their boxes and the box tests.
                const float sx = pose_lane(my_sx, g), sy = pose_lane(my_sy, g), sz = pose_lane(my_sz, g);
                const float reach = 120.0f * 1.0001f + sphere.w + pad + flex_reach;   // (8 * 15 A, the sphere's radius, the affine map's error; ANM: what the modes can add)
                bool any_near = false;
                for (int base = 0; base < n_rt && !any_near; base += 64) {
                    const TiledBox tb = base == 0 ? my_tile : (base + lane < n_rt ? tile_at(base + lane) : my_tile);
                    const float gx = fmaxf(0.f, fmaxf(tb.lox - sx, sx - tb.hix));
                    const float gy = fmaxf(0.f, fmaxf(tb.loy - sy, sy - tb.hiy));
                    const float gz = fmaxf(0.f, fmaxf(tb.loz - sz, sz - tb.hiz));
                    any_near = __builtin_amdgcn_ballot_w64(base + lane < n_rt && gx * gx + gy * gy + gz * gz <= reach * reach) != 0ull;
                }
                if (!any_near) {
                    if (COUNT && lane == 0) T->tile_tested[(listed0 + g) * (size_t)n_lt + lt] = 0;
                    continue;
                }
            }
            const Affine A = affine_of(g);
            float fx, fy, fz;
            bm_apply(A, loc.x, loc.y, loc.z, fx, fy, fz);
            if (ANM) {   // + sum_k amplitude_k mode_k (src/dfire.rs:288-301: after the rotation and translation, in the receptor's frame)
    #pragma unroll
                for (int k = 0; k < kBmMaxModes; k++) {
                    const float c = pose_lane(my_amp[k], g);
                    fx = __builtin_fmaf(c, mode_x[k], fx);
                    fy = __builtin_fmaf(c, mode_y[k], fy);
                    fz = __builtin_fmaf(c, mode_z[k], fz);
                }
            }
            const bool inside = fabsf(fx) <= ubound && fabsf(fy) <= ubound && fabsf(fz) <= ubound;

            // An atom outside the frame is more than the cutoff away from every receptor atom (the frame holds the receptor's
            // box + 16 A): it joins no box; the pairs it still meets inside blocks of its subtile read "miss", as they must.
            // The lane's point as a box, an excluded lane's as the empty box (lo = +inf, hi = -inf): ONE select -- a penalty of 0 or
            // +inf added for the minima, subtracted for the maxima -- instead of six.  (A select on VCC, `v_cndmask_b32_e32`, holds the
            // vector port for 23 cycles on gfx950, five plain instructions' worth -- tools/microbench/valu_rate.hip,
            // profiles/r06_valu_issue_rates.txt --; the form with the mask in a scalar register pair costs 4.4: bm_select.  An excluded
            // lane whose coordinate is itself infinite or NaN yields NaN on one side: v_min / v_max return the other operand, i.e. it
            // still joins no box.)
            const float pen = bm_select(__builtin_amdgcn_ballot_w64(valid && inside), 0.0f, INFINITY);
            sub = BoxRegs{fx + pen, fy + pen, fz + pen, fx - pen, fy - pen, fz - pen};
            box_reduce8(sub);
            {   // widen: the f32 positions are within box_pad of the exactly posed ones; the boxes are built from fl32(u), the true u within
                // 2^-24 |u| of it: |u| <= ubound for every atom in a box, so 2^-22 ubound on top of the pad is outwards (the relative
                // widening atom by atom, box_widen, was 18 vector instructions a pose and tile; an infinite side stays infinite)
                const float wide = pad + 2.384185791015625e-07f * ubound;
                sub.lox -= wide; sub.loy -= wide; sub.loz -= wide;
                sub.hix += wide; sub.hiy += wide; sub.hiz += wide;
            }
            whole = sub;   // (widening is monotone: the union of the widened subtile boxes IS the widened tile box)
            box_reduce64_from8(whole);
            whole.lox = lane63_f32(whole.lox); whole.loy = lane63_f32(whole.loy); whole.loz = lane63_f32(whole.loz);
            whole.hix = lane63_f32(whole.hix); whole.hiy = lane63_f32(whole.hiy); whole.hiz = lane63_f32(whole.hiz);
        } else {
            // (boxed above for all the item's poses: the subtile's box from the lane that holds it -- six ds_bpermute, the LDS crossbar,
            // not the vector port --, the tile's from the first lane of the pose)
            const int src = g * 8 + (lane >> 3);
            sub.lox = __shfl(my_sub.lox, src, 64); sub.loy = __shfl(my_sub.loy, src, 64); sub.loz = __shfl(my_sub.loz, src, 64);
            sub.hix = __shfl(my_sub.hix, src, 64); sub.hiy = __shfl(my_sub.hiy, src, 64); sub.hiz = __shfl(my_sub.hiz, src, 64);
            whole.lox = pose_lane(my_whole.lox, g * 8); whole.loy = pose_lane(my_whole.loy, g * 8); whole.loz = pose_lane(my_whole.loz, g * 8);
            whole.hix = pose_lane(my_whole.hix, g * 8); whole.hiy = pose_lane(my_whole.hiy, g * 8); whole.hiz = pose_lane(my_whole.hiz, g * 8);
        }
        const v2f sub_x{-sub.hix, sub.lox}, sub_y{-sub.hiy, sub.loy}, sub_z{-sub.hiz, sub.loz};

        // 64 x 64 tile boxes, 64 receptor tiles per ballot; then the 8 x 8 subtile boxes of every surviving tile
        uint32_t tested = 0;
        for (int base = 0; base < n_rt; base += 64) {
            bool tile_near = false;
            // (against the receptor box's own reach: the full cutoff unless its atoms' rows of the potential are zero; a counting launch
            // counts every pair inside the cutoff whatever the table holds)
            if (base == 0) tile_near = lane < n_rt && box_gap2(whole, my_tile) <= (COUNT ? kBmBoxCut : my_tile.pad0);
            else if (base + lane < n_rt) {
                const TiledBox tb = tile_at(base + lane);
                tile_near = box_gap2(whole, tb) <= (COUNT ? kBmBoxCut : tb.pad0);
            }
            unsigned long long rtmask = __builtin_amdgcn_ballot_w64(tile_near);
            if (rtmask) {
                // one surviving tile at a time, the next one's subtile boxes loaded while this one's are tested (the last trip loads
                // its own again: no branch around the loads)
                // (two copies of the step, the boxes alternating between two sets of registers: no moves)
                int RT_a = base + __ffsll(rtmask) - 1, RT_b = RT_a;
                rtmask &= rtmask - 1;
                BmCullBox nb_a = sub_at(RT_a * 8 + bj), nb_b;
                auto step = [&](int RT, const BmCullBox &nb, int &RT_next, BmCullBox &nb_next) {
                    const bool more = rtmask != 0ull;
                    RT_next = more ? base + __ffsll(rtmask) - 1 : RT;
                    rtmask &= rtmask - 1;
                    nb_next = sub_at(RT_next * 8 + bj);
#ifdef LD_BM_DIAG_NO_TRACKED   // (diagnostic builds: timing only, wrong sums -- no block with a receptor subtile that holds a tracked atom, i.e. 1k4c's beads)
                    const unsigned long long smask = __builtin_amdgcn_ballot_w64(bm_cull_gap2(sub_x, sub_y, sub_z, nb) <= (COUNT ? kBmBoxCut : nb.cut) && T->m.rec_sub_tracked[RT * 8 + bj] == 0);
#else
                    const unsigned long long smask = __builtin_amdgcn_ballot_w64(bm_cull_gap2(sub_x, sub_y, sub_z, nb) <= (COUNT ? kBmBoxCut : nb.cut));  // bit = ligand subtile (lane >> 3) * 8 + receptor subtile
#endif
                    if (smask) {   // hit `held` of this ballot stays in lane `held` until the ballot's tiles are done
                        held_lo = (uint32_t)bm_writelane((int)(uint32_t)smask, (int)held, (int)held_lo);
                        held_hi = (uint32_t)bm_writelane((int)(uint32_t)(smask >> 32), (int)held, (int)held_hi);
                        held_key = (uint32_t)bm_writelane(g << 16 | RT, (int)held, (int)held_key);
                        held++;
                    }
                    if (COUNT) tested += (uint32_t)__popcll(smask);
                    return more;
                };
                for (;;) {
                    if (!step(RT_a, nb_a, RT_b, nb_b)) break;
                    if (!step(RT_b, nb_b, RT_a, nb_a)) break;
                }
            }
            put_held();   // (a ballot's 64 receptor tiles make at most 64 hits)
        }
        if (COUNT && lane == 0) T->tile_tested[(listed0 + g) * (size_t)n_lt + lt] = tested;
    }
    LD_CT(const unsigned long long ct_d = __builtin_amdgcn_s_memrealtime(); ct_loop += ct_d - ct_c;)
    next_ticket = draw();   // (here, not at the item's start: memory operations return in order, and the item's loads would wait for it)
    flush();
    LD_CT(ct_flush += __builtin_amdgcn_s_memrealtime() - ct_d;)
    }
#ifdef LD_BM_DIAG_CULL_TIMES
    {
        const size_t w = (size_t)blockIdx.x * kBmCullWaves + wave;
        if (T->debug != nullptr && lane == 0 && w % 5 == 0 && w / 5 < 2048) {   // (every fifth wave: all workgroups of the launch are sampled)
            unsigned long long *d = T->debug + w / 5 * 8;
            d[0] = ct_start; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = ct_items; d[3] = ct_draw; d[4] = ct_box; d[5] = ct_loop; d[6] = ct_flush; d[7] = w;
        }
    }
#endif
#undef LD_CT
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_plan: every tile pair's entries cut into parts of P entries; a JOB = (tile pair, part, partial-sum row) is
// what one wave of dfire_bm_pairs walks.  P = kBmPartEntries for a large launch (the longer a job, the more batches
// share each staging of a block's table rows); a launch with few entries -- the late steps of a GSO run, when few
// glowworms still move -- is cut finer so that its jobs still spread over every wave of the chip.
// ---------------------------------------------------------------------------------------------
// entries per part of a tile pair with n entries: its ceil(n / P) parts are EQUAL (a multiple of 64 each; with parts of P and a
// remainder, a tile pair of 1070 entries -- the average of a 1k4c launch -- was one full job and one of 46 entries that paid
// the same set-ups for batches a quarter full)
__device__ __forceinline__ uint32_t bm_part_size(uint32_t n, uint32_t P) {
    const uint32_t parts = (n + P - 1) / P;
    return parts ? ((n + parts - 1) / parts + 63u) / 64u * 64u : P;
}

__global__ __launch_bounds__(1024) void dfire_bm_plan(const BmLaunch launch_arguments) {
    // One workgroup: P, and the list of (tile pair, part) pairs in any order (dfire_bm_order sorts the jobs).
    BmArgs *T = LD_BM_ARGS;
    __shared__ uint32_t s_total, s_at;
    const int tid = threadIdx.x;
    const uint32_t n_tp = (uint32_t)(T->m.lig.n_tiles * T->m.rec_n_tiles);
    if (tid == 0) s_total = s_at = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t tp = tid; tp < n_tp; tp += 1024) mine += T->tp_count[tp];
    if (mine) atomicAdd(&s_total, mine);
    __syncthreads();
    // about one (tile pair, part) pair per wave of the pair kernel, i.e. kBmJobRows jobs per wave
    const uint32_t waves = (uint32_t)(T->pairs_groups > 0 ? T->pairs_groups : 256) * kBmWavesPerCu;
#ifndef LD_BM_P_FACTOR
#define LD_BM_P_FACTOR 8
#endif
    uint32_t P = (uint32_t)(((unsigned long long)s_total * LD_BM_P_FACTOR / waves + 63u) / 64u * 64u);
    const uint32_t part_cap = T->part_cap ? T->part_cap : (uint32_t)kBmPartEntries;   // (the ANM form of the pair kernel holds 512 entries a job)
    P = P < 64u ? 64u : P > part_cap ? part_cap : P;
    for (uint32_t tp = tid; tp < n_tp; tp += 1024) {
        const uint32_t n = T->tp_count[tp];
        if (n == 0) continue;
        const uint32_t size = bm_part_size(n, P), parts = (n + size - 1) / size;
        const uint32_t at = atomicAdd(&s_at, parts);
        for (uint32_t k = 0; k < parts; k++) {
            T->jobs[2 * (at + k)] = tp;
            T->jobs[2 * (at + k) + 1] = k * size;
        }
    }
    __syncthreads();
    if (tid == 0) {
        T->job_count[0] = s_at;
        T->job_count[2] = P;
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_census + dfire_bm_order: how long is each job, and in which order should the waves of dfire_bm_pairs draw them?  A job's length
// is set by the block bits of its entries (16 on average of the 64 a mask has room for, between 0 and 8 in a row), not
// by the number of entries: jobs of equal entry count differ eightfold.  Every wave takes (tile pair, part) pairs, counts
// per row the block bits of the part's entries and the distinct blocks, and writes an estimate in units of 1/64 batch:
//   items + 96 per block present (staging, the last batch's empty lanes) + 224 (job set-up)
// dfire_bm_order (one workgroup) then lists the jobs that have any work by class of estimated length, longest first
// (counting sort in LDS): the launch ends on jobs of a few batches, and rows without a block are never drawn.
// ---------------------------------------------------------------------------------------------
constexpr int kBmOrderWaves = 16;
__global__ __launch_bounds__(kBmOrderWaves * 64) void dfire_bm_census(const BmLaunch launch_arguments) {
    static_assert(kBmJobRows == 8, "a job row is one byte of the block mask");
    BmArgs *T = LD_BM_ARGS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_pairs = T->job_count[0], P = T->job_count[2];
    for (uint32_t jd = blockIdx.x * kBmOrderWaves + wave; jd < n_pairs; jd += gridDim.x * kBmOrderWaves) {
        const size_t tp = T->jobs[2 * jd];
        const uint32_t lo = T->jobs[2 * jd + 1];
        const uint32_t n = T->tp_count[tp];
        const uint32_t size = bm_part_size(n, P), hi = n < lo + size ? n : lo + size;
        constexpr int kChunks = kBmPartEntries / 64;
        unsigned long long m[kChunks];
#pragma unroll
        for (int k = 0; k < kChunks; k++) {
            const uint32_t e = lo + (uint32_t)k * 64 + lane;
            m[k] = e < hi ? T->ent_mask[tp * T->cap + e] : 0ull;
        }
        uint32_t items[8], present_lo = 0, present_hi = 0;   // per row; the OR of the masks
#pragma unroll
        for (int r = 0; r < 8; r++) items[r] = 0;
#pragma unroll
        for (int k = 0; k < kChunks; k++) {
            present_lo |= (uint32_t)m[k];
            present_hi |= (uint32_t)(m[k] >> 32);
#pragma unroll
            for (int r = 0; r < 8; r++) items[r] += (uint32_t)__popc((uint32_t)(m[k] >> (8 * r)) & 0xffu);
        }
        // sums over the wave: two rows to a word (each below 2^14)
        uint32_t pk[4] = {items[0] | items[1] << 16, items[2] | items[3] << 16, items[4] | items[5] << 16, items[6] | items[7] << 16};
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) pk[q] += (uint32_t)__shfl_xor((int)pk[q], off, 64);
            present_lo |= (uint32_t)__shfl_xor((int)present_lo, off, 64);
            present_hi |= (uint32_t)__shfl_xor((int)present_hi, off, 64);
        }
        if (lane < 8) {
            const uint32_t word = lane < 2 ? pk[0] : lane < 4 ? pk[1] : lane < 6 ? pk[2] : pk[3];
            const uint32_t it = (lane & 1) ? word >> 16 : word & 0xffffu;
            const uint32_t blocks = (uint32_t)__popc(((lane < 4 ? present_lo : present_hi) >> (8 * (lane & 3))) & 0xffu);
            T->job_cost[(size_t)jd * kBmJobRows + lane] = it ? it + 96u * blocks + 224u : 0u;
            // the job's record, whole: what dfire_bm_pairs needs to start on it in ONE load behind its place in the order
            // (tile pair, first entry, one past its last entry, ligand subtile)
            reinterpret_cast<uint4 *>(T->job_rec)[(size_t)jd * kBmJobRows + lane] = uint4{(uint32_t)tp, lo, hi, (uint32_t)lane};
        }
    }
}

__global__ __launch_bounds__(kBmOrderWaves * 64) void dfire_bm_order(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    __shared__ uint32_t s_class[kBmCostClasses];
    const int tid = threadIdx.x;
    const uint32_t n_pairs = T->job_count[0];
    for (int c = tid; c < kBmCostClasses; c += kBmOrderWaves * 64) s_class[c] = 0;
    __syncthreads();
    const uint32_t n_jobs = n_pairs * (uint32_t)kBmJobRows;
    auto class_of = [](uint32_t cost) { const uint32_t c = cost >> 7; return c < (uint32_t)kBmCostClasses ? c : (uint32_t)kBmCostClasses - 1u; };
    // (eight loads in flight per thread: one at a time, the two passes over ~20 000 jobs took 60 us)
    constexpr uint32_t kStep = kBmOrderWaves * 64;
    for (uint32_t base = 0; base < n_jobs; base += 8 * kStep) {
        uint32_t cost[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = base + (uint32_t)u * kStep + (uint32_t)tid;
            cost[u] = j < n_jobs ? T->job_cost[j] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (cost[u]) atomicAdd(&s_class[class_of(cost[u])], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t at = 0;
        for (int c = kBmCostClasses - 1; c >= 0; c--) {
            const uint32_t k = s_class[c];
            s_class[c] = at;
            at += k;
        }
        T->job_count[3] = at;
    }
    __syncthreads();
    for (uint32_t base = 0; base < n_jobs; base += 8 * kStep) {
        uint32_t cost[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = base + (uint32_t)u * kStep + (uint32_t)tid;
            cost[u] = j < n_jobs ? T->job_cost[j] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (cost[u]) T->job_order[atomicAdd(&s_class[class_of(cost[u])], 1u)] = base + (uint32_t)u * kStep + (uint32_t)tid;
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_pairs: persistent workgroups (two per CU) of 4 independent waves; a wave draws jobs
// (tile pair, part of its entries, ligand subtile a) and walks the job's 8 blocks (a, b), each with its 64 table
// rows staged in the wave's own cube in LDS.  The waves share the cell LUT and nothing else: no barrier after set-up.
// The batch code exists once per wave of the workgroup (switch on the wave number): the LDS address of a cube row is a
// constant of the instruction that reads it, so a pair's table address is `code - 8`, no base to add.
// ---------------------------------------------------------------------------------------------
// ANM: the form for molecules that flex per pose (src/dfire.rs:288-320): a job holds half the entries, and in their place the modes
// of the job's ligand subtile and of the current block's receptor subtile (kappa x, f32, in the order a batch reads them).
template <bool ANM>
struct BmWaveSharedT {
    static constexpr int kPart = ANM ? kBmAnmPartEntries : kBmPartEntries;
    unsigned char row_bits[kPart];        // per entry of the job: which of the 8 blocks (a, .) it holds
    unsigned short items[kPart + 64];     // the entries that hold the current block, and 64 fillers (entry 0 of the part) behind them
    // (An entry's row of the pass -- where its affine map is -- is NOT here: an item's row is read from the entry list in global
    // memory two batches ahead of its use.  Until the end of round 5 it lay in LDS, 2.25 bytes an entry, and a job held 1024
    // entries at most; a launch's time is A + B / (entries a job), and set-ups were a sixth of a wave's life.)
    alignas(16) float modes[ANM ? 2 * kBmModeFloats : 4];   // ANM: [ligand subtile][receptor subtile] x kBmModeFloats
};
template <bool ANM>
struct BmSharedT {
    unsigned char lut[kBmLutBytes];   // indexed from the far end: cell' = floor(kBmCellZero + 1/2 - 64 d2), everything further reads cell' 0
    unsigned char cube[kBmWaves][kBmCubeBytes];   // (in front of the per-wave lists: every cube row within the 16-bit offset field of a DS instruction)
    BmWaveSharedT<ANM> w[kBmWaves];
};
typedef BmSharedT<false> BmShared;
static_assert(sizeof(BmSharedT<false>) * kBmGroupsPerCu <= 160 * 1024 && sizeof(BmSharedT<true>) * kBmGroupsPerCu <= 160 * 1024, "two workgroups per CU");
static_assert(offsetof(BmShared, cube) + sizeof(unsigned char[kBmWaves][kBmCubeBytes]) < 65536, "cube rows are addressed by instruction offsets");
static_assert(offsetof(BmSharedT<true>, cube) == offsetof(BmShared, cube), "one batch code for both forms");

// what a wave needs to evaluate queued items
struct BmWaveCtx {
    size_t tp;       // tile pair
    int ls;          // ligand subtile (global)
    int RT;          // receptor tile
    size_t lo;       // first entry of the job
};

// ---- the exact path.  A flagged cell reads its row's MARKER, (64 + i * 8 + j) << kBmMarkerShift (51), instead of a table value.  After a
// block's 64 adds the bits from 2^51 up of a lane's two sums are 0 (no flagged pair), 64 + pair (one: fourteen lanes in a
// hundred) or at least 128 (several: four lanes in a thousand).  One flagged pair is named by the sum itself and goes
// straight into the wave's list of pairs, a 64-bit item = row of the pass | ligand atom << 32 | receptor atom << 48 that
// needs nothing else of the job it came from: bm_exact_pairs evaluates the list between two jobs, a few hundred pairs at a
// time, inlined (as a function of its own every call moved a hundred registers through scratch: a fifth of the kernel's
// time), where little of the job loop's state is live.  With several flagged pairs the (entry, block) goes into a second list, and bm_recheck, at the
// job's end, finds its flagged pairs again: lane = item, the block's 64 pairs once more by the batch's own f32 arithmetic
// -- the same operations in the same order on the same operands, hence the same cells bit for bit.
// bm_exact_pairs: lane = flagged pair: exact_pair (f64, the reference's operation order, dfire_device.hpp), its value added
// to the pose's fixed-point sum by an atomic, its interface flags set.  The lists live in global memory (in practice a few
// lines per wave that never leave the L2).
#ifndef LD_BM_DRAIN_AT
#define LD_BM_DRAIN_AT 256
#endif
#ifndef LD_BM_EXACT_U
#define LD_BM_EXACT_U 4
#endif
constexpr int kBmDrainAt = LD_BM_DRAIN_AT;   // flagged pairs a wave collects before it evaluates them
__device__ __forceinline__ unsigned long long bm_pair_item(uint32_t row, int la, int ra) {
    return (unsigned long long)row | (unsigned long long)la << 32 | (unsigned long long)ra << 48;
}

__device__ __forceinline__ void bm_exact_pairs(BmArgs *T, unsigned long long *queue, uint32_t n_pairs, int lane) {
#ifdef LD_BM_DIAG_NO_EXACT   // (diagnostic builds: timing only, wrong sums -- the kernel without its exact path)
    return;
#endif
    // the wave reads back what it pushed itself: its stores are complete (through the write-through L1, in the XCD's L2), and
    // the loads below go past the L1.  (An agent-scope release here writes the whole L2 back, on every drain of every wave:
    // the launch took twice as long.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // Four pairs per lane and trip, phase by phase: the four items, then their rows, then everything the pairs read -- a
    // pair is a chain of five dependent loads, and one at a time the chains were most of this function's time.
    constexpr int U = LD_BM_EXACT_U;
    const ExactCtx ex0 = bm_exact_ctx(T, 0);
    const size_t flag_words = (size_t)(T->m.rec_flag_words + T->m.lig.flag_words);
    for (uint32_t base = 0; base < n_pairs; base += 64 * U) {
        unsigned long long item[U];
        bool act[U];
        int la[U], ra[U];
        size_t row[U], pose[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = base + (uint32_t)(u * 64 + lane);
            act[u] = k < n_pairs;
            item[u] = act[u] ? __hip_atomic_load(queue + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;   // (past the L1)
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            la[u] = (int)((item[u] >> 32) & 0xffffu);
            ra[u] = (int)(item[u] >> 48);
            act[u] = act[u] && la[u] < T->m.lig.n_real && ra[u] < T->m.rec_n_real;   // (padding atoms of a block: nothing to add)
            row[u] = (size_t)(item[u] & 0xffffffffull);
        }
        // a pair reads eight 16-byte pieces: its row of rt_exact (the pose), its two atoms' rows (scattered loads cost the
        // memory pipeline a cache line per lane whatever their width: the 24 loads of 8 and 4 bytes this replaced were the
        // exact path's time)
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d pq[U][4], lq[U][2], rq[U][2];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const v2d *prow = reinterpret_cast<const v2d *>(T->rt_exact + (act[u] ? row[u] : 0) * 8);
            const v2d *lrow = reinterpret_cast<const v2d *>(T->m.lig_exact + (size_t)(act[u] ? la[u] : 0) * 4);
            const v2d *rrow = reinterpret_cast<const v2d *>(T->m.rec_exact + (size_t)(act[u] ? ra[u] : 0) * 4);
#pragma unroll
            for (int c = 0; c < 4; c++) pq[u][c] = prow[c];
            lq[u][0] = lrow[0]; lq[u][1] = lrow[1];
            rq[u][0] = rrow[0]; rq[u][1] = rrow[1];
        }
        double pr[U][7], lc[U][3], rc[U][3];
        uint32_t lterm[U], rterm[U];
        int32_t rslot[U], lslot[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            pr[u][0] = pq[u][0].x; pr[u][1] = pq[u][0].y; pr[u][2] = pq[u][1].x; pr[u][3] = pq[u][1].y;
            pr[u][4] = pq[u][2].x; pr[u][5] = pq[u][2].y; pr[u][6] = pq[u][3].x;
            pose[u] = (size_t)__double_as_longlong(pq[u][3].y);
            lc[u][0] = lq[u][0].x; lc[u][1] = lq[u][0].y; lc[u][2] = lq[u][1].x;
            rc[u][0] = rq[u][0].x; rc[u][1] = rq[u][0].y; rc[u][2] = rq[u][1].x;
            const unsigned long long lw = (unsigned long long)__double_as_longlong(lq[u][1].y), rw = (unsigned long long)__double_as_longlong(rq[u][1].y);
            lterm[u] = (uint32_t)lw; lslot[u] = (int32_t)(lw >> 32);
            rterm[u] = (uint32_t)rw; rslot[u] = (int32_t)(rw >> 32);
        }
        // the reference's arithmetic for the four pairs first, then their four table reads TOGETHER, then the sums: with the read
        // inside each pair's branch a trip paid four dependent round trips to the L2 one after the other
        bool inside[U];
        uint32_t slot_at[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            inside[u] = false;
            slot_at[u] = 0u;
            if (!act[u]) continue;
            if (T->exact_pairs) atomicAdd(T->exact_pairs + row[u], 1u);
            // the ligand atom as the reference poses it (src/dfire.rs:282-302: pose_ligand_atom's operations), then exact_pair's
            const Quat q{pr[u][3], pr[u][4], pr[u][5], pr[u][6]};
            const Quat r = qmul(qmul(q, Quat{0.0, lc[u][0], lc[u][1], lc[u][2]}), qinverse(q));
            double px = r.x + pr[u][0], py = r.y + pr[u][1], pz = r.z + pr[u][2];
            double rx = rc[u][0], ry = rc[u][1], rz = rc[u][2];
            if (T->amp != nullptr) {   // molecules that flex: src/dfire.rs:288-320, the operations of pose_ligand_atom and exact_pair (dfire_device.hpp)
                // (an atom's modes and a row's amplitudes in 16-byte pieces of contiguous memory: mode by mode out of the [mode][xyz][atom]
                // arrays a pair was 60 scattered 8-byte loads, and the ANM form's waves spent a fifth of their time here)
                const v2d *amps = reinterpret_cast<const v2d *>(T->amp_exact + row[u] * (2 * kBmMaxModes));
                const v2d *lm = reinterpret_cast<const v2d *>(T->m.lig_modes_exact + (size_t)la[u] * (3 * kBmMaxModes));
                const v2d *rm = reinterpret_cast<const v2d *>(T->m.rec_modes_exact + (size_t)ra[u] * (3 * kBmMaxModes));
#pragma unroll
                for (int k2 = 0; k2 < kBmMaxModes / 2; k2++) {   // modes 2 k2 and 2 k2 + 1: x y z x y z
                    if (2 * k2 < T->m.anm_lig) {
                        const v2d c = amps[kBmMaxModes / 2 + k2], m0 = lm[3 * k2], m1 = lm[3 * k2 + 1], m2 = lm[3 * k2 + 2];
                        px += m0.x * c.x; py += m0.y * c.x; pz += m1.x * c.x;
                        if (2 * k2 + 1 < T->m.anm_lig) { px += m1.y * c.y; py += m2.x * c.y; pz += m2.y * c.y; }
                    }
                    if (2 * k2 < T->m.anm_rec) {
                        const v2d c = amps[k2], m0 = rm[3 * k2], m1 = rm[3 * k2 + 1], m2 = rm[3 * k2 + 2];
                        rx += m0.x * c.x; ry += m0.y * c.x; rz += m1.x * c.x;
                        if (2 * k2 + 1 < T->m.anm_rec) { rx += m1.y * c.y; ry += m2.x * c.y; rz += m2.y * c.y; }
                    }
                }
            }
            const double dx = 2.0 * rx - 2.0 * px, dy = 2.0 * ry - 2.0 * py, dz = 2.0 * rz - 2.0 * pz;
            const double D = dx * dx + dy * dy + dz * dz;   // = 4 d2 bit for bit (src/dfire.rs:331-333)
            if (!(D <= kCutScaled)) continue;   // d2 <= 225 (src/dfire.rs:334)
            uint32_t bin = 0;   // src/dfire.rs:336-337 as a count of the steps passed
            for (int b = 1; b <= 20; b++) bin += D >= ex0.step4[b] ? 1u : 0u;
            if (D <= ex0.iface_scaled) {   // d <= 3.9 (src/dfire.rs:339-342)
                uint32_t *flags = T->flags + pose[u] * flag_words;
                if (rslot[u] >= 0) atomicOr(&flags[rslot[u] >> 5], 1u << (rslot[u] & 31));
                if (lslot[u] >= 0) atomicOr(&flags[T->m.rec_flag_words + (lslot[u] >> 5)], 1u << (lslot[u] & 31));
            }
            inside[u] = true;
            slot_at[u] = (lterm[u] + rterm[u] + tiled_bin_term(bin)) / 8u;
        }
        double value[U];
#pragma unroll
        for (int u = 0; u < U; u++) value[u] = ex0.table[slot_at[u]];   // (slot 0 for the pairs that read nothing: a valid address)
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!inside[u]) continue;
            // a counting launch sums ones: the pair counts if it is within the cutoff
            const long long fix = T->count_mode ? 1ll : __double2ll_rn(value[u] * T->m.fix_scale);
#ifndef LD_BM_DIAG_NO_EXACT_ATOMIC   // (diagnostic builds: timing only, wrong sums)
            if (fix != 0) atomicAdd(reinterpret_cast<unsigned long long *>(T->exact_fix + row[u]), (unsigned long long)fix);
#else
            asm volatile("" :: "v"(fix));
#endif
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the list is read before it is written again
}

__device__ __forceinline__ uint32_t bm_cell(float Rs, float Rz, float Ry, float Rx, float l2, float lz, float ly, float lx) {
    // one half of the batch's packed chain: v_pk_add (Rs - l2), three v_pk_fma; v_cvt_u32_f32 (negative, NaN -> 0)
    float D = Rs - l2;
    D = __builtin_fmaf(Rz, lz, D);
    D = __builtin_fmaf(Ry, ly, D);
    D = __builtin_fmaf(Rx, lx, D);
    uint32_t c;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(c) : "v"(D));
    return c;
}

// (entry, block) items with several flagged pairs -> the list of pairs.  An item = entry of the pass | ligand subtile of the tile
// << 32 | receptor subtile of the tile << 35: like a pair item it needs nothing of the job it came from, so the wave collects
// 64 of them before it spends a call and 64 pair loops on them.
__device__ __forceinline__ unsigned long long bm_block_item(size_t entry, int a, int b) {
    return (unsigned long long)entry | (unsigned long long)a << 32 | (unsigned long long)b << 35;
}
__device__ __forceinline__ uint32_t bm_recheck(BmArgs *T, const unsigned char *lut, const unsigned long long *blocks, uint32_t n_blocks,
                                            unsigned long long *queue, uint32_t n_pairs, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    constexpr float seed = (float)kBmCellZero + 0.5f;
    const int n_rt = T->m.rec_n_tiles;
    for (uint32_t base = 0; base < n_blocks; base += 64) {
    if (n_pairs > (uint32_t)kBmQueuePairs - 4096u - (uint32_t)(8 * kBmPartEntries)) {   // room for 64 x 64 more, and still for a job's pushes
        bm_exact_pairs(T, queue, n_pairs, lane);
        n_pairs = 0;
    }
    const bool act = base + (uint32_t)lane < n_blocks;
    const unsigned long long item = act ? __hip_atomic_load(blocks + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const size_t entry = (size_t)(item & 0xffffffffull);
    const int a = (int)((item >> 32) & 7u), b = (int)((item >> 35) & 7u);
    const size_t tp = entry / T->cap;
    const int lt = (int)(tp / (unsigned)n_rt), RT = (int)(tp % (unsigned)n_rt), ls = lt * 8 + a;
    // the lane's pose and block, as the batch saw them
    const uint32_t row = act ? T->ent_row[entry] : 0u;   // (an idle lane's entry 0 may never have been written)
    const float4 *ap = reinterpret_cast<const float4 *>(T->rt) + (size_t)row * 3;
    const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
    const TiledBox box = T->m.rec_sub[(size_t)RT * 8 + b];
    const float cbx = 0.5f * (box.lox + box.hix), cby = 0.5f * (box.loy + box.hiy), cbz = 0.5f * (box.loz + box.hiz);
    const Affine A{a0.x, a0.y, a0.z, a0.w - cbx, a1.x, a1.y, a1.z, a1.w - cby, a2.x, a2.y, a2.z, a2.w - cbz};
    // ANM: the pose's amplitudes, and the deformation of an atom as the batch forms it (LD_BM_FLEX_ASM)
    const bool anm = T->amp != nullptr;
    float amp_rec[kBmMaxModes], amp_lig[kBmMaxModes];
    bool wild = false;
#pragma unroll
    for (int k = 0; k < kBmMaxModes; k++) amp_rec[k] = amp_lig[k] = 0.f;
    if (anm) {
        const float *am = T->amp + (size_t)row * kBmAmpFloats;
#pragma unroll
        for (int k = 0; k < kBmMaxModes; k++) {
            amp_rec[k] = am[k];
            amp_lig[k] = am[kBmMaxModes + k];
        }
        wild = am[20] != 0.f;
    }
    auto flexed = [&](const float *modes, const float *amps, int atom, int c) {   // modes: a subtile's kBmModeFloats (BmModel)
        const float *m = modes + ((atom >> 1) * 3 + c) * kBmMaxModes * 2 + (atom & 1);
        float d = amps[0] * m[0];   // (LD_BM_FLEX_ASM: a product, then nine fused multiply-adds)
#pragma unroll
        for (int k = 1; k < kBmMaxModes; k++) d = __builtin_fmaf(amps[k], m[2 * k], d);
        return d;
    };
    float lx[8], ly[8], lz[8], l2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float4 L = reinterpret_cast<const float4 *>(T->m.lig_local)[ls * 8 + i];
        bm_apply(A, L.x, L.y, L.z, lx[i], ly[i], lz[i]);
        if (anm) {
            const float *lm = T->m.lig_modes_f32 + (size_t)ls * kBmModeFloats;
            lx[i] += flexed(lm, amp_lig, i, 0);
            ly[i] += flexed(lm, amp_lig, i, 1);
            lz[i] += flexed(lm, amp_lig, i, 2);
        }
        l2[i] = __builtin_fmaf(lx[i], lx[i], __builtin_fmaf(ly[i], ly[i], lz[i] * lz[i]));
    }
    const float *rec = reinterpret_cast<const float *>(T->m.rec_pairs + (size_t)RT * 32 + b * 4);   // 4 records: x0 x1 y0 y1 z0 z1 . .
    // (in a block with tracked atoms the slots of bins 0 and 1 held markers too)
    const uint32_t near_code = T->m.lig_sub_tracked[ls] != 0 || T->m.rec_sub_tracked[RT * 8 + b] != 0 ? bm_code_of_bin(1) : 0xffffffffu;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float x = rec[q * 8 + h] - cbx, y = rec[q * 8 + 2 + h] - cby, z = rec[q * 8 + 4 + h] - cbz;
            if (anm) {   // (the batch: fma(1/2, 2 (r - c), d) = (r - c) + d, rounded once)
                const float *rm = T->m.rec_modes_f32 + ((size_t)RT * 8 + (size_t)b) * kBmModeFloats;
                x += flexed(rm, amp_rec, 2 * q + h, 0);
                y += flexed(rm, amp_rec, 2 * q + h, 1);
                z += flexed(rm, amp_rec, 2 * q + h, 2);
            }
            const float Rs = __builtin_fmaf(-x, x, __builtin_fmaf(-y, y, __builtin_fmaf(-z, z, seed)));
            const float Rx = x * 2.f, Ry = y * 2.f, Rz = z * 2.f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t cell = bm_cell(Rs, Rz, Ry, Rx, l2[i], lz[i], ly[i], lx[i]);
                const uint32_t code = lut[cell];
                const bool hit = act && (wild || code == kBmFlagged || (near_code != 0xffffffffu && code <= near_code));   // (a wild pose: every pair, the exact path decides)
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (hit) {
                    const uint32_t at = n_pairs + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    queue[at] = bm_pair_item(row, ls * 8 + i, RT * 64 + b * 8 + 2 * q + h);
                }
                n_pairs += (uint32_t)__popcll(m);
            }
        }
    }
    }
    return n_pairs;
}

// DEBUG: per-wave phase timers (LIGHTDOCK_BM_DEBUG) -- the production instantiation carries none.
template <bool DEBUG, bool ANM>
__global__ __launch_bounds__(kBmWaves * 64, kBmGroupsPerCu) void dfire_bm_pairs(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    __shared__ __attribute__((aligned(16))) BmSharedT<ANM> S;   // the kernel's only LDS object: at LDS address 0
    // The batch code (dfire_bm_batch.inc) reads the LUT at `cell` and a cube row at `code + a constant of the instruction`: both
    // assume S at LDS address 0 with the LUT first.  Another __shared__ object or a different placement would make it read wrong
    // codes silently: trap instead (the compiler folds the test away when the address is the 0 it assigns today).  The block also
    // clobbers v220..v255, i.e. it needs the 256 registers of two waves per SIMD.
    static_assert(offsetof(BmShared, lut) == 0, "the LUT's cell is its LDS address");
    static_assert(kBmWaves * kBmGroupsPerCu == 8, "dfire_bm_batch.inc clobbers v220..v255: 256 VGPRs a wave = two waves per SIMD");
    if ((uint32_t)(uintptr_t)&S != 0u) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_rt = T->m.rec_n_tiles;
    if (T->job_count[3] == 0) return;   // no job (a GSO step in which nothing moved): not 512 workgroups' copies of the LUT either -- that launch took 55 us
    {
        const uint8_t *lut = T->count_mode ? T->m.lut_full : T->m.lut;
        for (int i = tid; i < kBmLutBytes / 16; i += kBmWaves * 64) reinterpret_cast<uint4 *>(S.lut)[i] = reinterpret_cast<const uint4 *>(lut)[i];
    }
    BmWaveSharedT<ANM> &WS = S.w[wave];
    __syncthreads();
    const uint32_t n_jobs = T->job_count[3];
    unsigned long long *queue = T->queue + ((size_t)blockIdx.x * kBmWaves + wave) * kBmQueueCap;   // the wave's flagged pairs
    unsigned long long *queue_blocks = queue + kBmQueuePairs;                                        // (entry, block) items with several
    uint32_t queued = 0, queued_blocks = 0;   // wave-uniform: flagged pairs listed; (entry, block) items listed
    const unsigned char *table_rows = reinterpret_cast<const unsigned char *>(T->count_mode ? T->m.rows_ones : T->m.rows);
    const unsigned long long dbg_t0 = DEBUG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long dbg_jobs = 0, dbg_batches = 0, dbg_t_batch = 0, dbg_t_drain = 0, dbg_t_scan = 0, dbg_t_block = 0;
    auto now = [] { return DEBUG ? __builtin_amdgcn_s_memrealtime() : 0ull; };

    // A wave's FIRST job is its own number in the launch -- no draw: the 2048 waves of a launch all start at once, and their 2048
    // returning atomics on one counter (device scope: resolved behind the XCDs' L2s) queued for ~20 us before the first job's
    // set-up could begin, in every launch; a GSO step with few glowworms moving is one job per wave and little else.
    const uint32_t n_waves = gridDim.x * (uint32_t)kBmWaves;
    bool first_job = true;
    for (;;) {
        const unsigned long long dbg_tj = now();
        uint32_t job = blockIdx.x * (uint32_t)kBmWaves + (uint32_t)wave;
        if (!first_job) {
            if (lane == 0) job = atomicAdd(T->job_next, 1u);   // (drawing one job ahead was measured: the 2048 claimed jobs lengthen the tail)
            job = (uint32_t)__builtin_amdgcn_readfirstlane((int)job) + n_waves;
        }
        first_job = false;
        if (job >= n_jobs) break;
        job = T->job_order[job];   // longest first (dfire_bm_order)
        // (the job's record as dfire_bm_census wrote it: one load; its pieces one by one -- the (tile pair, part) pair, then the tile
        // pair's entry count -- were two more dependent round trips per job)
        const uint4 rec = reinterpret_cast<const uint4 *>(T->job_rec)[job];
        const size_t tp = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)rec.x);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.y), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.z);
        const int a = __builtin_amdgcn_readfirstlane((int)rec.w);   // ligand subtile a of the tile = partial-sum row of the entry
        const int n_chunks = (int)((hi - lo + 63) / 64);
        const int lt = (int)(tp / (unsigned)n_rt), RT = (int)(tp % (unsigned)n_rt);
        const int ls = lt * 8 + a;

        // What the job's blocks need that does not depend on the entry, for all 8 receptor subtiles of the tile at once (one
        // latency, together with the masks): lane = (receptor subtile b, atom j).
        const uint32_t roff_all = T->m.rec_rowoff[(size_t)RT * 64 + lane];           // the atom's column in a table row block
        uint32_t any_bits = 0;
        {   // the job's block masks: all loads in flight at once.  UNCONDITIONAL, from a per-lane pointer formed once: guarded by
            // `e < hi` each load was a basic block of its own -- exec saved, two scalar loads of the launch arguments and a wait
            // for them, seven scalar instructions of address arithmetic, the load -- 4 of a job set-up's 8 us.  What the lanes
            // beyond the part's end read (the next part's entries, another tile pair's, never-written memory: the workspace has a
            // part's room behind its end) is masked out of `bits` and otherwise unused.
            constexpr int kChunks = BmWaveSharedT<ANM>::kPart / 64;   // (28 chunks of 64 entries; the ANM form 16)
            const unsigned long long *mask_at = T->ent_mask + (tp * T->cap + lo + (size_t)lane);
            unsigned long long m[kChunks];
#pragma unroll
            for (int k = 0; k < kChunks; k++) m[k] = mask_at[k * 64];
            const uint32_t n_mine = hi - lo;   // entries of the part
#pragma unroll
            for (int k = 0; k < kChunks; k++) {
                const uint32_t bits = (uint32_t)(k * 64 + lane) < n_mine ? (uint32_t)(m[k] >> (8 * a)) & 0xffu : 0u;
                if (k < n_chunks) WS.row_bits[k * 64 + lane] = (unsigned char)bits;
                any_bits |= bits;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) any_bits |= (uint32_t)__shfl_xor((int)any_bits, off, 64);
        any_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)any_bits);
        if (any_bits == 0) continue;
        if (DEBUG) dbg_jobs++;

        // the ligand subtile's local coordinates (uniform)
        // (kept in LDS, read back per batch as broadcasts: 24 wave-uniform values in vector registers for the whole job are what
        // pushed the block set-up into scratch)
        // (in SCALAR registers, two atoms a pair -- the operands of the packed posing instructions.  Kept in LDS and read back per batch
        // they cost every batch an LDS round trip at its head; 24 wave-uniform values in vector registers pushed the block set-up into scratch.)
        v2f LocX[4], LocY[4], LocZ[4];
        {
            const float mine = T->m.lig_local[(size_t)(ls * 8 + (lane & 7)) * 4 + (lane < 24 ? lane >> 3 : 0)];   // lanes 0..7 x, 8..15 y, 16..23 z
            auto from = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), l)); };
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) {
                LocX[p2] = v2f{from(2 * p2), from(2 * p2 + 1)};
                LocY[p2] = v2f{from(8 + 2 * p2), from(9 + 2 * p2)};
                LocZ[p2] = v2f{from(16 + 2 * p2), from(17 + 2 * p2)};
            }
        }
        if (ANM) {   // the job's ligand subtile's modes -> LDS (240 floats: 60 lanes x 16 bytes)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous job's last reads of them are done
            if (lane < kBmModeFloats / 4) reinterpret_cast<float4 *>(WS.modes)[lane] = reinterpret_cast<const float4 *>(T->m.lig_modes_f32 + (size_t)ls * kBmModeFloats)[lane];
        }
        // table rows of a block -> LDS by LDS-DMA: an instruction copies 5 rows, lane = (row of the five, one of its 11 pieces of
        // 16 bytes) -- the lane's two numbers are the same for every instruction, 55 lanes take part (piece p of an instruction
        // lands at its LDS address + 16 p: five rows of 176 bytes, contiguous).  The lane keeps the row block of ligand atom
        // lane % 8 (where its type's rows start in the table).
        constexpr int kRowPieces = kBmRowBytes / 16, kDmaRows = 64 / kRowPieces, kDma = (kBmCubeRows + kDmaRows - 1) / kDmaRows;
        static_assert(kRowPieces == 11 && kDmaRows == 5 && kDma == 13, "five rows per copy");
        const bool lig_tracked = T->m.lig_sub_tracked[ls] != 0;
        const uint32_t my_tracked = T->m.rec_sub_tracked[RT * 8 + (lane & 7)];
        const uint32_t lig_rowbase = T->m.lig_rowbase[ls * 8 + (lane & 7)];
        // the job's partial sums, one per entry of the part: the wave's own 8 KB of global memory -- written and read again within
        // microseconds, by this wave only, they never leave the L2 (as a sparse [tile pair][row][entry] array they were 33 M
        // scattered read-modify-writes of HBM per launch)
        long long *const my_partial = T->ent_partial + ((size_t)blockIdx.x * kBmWaves + wave) * kBmPartEntries;
        const size_t row_base_entry = tp * T->cap + lo;
        const uint32_t *const job_rows = T->ent_row + row_base_entry;   // (an item's row of the pass)
        const uint32_t dma_rowsel = (uint32_t)(lane / kRowPieces) * 4u, dma_piece = (uint32_t)(lane % kRowPieces) * 16u;   // (constants of the lane: row of the five, piece)
        if (DEBUG) dbg_t_scan += now() - dbg_tj;   // job set-up
        // a batch's results, on their way out one batch late (flush_pending)
        // (the sums' address, opaque to the compiler: read through the kernel arguments at its use it was two scalar loads and a wait
        // for them inside every batch's write-out)
        typedef __attribute__((address_space(1))) unsigned long long global_u64;
        // the pass's (ligand tile, row) sums: [ligand tile][row of the pass] -- the lanes of a batch are entries of ONE tile pair, in
        // the order the culling kernel listed them: runs of up to 8 consecutive rows (its items are 8 poses), so with the rows
        // contiguous a run's atomics are ONE 64-byte request at the memory side, where the atomics execute, instead of one each
        // (as [row][ligand tile], until round 6, every lane's atomic was a request of its own: MI355X_MICROARCH.md "Global float
        // atomics": 64 lanes in 64 different rows run at a seventeenth of the rate of contiguous ones)
        unsigned long long job_tile_sum_bits = (unsigned long long)(uintptr_t)(T->tile_sum + (size_t)lt * T->cap);
        asm volatile("" : "+s"(job_tile_sum_bits));
        global_u64 *const job_tile_sum = (global_u64 *)job_tile_sum_bits;   // (a GLOBAL pointer: as a generic one the atomic became a flat instruction)
        long long pending_val = 0;
        uint32_t pending_item = 0xffffffffu, pending_row = 0;
        uint32_t pending_push = 0xffffffffu;       // where in the wave's lists (pairs, then (entry, block) items) the lane's flagged item goes, or none
        unsigned long long pending_push_item = 0ull;
        auto flush_pending = [&]() {
            if (pending_push != 0xffffffffu) queue[pending_push] = pending_push_item;
            pending_push = 0xffffffffu;
            // The (entry, ligand subtile)'s sum is complete with the entry's last block of the job: it goes to the pose's
            // (ligand tile, row) sum by an integer atomic -- order-free, and no gather over 24 M scattered partial sums afterwards.
            // (All of a job's atomics in one burst at the job's end, out of the scratch, was measured in round 6: 6.0 against 6.35 M
            // evaluations/s -- tools/experiments/r06_atomics_at_job_end.patch.)
            if (pending_item != 0xffffffffu) {
                // (LD_BM_DIAG_*: diagnostic builds -- timing only, wrong sums: what the kernel takes without one of its memory streams)
#ifndef LD_BM_DIAG_NO_ATOMIC
                if (pending_item & 0x4000u) __hip_atomic_fetch_add(job_tile_sum + pending_row, (unsigned long long)pending_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                if (pending_item & 0x4000u) asm volatile("" :: "v"(pending_val), "v"(pending_row));
#endif
#ifndef LD_BM_DIAG_NO_PARTIAL
                else my_partial[pending_item & (uint32_t)kBmEntryMask] = pending_val;
#endif
            }
            pending_item = 0xffffffffu;
        };
        for (int b = 0; b < 8; b++) {
            if (!((any_bits >> b) & 1u)) continue;
            const unsigned long long dbg_tblk = now();
            {   // stage the block's rows; they land while the entries are scanned
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous block's reads are done
                // lane r: where row r = (i, j) = (r / 8, r % 8) of the block starts in the table
                const uint32_t row_src = (uint32_t)__shfl((int)lig_rowbase, lane >> 3, 64) + (uint32_t)__shfl((int)roff_all, b * 8 + (lane & 7), 64);
                // thirteen copies of five rows: every lane fetches its rows' sources with thirteen ds_bpermute in flight, one wait,
                // then the copies (LD_BM_DMA_ASM, dfire_bm_batch.inc)
                static_assert(kDma == 13 && kDmaRows * kBmRowBytes == 880, "LD_BM_DMA_ASM is written for 13 copies of 880 bytes");
                uint32_t dma_tmp[kDma];
                unsigned long long dma_exec;
                uint32_t dma_m0;
                const uint32_t cube_lds = (uint32_t)(uintptr_t)S.cube[wave];
#ifndef LD_BM_DIAG_NO_DMA   // (diagnostic builds: timing only, wrong sums)
                LD_BM_DMA_ASM(dma_exec, dma_m0, dma_tmp, dma_rowsel, row_src, dma_piece, table_rows, cube_lds, 0x007fffffffffffffull, 0x00000fffffffffffull);
#endif
            }
            if (ANM) {   // the block's receptor subtile's modes -> LDS, behind the job's ligand modes
                if (lane < kBmModeFloats / 4)
                    reinterpret_cast<float4 *>(WS.modes + kBmModeFloats)[lane] = reinterpret_cast<const float4 *>(T->m.rec_modes_f32 + ((size_t)RT * 8 + (size_t)b) * kBmModeFloats)[lane];
            }
            const bool tracked = lig_tracked || __builtin_amdgcn_readlane((int)my_tracked, b) != 0;
            constexpr float seed = (float)kBmCellZero + 0.5f;
            // ---- the job's entries that hold block (a, b), in entry order (all 16 chunks' bytes in flight, then the ballots)
            uint32_t n_items = 0;
            {
                constexpr int kChunks = BmWaveSharedT<ANM>::kPart / 64;
                uint32_t bits16[kChunks];
#pragma unroll
                for (int k = 0; k < kChunks; k++) bits16[k] = k < n_chunks ? (uint32_t)WS.row_bits[k * 64 + lane] : 0u;
#pragma unroll
                for (int k = 0; k < kChunks; k++) {
                    const uint32_t bits = bits16[k];
                    const bool act = (bits >> b) & 1u;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(act);
                    if (act) {   // (the entry's number only: whether this is its first or last block of the job is worked out per batch, read_row)
                        const uint32_t at = n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        WS.items[at] = (unsigned short)(k * 64 + lane);
                    }
                    n_items += (uint32_t)__popcll(m);
                }
                // 64 fillers behind the list's end: entry 0 of the part (a real entry, so everything a lane loads for it is valid
                // memory; what the lane computes is dropped, `valid`).  A batch's lanes beyond the end read these instead of picking
                // "in range ? at : first" with a compare and a v_cndmask_b32 -- 23 cycles of the vector port on gfx950, five times
                // a plain instruction (tools/microbench/valu_rate.hip, profiles/r06_valu_issue_rates.txt) -- per look-up.
                WS.items[n_items + (uint32_t)lane] = 0;
            }
            // what a lane of a batch needs from memory, loaded one batch ahead: the pose's affine map out of the [row][12] table
            // (L2: the pass's table is 48 bytes a pose) and the entry's partial sum so far
            struct BatchLoads {
                float4 a0, a1, a2;   // the entry's affine map
                long long prev;      // the entry's partial of this row so far
                uint32_t item, row;
                float4 amp[ANM ? kBmAmpFloats / 4 : 1];   // ANM: the pose's amplitudes, receptor's then ligand's; [20] != 0: a wild pose
            };
            // The way from a batch's place in the item list to its loads goes through LDS twice -- the item, then the entry's row of
            // the pass --: looked up when the loads were due, that was two LDS round trips at the head of every batch, with the
            // other wave of the SIMD keeping the LDS queue full.  They run AHEAD now: while batch k is walked, the loads of batch
            // k + 1 are in flight (issued from a row that is already in a register), the row of batch k + 2 and the item of
            // batch k + 3 are being read -- behind the batch's own LDS traffic, whose last wait covers them.
            auto first_of = [&](uint32_t first_item) { return first_item < n_items ? first_item : 0u; };   // (beyond the block's end: its first items again)
            auto read_item = [&](uint32_t first_item) { return (uint32_t)WS.items[first_item + (uint32_t)lane]; };   // (beyond the end: the fillers)
            // the entry's row of the pass, and in bits 30 / 31 of a second word whether block b is the entry's LAST / FIRST of this job
            // (last: the (entry, row)'s sum is complete; first: nothing to add to yet) -- from the entry's byte of block bits
            struct RowOfItem {
                uint32_t row;     // the entry's row of the pass (ANM form: as loaded from the entry list -- nothing may be computed from it before the loop's wait)
                uint32_t flags;   // bit 31: block b is the entry's FIRST of this job, bit 30: its LAST
            };
            auto read_row = [&](uint32_t item) {
                const uint32_t el = item & (uint32_t)kBmEntryMask;
                RowOfItem r;
                const uint32_t bits = (uint32_t)WS.row_bits[el];
                r.row = job_rows[el];
                // (x - 1) has bit 31 set exactly when x = 0, for x below 2^31
                r.flags = (((bits & ((1u << b) - 1u)) - 1u) & 0x80000000u) | ((((bits >> (b + 1)) - 1u) >> 1) & 0x40000000u);
                return r;
            };
            auto issue_loads = [&](uint32_t item_el, const RowOfItem &of) {
                BatchLoads L;
                const uint32_t item = (item_el & (uint32_t)kBmEntryMask) | (of.flags >> 16 & 0xc000u);   // entry | first << 15 | last << 14, as the code below reads it
                const uint32_t row = of.row & 0x3ffffu;
                L.item = item;
                L.row = row;
#ifndef LD_BM_DIAG_ROW_OF_LANE
                const float4 *ap = reinterpret_cast<const float4 *>(T->rt) + (size_t)row * 3;
#else
                const float4 *ap = reinterpret_cast<const float4 *>(T->rt) + (size_t)(row & 63u) * 3;   // (every load from the same 3 KB)
#endif
                L.a0 = ap[0];
                L.a1 = ap[1];
                L.a2 = ap[2];
                if (ANM) {
                    const float4 *am = reinterpret_cast<const float4 *>(T->amp) + (size_t)row * (kBmAmpFloats / 4);
#pragma unroll
                    for (int k = 0; k < kBmAmpFloats / 4; k++) L.amp[k] = am[k];
                }
                L.prev = 0;
#ifndef LD_BM_DIAG_NO_PARTIAL
                if (!(item & 0x8000u)) L.prev = my_partial[item & (uint32_t)kBmEntryMask];
#endif
                return L;
            };
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the item list as every lane wrote it
            uint32_t look_item, look2_item;   // batch 1's item, batch 2's item
            RowOfItem look_row;               // batch 1's row
            BatchLoads next;
            {
                const uint32_t item0 = read_item(0u);
                look_item = read_item(first_of(64u));
                look2_item = read_item(first_of(128u));
                next = issue_loads(item0, read_row(item0));
                look_row = read_row(look_item);
            }
            // (behind the first batch's loads: their latency covers it)
            // receptor subtile b of the tile: 4 pair records, wave-uniform, out of the registers loaded at the job's start
            // The block's distance arithmetic has its origin at the centre c of the receptor subtile's box:
            //   E = seed - |l - r|^2 = (seed - |r - c|^2) - |l - c|^2 + 2 (r - c) . (l - c),     cell' = (u32)E
            // four packed operations per step instead of six (the differences need not be formed), all operands small
            // enough (below 2^17 for every pair within reach of the cutoff) that the roundings stay inside eps.  The LUT
            // is indexed from the far end: a pair beyond its last cell has E < 0, which v_cvt_u32_f32 turns into cell' 0
            // ("miss") like a NaN -- no clamp.  (bm_drain repeats this arithmetic: keep the two in step.)
            // The sixteen operands (and the centre) come from the receptor's table of them (BmModel::rec_ops, formed on the host) by
            // SCALAR loads, straight into the scalar register pairs the batch's packed instructions take: no vector instruction.
            // (Until round 5 a block's set-up formed them itself: 30 v_readlane of the tile's records, 45 packed operations, 32
            // v_readfirstlane -- 110 of a set-up's ~500 vector instructions; before that, in vector registers, the compiler
            // re-derived all sixteen in every batch.)
            typedef const __attribute__((address_space(4))) float const_f32;
            const_f32 *ops = (const_f32 *)(uintptr_t)(T->m.rec_ops + ((size_t)RT * 8 + (size_t)b) * kBmOpsFloats);
            v2f Rx[4], Ry[4], Rz[4], Rs[4];   // seed - |r - c|^2, and 2 (r - c)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                Rs[q] = v2f{ops[2 * q], ops[2 * q + 1]};
                Rz[q] = v2f{ops[8 + 2 * q], ops[9 + 2 * q]};
                Ry[q] = v2f{ops[16 + 2 * q], ops[17 + 2 * q]};
                Rx[q] = v2f{ops[24 + 2 * q], ops[25 + 2 * q]};
            }
            const float cbx = ops[32], cby = ops[33], cbz = ops[34];
            if (DEBUG) dbg_t_block += now() - dbg_tblk;   // block set-up

            // ---- one batch: lane = entry.  Three pieces: the posing (code all waves share), the 64 pairs (once per wave of the
            // workgroup: WAVE is a constant of that code, see the switch below), the markers (shared again).
            // (Until round 6 the whole batch sat inside the switch and the next batch's loads were issued in front of it, into a second
            // set of registers: 13 register moves a batch carried `next` into `cur`.  Posing in front of the loads frees the map's
            // twelve registers before the loads are issued -- the loop-carried values are the same registers --, and only what the
            // batch needs at its END (the partial so far, the item, the row) is copied.)
            struct Posed {
                v2f LX[4], LY[4], LZ[4], L2[4];       // atoms (2p, 2p + 1): l - c and |l - c|^2
                v2f fRs[4], fRz[4], fRy[4], fRx[4];   // ANM: the receptor subtile's operands of THIS lane's pose
                bool wild;
            };
            auto pose_batch = [&](const BatchLoads &cur, Posed &P) {
                // the lane's 8 ligand atoms posed two at a time (packed; the operations and their nesting are bm_apply's, so the
                // culling kernel's boxes and the exact path see the same bits), relative to the block's centre
                const v2f A0xy{cur.a0.x, cur.a0.y}, A0zw{cur.a0.z, cur.a0.w - cbx}, A1xy{cur.a1.x, cur.a1.y}, A1zw{cur.a1.z, cur.a1.w - cby};
                const v2f A2xy{cur.a2.x, cur.a2.y}, A2zw{cur.a2.z, cur.a2.w - cbz};
                P.wild = false;
                if constexpr (!ANM) {
#ifndef LD_BM_DIAG_NO_POSE
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        LD_BM_POSE_ASM(P.LX[p], P.LY[p], P.LZ[p], P.L2[p], A0xy, A0zw, A1xy, A1zw, A2xy, A2zw, LocX[p], LocY[p], LocZ[p]);
                    }
#else   // (diagnostic builds: timing only, wrong sums -- what ANY scheme that reuses an entry's posed atoms across a job's blocks could save at most)
#pragma unroll
                    for (int p = 0; p < 4; p++) { P.LX[p] = A0xy; P.LY[p] = A1xy; P.LZ[p] = A2xy; P.L2[p] = A0zw; }
#endif
                } else {
                    // Both subtiles flex with the lane's pose (src/dfire.rs:288-320): atom += sum_k amplitude_k x mode_k, in the receptor's
                    // frame.  The modes of the two subtiles lie in LDS in the order they are read here -- per pair of atoms and
                    // coordinate ten values a pair, two modes per 16-byte broadcast read --, the amplitudes came with the map; a
                    // packed multiply-add per mode and pair of atoms, the amplitude's half picked by the operand select.
                    const v2f *amp2 = reinterpret_cast<const v2f *>(cur.amp);   // [0..4] the receptor's amplitudes two by two, [5..9] the ligand's
                    P.wild = reinterpret_cast<const float *>(cur.amp)[20] != 0.f;
                    const uint32_t modes_lds = (uint32_t)(uintptr_t)WS.modes;
                    {
                        v2f D[12];   // the ligand subtile's deformation: atoms (2p, 2p + 1), coordinate c at [3 p + c] (LD_BM_FLEX_ASM, dfire_bm_batch.inc)
                        LD_BM_FLEX_ASM(D, (amp2 + kBmMaxModes / 2), modes_lds);
#pragma unroll
                        for (int p = 0; p < 4; p++) {
                            LD_BM_POSE_FLEX_ASM(P.LX[p], P.LY[p], P.LZ[p], P.L2[p], A0xy, A0zw, A1xy, A1zw, A2xy, A2zw, LocX[p], LocY[p], LocZ[p], D[3 * p], D[3 * p + 1], D[3 * p + 2]);
                        }
                    }
                    // the receptor subtile: (r - c) = Rx / 2 exactly, + the lane's deformation, then the operands as the rigid form's table
                    // holds them: seed - |r - c|^2 and 2 (r - c)
                    {
                        v2f D[12];
                        LD_BM_FLEX_ASM(D, amp2, modes_lds + (uint32_t)(kBmModeFloats * sizeof(float)));
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const v2f half{0.5f, 0.5f}, two{2.f, 2.f};
                            const v2f x = __builtin_elementwise_fma(half, Rx[q], D[3 * q]), y = __builtin_elementwise_fma(half, Ry[q], D[3 * q + 1]);
                            const v2f z = __builtin_elementwise_fma(half, Rz[q], D[3 * q + 2]);
                            P.fRs[q] = __builtin_elementwise_fma(-x, x, __builtin_elementwise_fma(-y, y, __builtin_elementwise_fma(-z, z, v2f{seed, seed})));
                            P.fRx[q] = x * two;
                            P.fRy[q] = y * two;
                            P.fRz[q] = z * two;
                        }
                    }
                }
            };
            // The batch's 64 pairs: dfire_bm_batch.inc (generated, tools/gen_bm_batch_asm.py).  Fixed-point sum: table values are
            // integers (2^-k units, exact adds in any order); a flagged cell's slot holds the row's marker.
            auto walk_batch = [&](auto wave_constant, const Posed &P, unsigned long long &acc0, unsigned long long &acc1) {
                constexpr int WAVE = decltype(wave_constant)::value;
                constexpr uint32_t kCube = (uint32_t)(offsetof(BmShared, cube) + (size_t)WAVE * kBmCubeBytes);   // a constant LDS address
                (void)Rs; (void)Rz; (void)Ry; (void)Rx;   // (named here so that the generic lambda captures them: their only other use is inside an asm operand list)
#if defined(LD_BM_DIAG_ANM_COST)
                // Timing experiment (wrong sums): what a block-major batch would cost for molecules that FLEX per pose
                // (src/dfire.rs:288-320: 10 + 10 normal modes) -- per lane the ligand subtile's deformation (8 atoms x 3 coordinates x 10
                // modes = 240 multiply-adds = 120 packed), the receptor subtile's (120 packed) and its sixteen operands per lane (36
                // packed: the batch then takes them from vector registers).
                {
                    v2f t0{P.LX[0].x, P.LX[1].y}, t1{P.LY[0].x, P.LY[1].y}, t2{P.LZ[0].x, P.LZ[1].y}, t3{P.L2[0].x, P.L2[1].y};
#ifndef LD_BM_DIAG_ANM_LDS
                    asm volatile(".rept 69\n\tv_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t.endr"
                                 : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(P.LX[0]), "v"(P.LY[0]));
#else
                    // ... and the 480 mode components of the two subtiles are wave-uniform: delivered from LDS, 120 broadcast reads of 16
                    // bytes (one per four multiply-adds' worth of operands), four in flight
                    {
                        const uint32_t lds_at = (uint32_t)(uintptr_t)&WS.items[0];   // (any 1 KB of the wave's LDS: the values do not matter here)
                        asm volatile(".rept 30\n\t"
                                     "ds_read_b128 v[220:223], %6\n\tds_read_b128 v[224:227], %6 offset:16\n\tds_read_b128 v[228:231], %6 offset:32\n\tds_read_b128 v[232:235], %6 offset:48\n\t"
                                     "s_waitcnt lgkmcnt(0)\n\t"
                                     "v_pk_fma_f32 %0, %4, v[220:221], %0\n\tv_pk_fma_f32 %1, %4, v[222:223], %1\n\tv_pk_fma_f32 %2, %5, v[224:225], %2\n\tv_pk_fma_f32 %3, %5, v[226:227], %3\n\t"
                                     "v_pk_fma_f32 %0, %4, v[228:229], %0\n\tv_pk_fma_f32 %1, %4, v[230:231], %1\n\tv_pk_fma_f32 %2, %5, v[232:233], %2\n\tv_pk_fma_f32 %3, %5, v[234:235], %3\n\t"
                                     ".endr\n\t"
                                     ".rept 9\n\tv_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t.endr"
                                     : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(P.LX[0]), "v"(P.LY[0]), "v"(lds_at)
                                     : "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "memory");
                    }
#endif
                    v2f vRs[4], vRz[4], vRy[4], vRx[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        vRs[q] = Rs[q] + t0 * v2f{0.f, 0.f}; vRz[q] = Rz[q] + t1 * v2f{0.f, 0.f}; vRy[q] = Ry[q] + t2 * v2f{0.f, 0.f}; vRx[q] = Rx[q] + t3 * v2f{0.f, 0.f};
                    }
                    LD_BM_BATCH_ASM_V(acc0, acc1, vRs, vRz, vRy, vRx, P.L2, P.LZ, P.LY, P.LX, kCube);
                }
#elif !defined(LD_BM_DIAG_NO_PAIRS)
                if constexpr (ANM) {
                    LD_BM_BATCH_ASM_V(acc0, acc1, P.fRs, P.fRz, P.fRy, P.fRx, P.L2, P.LZ, P.LY, P.LX, kCube);
                } else {
                    LD_BM_BATCH_ASM(acc0, acc1, Rs, Rz, Ry, Rx, P.L2, P.LZ, P.LY, P.LX, kCube);
                }
#else
                asm volatile("" : "+v"(acc0), "+v"(acc1) : "v"(P.L2[0]), "v"(P.L2[1]), "v"(P.L2[2]), "v"(P.L2[3]), "v"(P.LX[0]), "v"(P.LY[0]), "v"(P.LZ[0]), "s"(Rs[0]), "s"(Rx[3]));
#endif
            };
            // what a batch keeps of its loads until its END: the entry's partial so far, its item and row of the pass
            auto finish_batch = [&](long long cur_prev, uint32_t cur_item, uint32_t cur_row, bool wild, unsigned long long acc0, unsigned long long acc1, uint32_t done) {
                const int count = n_items - done >= 64u ? 64 : (int)(n_items - done);
                // the lanes beyond `count` hold a filler (entry 0 of the part): nothing of theirs leaves.  All-ones for them, by
                // arithmetic (a compare and a v_cndmask_b32 per use cost the vector port 27 cycles, this 9)
                // (through an asm statement: written as C++ the compiler turns the shift back into the compare and the select)
                uint32_t invalid;
                asm("v_ashrrev_i32 %0, 31, %1" : "=v"(invalid) : "v"(count - 1 - lane));
                const uint32_t el = cur_item & (uint32_t)kBmEntryMask;
                // each sum = marker bits + the true sum, |true sum| < 2^50 (32 pairs; the scale is chosen for that)
                // (the marker bits live in the sums' upper words -- bit 51 is bit 19 there, and rounding by 2^50 never carries out of
                // the lower word: 32-bit arithmetic does what 64-bit shifts and subtractions did in twice the instructions)
                static_assert(kBmMarkerShift > 33, "the markers sit in the upper word");
                constexpr int kHiShift = kBmMarkerShift - 32;
                const int mark0 = ((int)(uint32_t)(acc0 >> 32) + (1 << (kHiShift - 1))) >> kHiShift, mark1 = ((int)(uint32_t)(acc1 >> 32) + (1 << (kHiShift - 1))) >> kHiShift;
                // the true sum = both sums less their marker bits: a subtraction in the upper word only (wrap-around there is the 64-bit
                // subtraction's), folded into the entry's partial so far -- one 64-bit add, one 32-bit subtract
                const unsigned long long marker_bits = (unsigned long long)((uint32_t)(mark0 + mark1) << kHiShift) << 32;
                // A sum's marker bits are 0 (no flagged pair among its 32), 64 + pair (one) or at least 128 (several): so the two sums'
                // marker bits ADDED are 0, 64 + pair, or at least 128 -- one range test each instead of tests of both sums and their
                // combinations.  (ANM: a wild pose -- amplitudes beyond what the f32 arithmetic's error bound covers -- sends its whole
                // block to the exact path through the list of (entry, block) items, and nothing of what it summed here counts.)
                uint32_t marks = (uint32_t)(mark0 + mark1);
                if constexpr (ANM) marks = wild ? 128u : marks;
                marks &= ~invalid;
                const bool one = marks - 64u < 64u, several = marks >= 128u;
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(one), m2 = __builtin_amdgcn_ballot_w64(several);
                // the lane's one pair in a flagged cell (0.1 % of all pairs; some lane of nearly every batch has one): the exact path.
                // Like the lane's sum, the item leaves at the start of the NEXT batch (flush_pending): stored here, just before the
                // loop's wait for the next batch's loads, that wait was for this store's acknowledgement -- in every batch.
                pending_push = 0xffffffffu;
                if (one) {
                    const int pair = (int)marks - 64;
                    pending_push = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                    pending_push_item = bm_pair_item(cur_row, ls * 8 + (pair >> 3), RT * 64 + b * 8 + (pair & 7));
                }
                queued += (uint32_t)__popcll(m1);
                if (__builtin_expect(m2 != 0ull, 0)) {
                    if (several) {   // (a lane has one flagged pair or several, never both: one slot serves both lists -- queue_blocks = queue + kBmQueuePairs)
                        pending_push = (uint32_t)kBmQueuePairs + queued_blocks + __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
                        pending_push_item = bm_block_item(row_base_entry + el, a, b);
                    }
                    queued_blocks += (uint32_t)__popcll(m2);
                }
                // The lane's sum goes out at the start of the NEXT batch (flush_pending): memory operations complete in order, and
                // the wait for the next batch's loads at the loop's top would wait for a store or atomic issued here, just before it,
                // as well -- a round trip to the L2 per batch.  Issued in front of the following loads, it has a whole batch to complete.
                if constexpr (ANM) pending_val = wild ? cur_prev : (long long)((unsigned long long)cur_prev + acc0 + acc1 - marker_bits);
                else pending_val = (long long)((unsigned long long)cur_prev + acc0 + acc1 - marker_bits);
                pending_item = cur_item | invalid;
                pending_row = cur_row;
            };
            for (uint32_t done = 0; done < n_items; done += 64) {
                if (DEBUG) dbg_batches++;
                const unsigned long long dbg_tb = now();
                // this batch's loads are in (the compiler's own count: behind them only the previous batch's push into the wave's
                // list may still be in flight); the first time also the block's rows, which the compiler does not know it waits for
                if (done == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (the empty statement READS what the loads return: the compiler's wait for them goes here, in front of the writes
                // and loads issued below, counted exactly.  Without it the wait sat at the values' first use, behind the next
                // batch's loads, as `vmcnt(3)` -- the count of the shortest path -- which on the usual path, four loads and
                // a write or two, waited for the first of the loads just issued: a round trip to the L2 in every batch.)
                asm volatile("" :: "v"(next.a0.x), "v"(next.a0.y), "v"(next.a0.z), "v"(next.a0.w), "v"(next.a1.x), "v"(next.a1.y), "v"(next.a1.z), "v"(next.a1.w),
                             "v"(next.a2.x), "v"(next.a2.y), "v"(next.a2.z), "v"(next.a2.w), "v"(next.prev));
                // (also the row of the batch after this one, a load from the entry list: its wait belongs here too, not behind the stores below)
                asm volatile("" :: "v"(look_row.row));
                if constexpr (ANM) asm volatile("" :: "v"(next.amp[0].x), "v"(next.amp[1].x), "v"(next.amp[2].x), "v"(next.amp[3].x), "v"(next.amp[4].x), "v"(next.amp[5].x));
#ifdef LD_BM_DIAG_WAIT   // (diagnostic builds: the drain timer holds the time a wave waits at the head of its batches for their loads)
                if (DEBUG) dbg_t_drain += now() - dbg_tb;
#endif
                flush_pending();   // (in front of the next batch's loads)
                // the rows' markers, over what the copy left in their slots (they name the PAIR: not part of the table); in a block
                // with an atom that has an interface-flag slot also in place of bins 0 and 1: those pairs go to the exact path, which
                // sets the flags (src/dfire.rs:339-342) and adds the value
                if (done == 0) {
                    long long *row = reinterpret_cast<long long *>(S.cube[wave] + lane * kBmRowBytes);
                    const long long marker = (long long)(64 + lane) << kBmMarkerShift;
                    row[kBmFlagged / 8] = marker;
                    if (tracked) row[0] = row[1] = marker;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // the posing consumes the map: its registers are free for the next batch's loads
                Posed P;
                pose_batch(next, P);
                const long long cur_prev = next.prev;
                const uint32_t cur_item = next.item, cur_row = next.row;
                // (in the code all waves share, and unconditional -- the block's last batch asks for its first items again -- so that the
                // loads land in the registers the next trip reads them from: behind a branch the compiler moved them there
                // right away, i.e. waited for them)
                next = issue_loads(look_item, look_row);
                const RowOfItem in_row = read_row(look2_item);                  // batch k + 2's row
                const uint32_t in_item = read_item(first_of(done + 192u));      // batch k + 3's item
                unsigned long long acc0 = 0ull, acc1 = 0ull;   // over the pairs with receptor atoms 0 2 4 6 / 1 3 5 7 of the subtile
                switch (wave) {
                    case 0: walk_batch(std::integral_constant<int, 0>{}, P, acc0, acc1); break;
                    case 1: walk_batch(std::integral_constant<int, 1>{}, P, acc0, acc1); break;
                    case 2: walk_batch(std::integral_constant<int, 2>{}, P, acc0, acc1); break;
                    default: walk_batch(std::integral_constant<int, 3>{}, P, acc0, acc1); break;
                }
                finish_batch(cur_prev, cur_item, cur_row, P.wild, acc0, acc1, done);
                look_item = look2_item;
                look_row = in_row;
                look2_item = in_item;
                if (DEBUG) dbg_t_batch += now() - dbg_tb;
#ifdef LD_BM_DIAG_FIRST   // (diagnostic builds: the drain timer holds the time of every block's FIRST batch -- the wait for the rows and the first loads)
                if (DEBUG && done == 0) dbg_t_drain += now() - dbg_tb;
#endif
            }
            flush_pending();   // (the next block's first loads read what this block's last batch wrote)
        }
        // The exact path, at the job's end only: no call inside the block and batch loops (the compiler keeps what lives across
        // a call site in scratch for the whole job), and the lists have room for everything one job can push.
        if (queued_blocks >= 64u) {
            const unsigned long long td = now();
            queued = bm_recheck(T, S.lut, queue_blocks, queued_blocks, queue, queued, lane);
            queued_blocks = 0;
            if (DEBUG) dbg_t_drain += now() - td;
        }
        // (a few hundred pairs at a time: often enough that the other waves' batches hide its memory latencies -- everything at
        // the wave's end was a 230 us tail of the whole launch -- and seldom enough that the calls do not count.  A kernel of
        // its own for the lists, after this one, took 120 us for what costs the waves 45 us each here: measured.)
        if (queued >= (uint32_t)kBmDrainAt) {
            const unsigned long long td = now();
            bm_exact_pairs(T, queue, queued, lane);
            queued = 0;
            if (DEBUG) dbg_t_drain += now() - td;
        }
    }
    {
        const unsigned long long td = now();
        if (queued_blocks) queued = bm_recheck(T, S.lut, queue_blocks, queued_blocks, queue, queued, lane);
        if (queued) bm_exact_pairs(T, queue, queued, lane);
        if (DEBUG) dbg_t_drain += now() - td;
    }
#ifndef LD_BM_DIAG_CULL_TIMES
    if (DEBUG && T->debug != nullptr && lane == 0) {
        unsigned long long *d = T->debug + ((size_t)blockIdx.x * kBmWaves + wave) * 8;
        d[0] = dbg_t0;
        d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = dbg_jobs;
        d[3] = dbg_batches;
        d[4] = dbg_t_batch;
        d[5] = dbg_t_drain;
        d[6] = dbg_t_block;
        d[7] = dbg_t_scan;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_gather: eight lanes = row of the pass: the pose's (ligand tile, row) sums -- integers, filled by the pair kernel's
// atomics, each below 2^61 -- added as f64 in a fixed order (a pose's total can pass 63 bits for an extreme table), plus the
// exact path's sum.  A counting launch (count_mode) sums ones.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dfire_bm_gather(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    const int n_lt = T->m.lig.n_tiles;
    const size_t n_rows = bm_rows(T);
    // eight lanes per row, each every eighth ligand tile (a small launch -- one swarm -- is all latency: few dependent loads a lane);
    // the sums lie [ligand tile][row]: a load's 64 lanes read 8 tiles x 8 consecutive rows, eight 64-byte pieces.  The sums are
    // integers, their f64 images added over the eight lanes in a fixed tree.
    const int sub = (int)threadIdx.x & 7;
    const size_t rows_per_trip = (size_t)gridDim.x * 32;
    for (size_t first = (size_t)blockIdx.x * 32; first < n_rows; first += rows_per_trip) {   // (whole waves stay together for the shuffles)
        const size_t row = first + threadIdx.x / 8;
        const bool valid = row < n_rows;
        const long long pp = valid ? bm_pose_of(T, row) : -1;
        double units = 0.0;
        uint32_t tested = 0;
        if (pp >= 0)
            for (int lt = sub; lt < n_lt; lt += 8) {
                units += (double)T->tile_sum[(size_t)lt * T->cap + row];   // [ligand tile][row of the pass]
                if (T->count_mode && T->tile_tested) tested += T->tile_tested[row * (size_t)n_lt + lt];
            }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            units += __shfl_xor(units, off, 64);
            tested += (uint32_t)__shfl_xor((int)tested, off, 64);
        }
        if (pp < 0 || sub != 0) continue;
        const size_t pose = (size_t)pp;
        units += (double)T->exact_fix[row];
        if (T->count_mode) {   // (pair counts stay far below 2^53: exact)
            T->count_partial[pose] = (uint32_t)units;
            if (T->tested_partial) T->tested_partial[pose] = tested;
            if (T->exact_partial) T->exact_partial[pose] = T->exact_pairs[row];
        } else {
            T->partial[2 * pose] = units * (1.0 / T->m.fix_scale);
            T->partial[2 * pose + 1] = 0.0;
        }
    }
}

}  // namespace

size_t bm_pairs_lds_bytes() { return sizeof(BmShared); }

hipError_t launch_bm_pose(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(dfire_bm_pose, dim3((unsigned)std::min<size_t>((t.n_poses + 255) / 256, 2048)), dim3(256), 0, stream, t);
    return hipGetLastError();
}

hipError_t launch_bm_cull(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    if (t.m.rec_n_tiles > 1024 || t.n_poses > kBmMaxPassPoses) return hipErrorInvalidValue;
    const size_t lds = (size_t)t.m.rec_n_tiles * 9 * sizeof(TiledBox) + (size_t)kBmCullWaves * bm_cull_wave_lds(t.m.rec_n_tiles);
    // persistent workgroups (each fills its LDS with the receptor's boxes once): as many as fit the chip at this LDS size
    const size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 512)));
    const size_t cus = t.pairs_groups > 0 ? (size_t)t.pairs_groups : 256;
    const size_t blocks = std::min<size_t>(((t.n_poses + kBmCullPoses - 1) / kBmCullPoses * (size_t)t.m.lig.n_tiles + kBmCullWaves - 1) / kBmCullWaves, cus * per_cu);
    const bool anm = t.amp != nullptr, count = t.tile_tested != nullptr;
    const void *kernel = anm ? (count ? reinterpret_cast<const void *>(&dfire_bm_cull<true, true>) : reinterpret_cast<const void *>(&dfire_bm_cull<false, true>))
                             : (count ? reinterpret_cast<const void *>(&dfire_bm_cull<true, false>) : reinterpret_cast<const void *>(&dfire_bm_cull<false, false>));
    if (lds > 64 * 1024) {   // (a receptor of more than ~100 tiles)
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (anm) {
        const size_t items = (t.n_poses + 63) / 64 * (size_t)t.m.rec_n_tiles;
        hipLaunchKernelGGL(dfire_bm_rec_boxes, dim3((unsigned)std::min<size_t>((items + 3) / 4, cus * 8)), dim3(256), 0, stream, t);
        if (count) hipLaunchKernelGGL((dfire_bm_cull<true, true>), dim3((unsigned)blocks), dim3(kBmCullWaves * 64), lds, stream, t);
        else hipLaunchKernelGGL((dfire_bm_cull<false, true>), dim3((unsigned)blocks), dim3(kBmCullWaves * 64), lds, stream, t);
    } else {
        if (count) hipLaunchKernelGGL((dfire_bm_cull<true, false>), dim3((unsigned)blocks), dim3(kBmCullWaves * 64), lds, stream, t);
        else hipLaunchKernelGGL((dfire_bm_cull<false, false>), dim3((unsigned)blocks), dim3(kBmCullWaves * 64), lds, stream, t);
    }
    return hipGetLastError();
}

static unsigned bm_pairs_groups(const BmLaunch &t) {   // persistent: what the chip holds
    unsigned groups = (t.pairs_groups > 0 ? (unsigned)t.pairs_groups : 256u) * kBmGroupsPerCu;
#ifdef LD_DIAG_BUILD   // (diagnostic builds only, tools/build_variant.sh)
    if (const char *e = std::getenv("LIGHTDOCK_BM_HALF_OCCUPANCY")) {   // one workgroup per CU (one wave per SIMD)
        if (std::atoi(e) == 1) groups /= kBmGroupsPerCu;
    }
#endif
    return groups;
}

hipError_t launch_bm_pairs(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(dfire_bm_plan, dim3(1), dim3(1024), 0, stream, t);
    hipLaunchKernelGGL(dfire_bm_census, dim3(128), dim3(kBmOrderWaves * 64), 0, stream, t);
    hipLaunchKernelGGL(dfire_bm_order, dim3(1), dim3(kBmOrderWaves * 64), 0, stream, t);
    const unsigned groups = bm_pairs_groups(t);
    if (t.amp != nullptr) {   // molecules that flex per pose
        if (t.debug != nullptr) hipLaunchKernelGGL((dfire_bm_pairs<true, true>), dim3(groups), dim3(kBmWaves * 64), 0, stream, t);
        else hipLaunchKernelGGL((dfire_bm_pairs<false, true>), dim3(groups), dim3(kBmWaves * 64), 0, stream, t);
    } else {
        if (t.debug != nullptr) hipLaunchKernelGGL((dfire_bm_pairs<true, false>), dim3(groups), dim3(kBmWaves * 64), 0, stream, t);
        else hipLaunchKernelGGL((dfire_bm_pairs<false, false>), dim3(groups), dim3(kBmWaves * 64), 0, stream, t);
    }
    return hipGetLastError();
}

hipError_t launch_bm_gather(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(dfire_bm_gather, dim3((unsigned)std::min<size_t>((t.n_poses + 31) / 32, 8192)), dim3(256), 0, stream, t);   // 8 lanes per row
    return hipGetLastError();
}

}  // namespace ld
