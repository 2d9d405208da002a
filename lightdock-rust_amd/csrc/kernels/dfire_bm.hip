// dfire_bm.hip -- K1 for DFIRE, block-major (gfx950 / MI355X).  Interface, data flow and numerics: dfire_bm.hpp.
//
// The pair work of a launch is ordered by 8 x 8 atom-pair BLOCK, not by pose: a block's 64 table rows
// T[type_i][type_j][bin] sit in LDS while every pose in which the block is within the cutoff walks its 64 pairs, one pose
// per lane, the atom pair wave-uniform.  The potential never travels through the vector L1 as a gather.
// Compiled with -ffp-contract=off; every f64 operation is the reference's (src/dfire.rs:325-345); the f32 filter uses
// explicit fmas.  No MFMA (lookup/reduction).
#include "dfire_bm.hpp"

#include <cmath>

#include "dfire_device.hpp"

namespace ld {

namespace {

// The launch arguments stay where the dispatch put them, in the kernarg segment (constant address space): every field
// is a scalar load at its point of use, nothing is copied to registers up front or to scratch when a non-inlined function
// wants the whole block.
typedef const __attribute__((address_space(4))) BmLaunch BmArgs;
#define LD_BM_ARGS ((BmArgs *)__builtin_amdgcn_kernarg_segment_ptr())

constexpr float kBmBoxCut = 14400.0f * 1.00005f;  // (8 * 15 A)^2 in record units, padded for the rounding of the box test

__device__ __forceinline__ uint32_t bm_cvt_u32(float f) {  // v_cvt_u32_f32 saturates: negative and NaN -> 0
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// listed row -> pose row, or -1 beyond the list of this launch / inactive
__device__ __forceinline__ long long bm_pose_of(BmArgs *T, size_t listed) {
    if (T->pose_count != nullptr && T->first + listed >= (size_t)*T->pose_count) return -1;
    const size_t pose = T->pose_list ? (size_t)T->pose_list[T->first + listed] : T->first + listed;
    if (T->active != nullptr && T->active[pose] == 0) return -1;
    return (long long)pose;
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
__device__ __forceinline__ Affine bm_load_affine(const float *rt, size_t pose) {
    const float4 a = reinterpret_cast<const float4 *>(rt + pose * 12)[0];
    const float4 b = reinterpret_cast<const float4 *>(rt + pose * 12)[1];
    const float4 c = reinterpret_cast<const float4 *>(rt + pose * 12)[2];
    return Affine{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
}

__device__ __forceinline__ TiledLigand bm_ligand(BmArgs *T) {
    TiledLigand l;
    l.n_real = T->m.lig.n_real;
    l.n_tiles = T->m.lig.n_tiles;
    l.x = T->m.lig.x;
    l.y = T->m.lig.y;
    l.z = T->m.lig.z;
    return l;
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
    ex.pose_flags = T->flags + pose * (size_t)(T->m.rec_flag_words + T->m.lig.flag_words);
    ex.rec_flag_words = T->m.rec_flag_words;
    return ex;
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_pose: pose row -> f32 affine map into the record frame.  v' = q v q^-1 + t (src/qt.rs:48-61) is the
// rotation matrix of q / |q|; computed in f64, rounded once.  Its error is part of eps (dfire_bm_error_bound).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dfire_bm_pose(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    const size_t listed = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (listed >= T->n_poses) return;
    const long long p = bm_pose_of(T, listed);
    if (p < 0) return;
    const size_t pose = (size_t)p;
    const double *row = T->poses + pose * T->stride;
    const double tx = row[0], ty = row[1], tz = row[2], w = row[3], x = row[4], y = row[5], z = row[6];
    const double n2 = w * w + x * x + y * y + z * z;
    const double k = kBmKappa / n2;
    float *o = T->rt + pose * 12;
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
    if (T->exact_fix) T->exact_fix[pose] = 0;
    if (T->exact_count) T->exact_count[pose] = 0;
    if (T->exact_pairs) T->exact_pairs[pose] = 0;
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_cull: wave = (pose, ligand tile)
// ---------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(64) void dfire_bm_cull(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    __shared__ unsigned long long s_mask[256];
    __shared__ uint32_t s_rt[256];
    const int lane = threadIdx.x;
    const int n_lt = T->m.lig.n_tiles, n_rt = T->m.rec_n_tiles;
    const size_t listed = blockIdx.x / (unsigned)n_lt;
    const int lt = (int)(blockIdx.x % (unsigned)n_lt);
    const long long pp = bm_pose_of(T, listed);
    if (pp < 0) return;
    const size_t pose = (size_t)pp;
    const size_t slot = pose * (size_t)n_lt + lt;

    const int la = lt * 64 + lane;
    const float4 loc = reinterpret_cast<const float4 *>(T->m.lig_local)[la];
    const bool valid = loc.w != 0.f;
    const Affine A = bm_load_affine(T->rt, pose);
    float fx, fy, fz;
    bm_apply(A, loc.x, loc.y, loc.z, fx, fy, fz);
    const bool inside = fabsf(fx) <= T->m.ubound && fabsf(fy) <= T->m.ubound && fabsf(fz) <= T->m.ubound;

    // An atom outside the frame is more than the cutoff away from every receptor atom (the frame holds the receptor's
    // box + 16 A): it joins no box; the pairs it still meets inside blocks of its subtile read "miss", as they must.
    BoxRegs sub = lane_box(valid && inside, fx, fy, fz);
    box_reduce8(sub);
    BoxRegs whole = sub;
    box_reduce64_from8(whole);
    {   // widen: the f32 positions are within box_pad of the exactly posed ones (and a relative term for huge frames)
        const float pad = T->m.box_pad;
        auto widen = [pad](BoxRegs &b) {
            box_widen(b);
            b.lox -= pad; b.loy -= pad; b.loz -= pad;
            b.hix += pad; b.hiy += pad; b.hiz += pad;
        };
        widen(sub);
        widen(whole);
    }
    whole.lox = lane63_f32(whole.lox); whole.loy = lane63_f32(whole.loy); whole.loz = lane63_f32(whole.loz);
    whole.hix = lane63_f32(whole.hix); whole.hiy = lane63_f32(whole.hiy); whole.hiz = lane63_f32(whole.hiz);

    // 64 x 64 tile boxes, 64 receptor tiles per ballot; then the 8 x 8 subtile boxes of every surviving tile
    const int bj = lane & 7;
    int n_vis = 0;
    uint32_t tested = 0;
    for (int base = 0; base < n_rt; base += 64) {
        bool tile_near = false;
        if (base + lane < n_rt) tile_near = box_gap2(whole, T->m.rec_tile[base + lane]) <= kBmBoxCut;
        unsigned long long rtmask = __ballot(tile_near);
        while (rtmask) {
            const int RT = base + __ffsll(rtmask) - 1;
            rtmask &= rtmask - 1;
            const TiledBox nb = T->m.rec_sub[(size_t)RT * 8 + bj];
            const unsigned long long smask = __ballot(box_gap2(sub, nb) <= kBmBoxCut);  // bit = ligand subtile (lane >> 3) * 8 + receptor subtile
            if (smask) {
                if (lane == 0) {
                    s_mask[n_vis] = smask;
                    s_rt[n_vis] = (uint32_t)RT;
                }
                n_vis++;
                if (COUNT) tested += (uint32_t)__popcll(smask);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // one entry per surviving tile pair, the appends of a wave in one atomic instruction
    for (int v0 = 0; v0 < n_vis; v0 += 64) {
        const int v = v0 + lane;
        if (v < n_vis) {
            const uint32_t RT = s_rt[v];
            const size_t tp = (size_t)lt * n_rt + RT;
            const uint32_t idx = atomicAdd(&T->tp_count[tp], 1u);
            T->ent_pose[tp * T->cap + idx] = (uint32_t)pose;
            T->ent_mask[tp * T->cap + idx] = s_mask[v];
            T->vis_entry[slot * (size_t)n_rt + v] = RT << 24 | idx;
        }
    }
    if (lane == 0) {
        T->vis_count[slot] = (uint32_t)n_vis;
        if (COUNT) T->tile_tested[slot] = tested;
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_pairs: workgroup = (tile pair, ligand subtile a, part of the pair's entries); 8 waves
// ---------------------------------------------------------------------------------------------
struct BmShared {
    unsigned char cube[2][kBmCubeBytes];
    unsigned char lut[kBmLutBytes];
    unsigned char row_bits[kBmWaves][kBmPartEntries / kBmWaves];  // per entry of a wave's range: which of the 8 blocks (a, .) it holds
    uint32_t pend[kBmWaves][128];                                   // ring: entries waiting for the next batch
    uint32_t queue[kBmWaves][kBmQueue];                             // pairs for the exact path
};

// what a wave needs to evaluate queued pairs exactly
struct BmWaveCtx {
    size_t tp;       // tile pair
    int ls;          // ligand subtile (global)
    int RT;          // receptor tile
    size_t my_lo;    // first entry of this wave's range
};

// Queue item: entry (local to the wave's range) | (i * 8 + j) << 9 | b << 15
template <bool COUNT>
__device__ __noinline__ void bm_drain(BmArgs *T, const BmWaveCtx &W, const uint32_t *queue, uint32_t queued, int lane) {
    for (uint32_t k = (uint32_t)lane; k < queued; k += 64) {
        const uint32_t item = queue[k];
        const size_t e = W.my_lo + (item & 511u);
        const int i = (int)((item >> 12) & 7u), j = (int)((item >> 9) & 7u), b = (int)(item >> 15);
        const int la = W.ls * 8 + i, ra = W.RT * 64 + b * 8 + j;
        if (la >= T->m.lig.n_real || ra >= T->m.rec_n_real) continue;
        const size_t pose = T->ent_pose[W.tp * T->cap + e];
        const ExactCtx ex = bm_exact_ctx(T, pose);
        const Vec3 p = pose_ligand_atom(bm_ligand(T), 0, 0, T->poses + pose * T->stride, la);
        uint32_t cnt = 0;
        const double v = exact_pair(ex, p, T->m.lig.tindex[la], la, ra, cnt);
        // order-free: 2^-40 fixed point (the one place where a table value is rounded: below the noise of any f64 sum order)
        const long long fix = __double2ll_rn(v * kBmFixScale);
        if (fix != 0) atomicAdd(reinterpret_cast<unsigned long long *>(T->exact_fix + pose), (unsigned long long)fix);
        if (COUNT) {
            if (cnt) atomicAdd(T->exact_count + pose, cnt);
            atomicAdd(T->exact_pairs + pose, 1u);
        }
    }
}

// A whole batch in f64 (its queue overflowed: the poses of an absurd batch, e.g. molecules on top of each other):
// item k's 64 pairs across the lanes, summed into lane k's accumulator.
template <bool COUNT>
__device__ __noinline__ void bm_exact_batch(BmArgs *T, const BmWaveCtx &W, int b, uint32_t el, int count, int lane, double &acc, uint32_t &cnt) {
    const int i = lane >> 3, j = lane & 7;
    const int la = W.ls * 8 + i, ra = W.RT * 64 + b * 8 + j;
    const bool real = la < T->m.lig.n_real && ra < T->m.rec_n_real;
    acc = 0.0;
    cnt = 0;
    for (int k = 0; k < count; k++) {
        const size_t e = W.my_lo + (size_t)__builtin_amdgcn_readlane((int)el, k);
        const size_t pose = T->ent_pose[W.tp * T->cap + e];
        double v = 0.0;
        uint32_t c = 0;
        if (real) {
            const ExactCtx ex = bm_exact_ctx(T, pose);
            const Vec3 p = pose_ligand_atom(bm_ligand(T), 0, 0, T->poses + pose * T->stride, la);
            v = exact_pair(ex, p, T->m.lig.tindex[la], la, ra, c);
        }
        v = wave_sum(v);
        v = __shfl(v, 0, 64);
        if (COUNT) {
            c = wave_sum_u32(c);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        }
        if (lane == k) {
            acc = v;
            cnt = c;
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(kBmWaves * 64, 4) void dfire_bm_pairs(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    __shared__ __attribute__((aligned(16))) BmShared S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_rt = T->m.rec_n_tiles;
    const unsigned parts = (unsigned)((T->cap + kBmPartEntries - 1) / kBmPartEntries);
    const unsigned job = blockIdx.x / parts, part = blockIdx.x % parts;
    const size_t tp = job >> 3;
    const int a = (int)(job & 7u);
    const uint32_t n = T->tp_count[tp];
    const uint32_t lo = part * (uint32_t)kBmPartEntries;
    if (lo >= n) return;
    const uint32_t hi = n < lo + (uint32_t)kBmPartEntries ? n : lo + (uint32_t)kBmPartEntries;
    const int lt = (int)(tp / (unsigned)n_rt), RT = (int)(tp % (unsigned)n_rt);
    const int ls = lt * 8 + a;

    // ---- set-up: LUT, zero slots behind the cubes, this wave's entries
    {
        const uint8_t *lut = COUNT ? T->m.lut_full : T->m.lut;
        for (int i = tid; i < kBmLutBytes / 16; i += kBmWaves * 64) reinterpret_cast<uint4 *>(S.lut)[i] = reinterpret_cast<const uint4 *>(lut)[i];
        if (tid < 8) reinterpret_cast<uint32_t *>(S.cube[tid >> 2] + 64 * kBmRowBytes)[tid & 3] = 0u;
    }
    const uint32_t per_wave = ((hi - lo + kBmWaves * 64 - 1) / (kBmWaves * 64)) * 64;   // <= 512
    const uint32_t my_lo = lo + (uint32_t)wave * per_wave;
    const uint32_t my_hi = my_lo + per_wave < hi ? my_lo + per_wave : hi;
    const int n_chunks = my_lo < my_hi ? (int)((my_hi - my_lo + 63) / 64) : 0;
    for (int k = 0; k < n_chunks; k++) {
        const uint32_t e = my_lo + (uint32_t)k * 64 + lane;
        const unsigned long long m = e < my_hi ? T->ent_mask[tp * T->cap + e] : 0ull;
        S.row_bits[wave][k * 64 + lane] = (unsigned char)(m >> (8 * a));
    }
    BmWaveCtx W{tp, ls, RT, my_lo};

    // the ligand subtile's local coordinates (uniform)
    float Lx[8], Ly[8], Lz[8];
    bool Lreal[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float4 v = reinterpret_cast<const float4 *>(T->m.lig_local)[ls * 8 + i];
        Lx[i] = v.x; Ly[i] = v.y; Lz[i] = v.z;
        Lreal[i] = v.w != 0.f;
    }

    // ---- table rows of a block -> LDS: 704 pieces of 16 bytes, one LDS-DMA instruction per KiB
    const int piece0 = wave * 64 + lane, piece1 = (wave + 8) * 64 + lane;   // wave w copies KiB w and, if w < 3, KiB w + 8
    const int row0 = piece0 / 11, row1 = piece1 / 11;
    const uint32_t src0 = T->m.lig_rowbase[ls * 8 + (row0 >> 3)] + (uint32_t)(piece0 % 11) * 16u;
    const uint32_t src1 = wave < 3 ? T->m.lig_rowbase[ls * 8 + (row1 >> 3)] + (uint32_t)(piece1 % 11) * 16u : 0u;
    auto stage_cube = [&](int b) {
        const unsigned char *rows = reinterpret_cast<const unsigned char *>(T->m.rows);
        const uint32_t *roff = T->m.rec_rowoff + (size_t)RT * 64 + b * 8;
        unsigned char *dst = S.cube[b & 1];
        __builtin_amdgcn_global_load_lds((const global_u32 *)(rows + src0 + roff[row0 & 7]), (lds_u32 *)(dst + wave * 1024), 16, 0, 0);
        if (wave < 3)
            __builtin_amdgcn_global_load_lds((const global_u32 *)(rows + src1 + roff[row1 & 7]), (lds_u32 *)(dst + (wave + 8) * 1024), 16, 0, 0);
    };
    stage_cube(0);

    uint32_t queued = 0;   // wave-uniform

    auto run_block = [&](auto buf_tag, int b) {
        constexpr int BUF = decltype(buf_tag)::value;
        const unsigned char *cube = S.cube[BUF];
        // receptor subtile b of the tile: 4 pair records, wave-uniform
        const PackedRecPair *rp = T->m.rec_pairs + (size_t)RT * 32 + b * 4;
        v2f Rx[4], Ry[4], Rz[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            Rx[q] = v2f{rp[q].x0, rp[q].x1};
            Ry[q] = v2f{rp[q].y0, rp[q].y1};
            Rz[q] = v2f{rp[q].z0, rp[q].z1};
        }
        uint32_t pend_n = 0, pend_head = 0;
        for (int k = 0; k <= n_chunks; k++) {
            if (k < n_chunks) {
                const uint32_t bits = S.row_bits[wave][k * 64 + lane];
                const bool act = (bits >> b) & 1u;
                const unsigned long long m = __ballot(act);
                if (act) {
                    const uint32_t at = pend_head + pend_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    S.pend[wave][at & 127u] = (uint32_t)(k * 64 + lane) | bits << 16;
                }
                pend_n += (uint32_t)__popcll(m);
            }
            if (!(pend_n >= 64u || (k == n_chunks && pend_n > 0u))) continue;
            // ---- one batch: lane = entry
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int count = pend_n >= 64u ? 64 : (int)pend_n;
            const bool valid = lane < count;
            const uint32_t item = S.pend[wave][(pend_head + (valid ? lane : 0)) & 127u];
            pend_head += (uint32_t)count;
            pend_n -= (uint32_t)count;
            const uint32_t el = item & 0xffffu, bits = item >> 16;
            const size_t e = (size_t)my_lo + el;
            const size_t pose = T->ent_pose[tp * T->cap + e];
            const Affine A = bm_load_affine(T->rt, pose);
            const size_t pslot = (tp * 8 + (size_t)a) * T->cap + e;
            const bool first = (int)__builtin_ctz(bits) == b;   // the entry's first block in this row: nothing to add to yet
            double prev = 0.0;
            uint32_t prev_cnt = 0;
            if (!first) {
                prev = T->ent_partial[pslot];
                if (COUNT) prev_cnt = T->ent_count[pslot];
            }
            float lx[8], ly[8], lz[8];
#pragma unroll
            for (int i = 0; i < 8; i++) bm_apply(A, Lx[i], Ly[i], Lz[i], lx[i], ly[i], lz[i]);
            double acc = 0.0;
            uint32_t cnt = 0;
            const uint32_t queued_before = queued;
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const v2f dx = Rx[q] - v2f{lx[i], lx[i]}, dy = Ry[q] - v2f{ly[i], ly[i]}, dz = Rz[q] - v2f{lz[i], lz[i]};
                    v2f D = __builtin_elementwise_fma(dz, dz, v2f{0.5f, 0.5f});
                    D = __builtin_elementwise_fma(dy, dy, D);
                    D = __builtin_elementwise_fma(dx, dx, D);
                    const uint32_t c0 = bm_cvt_u32(fminf(D.x, kBmCellMax)), c1 = bm_cvt_u32(fminf(D.y, kBmCellMax));
                    const uint32_t w0 = S.lut[c0], w1 = S.lut[c1];
                    acc += *reinterpret_cast<const double *>(cube + (i * 8 + 2 * q) * kBmRowBytes + w0);
                    acc += *reinterpret_cast<const double *>(cube + (i * 8 + 2 * q + 1) * kBmRowBytes + w1);
                    asm volatile("" : "+v"(acc));   // add here, not 64 values later (the scheduler would park them all in registers)
                    if (COUNT && Lreal[i]) cnt += (w0 != 0u && w0 < kBmFlagged ? 1u : 0u) + (w1 != 0u && w1 < kBmFlagged ? 1u : 0u);
                    const uint32_t wm = w0 > w1 ? w0 : w1;
                    if (__builtin_expect(__ballot(wm >= kBmFlagged) != 0ull, 0)) {
                        // (rare: keep the compiler from preparing any of this outside the branch for all 32 steps)
                        uint32_t el_here = el;
                        asm volatile("" : "+v"(el_here));
                        const bool f0 = valid && w0 >= kBmFlagged, f1 = valid && w1 >= kBmFlagged;
                        const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
                        const uint32_t n0 = (uint32_t)__popcll(m0);
                        const uint32_t i0 = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                        const uint32_t i1 = queued + n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                        const uint32_t qi = el_here | (uint32_t)(i * 8 + 2 * q) << 9 | (uint32_t)b << 15;
                        if (f0 && i0 < (uint32_t)kBmQueue) S.queue[wave][i0] = qi;
                        if (f1 && i1 < (uint32_t)kBmQueue) S.queue[wave][i1] = qi + (1u << 9);
                        queued += n0 + (uint32_t)__popcll(m1);
                    }
                }
            }
            if (__builtin_expect(queued > (uint32_t)kBmQueue, 0)) {
                queued = queued_before;   // forget what this batch queued: all of it again in f64
                bm_exact_batch<COUNT>(T, W, b, el, count, lane, acc, cnt);
            }
            if (valid) {
                T->ent_partial[pslot] = prev + acc;
                if (COUNT) T->ent_count[pslot] = prev_cnt + cnt;
            }
            if (queued > (uint32_t)kBmQueue / 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                bm_drain<COUNT>(T, W, S.queue[wave], queued, lane);
                queued = 0;
            }
        }
    };

    for (int b = 0; b < 8; b++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of cube b has landed (and its stores are out)
        __syncthreads();                                   // everybody's has; nobody reads cube b - 1 any more
        if (b + 1 < 8) stage_cube(b + 1);
        if (b & 1) run_block(std::integral_constant<int, 1>{}, b);
        else run_block(std::integral_constant<int, 0>{}, b);
    }
    if (queued) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bm_drain<COUNT>(T, W, S.queue[wave], queued, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_gather: wave = pose; lanes over the ligand tiles, fixed order, then a fixed tree
// ---------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(64) void dfire_bm_gather(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    const int lane = threadIdx.x;
    const long long pp = bm_pose_of(T, blockIdx.x);
    if (pp < 0) return;
    const size_t pose = (size_t)pp;
    const int n_lt = T->m.lig.n_tiles, n_rt = T->m.rec_n_tiles;
    double s = 0.0;
    uint32_t cnt = 0, tested = 0;
    for (int lt = lane; lt < n_lt; lt += 64) {
        const size_t slot = pose * (size_t)n_lt + lt;
        if (COUNT) tested += T->tile_tested[slot];
        const uint32_t n_vis = T->vis_count[slot];
        for (uint32_t v = 0; v < n_vis; v++) {
            const uint32_t ent = T->vis_entry[slot * (size_t)n_rt + v];
            const size_t tp = (size_t)lt * n_rt + (ent >> 24);
            const size_t idx = ent & 0xffffffu;
            const unsigned long long m = T->ent_mask[tp * T->cap + idx];
            for (int a = 0; a < 8; a++) {
                if (((m >> (8 * a)) & 0xffull) == 0ull) continue;
                const size_t pslot = (tp * 8 + (size_t)a) * T->cap + idx;
                s += T->ent_partial[pslot];
                if (COUNT) cnt += T->ent_count[pslot];
            }
        }
    }
    s = wave_sum(s);
    if (COUNT) {
        cnt = wave_sum_u32(cnt);
        tested = wave_sum_u32(tested);
    }
    if (lane == 0) {
        s += (double)T->exact_fix[pose] * (1.0 / kBmFixScale);
        T->partial[2 * pose] = s;
        T->partial[2 * pose + 1] = 0.0;
        if (COUNT) {
            T->count_partial[pose] = cnt + T->exact_count[pose];
            if (T->tested_partial) T->tested_partial[pose] = tested;
            if (T->exact_partial) T->exact_partial[pose] = T->exact_pairs[pose];
        }
    }
}

}  // namespace

size_t bm_pairs_lds_bytes() { return sizeof(BmShared); }

hipError_t launch_bm_pose(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(dfire_bm_pose, dim3((unsigned)((t.n_poses + 255) / 256)), dim3(256), 0, stream, t);
    return hipGetLastError();
}

hipError_t launch_bm_cull(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    const size_t blocks = t.n_poses * (size_t)t.m.lig.n_tiles;
    if (blocks > 0x7fffffffULL || t.m.rec_n_tiles > 255) return hipErrorInvalidValue;
    if (t.ent_count != nullptr) hipLaunchKernelGGL((dfire_bm_cull<true>), dim3((unsigned)blocks), dim3(64), 0, stream, t);
    else hipLaunchKernelGGL((dfire_bm_cull<false>), dim3((unsigned)blocks), dim3(64), 0, stream, t);
    return hipGetLastError();
}

hipError_t launch_bm_pairs(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    const size_t parts = (t.cap + kBmPartEntries - 1) / kBmPartEntries;
    const size_t blocks = (size_t)t.m.lig.n_tiles * t.m.rec_n_tiles * 8 * parts;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    if (t.ent_count != nullptr) hipLaunchKernelGGL((dfire_bm_pairs<true>), dim3((unsigned)blocks), dim3(kBmWaves * 64), 0, stream, t);
    else hipLaunchKernelGGL((dfire_bm_pairs<false>), dim3((unsigned)blocks), dim3(kBmWaves * 64), 0, stream, t);
    return hipGetLastError();
}

hipError_t launch_bm_gather(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    if (t.ent_count != nullptr) hipLaunchKernelGGL((dfire_bm_gather<true>), dim3((unsigned)t.n_poses), dim3(64), 0, stream, t);
    else hipLaunchKernelGGL((dfire_bm_gather<false>), dim3((unsigned)t.n_poses), dim3(64), 0, stream, t);
    return hipGetLastError();
}

}  // namespace ld
