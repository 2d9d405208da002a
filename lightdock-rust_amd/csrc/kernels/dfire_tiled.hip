// dfire_tiled.hip -- K1 for DFIRE with bounding-box culling (gfx950 / MI355X).
//
// DFIRE only counts pairs closer than 15 A (src/dfire.rs:334): about 1 % of the 11.2 M atom
// pairs of the 1k4c example.  This kernel evaluates the same sum as src/dfire.rs:325-345 but
// discards whole blocks of pairs by box distance before touching them.
//
// Shape: "ligand tile stationary, receptor tiles streamed".
//   wave64 = (pose, ligand tile of 64 atoms); a workgroup is 1..16 such waves of one pose.
//   1. The wave poses its 64 ligand atoms in registers (q v q^-1 + t, then ANM;
//      src/dfire.rs:282-302), parks the 32-byte records in its private 2 KiB LDS slice and
//      builds 8 subtile boxes (8 atoms each) + the tile box with xor-shuffle butterflies.
//   2. 64 lanes test the tile box against 64 receptor tile boxes per ballot.
//   3. Every surviving receptor tile is copied L2 -> LDS by LDS-DMA (64 lanes x 2 x 16 B, no
//      VGPRs), and while the copy is in flight 64 lanes test the 8 x 8 subtile-box pairs in
//      one ballot.
//   4. The surviving subtile pairs are done row by row (row = one ligand subtile): the ligand
//      record is read once per row, each trip of the inner loop takes two receptor subtiles.
//      A block is 64 distinct atom pairs: lane (i, j) takes ligand atom i and receptor atom j
//      and runs the reference's pair body: f64 d2 in the reference's operation order, distance
//      bin through an exact cell LUT (which also encodes the cutoff), gather of
//      potential[type_i][type_j][bin] from a table re-laid out in 128-byte patches, interface
//      flags on a rare slow path.  The two gathers of a trip are retired one trip later.
//   5. wave64 shuffle reduction; one partial per (pose, workgroup).
// The receptor image (records + boxes) is static in HBM/L2 and shared by all poses; with
// receptor ANM (src/dfire.rs:304-320) dfire_prepare_receptor writes one image per pose first.
//
// Box tests are conservative (boxes rounded outwards, cutoff padded): no in-cutoff pair is
// ever dropped, and the pair body is bit-identical to the all-pairs kernel; only the order of
// the f64 += differs.  Compiled with -ffp-contract=off.  No MFMA (lookup/reduction).
#include "dfire_tiled.hpp"

#include <cmath>

namespace ld {

namespace {

struct Quat {
    double w, x, y, z;
};
__device__ __forceinline__ Quat qmul(const Quat &a, const Quat &b) {  // src/qt.rs:174-185
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}
__device__ __forceinline__ Quat qinverse(const Quat &q) {  // src/qt.rs:48-50
    const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return Quat{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}

// Everything inside the kernel lives in coordinates scaled by 2: (2a - 2b)^2 = 4 (a - b)^2 holds
// bit for bit in IEEE arithmetic (power-of-two scaling commutes with rounding), so
// D = 4 * d2 exactly, the cutoff d2 <= 225 is D <= 900, and DFIRE's 0.25 A^2 binning cell is
// simply (int)D -- one conversion, no multiply.
constexpr int kSliceRecords = 64 + 64;  // per wave: ligand tile, receptor tile
constexpr double kCutScaled = 900.0;       // 4 * 15^2, src/dfire.rs:334
constexpr float kCut2Padded = 900.04f;     // the same for the f32 box tests, padded for their rounding

__device__ __forceinline__ float round_down(double v) {
    float f = (float)v;
    return ((double)f > v) ? nextafterf(f, -INFINITY) : f;
}
__device__ __forceinline__ float round_up(double v) {
    float f = (float)v;
    return ((double)f < v) ? nextafterf(f, INFINITY) : f;
}
__device__ __forceinline__ float axis_gap(float lo_a, float hi_a, float lo_b, float hi_b) {
    return fmaxf(0.0f, fmaxf(lo_a - hi_b, lo_b - hi_a));
}
__device__ __forceinline__ float uniform_f32(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
struct BoxRegs {
    float lox, loy, loz, hix, hiy, hiz;
};
__device__ __forceinline__ float box_gap2(const BoxRegs &a, const TiledBox &b) {
    const float gx = axis_gap(a.lox, a.hix, b.lox, b.hix);
    const float gy = axis_gap(a.loy, a.hiy, b.loy, b.hiy);
    const float gz = axis_gap(a.loz, a.hiz, b.loz, b.hiz);
    return gx * gx + gy * gy + gz * gz;
}
// min/max over lane groups: masks 1,2,4 -> groups of 8 lanes; 8,16,32 -> the whole wave
template <int FROM, int TO>
__device__ __forceinline__ void box_butterfly(BoxRegs &b) {
#pragma unroll
    for (int m = FROM; m < TO; m <<= 1) {
        b.lox = fminf(b.lox, __shfl_xor(b.lox, m, 64)); b.hix = fmaxf(b.hix, __shfl_xor(b.hix, m, 64));
        b.loy = fminf(b.loy, __shfl_xor(b.loy, m, 64)); b.hiy = fmaxf(b.hiy, __shfl_xor(b.hiy, m, 64));
        b.loz = fminf(b.loz, __shfl_xor(b.loz, m, 64)); b.hiz = fmaxf(b.hiz, __shfl_xor(b.hiz, m, 64));
    }
}
__device__ __forceinline__ BoxRegs point_box(bool valid, double x, double y, double z) {
    BoxRegs b;
    b.lox = valid ? round_down(x) : INFINITY; b.hix = valid ? round_up(x) : -INFINITY;
    b.loy = valid ? round_down(y) : INFINITY; b.hiy = valid ? round_up(y) : -INFINITY;
    b.loz = valid ? round_down(z) : INFINITY; b.hiz = valid ? round_up(z) : -INFINITY;
    return b;
}
__device__ __forceinline__ TiledBox to_box(const BoxRegs &b) {
    return TiledBox{b.lox, b.loy, b.loz, 0.f, b.hix, b.hiy, b.hiz, 0.f};
}


__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Receptor image: one wave per (pose, receptor tile); lane = atom.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dfire_prepare_receptor(const PrepareReceptorLaunch P) {
    const size_t pose = blockIdx.x / (unsigned)P.n_tiles;
    const int tile = blockIdx.x % (unsigned)P.n_tiles;
    if (P.active != nullptr && P.active[pose] == 0) return;
    const int lane = threadIdx.x;
    const int a = tile * 64 + lane;
    const size_t pad = (size_t)P.n_tiles * 64;
    double x = P.x[a], y = P.y[a], z = P.z[a];
    if (P.num_anm > 0) {  // src/dfire.rs:304-320
        const double *rec_nm = P.poses + pose * P.stride + 7;
        for (int k = 0; k < P.num_anm; k++) {
            const double c = rec_nm[k];
            const double *m = P.modes + (size_t)k * 3 * pad;
            x += m[a] * c;
            y += m[pad + a] * c;
            z += m[2 * pad + a] * c;
        }
    }
    TiledAtom r;  // records carry 2*x, 2*y, 2*z (exact), see "scaled coordinates" in the header comment
    r.x = 2.0 * x;
    r.y = 2.0 * y;
    r.z = 2.0 * z;
    r.tindex = P.tindex[a];
    r.slot = P.slot[a];
    P.atoms_out[pose * pad + a] = r;
    BoxRegs b = point_box(a < P.n_real, r.x, r.y, r.z);
    box_butterfly<1, 8>(b);
    if ((lane & 7) == 0) P.sub_out[(pose * (size_t)P.n_tiles + tile) * 8 + (lane >> 3)] = to_box(b);
    box_butterfly<8, 64>(b);
    if (lane == 0) P.tile_out[pose * (size_t)P.n_tiles + tile] = to_box(b);
}

// ---------------------------------------------------------------------------------------------
// Pair kernel
// ---------------------------------------------------------------------------------------------
struct PairCtx {
    const double *bin_step;  // scaled by 4
    double iface_scaled;     // 4 * iface_d2
    uint32_t *pose_flags;
    int rec_flag_words;
};

// 16-byte halves of a record: two ds_read_b128 per record
struct alignas(16) RecLo {
    double x, y;
};
struct alignas(16) RecHi {
    double z;
    uint32_t tindex;
    int32_t slot;
};
typedef double vec2d __attribute__((ext_vector_type(2)));
typedef float vec4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(1))) unsigned int global_u32;
__device__ __forceinline__ void read_record(const TiledAtom *p, RecLo &lo, RecHi &hi) {
    // two 16-byte vector loads (ds_read_b128 each; member-wise loads become the slower ds_read2_b64)
    const vec2d a = *reinterpret_cast<const vec2d *>(p);
    const vec2d b = *reinterpret_cast<const vec2d *>(reinterpret_cast<const unsigned char *>(p) + 16);
    lo.x = a.x;
    lo.y = a.y;
    hi.z = b.x;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(b.y);
    hi.tindex = (uint32_t)bits;
    hi.slot = (int32_t)(bits >> 32);
}

// the rare tail of a pair (LUT word flagged kTiledLutSlow): the cutoff itself, the exact position
// of a bin step inside the cell, interface flags.  Returns the pair's LUT term.
__device__ __forceinline__ uint32_t pair_slow_path(const PairCtx &c, uint32_t word, double D, int32_t lslot, int32_t rslot) {
    if (!(D <= kCutScaled)) return kTiledLutMiss;  // d2 <= 225 (src/dfire.rs:334)
    uint32_t bin = word & 0x1fu;
    if (word & 0x80u) bin += D >= c.bin_step[bin + 1] ? 1u : 0u;
    if ((word & 0x40u) && D <= c.iface_scaled) {  // d <= 3.9 (src/dfire.rs:339-342)
        if (rslot >= 0) atomicOr(&c.pose_flags[rslot >> 5], 1u << (rslot & 31));
        if (lslot >= 0) atomicOr(&c.pose_flags[c.rec_flag_words + (lslot >> 5)], 1u << (lslot & 31));
    }
    return tiled_bin_term(bin);
}

// potential[...] through a raw buffer: the offset is 32-bit (no 64-bit address arithmetic) and an
// offset past the end (kTiledLutMiss) reads 0.0 without a memory request
typedef unsigned int vec2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double table_entry(__amdgpu_buffer_rsrc_t table, uint32_t byte_offset) {
    const vec2u v = __builtin_amdgcn_raw_buffer_load_b64(table, (int)byte_offset, 0, 0);
    return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}

// MAXT/MINW = launch bounds.  The 256-thread instantiation needs 50 VGPRs without any cap; 8
// workgroups per CU are set by its 20 192 bytes of LDS.
template <bool COUNT, int MAXT, int MINW>
__global__ __launch_bounds__(MAXT, MINW) void dfire_tiled_pairs(const TiledLaunch T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- LDS carve (every offset a multiple of 16) ---------------------------------------
    uint32_t *lut = reinterpret_cast<uint32_t *>(smem);
    size_t off = kDfireLutCells * sizeof(uint32_t);
    double *bin_step = reinterpret_cast<double *>(smem + off);
    off += kDfireSteps * sizeof(double);
    TiledAtom *slices = reinterpret_cast<TiledAtom *>(smem + off);  // per wave: kSliceRecords records
    // 3616 + 192 + 4 x 4096 = 20 192 bytes for 4 waves: 8 workgroups (32 waves) per CU.  The
    // per-wave results of the final reduction reuse the first 16 bytes of each wave's own slice.

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Block id -> (pose, group) through a pseudo-random permutation.  The ligand tiles at the
    // interface carry most of the work; the hardware deals block ids round-robin to XCDs and
    // CUs, and with the plain pose-major order (or any map that keeps the low bits of the block
    // id in the group index, e.g. a multiplicative one when groups-per-pose is a power of two)
    // the same heavy group keeps landing on the same XCDs: measured up to 2x slower.  The
    // permutation is a bijective mixer on the next power of two, cycle-walked into [0, total):
    // every XCD and CU sees the same mix of heavy and light workgroups at all times.
    const unsigned long long total_items = (unsigned long long)T.n_poses * (unsigned)T.n_groups;
    unsigned long long item_id = blockIdx.x;
    {
        const int bits = 64 - __builtin_clzll(total_items | 1ull);  // total_items < 2^bits
        const unsigned long long mask = (1ull << bits) - 1ull;
        const int half = (bits + 1) / 2;
        do {  // each step is a bijection on `bits`-bit numbers; expected < 2 rounds
            item_id = (item_id * 0x9E3779B97F4A7C15ull) & mask;
            item_id ^= item_id >> half;
            item_id = (item_id * 0xD6E8FEB86659FD93ull) & mask;
            item_id ^= item_id >> half;
        } while (item_id >= total_items);
    }
    const size_t pose = (size_t)(item_id / (unsigned)T.n_groups);
    const int group = (int)(item_id % (unsigned)T.n_groups);
    if (T.active != nullptr && T.active[pose] == 0) return;

    // 904 words = 226 x 16 bytes: one load per thread and one round trip for a 256-thread workgroup
    for (int i = tid; i < kDfireLutCells / 4; i += blockDim.x)
        reinterpret_cast<uint4 *>(lut)[i] = reinterpret_cast<const uint4 *>(T.lut)[i];
    if (tid < kDfireSteps) bin_step[tid] = 4.0 * T.bin_step[tid];  // scaled coordinates
    __syncthreads();

    TiledAtom *ligt = slices + wave * kSliceRecords;
    TiledAtom *rect = ligt + 64;
    const int li = lane >> 3, lj = lane & 7;
    double acc = 0.0, pend0 = 0.0, pend1 = 0.0;
    uint32_t cnt = 0, tested = 0;

    // work item = (ligand tile, part): `split` waves share one ligand tile and take every
    // split-th surviving receptor tile, so the tiles at the interface do not make one long wave
    const int item = group * T.waves + wave;
    const int LT = item / T.split;
    const int part = item % T.split;
    if (LT < T.lig.n_tiles) {
        const double *row = T.poses + pose * T.stride;
        const TiledAtom *rec_atoms = T.rec.atoms + pose * T.rec.pose_stride_atoms;
        const TiledBox *rec_sub = T.rec.sub_boxes + pose * T.rec.pose_stride_sub;
        const TiledBox *rec_tile = T.rec.tile_boxes + pose * T.rec.pose_stride_tile;

        // ---- 1. pose this lane's ligand atom ---------------------------------------------------
        const int la = LT * 64 + lane;
        const bool valid = la < T.lig.n_real;
        TiledAtom me;
        {
            const double tx = row[0], ty = row[1], tz = row[2];
            const Quat q{row[3], row[4], row[5], row[6]};
            const Quat qinv = qinverse(q);
            const Quat v{0.0, T.lig.x[la], T.lig.y[la], T.lig.z[la]};
            const Quat r = qmul(qmul(q, v), qinv);
            double px = r.x + tx, py = r.y + ty, pz = r.z + tz;
            if (T.use_anm && T.lig.num_anm > 0) {
                const double *lig_nm = row + 7 + T.anm_rec;
                const size_t pad = (size_t)T.lig.n_tiles * 64;
                for (int k = 0; k < T.lig.num_anm; k++) {
                    const double c = lig_nm[k];
                    const double *m = T.lig.modes + (size_t)k * 3 * pad;
                    px += m[la] * c;
                    py += m[pad + la] * c;
                    pz += m[2 * pad + la] * c;
                }
            }
            me.x = valid ? 2.0 * px : 1.0e30;  // padding: far away, on the other side of the receptor's padding
            me.y = valid ? 2.0 * py : 0.0;
            me.z = valid ? 2.0 * pz : 0.0;
            me.tindex = T.lig.tindex[la];
            me.slot = T.lig.slot[la];
        }
        ligt[lane] = me;
        BoxRegs sub = point_box(valid, me.x, me.y, me.z);
        box_butterfly<1, 8>(sub);  // lanes 8a..8a+7 now hold the box of ligand subtile a
        BoxRegs whole = sub;
        box_butterfly<8, 64>(whole);
        // the tile box is the same in every lane: keep it in scalar registers (6 VGPRs less across
        // the loops below; the kernel sits exactly at the 64 VGPRs that allow 8 waves per SIMD)
        whole.lox = uniform_f32(whole.lox); whole.loy = uniform_f32(whole.loy); whole.loz = uniform_f32(whole.loz);
        whole.hix = uniform_f32(whole.hix); whole.hiy = uniform_f32(whole.hiy); whole.hiz = uniform_f32(whole.hiz);

        PairCtx ctx;
        ctx.bin_step = bin_step;
        ctx.iface_scaled = T.iface_scaled;  // 4 * iface_d2, from the host: a kernel argument stays in SGPRs
        ctx.pose_flags = T.flags + pose * (size_t)(T.rec.flag_words + T.lig.flag_words);
        ctx.rec_flag_words = T.rec.flag_words;
        const __amdgpu_buffer_rsrc_t table = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(T.table), 0, (int)(kTiledTableDoubles * sizeof(double)), 0x00020000);

        // ---- 2. receptor tiles, 64 per ballot ---------------------------------------------------
        for (int base = 0; base < T.rec.n_tiles; base += 64) {
            bool tile_near = false;
            if (base + lane < T.rec.n_tiles) tile_near = box_gap2(whole, rec_tile[base + lane]) <= kCut2Padded;
            unsigned long long rtmask = __ballot(tile_near);
            if (rtmask == 0) continue;

            // ---- 3. stream the surviving tiles through the LDS slice
            int turn = 0;
            while (rtmask) {
                const int RT = base + __ffsll(rtmask) - 1;
                rtmask &= rtmask - 1;
                if (turn++ % T.split != part) continue;
                // 2 KiB of records straight from L2/HBM into this wave's LDS slice (LDS-DMA: no
                // VGPRs, no ds_write); lane l moves bytes [16 l, 16 l + 16) of each KiB
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // reads of the previous tile are done
                const unsigned char *gsrc = reinterpret_cast<const unsigned char *>(rec_atoms + (size_t)RT * 64) + lane * 16;
                __builtin_amdgcn_global_load_lds((const global_u32 *)gsrc, (lds_u32 *)rect, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const global_u32 *)(gsrc + 1024),
                                                 (lds_u32 *)(reinterpret_cast<unsigned char *>(rect) + 1024), 16, 0, 0);
                const TiledBox nb = rec_sub[(size_t)RT * 8 + lj];
                const bool sub_near = box_gap2(sub, nb) <= kCut2Padded;  // ligand subtile li x receptor subtile lj
                unsigned long long smask = __ballot(sub_near);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA has landed
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // rect/ligt written before anyone reads
                if (COUNT) tested += (uint32_t)__popcll(smask);

                // ---- 4. surviving subtile pairs, row by row ---------------------------------------
                // bit k of smask = (ligand subtile k >> 3, receptor subtile k & 7).  The kernel is
                // bound by LDS reads of the records as much as by VALU, so the blocks of one
                // ligand subtile (a row of the mask, 3 on average) are done together: the ligand
                // record is read once per row and stays in registers, and each trip of the inner
                // loop takes two receptor subtiles of the row (4 instead of 8 ds_read_b128).  An
                // odd one left over is paired with the far-away subtile (every pair misses).  One
                // straight-line body; both blocks of a trip are independent.
                while (smask) {
                    const int row = (__ffsll(smask) - 1) >> 3;
                    unsigned rbits = (unsigned)(smask >> (row * 8)) & 0xffu;
                    smask &= ~(0xffull << (row * 8));
                    RecLo Llo;
                    RecHi Lhi;
                    read_record(&ligt[row * 8 + li], Llo, Lhi);
                    while (rbits) {
                        const int j0 = __ffs(rbits) - 1;
                        rbits &= rbits - 1;
                        const bool two = rbits != 0;
                        const int j1 = two ? __ffs(rbits) - 1 : 0;
                        rbits &= rbits - 1;  // 0 stays 0
                        RecLo R0lo;
                        RecHi R0hi;
                        read_record(&rect[j0 * 8 + lj], R0lo, R0hi);
                        uint32_t t1 = kTiledLutMiss, off1 = 0;
                        double D1 = 0.0;
                        int32_t rslot1 = -1;
                        if (two) {  // wave-uniform: an odd block left over in its row goes alone
                            RecLo R1lo;
                            RecHi R1hi;
                            read_record(&rect[j1 * 8 + lj], R1lo, R1hi);
                            const double dx1 = R1lo.x - Llo.x, dy1 = R1lo.y - Llo.y, dz1 = R1hi.z - Lhi.z;
                            D1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
                            t1 = lut[min((unsigned)(int)D1, (unsigned)(kDfireLutCells - 1))];
                            off1 = Lhi.tindex + R1hi.tindex;
                            rslot1 = R1hi.slot;
                        }
                        // (x1 - la[0])^2 + (y1 - la[1])^2 + (z1 - la[2])^2, src/dfire.rs:331-333 (x4)
                        const double dx0 = R0lo.x - Llo.x, dy0 = R0lo.y - Llo.y, dz0 = R0hi.z - Lhi.z;
                        const double D0 = dx0 * dx0 + dy0 * dy0 + dz0 * dz0;
                        // cell -> table term; cells past the cutoff give kTiledLutMiss (no compare on D)
                        uint32_t t0 = lut[min((unsigned)(int)D0, (unsigned)(kDfireLutCells - 1))];
                        const bool slow0 = (int)t0 >= (int)kTiledLutSlow, slow1 = (int)t1 >= (int)kTiledLutSlow;
                        if (__builtin_expect(slow0 || slow1, 0)) {
                            if (slow0) t0 = pair_slow_path(ctx, t0, D0, Lhi.slot, R0hi.slot);
                            if (slow1) t1 = pair_slow_path(ctx, t1, D1, Lhi.slot, rslot1);
                        }
                        // retire the previous trip's gathers only now, so their L2 latency hides
                        // behind this trip's LDS reads and arithmetic; the asm pins the order "add
                        // the old value, then issue the new load into the same register"
                        acc += pend0;
                        acc += pend1;
                        asm volatile("" : "+v"(acc) : : "memory");
                        pend0 = table_entry(table, Lhi.tindex + R0hi.tindex + t0);  // src/dfire.rs:338, re-laid out
                        pend1 = table_entry(table, off1 + t1);
                        if (COUNT) cnt += (t0 < kTiledLutSlow ? 1u : 0u) + (t1 < kTiledLutSlow ? 1u : 0u);
                    }
                }
            }
        }
    }

    // ---- 5. reduction ----------------------------------------------------------------------------
    acc += pend0;
    acc += pend1;
    acc = wave_sum(acc);
    if (COUNT) cnt = wave_sum_u32(cnt);
    struct WaveResult {
        double sum;
        uint32_t count, tested;
    };
    static_assert(sizeof(WaveResult) == 16, "WaveResult overlays the head of a slice");
    if (lane == 0) {  // this wave is done with its slice
        WaveResult r;
        r.sum = acc;
        r.count = cnt;
        r.tested = tested;
        *reinterpret_cast<WaveResult *>(ligt) = r;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        uint32_t c = 0, t = 0;
        for (int w = 0; w < T.waves; w++) {
            const WaveResult r = *reinterpret_cast<const WaveResult *>(slices + w * kSliceRecords);
            s += r.sum;
            c += r.count;
            t += r.tested;
        }
        const size_t slot = pose * (size_t)T.n_groups + group;
        T.partial[2 * slot] = s;
        T.partial[2 * slot + 1] = 0.0;
        if (COUNT) {
            T.count_partial[slot] = c;
            if (T.tested_partial) T.tested_partial[slot] = t;
        }
    }
}

}  // namespace

size_t tiled_kernel_lds_bytes(const TiledLaunch &t) {
    size_t b = kDfireLutCells * sizeof(uint32_t) + kDfireSteps * sizeof(double);
    b += (size_t)t.waves * kSliceRecords * sizeof(TiledAtom);
    return b;
}

hipError_t launch_dfire_tiled(const TiledLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    const size_t blocks = t.n_poses * (size_t)t.n_groups;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = tiled_kernel_lds_bytes(t);
    const dim3 grid((unsigned)blocks), block((unsigned)t.waves * 64);
    if (t.waves <= 4) {
        if (t.count_partial != nullptr) hipLaunchKernelGGL((dfire_tiled_pairs<true, 256, 1>), grid, block, lds, stream, t);
        else hipLaunchKernelGGL((dfire_tiled_pairs<false, 256, 1>), grid, block, lds, stream, t);
    } else {
        if (t.count_partial != nullptr) hipLaunchKernelGGL((dfire_tiled_pairs<true, 1024, 1>), grid, block, lds, stream, t);
        else hipLaunchKernelGGL((dfire_tiled_pairs<false, 1024, 1>), grid, block, lds, stream, t);
    }
    return hipGetLastError();
}

hipError_t launch_prepare_receptor(const PrepareReceptorLaunch &p, hipStream_t stream) {
    if (p.n_poses == 0 || p.n_tiles == 0) return hipSuccess;
    const size_t blocks = p.n_poses * (size_t)p.n_tiles;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dfire_prepare_receptor, dim3((unsigned)blocks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ld
