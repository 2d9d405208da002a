// dfire_packed.hip -- K1 for DFIRE: box culling + packed-f32 pair test with an exact f64 path
// (gfx950 / MI355X).  Interface and numerics: dfire_packed.hpp.
//
// Shape: "ligand tile stationary, receptor tiles streamed", as dfire_tiled.hip.
//   wave64 = workgroup = (pose, ligand tile of 64 atoms)  (kPackedWaves = 1, see dfire_packed.hpp).
//   1. The wave poses its 64 ligand atoms in f64 (q v q^-1 + t, then ANM; src/dfire.rs:282-302),
//      parks 16-byte f32 records {u, type term} in its LDS slice and builds 8 subtile boxes + the
//      tile box with butterflies.
//   2. 64 lanes test the tile box against 64 receptor tile boxes per ballot.
//   3. Every surviving receptor tile (1 KiB of pair records) is copied L2 -> LDS by LDS-DMA while
//      64 lanes test the 8 x 8 subtile-box pairs.
//   4. The surviving 8 x 8 blocks of the tile pair are taken two per trip, any two: each half of
//      the wave does one block, lane (i, q) = ligand atom i x the receptor atoms (2 q, 2 q + 1) of
//      the block's receptor subtile, read as ONE 32-byte pair record.  128 atom pairs per trip in
//      3 packed subtractions and 3 packed fmas; per pair: convert, clamp, LUT word, offset,
//      gather from the 2 x 2 x 4-patch table (out-of-range offset = 0.0 without a memory
//      request), f64 add one trip later.
//   5. The pairs the f32 test could not decide are recomputed in f64 from the queue; wave64 shuffle
//      reduction; one partial per (pose, workgroup), folded by pose_energy_finish.
// Compiled with -ffp-contract=off: every f64 operation is the reference's; the f32 filter uses
// explicit fmas.  No MFMA (lookup/reduction).
#include "dfire_packed.hpp"

#include <cmath>

#include "dfire_device.hpp"

namespace ld {

namespace {

// ---------------------------------------------------------------------------------------------
// Receptor image: one wave per (receptor tile, 16 poses); lane = atom, the tile's modes stay in registers.
// ---------------------------------------------------------------------------------------------
constexpr int kPreparePoses = 16;   // poses per workgroup: the tile's modes are read once for all of them
constexpr int kPrepareModes = 10;   // modes kept in registers; any further ones are read per pose
__global__ __launch_bounds__(64) void dfire_packed_prepare(const PackedPrepareLaunch P) {
    const int tile = blockIdx.x % (unsigned)P.n_tiles;
    const size_t pose0 = (size_t)(blockIdx.x / (unsigned)P.n_tiles) * kPreparePoses;
    const int lane = threadIdx.x;
    const int a = tile * 64 + lane;
    const size_t pad = (size_t)P.n_tiles * 64;
    const double x0 = P.x[a], y0 = P.y[a], z0 = P.z[a];
    double mx[kPrepareModes], my[kPrepareModes], mz[kPrepareModes];
#pragma unroll
    for (int k = 0; k < kPrepareModes; k++) {
        const bool have = k < P.num_anm;
        const double *m = P.modes + (size_t)(have ? k : 0) * 3 * pad;
        mx[k] = have ? m[a] : 0.0;
        my[k] = have ? m[pad + a] : 0.0;
        mz[k] = have ? m[2 * pad + a] : 0.0;
    }
    const bool real = a < P.n_real;  // padding sits at x = -1e30 (scorer.cpp)
    const uint32_t my_term = P.tindex[a];
    const unsigned long long tracked = __ballot(real && P.slot[a] >= 0);  // atoms with an interface-flag slot
    for (int i = 0; i < kPreparePoses; i++) {
    const size_t pose = pose0 + i;
    if (pose >= P.n_poses) break;
    if (P.active != nullptr && P.active[pose] == 0) continue;
    double x = x0, y = y0, z = z0;
    if (P.num_anm > 0) {  // src/dfire.rs:304-320
        const double *rec_nm = P.poses + pose * P.stride + 7;
#pragma unroll
        for (int k = 0; k < kPrepareModes; k++) {
            if (k < P.num_anm) {
                const double c = rec_nm[k];
                x += mx[k] * c;
                y += my[k] * c;
                z += mz[k] * c;
            }
        }
        for (int k = kPrepareModes; k < P.num_anm; k++) {
            const double c = rec_nm[k];
            const double *m = P.modes + (size_t)k * 3 * pad;
            x += m[a] * c;
            y += m[pad + a] * c;
            z += m[2 * pad + a] * c;
        }
    }
    const float fx = frame_coord(x, P.cx, P.kappa), fy = frame_coord(y, P.cy, P.kappa), fz = frame_coord(z, P.cz, P.kappa);
    const bool inside = fabsf(fx) <= P.ubound && fabsf(fy) <= P.ubound && fabsf(fz) <= P.ubound;
    // record (4 j + q) of the tile holds the atoms (2 q, 2 q + 1) of its subtile j
    float *rec = reinterpret_cast<float *>(P.pairs_out + (pose * (size_t)P.n_tiles + tile) * 32 + (lane >> 1));
    const int h = lane & 1;
    rec[h] = real ? fx : -1.0e30f;
    rec[2 + h] = real ? fy : 0.f;
    rec[4 + h] = real ? fz : 0.f;
    reinterpret_cast<uint32_t *>(rec)[6 + h] = my_term | (real && !inside ? kPackedSlow : 0u);
    BoxRegs b = lane_box(real, fx, fy, fz);
    box_reduce8(b);
    {
        BoxRegs sub = b;
        box_widen(sub);
        if ((lane & 7) == 0) P.sub_out[(pose * (size_t)P.n_tiles + tile) * 8 + (lane >> 3)] = to_box(sub);
    }
    box_reduce64_from8(b);
    box_widen(b);
    // atoms with an interface-flag slot (restraint atoms, membrane beads): one bit per atom of the tile
    if (lane == 63) {
        TiledBox t = to_box(b);
        t.pad0 = __uint_as_float((uint32_t)tracked);
        t.pad1 = __uint_as_float((uint32_t)(tracked >> 32));
        P.tile_out[pose * (size_t)P.n_tiles + tile] = t;
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Pair kernel
// ---------------------------------------------------------------------------------------------
struct alignas(16) LigRecord {
    float x, y, z;
    uint32_t tindex;
};

// potential[...] through a raw buffer: 32-bit offsets, and an offset past the end (kPackedMiss)
// reads 0.0 without a memory request
__device__ __forceinline__ double table_entry(__amdgpu_buffer_rsrc_t table, uint32_t byte_offset) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(table, (int)byte_offset, 0, 0);
    return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}

// v_cvt_u32_f32 saturates: negative and NaN -> 0, too large -> 0xffffffff (a C++ cast leaves those undefined)
__device__ __forceinline__ uint32_t cvt_u32_sat(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// (a.lo - b.lo, a.hi - b.lo) and (a.lo - b.hi, a.hi - b.hi): one ligand coordinate against the same
// coordinate of two receptor atoms, the broadcast done by the operand selects of v_pk_add_f32
__device__ __forceinline__ v2f pk_sub_lo(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_sub_hi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The lean form of the cutoff cell sends the pairs just beyond 15 A to the bin slot 21 of their
// type pair, an unused entry of the patch that holds 0.0 (dfire_tiled.hpp: 24 slots for 21 bins).
// Counting launches must not count them: slot 21 is the second entry (bin % 4 == 1) of the sixth
// bin group, whatever the types.
__device__ __forceinline__ bool beyond_cutoff_slot(uint32_t byte_offset) {
    const uint32_t d = byte_offset / 8u;
    return (d % kTiledRecStride) / kTiledPatchDoubles == 5u && (d % 4u) == 1u;
}

// Second argument of __launch_bounds__ in HIP: waves per SIMD the register allocation must allow.  One-wave
// workgroups: the LUT in LDS limits the CU to 24 (unit cells) or 14 (half-unit cells) waves anyway, and the
// kernel runs equally fast at 3 to 6 waves per SIMD, so it gets the registers it asks for.
#ifndef LD_PACKED_WAVES_PER_SIMD
#define LD_PACKED_WAVES_PER_SIMD (kPackedWaves == 1 ? (SC == 1 ? 5 : 3) : 8)
#endif
template <bool COUNT, int SC>
__global__ __launch_bounds__(kPackedWaves * 64, LD_PACKED_WAVES_PER_SIMD) void dfire_packed_pairs(const PackedLaunch T) {
    // separate LDS objects: the backend tells the LDS-DMA target apart from the other arrays
    __shared__ __attribute__((aligned(16))) uint32_t s_lut[kPackedLutCells * SC];
    __shared__ __attribute__((aligned(16))) double s_step4[kDfireSteps];
    // ligand tile: 64 records + a ninth subtile that lies far away (the partner of an odd block left over)
    __shared__ __attribute__((aligned(16))) LigRecord s_lig[kPackedWaves][64 + 8];
    __shared__ __attribute__((aligned(32))) PackedRecPair s_rec[kPackedWaves][32];
    struct WaveResult {
        double sum;
        uint32_t count, tested, exact, pad;
    };
    __shared__ WaveResult s_res[kPackedWaves];
    // pairs for the exact path: ligand atom of the tile | receptor atom << 6
    __shared__ uint32_t s_queue[kPackedWaves][kPackedQueue];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block id -> (pose, group) through a bijective mixer, see dfire_tiled.hip
    const unsigned long long total_items = (unsigned long long)T.n_poses * (unsigned)T.n_groups;
    unsigned long long item_id = blockIdx.x;
    {
        const int bits = 64 - __builtin_clzll(total_items | 1ull);
        const unsigned long long mask = (1ull << bits) - 1ull;
        const int half = (bits + 1) / 2;
        do {
            item_id = (item_id * 0x9E3779B97F4A7C15ull) & mask;
            item_id ^= item_id >> half;
            item_id = (item_id * 0xD6E8FEB86659FD93ull) & mask;
            item_id ^= item_id >> half;
        } while (item_id >= total_items);
    }
    const size_t listed = (size_t)(item_id / (unsigned)T.n_groups);
    const int group = (int)(item_id % (unsigned)T.n_groups);
    if (T.pose_count != nullptr && listed >= (size_t)*T.pose_count) return;  // beyond the list of this step
    const size_t pose = T.pose_list ? (size_t)T.pose_list[listed] : listed;
    if (T.active != nullptr && T.active[pose] == 0) return;

    for (int i = tid; i < kPackedLutCells * SC / 4; i += kPackedWaves * 64)
        reinterpret_cast<uint4 *>(s_lut)[i] = reinterpret_cast<const uint4 *>(T.lut)[i];
    if (tid < kDfireSteps) s_step4[tid] = 4.0 * T.bin_step[tid];
    if (lane < 8) s_lig[wave][64 + lane] = LigRecord{1.0e30f, 0.f, 0.f, 0u};
    __syncthreads();

    LigRecord *ligt = s_lig[wave];
    PackedRecPair *rect = s_rec[wave];
    const int bj = lane & 7;                                 // box tests: lane = ligand subtile (lane >> 3) x receptor subtile bj
    const int ph = lane >> 5, pi = (lane >> 2) & 7, pq = lane & 3;  // pair loop: block of the trip, ligand atom, record
    double acc = 0.0, pend0 = 0.0, pend1 = 0.0;
    uint32_t cnt = 0, tested = 0, n_exact = 0;
    const uint32_t half_shift = (uint32_t)ph * 32u;
    uint32_t queued = 0;  // wave-uniform; beyond kPackedQueue the wave redoes its tile in f64 (overflow pass)
    constexpr float kCellMax = kPackedCellMax * SC;
    constexpr float kBoxCut = kCut2Padded * SC;  // boxes live in the record frame

    const int item = group * kPackedWaves + wave;
    const int LT = item / T.split;
    const int part = item % T.split;
    if (LT < T.lig.n_tiles) {
        const double *row = T.poses + pose * T.stride;
        const PackedRecPair *rec_pairs = T.rec.pairs + pose * T.rec.pose_stride_pairs;
        const TiledBox *rec_sub = T.rec.sub_boxes + pose * T.rec.pose_stride_sub;
        const TiledBox *rec_tile = T.rec.tile_boxes + pose * T.rec.pose_stride_tile;

        // ---- 1. pose this lane's ligand atom ---------------------------------------------------
        const int la = LT * 64 + lane;
        const bool valid = la < T.lig.n_real;
        float fx, fy, fz;
        {
            const Vec3 p = pose_ligand_atom(T.lig, T.use_anm, T.anm_rec, row, la);
            fx = frame_coord(p.x, T.cx, T.kappa);
            fy = frame_coord(p.y, T.cy, T.kappa);
            fz = frame_coord(p.z, T.cz, T.kappa);
            const bool inside = fabsf(fx) <= T.ubound && fabsf(fy) <= T.ubound && fabsf(fz) <= T.ubound;
            LigRecord me;
            me.x = valid ? fx : 1.0e30f;  // padding: far away, opposite the receptor's
            me.y = valid ? fy : 0.f;
            me.z = valid ? fz : 0.f;
            me.tindex = T.lig.tindex[la] | (valid && !inside ? kPackedSlow : 0u);
            ligt[lane] = me;
        }
        // ligand atoms with an interface-flag slot, one bit per atom of the tile
        const unsigned long long lig_tracked = T.lig.flag_words > 0 ? __ballot(valid && T.lig.slot[la] >= 0) : 0ull;
        BoxRegs sub = lane_box(valid, fx, fy, fz);
        box_reduce8(sub);  // lanes 8a..8a+7 now hold the box of ligand subtile a
        BoxRegs whole = sub;
        box_reduce64_from8(whole);
        box_widen(sub);
        box_widen(whole);
        whole.lox = lane63_f32(whole.lox); whole.loy = lane63_f32(whole.loy); whole.loz = lane63_f32(whole.loz);
        whole.hix = lane63_f32(whole.hix); whole.hiy = lane63_f32(whole.hiy); whole.hiz = lane63_f32(whole.hiz);

        const __amdgpu_buffer_rsrc_t table = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(T.table), 0, (int)(kTiledTableDoubles * sizeof(double)), 0x00020000);
        // ---- 2. receptor tiles, 64 per ballot; `tile_body(RT, tracked atoms of RT)` for every surviving
        // tile of this wave's share
        auto for_each_tile = [&](auto &&tile_body) {
            for (int base = 0; base < T.rec.n_tiles; base += 64) {
                bool tile_near = false;
                uint32_t trk_lo = 0, trk_hi = 0;  // this lane's tile: its atoms with an interface-flag slot
                if (base + lane < T.rec.n_tiles) {
                    const TiledBox tb = rec_tile[base + lane];
                    tile_near = box_gap2(whole, tb) <= kBoxCut;
                    trk_lo = __float_as_uint(tb.pad0);
                    trk_hi = __float_as_uint(tb.pad1);
                }
                unsigned long long rtmask = __ballot(tile_near);
                int turn = 0;
                while (rtmask) {
                    const int RT = base + __ffsll(rtmask) - 1;
                    rtmask &= rtmask - 1;
                    if (turn++ % T.split != part) continue;
                    const unsigned long long rec_tracked = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)trk_hi, RT - base) << 32) |
                                                           (uint32_t)__builtin_amdgcn_readlane((int)trk_lo, RT - base);
                    tile_body(RT, rec_tracked);
                }
            }
        };

        // ---- 3. stream the surviving tiles through the LDS slice
        for_each_tile([&](const int RT, const unsigned long long rec_tracked) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // reads of the previous tile are done
                const unsigned char *gsrc = reinterpret_cast<const unsigned char *>(rec_pairs + (size_t)RT * 32) + lane * 16;
                __builtin_amdgcn_global_load_lds((const global_u32 *)gsrc, (lds_u32 *)rect, 16, 0, 0);
                const TiledBox nb = rec_sub[(size_t)RT * 8 + bj];
                const bool sub_near = box_gap2(sub, nb) <= kBoxCut;  // ligand subtile bi x receptor subtile bj
                unsigned long long smask = __ballot(sub_near);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA has landed
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (COUNT) tested += (uint32_t)__popcll(smask);

                // ---- 4. surviving blocks, two per trip ---------------------------------------------
                // bit k of smask = (ligand subtile k >> 3, receptor subtile k & 7).  The lower half
                // of the wave takes the first block left, the upper half the second; an odd block
                // left over is paired with the far-away records behind the tile (all misses).
                while (smask) {
                    // bit k of smask = block (ligand subtile k >> 3, receptor subtile k & 7); 64 = the far-away
                    // ligand subtile x receptor subtile 0.  Each half of the wave picks its block out of the
                    // scalar pair with one 64-bit shift.
                    const int k0 = __ffsll(smask) - 1;
                    asm("s_bitset0_b64 %0, %1" : "+s"(smask) : "s"(k0));
                    const int k1 = smask ? __ffsll(smask) - 1 : 64;
                    asm("s_bitset0_b64 %0, %1" : "+s"(smask) : "s"(k1));  // bit 64 = bit 0, which is clear by now
                    const uint32_t kk = (uint32_t)((((unsigned long long)(uint32_t)k1 << 32) | (uint32_t)k0) >> half_shift);
                    const uint32_t lsub = kk >> 3, rsub = kk & 7u;
                    const LigRecord *lp = ligt + lsub * 8 + pi;
                    const PackedRecPair *rp = rect + rsub * 4 + pq;
                    const v4f Lv = *reinterpret_cast<const v4f *>(lp);
                    const v4f Ra = *reinterpret_cast<const v4f *>(rp);
                    const v4f Rb = *reinterpret_cast<const v4f *>(reinterpret_cast<const unsigned char *>(rp) + 16);
                    const uint32_t Lt = __float_as_uint(Lv.w);
                    const v2f Lxy = {Lv.x, Lv.y}, Lzt = {Lv.z, Lv.w};
                    const v2f dx = pk_sub_lo(v2f{Ra.x, Ra.y}, Lxy), dy = pk_sub_hi(v2f{Ra.z, Ra.w}, Lxy), dz = pk_sub_lo(v2f{Rb.x, Rb.y}, Lzt);
                    // D' = SC (4 d2) + 1/2 for both receptor atoms, clamped into the LUT
                    v2f Dp = __builtin_elementwise_fma(dz, dz, v2f{0.5f, 0.5f});
                    Dp = __builtin_elementwise_fma(dy, dy, Dp);
                    Dp = __builtin_elementwise_fma(dx, dx, Dp);
                    const uint32_t c0 = cvt_u32_sat(fminf(Dp.x, kCellMax)), c1 = cvt_u32_sat(fminf(Dp.y, kCellMax));  // gfx950 has no packed min
                    const uint32_t w0 = s_lut[c0], w1 = s_lut[c1];
                    uint32_t off0 = Lt + __float_as_uint(Rb.z) + w0;  // src/dfire.rs:338, re-laid out
                    uint32_t off1 = Lt + __float_as_uint(Rb.w) + w1;
                    // flagged cell or an atom outside the f32 frame: bit 30 (or 31) of the sum
                    // (branches on ballots: every lane goes along, `queued` stays wave-uniform)
                    if (__builtin_expect(__ballot((off0 > off1 ? off0 : off1) >= kPackedSlow) != 0ull, 0)) {
                        // Lean cells first, both pairs, no branches: at most one step, in the middle of the
                        // cell (D' = cell + 1/2), the f32 distance further than eps from it, and no
                        // interface flag to set (the cell lies above the interface distance, or neither
                        // atom is tracked).  Bit 31 of the sum is clear exactly when the word is the only
                        // flagged addend.
                        bool need0, need1;
                        {
                            // position inside the cell relative to its middle (cell = floor(D') below the clamp)
                            const float d0 = __builtin_amdgcn_fractf(Dp.x) - 0.5f, d1 = __builtin_amdgcn_fractf(Dp.y) - 0.5f;
                            const float epsc = T.eps * SC;
                            const uint32_t tag0 = w0 & 0xff000000u, tag1 = w1 & 0xff000000u;
                            constexpr uint32_t lean_plain = kPackedSlow | kPackedCodeLean << 24, lean_flags = lean_plain | kPackedCodeFlags << 24;
                            const bool clear0 = (int)off0 >= 0 && fabsf(d0) > epsc, clear1 = (int)off1 >= 0 && fabsf(d1) > epsc;
                            bool lean0 = tag0 == lean_plain && clear0, lean1 = tag1 == lean_plain && clear1;
                            // cells below the interface distance (clashing atoms): lean too, unless one of the two atoms
                            // has an interface-flag slot
                            if (__ballot(tag0 == lean_flags || tag1 == lean_flags) != 0ull) {
                                const uint32_t la_bit = lsub * 8u + (uint32_t)pi, ra_bit = rsub * 8u + 2u * (uint32_t)pq;
                                const bool lt = (lig_tracked >> (la_bit & 63u)) & 1ull;   // (the far-away partner of an odd block has lsub = 8: only misses)
                                const bool t0 = lt || ((rec_tracked >> (ra_bit & 63u)) & 1ull), t1 = lt || ((rec_tracked >> ((ra_bit + 1) & 63u)) & 1ull);
                                lean0 = lean0 || (tag0 == lean_flags && !t0 && clear0);
                                lean1 = lean1 || (tag1 == lean_flags && !t1 && clear1);
                            }
                            // word = flags | growth << 12 | term below the step
                            const uint32_t f0 = off0 - (w0 & 0xfffff000u) + (d0 < 0.f ? 0u : (w0 >> 12) & 0xfffu);
                            const uint32_t f1 = off1 - (w1 & 0xfffff000u) + (d1 < 0.f ? 0u : (w1 >> 12) & 0xfffu);
                            need0 = off0 >= kPackedSlow && !lean0;
                            need1 = off1 >= kPackedSlow && !lean1;
                            off0 = lean0 ? f0 : off0;
                            off1 = lean1 ? f1 : off1;
                        }
                        if (__builtin_expect(__ballot(need0 || need1) != 0ull, 0)) {
                            // queue them for the exact path (after the loops) and read misses for now
                            const bool real_block = lsub < 8u;  // not the far-away partner of an odd block
                            need0 = need0 && real_block;
                            need1 = need1 && real_block;
                            const unsigned long long m0 = __ballot(need0), m1 = __ballot(need1);
                            const uint32_t n0 = (uint32_t)__popcll(m0);
                            const uint32_t i0 = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                            const uint32_t i1 = queued + n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                            const uint32_t item = (lsub * 8u + (uint32_t)pi) | (uint32_t)(RT * 64 + (int)(rsub * 8u) + 2 * pq) << 6;
                            if (need0 && i0 < kPackedQueue) s_queue[wave][i0] = item;
                            if (need1 && i1 < kPackedQueue) s_queue[wave][i1] = item + 64u;
                            queued += n0 + (uint32_t)__popcll(m1);
                            off0 = need0 ? kPackedMiss : off0;
                            off1 = need1 ? kPackedMiss : off1;
                        }
                    }
                    // retire the previous trip's gathers only now (their L2 latency hides behind
                    // this trip's LDS reads and arithmetic), then issue this trip's
                    acc += pend0;
                    acc += pend1;
                    asm volatile("" : "+v"(acc) : : "memory");
                    pend0 = table_entry(table, off0);
                    pend1 = table_entry(table, off1);
                    if (COUNT) cnt += (off0 < kPackedMiss && !beyond_cutoff_slot(off0) ? 1u : 0u) + (off1 < kPackedMiss && !beyond_cutoff_slot(off1) ? 1u : 0u);
                }
        });

        // ---- 4b. the pairs the f32 test could not decide, in f64 --------------------------------------
        ExactCtx ex;
        ex.rx = T.rec.x;
        ex.ry = T.rec.y;
        ex.rz = T.rec.z;
        ex.modes = T.rec.modes;
        ex.rec_nm = row + 7;
        ex.pad = (size_t)T.rec.n_tiles * 64;
        ex.num_anm = T.rec.num_anm;
        ex.rec_tindex = T.rec.tindex;
        ex.rec_slot = T.rec.slot;
        ex.lig_slot = T.lig.slot;
        ex.step4 = s_step4;
        ex.table = T.table;
        ex.iface_scaled = T.iface_scaled;
        ex.pose_flags = T.flags + pose * (size_t)(T.rec.flag_words + T.lig.flag_words);
        ex.rec_flag_words = T.rec.flag_words;
        if (__builtin_expect(queued > (uint32_t)kPackedQueue, 0)) {
            // Overflow pass (atoms outside the f32 frame, i.e. absurd poses): everything of this wave
            // again, every pair of the surviving tiles in f64.  Replaces what the pair loop summed.
            acc = 0.0;
            pend0 = 0.0;
            pend1 = 0.0;
            cnt = 0;
            const Vec3 p = pose_ligand_atom(T.lig, T.use_anm, T.anm_rec, row, la);
            const uint32_t lig_term = T.lig.tindex[la];
            for_each_tile([&](const int RT, const unsigned long long) {
                if (!valid) return;
                for (int r = 0; r < 64; r++) {
                    const int ra = RT * 64 + r;
                    if (ra < T.rec.n_real) acc += exact_pair(ex, p, lig_term, la, ra, cnt);
                }
            });
            if (COUNT) n_exact = 0xffffffffu / 64u;  // marks the pass in the diagnostics
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the queue as every lane wrote it
            for (uint32_t i = (uint32_t)lane; i < queued; i += 64) {
                const uint32_t item = s_queue[wave][i];
                const int qa = LT * 64 + (int)(item & 63u), ra = (int)(item >> 6);
                const Vec3 p = pose_ligand_atom(T.lig, T.use_anm, T.anm_rec, row, qa);
                acc += exact_pair(ex, p, T.lig.tindex[qa] & ~kPackedSlow, qa, ra, cnt);
                if (COUNT) n_exact++;
            }
        }
    }

    // ---- 5. reduction ----------------------------------------------------------------------------
    acc += pend0;
    acc += pend1;
    acc = wave_sum(acc);
    if (COUNT) {
        cnt = wave_sum_u32(cnt);
        n_exact = wave_sum_u32(n_exact);
    }
    if (lane == 0) s_res[wave] = WaveResult{acc, cnt, tested, n_exact, 0};
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        uint32_t c = 0, t = 0, e = 0;
        for (int w = 0; w < kPackedWaves; w++) {
            s += s_res[w].sum;
            c += s_res[w].count;
            t += s_res[w].tested;
            e += s_res[w].exact;
        }
        const size_t slot = pose * (size_t)T.n_groups + group;
        T.partial[2 * slot] = s;
        T.partial[2 * slot + 1] = 0.0;
        if (COUNT) {
            T.count_partial[slot] = c;
            if (T.tested_partial) T.tested_partial[slot] = t;
            if (T.exact_partial) T.exact_partial[slot] = e;
        }
    }
}

}  // namespace

size_t packed_kernel_lds_bytes(int cells_per_unit) {
    return (size_t)kPackedLutCells * cells_per_unit * sizeof(uint32_t) + kDfireSteps * sizeof(double) +
           (size_t)kPackedWaves * (72 * sizeof(LigRecord) + 32 * sizeof(PackedRecPair) + kPackedQueue * 4) + kPackedWaves * 24;
}

hipError_t launch_dfire_packed(const PackedLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    const size_t blocks = t.n_poses * (size_t)t.n_groups;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(kPackedWaves * 64);
    if (t.cells_per_unit == 2) {
        if (t.count_partial != nullptr) hipLaunchKernelGGL((dfire_packed_pairs<true, 2>), grid, block, 0, stream, t);
        else hipLaunchKernelGGL((dfire_packed_pairs<false, 2>), grid, block, 0, stream, t);
    } else {
        if (t.count_partial != nullptr) hipLaunchKernelGGL((dfire_packed_pairs<true, 1>), grid, block, 0, stream, t);
        else hipLaunchKernelGGL((dfire_packed_pairs<false, 1>), grid, block, 0, stream, t);
    }
    return hipGetLastError();
}

hipError_t launch_packed_prepare(const PackedPrepareLaunch &p, hipStream_t stream) {
    if (p.n_poses == 0 || p.n_tiles == 0) return hipSuccess;
    const size_t blocks = ((p.n_poses + kPreparePoses - 1) / kPreparePoses) * (size_t)p.n_tiles;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dfire_packed_prepare, dim3((unsigned)blocks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ld
