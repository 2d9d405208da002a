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

#include "dfire_device.hpp"

namespace ld {

namespace {

// The launch arguments stay where the dispatch put them, in the kernarg segment (constant address space): every field
// is a scalar load at its point of use, nothing is copied to registers up front or to scratch when a non-inlined function
// wants the whole block.
typedef const __attribute__((address_space(4))) BmLaunch BmArgs;
#define LD_BM_ARGS ((BmArgs *)__builtin_amdgcn_kernarg_segment_ptr())

#ifndef LD_BM_CULL_WAVES
#define LD_BM_CULL_WAVES 4
#endif
constexpr int kBmCullWaves = LD_BM_CULL_WAVES;   // independent waves per dfire_bm_cull workgroup
#ifndef LD_BM_CULL_POSES
#define LD_BM_CULL_POSES 8
#endif
constexpr int kBmCullPoses = LD_BM_CULL_POSES;   // poses a wave of dfire_bm_cull walks with its ligand tile
constexpr int kBmCullQueues = kBmCullQueueWords;   // (1k4c: 8 queues 695 us, 16 581, 32 388, 64 320, 128 ~310, 256 296, 512 316; static 337)   // counters the waves of dfire_bm_cull draw their items from
constexpr int kBmCullHitTiles = 4;              // hit list of a dfire_bm_cull wave: room for this many poses that reach every receptor tile
__host__ __device__ inline int bm_cull_hit_cap(int n_rt) { return kBmCullHitTiles * n_rt > 192 ? kBmCullHitTiles * n_rt : 192; }   // (a flush costs one atomic per tile pair)
__host__ __device__ inline size_t bm_cull_wave_lds(int n_rt) { return ((size_t)bm_cull_hit_cap(n_rt) * 12 + (size_t)n_rt * 8 + 15) / 16 * 16; }
// dfire_bm_gather: threads per pose = span (a power of two covering the ligand's tiles) x chunks (up to 64 threads per pose)
__host__ __device__ inline int bm_gather_span(int n_lt) {
    int span = 1;
    while (span < n_lt && span < 512) span <<= 1;
    return span;
}
__host__ __device__ inline int bm_gather_chunks(int n_lt, int n_rt) {
    int chunks = 1;
    while (chunks * 4 < n_rt && bm_gather_span(n_lt) * chunks * 2 <= 64) chunks <<= 1;
    return chunks;
}
constexpr float kBmBoxCut = 14400.0f * 1.00005f;  // (8 * 15 A)^2 in record units, padded for the rounding of the box test

// listed row -> pose row, or -1 beyond the list of this launch / inactive
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
    const size_t rows = bm_rows(T);   // (a launch sized for every glowworm of a GSO costs what the glowworms that moved cost)
    for (size_t listed = (size_t)blockIdx.x * 256 + threadIdx.x; listed < rows; listed += (size_t)gridDim.x * 256) {
        const long long p = bm_pose_of(T, listed);
        if (p < 0) continue;
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
        {   // the pose's interface-flag words (the exact path sets bits, pose_energy_finish reads them)
            const int words = T->m.rec_flag_words + T->m.lig.flag_words;
            uint32_t *f = T->flags + pose * (size_t)words;
            for (int k = 0; k < words; k++) f[k] = 0u;
        }
        if (T->exact_fix) T->exact_fix[pose] = 0;
        if (T->exact_count) T->exact_count[pose] = 0;
        if (T->exact_pairs) T->exact_pairs[pose] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_cull: wave = (ligand tile, kBmCullPoses consecutive poses of the launch).  Phase 1, pose by pose: the tile's
// atoms posed in f32, boxes, the 64 x 64 and 8 x 8 box tests; the block masks go to LDS.  Phase 2, lane = receptor
// tile: ONE atomic per tile pair for all the poses of the wave (the lists of a small complex have few heads: one
// returning atomic per pose and tile pair serialises on them), then the entries.
// ---------------------------------------------------------------------------------------------
// bit (a * kBmHalves + h): the mask holds a block of ligand subtile a in the h-th part of its row
__device__ __forceinline__ uint32_t bm_rows_of(unsigned long long mask) {
    uint32_t rows = 0;
#pragma unroll
    for (int k = 0; k < 8 * kBmHalves; k++) rows |= ((mask >> (k * (8 / kBmHalves))) & ((1ull << (8 / kBmHalves)) - 1ull)) ? 1u << k : 0u;
    return rows;
}

template <bool COUNT>
__global__ __launch_bounds__(kBmCullWaves * 64) void dfire_bm_cull(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    // LDS: the receptor's subtile and tile boxes (read by every item; a global load per surviving tile was most of an
    // item's time), then [wave][pose of the wave][receptor tile]: block mask, 0 = not within reach
    extern __shared__ unsigned long long s_cull[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int n_lt = T->m.lig.n_tiles, n_rt = T->m.rec_n_tiles;
    TiledBox *s_sub = reinterpret_cast<TiledBox *>(s_cull);            // [n_rt * 8]
    TiledBox *s_tile = s_sub + (size_t)n_rt * 8;                        // [n_rt]
    // per wave: the HITS (pose of the wave, receptor tile, block mask) of the item so far, room for kBmCullHitTiles * n_rt of
    // them (a pose adds at most n_rt; the list is flushed when the next pose might not fit), and per receptor tile a counter
    // and the first entry of the wave in that tile pair's list
    const int hit_cap = bm_cull_hit_cap(n_rt);
    unsigned char *s_wave = reinterpret_cast<unsigned char *>(s_tile + n_rt) + (size_t)wave * bm_cull_wave_lds(n_rt);
    unsigned long long *s_hmask = reinterpret_cast<unsigned long long *>(s_wave);          // [hit_cap]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_hmask + hit_cap);                      // [n_rt]
    uint32_t *s_base = s_cnt + n_rt;                                                        // [n_rt]
    unsigned short *s_hkey = reinterpret_cast<unsigned short *>(s_base + n_rt);             // [hit_cap]: pose of the wave << 8 | receptor tile
    unsigned short *s_hrank = s_hkey + hit_cap;                                             // [hit_cap]: place among the wave's hits of that tile pair
    {
        static_assert(sizeof(TiledBox) == 32, "two 16-byte pieces");
        const uint4 *src_sub = reinterpret_cast<const uint4 *>(T->m.rec_sub), *src_tile = reinterpret_cast<const uint4 *>(T->m.rec_tile);
        for (int k = threadIdx.x; k < n_rt * 16; k += kBmCullWaves * 64) reinterpret_cast<uint4 *>(s_sub)[k] = src_sub[k];
        for (int k = threadIdx.x; k < n_rt * 2; k += kBmCullWaves * 64) reinterpret_cast<uint4 *>(s_tile)[k] = src_tile[k];
        __syncthreads();
    }
    const size_t rows = bm_rows(T);
    const size_t n_items = (rows + kBmCullPoses - 1) / kBmCullPoses * (size_t)n_lt;
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
    for (;;) {
    const size_t item = queue_first + (uint32_t)__builtin_amdgcn_readfirstlane((int)next_ticket);
    if (item >= queue_end) break;
    const size_t group = item / (unsigned)n_lt;
    const int lt = (int)(item % (unsigned)n_lt);
    const size_t listed0 = group * kBmCullPoses;

    const int la = lt * 64 + lane;
    const float4 loc = reinterpret_cast<const float4 *>(T->m.lig_local)[la];
    const bool valid = loc.w != 0.f;
    const float4 sphere = reinterpret_cast<const float4 *>(T->m.lig_tile_sphere)[lt];
    // this lane's receptor tile box (the first 64 tiles; larger receptors read the rest per pose)
    TiledBox my_tile = TiledBox{INFINITY, INFINITY, INFINITY, 0.f, -INFINITY, -INFINITY, -INFINITY, 0.f};
    if (lane < n_rt) my_tile = s_tile[lane];

    // the item's poses and their affine maps: lane g loads those of pose g, all in flight together
    long long my_pose = -1;
    float4 my_a0 = float4{0.f, 0.f, 0.f, 0.f}, my_a1 = my_a0, my_a2 = my_a0;
    if (lane < kBmCullPoses && listed0 + lane < rows) my_pose = bm_pose_of(T, listed0 + lane);
    if (my_pose >= 0) {
        const float4 *ap = reinterpret_cast<const float4 *>(T->rt + (size_t)my_pose * 12);
        my_a0 = ap[0];
        my_a1 = ap[1];
        my_a2 = ap[2];
    }
    auto pose_lane = [](float v, int g) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), g)); };
    auto affine_of = [&](int g) {
        return Affine{pose_lane(my_a0.x, g), pose_lane(my_a0.y, g), pose_lane(my_a0.z, g), pose_lane(my_a0.w, g),
                      pose_lane(my_a1.x, g), pose_lane(my_a1.y, g), pose_lane(my_a1.z, g), pose_lane(my_a1.w, g),
                      pose_lane(my_a2.x, g), pose_lane(my_a2.y, g), pose_lane(my_a2.z, g), pose_lane(my_a2.w, g)};
    };

    uint32_t n_hits = 0;                 // wave-uniform: hits listed and not flushed yet
    uint32_t my_first = 0, my_nvis = 0;  // lane g: where pose g's hits start in the list; how many it has
    // ---- the list -> entries.  One LDS atomic per hit (its place among the wave's hits of the tile pair), ONE global atomic per
    // tile pair for the whole wave (the lists of a small complex have few heads: one returning atomic per pose and tile pair
    // serialises on them), then lane = hit writes the entry.
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = lane; k < n_rt; k += 64) s_cnt[k] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t h = (uint32_t)lane; h < n_hits; h += 64) s_hrank[h] = (unsigned short)atomicAdd(&s_cnt[s_hkey[h] & 0xffu], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = lane; k < n_rt; k += 64) {
            const uint32_t total = s_cnt[k];
            if (total) s_base[k] = atomicAdd(&T->tp_count[(size_t)lt * n_rt + k], total);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t h0 = 0; h0 < n_hits; h0 += 64) {
            const uint32_t h = h0 + (uint32_t)lane;
            const bool act = h < n_hits;
            const uint32_t key = act ? s_hkey[h] : 0u;
            const int g = (int)(key >> 8), RT = (int)(key & 0xffu);
            // what lane g holds about pose g (every lane takes part in the shuffles)
            const uint32_t pose = (uint32_t)__shfl((int)(uint32_t)my_pose, g, 64);
            const uint32_t first = (uint32_t)__shfl((int)my_first, g, 64);
            const float4 a0 = float4{__shfl(my_a0.x, g, 64), __shfl(my_a0.y, g, 64), __shfl(my_a0.z, g, 64), __shfl(my_a0.w, g, 64)};
            const float4 a1 = float4{__shfl(my_a1.x, g, 64), __shfl(my_a1.y, g, 64), __shfl(my_a1.z, g, 64), __shfl(my_a1.w, g, 64)};
            const float4 a2 = float4{__shfl(my_a2.x, g, 64), __shfl(my_a2.y, g, 64), __shfl(my_a2.z, g, 64), __shfl(my_a2.w, g, 64)};
            if (act) {
                const unsigned long long mask = s_hmask[h];
                const uint32_t idx = s_base[RT] + s_hrank[h];
                const size_t at = ((size_t)lt * n_rt + RT) * T->cap + idx;
                T->ent_pose[at] = pose;
                T->ent_mask[at] = mask;
                float4 *ap = reinterpret_cast<float4 *>(T->ent_rt) + at * 3;   // (what a pair batch poses the entry with)
                ap[0] = a0;
                ap[1] = a1;
                ap[2] = a2;
                T->vis_entry[((size_t)pose * n_lt + lt) * (size_t)n_rt + (h - first)] =
                    (unsigned long long)RT << 48 | (unsigned long long)bm_rows_of(mask) << 32 | idx;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the list is read before it is written again
        n_hits = 0;
    };

    long long pose_of[kBmCullPoses];   // wave-uniform
#pragma unroll
    for (int g = 0; g < kBmCullPoses; g++) {
        pose_of[g] = (long long)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_pose, g) |
                                 (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)my_pose >> 32), g) << 32);
        if (pose_of[g] < 0) continue;
        const size_t pose = (size_t)pose_of[g];
        const Affine A = affine_of(g);
        if (n_hits + (uint32_t)n_rt > (uint32_t)hit_cap) flush();
        const uint32_t first_hit = n_hits;
        if (lane == g) my_first = first_hit;
        // the hits of one ballot of receptor tiles, lane-distributed, written to the list together (one LDS store per lane
        // instead of a scalar branch, five register moves and two stores per hit)
        uint32_t held = 0;                  // wave-uniform
        unsigned long long held_mask = 0ull;
        uint32_t held_key = 0;
        auto put_held = [&]() {
            if ((uint32_t)lane < held) {
                s_hmask[n_hits + (uint32_t)lane] = held_mask;
                s_hkey[n_hits + (uint32_t)lane] = (unsigned short)held_key;
            }
            n_hits += held;
            held = 0;
        };
        {   // A tile whose bounding sphere stays beyond the cutoff of every receptor tile's box has nothing to list (most tiles
            // of a large ligand, in most poses): one point posed and one test per receptor tile instead of 64 atoms posed,
            // their boxes and the box tests.
            float sx, sy, sz;
            bm_apply(A, sphere.x, sphere.y, sphere.z, sx, sy, sz);
            const float reach = 120.0f * 1.0001f + sphere.w + pad;   // (8 * 15 A, the sphere's radius, the affine map's error)
            bool any_near = false;
            for (int base = 0; base < n_rt && !any_near; base += 64) {
                const TiledBox tb = base == 0 ? my_tile : (base + lane < n_rt ? s_tile[base + lane] : my_tile);
                const float gx = fmaxf(0.f, fmaxf(tb.lox - sx, sx - tb.hix));
                const float gy = fmaxf(0.f, fmaxf(tb.loy - sy, sy - tb.hiy));
                const float gz = fmaxf(0.f, fmaxf(tb.loz - sz, sz - tb.hiz));
                any_near = __ballot(base + lane < n_rt && gx * gx + gy * gy + gz * gz <= reach * reach) != 0ull;
            }
            if (!any_near) {
                if (COUNT && lane == 0) T->tile_tested[pose * (size_t)n_lt + lt] = 0;
                continue;
            }
        }
        float fx, fy, fz;
        bm_apply(A, loc.x, loc.y, loc.z, fx, fy, fz);
        const bool inside = fabsf(fx) <= ubound && fabsf(fy) <= ubound && fabsf(fz) <= ubound;

        // An atom outside the frame is more than the cutoff away from every receptor atom (the frame holds the receptor's
        // box + 16 A): it joins no box; the pairs it still meets inside blocks of its subtile read "miss", as they must.
        BoxRegs sub = lane_box(valid && inside, fx, fy, fz);
        box_reduce8(sub);
        {   // widen: the f32 positions are within box_pad of the exactly posed ones
            box_widen(sub);
            sub.lox -= pad; sub.loy -= pad; sub.loz -= pad;
            sub.hix += pad; sub.hiy += pad; sub.hiz += pad;
        }
        BoxRegs whole = sub;   // (widening is monotone: the union of the widened subtile boxes IS the widened tile box)
        box_reduce64_from8(whole);
        whole.lox = lane63_f32(whole.lox); whole.loy = lane63_f32(whole.loy); whole.loz = lane63_f32(whole.loz);
        whole.hix = lane63_f32(whole.hix); whole.hiy = lane63_f32(whole.hiy); whole.hiz = lane63_f32(whole.hiz);

        // 64 x 64 tile boxes, 64 receptor tiles per ballot; then the 8 x 8 subtile boxes of every surviving tile
        uint32_t tested = 0;
        for (int base = 0; base < n_rt; base += 64) {
            bool tile_near = false;
            if (base == 0) tile_near = lane < n_rt && box_gap2(whole, my_tile) <= kBmBoxCut;
            else if (base + lane < n_rt) tile_near = box_gap2(whole, s_tile[base + lane]) <= kBmBoxCut;
            unsigned long long rtmask = __ballot(tile_near);
            while (rtmask) {
                // four surviving tiles at a time: their subtile boxes are loaded together
                int RTs[4], nk = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    RTs[k] = rtmask ? base + __ffsll(rtmask) - 1 : RTs[0];
                    if (rtmask) nk++;
                    rtmask &= rtmask - 1;
                }
                TiledBox nb[4];
#pragma unroll
                for (int k = 0; k < 4; k++) nb[k] = s_sub[RTs[k] * 8 + bj];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k >= nk) break;
                    const unsigned long long smask = __ballot(box_gap2(sub, nb[k]) <= kBmBoxCut);  // bit = ligand subtile (lane >> 3) * 8 + receptor subtile
                    if (smask) {   // hit `held` of this ballot stays in lane `held` until the ballot's tiles are done
                        if (lane == (int)held) {
                            held_mask = smask;
                            held_key = (uint32_t)(g << 8 | RTs[k]);
                        }
                        held++;
                    }
                    if (COUNT) tested += (uint32_t)__popcll(smask);
                }
            }
            put_held();   // (a ballot's 64 receptor tiles make at most 64 hits)
        }
        if (COUNT && lane == 0) T->tile_tested[pose * (size_t)n_lt + lt] = tested;
        if (lane == g) my_nvis = n_hits - first_hit;
    }
    next_ticket = draw();   // (here, not at the item's start: memory operations return in order, and the item's loads would wait for it)
    flush();
    if (my_pose >= 0) T->vis_count[(size_t)my_pose * n_lt + lt] = my_nvis;
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_plan: every tile pair's entries cut into parts of P entries; a JOB = (tile pair, part, partial-sum row) is
// what one wave of dfire_bm_pairs walks.  P = kBmPartEntries for a large launch (the longer a job, the more batches
// share each staging of a block's table rows); a launch with few entries -- the late steps of a GSO run, when few
// glowworms still move -- is cut finer so that its jobs still spread over every wave of the chip.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dfire_bm_plan(const BmLaunch launch_arguments) {
    // One workgroup.  Jobs are listed longest first (by entries, in 16 classes): the persistent waves of
    // dfire_bm_pairs draw them in that order, so the launch ends on its shortest jobs.
    BmArgs *T = LD_BM_ARGS;
    __shared__ uint32_t s_class[18];   // [c]: parts of more than (c - 1) P / 16 and at most c P / 16 entries; then cursors
    __shared__ uint32_t s_total;
    const int tid = threadIdx.x;
    const uint32_t n_tp = (uint32_t)(T->m.lig.n_tiles * T->m.rec_n_tiles);
    if (tid < 18) s_class[tid] = 0;
    if (tid == 0) s_total = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t tp = tid; tp < n_tp; tp += 1024) mine += T->tp_count[tp];
    if (mine) atomicAdd(&s_total, mine);
    __syncthreads();
    // about one (tile pair, part) pair per wave of the pair kernel, i.e. kBmJobRows jobs per wave
    const uint32_t waves = (uint32_t)(T->pairs_groups > 0 ? T->pairs_groups : 256) * kBmWaves;
#ifndef LD_BM_P_FACTOR
#define LD_BM_P_FACTOR 4
#endif
    uint32_t P = (s_total * LD_BM_P_FACTOR / waves + 63u) / 64u * 64u;
    P = P < 64u ? 64u : P > (uint32_t)kBmPartEntries ? (uint32_t)kBmPartEntries : P;
    const uint32_t step = P / 16u;
    for (uint32_t tp = tid; tp < n_tp; tp += 1024) {
        const uint32_t n = T->tp_count[tp];
        if (n == 0) continue;
        const uint32_t full = n / P, rest = n % P;
        if (full) atomicAdd(&s_class[16], full);
        if (rest) atomicAdd(&s_class[(rest + step - 1) / step], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t at = 0;
        for (int c = 16; c >= 1; c--) {
            const uint32_t k = s_class[c];
            s_class[c] = at;
            at += k;
        }
        T->job_count[0] = at;
        T->job_count[2] = P;
    }
    __syncthreads();
    for (uint32_t tp = tid; tp < n_tp; tp += 1024) {
        const uint32_t n = T->tp_count[tp];
        if (n == 0) continue;
        const uint32_t full = n / P, rest = n % P;
        if (full) {
            const uint32_t at = atomicAdd(&s_class[16], full);
            for (uint32_t k = 0; k < full; k++) {
                T->jobs[2 * (at + k)] = tp;
                T->jobs[2 * (at + k) + 1] = k * P;
            }
        }
        if (rest) {
            const uint32_t at = atomicAdd(&s_class[(rest + step - 1) / step], 1u);
            T->jobs[2 * at] = tp;
            T->jobs[2 * at + 1] = full * P;
        }
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
    static_assert(kBmJobRows == 8 && kBmHalves == 1 && kBmSplit == 1, "a job row is one byte of the block mask");
    BmArgs *T = LD_BM_ARGS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_pairs = T->job_count[0], P = T->job_count[2];
    for (uint32_t jd = blockIdx.x * kBmOrderWaves + wave; jd < n_pairs; jd += gridDim.x * kBmOrderWaves) {
        const size_t tp = T->jobs[2 * jd];
        const uint32_t lo = T->jobs[2 * jd + 1];
        const uint32_t n = T->tp_count[tp];
        const uint32_t hi = n < lo + P ? n : lo + P;
        unsigned long long m[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t e = lo + (uint32_t)k * 64 + lane;
            m[k] = e < hi ? T->ent_mask[tp * T->cap + e] : 0ull;
        }
        uint32_t items[8], present_lo = 0, present_hi = 0;   // per row; the OR of the masks
#pragma unroll
        for (int r = 0; r < 8; r++) items[r] = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
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
// dfire_bm_pairs: persistent workgroups (one per CU) of 8 independent waves; a wave draws jobs
// (tile pair, part of its entries, ligand subtile a) and walks the job's 8 blocks (a, b), each with its 64 table
// rows staged in the wave's own slice of LDS.  The waves share the cell LUT and nothing else: no barrier after set-up.
// ---------------------------------------------------------------------------------------------
struct BmWaveShared {
    unsigned char cube[kBmCubeBytes];
    unsigned char row_bits[kBmPartEntries];        // per entry of the job: which of the 8 blocks (a, .) it holds
    unsigned short items[kBmPartEntries + 64];     // the entries that hold the current block (| 0x8000: its first block of the row)
    uint32_t queue[kBmQueue];                      // pairs for the exact path
    float4 lig_local[kBmLig];                      // the job's ligand atoms: local coordinates, w = 1 for a real atom
};
struct BmShared {
    unsigned char lut[kBmLutBytes];   // indexed from the far end: cell' = floor(kBmCellZero + 1/2 - 64 d2), everything further reads cell' 0
    BmWaveShared w[kBmWaves];
};

// what a wave needs to evaluate queued pairs exactly
struct BmWaveCtx {
    size_t tp;       // tile pair
    int ls;          // ligand subtile (global)
    int RT;          // receptor tile
    size_t lo;       // first entry of the job
};

// Queue item: entry (local to the job) | (i * 8 + j) << 10 | b << 16 | kBmFlagsOnly (the pair's table value is in the sum already:
// only its interface flags are wanted)
constexpr uint32_t kBmFlagsOnly = 1u << 19;
template <bool COUNT>
__device__ __noinline__ void bm_drain(BmArgs *T, const BmWaveCtx &W, const uint32_t *queue, uint32_t queued, int lane) {
    for (uint32_t k = (uint32_t)lane; k < queued; k += 64) {
        const uint32_t item = queue[k];
        const size_t e = W.lo + (item & 1023u);
        const int i = (int)((item >> 13) & 7u), j = (int)((item >> 10) & 7u), b = (int)((item >> 16) & 7u);
        const int la = W.ls * 8 + i, ra = W.RT * 64 + b * 8 + j;
        if (la >= T->m.lig.n_real || ra >= T->m.rec_n_real) continue;
        const size_t pose = T->ent_pose[W.tp * T->cap + e];
        const ExactCtx ex = bm_exact_ctx(T, pose);
        const Vec3 p = pose_ligand_atom(bm_ligand(T), 0, 0, T->poses + pose * T->stride, la);
        uint32_t cnt = 0;
        const double v = exact_pair(ex, p, T->m.lig.tindex[la], la, ra, cnt);   // (sets the interface flags)
        if (item & kBmFlagsOnly) continue;
        // order-free: 2^-40 fixed point (the one place where a table value is rounded: below the noise of any f64 sum order)
        const long long fix = __double2ll_rn(v * kBmFixScale);
        if (fix != 0) atomicAdd(reinterpret_cast<unsigned long long *>(T->exact_fix + pose), (unsigned long long)fix);
        if (COUNT) {
            if (cnt) atomicAdd(T->exact_count + pose, cnt);
            atomicAdd(T->exact_pairs + pose, 1u);
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(kBmWaves * 64, (kBmWaves + 3) / 4) void dfire_bm_pairs(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    __shared__ __attribute__((aligned(16))) BmShared S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_rt = T->m.rec_n_tiles;
    {
        const uint8_t *lut = COUNT ? T->m.lut_full : T->m.lut;
        for (int i = tid; i < kBmLutBytes / 16; i += kBmWaves * 64) reinterpret_cast<uint4 *>(S.lut)[i] = reinterpret_cast<const uint4 *>(lut)[i];
    }
    BmWaveShared &WS = S.w[wave];
    if (lane < 4) reinterpret_cast<uint32_t *>(WS.cube + kBmCubeRows * kBmRowBytes)[lane] = 0u;   // the zero slot behind the last row
    __syncthreads();
    const unsigned char *cube = WS.cube;
    const uint32_t n_jobs = T->job_count[3], part_entries = T->job_count[2];
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dbg_jobs = 0, dbg_batches = 0, dbg_t_batch = 0, dbg_t_drain = 0, dbg_t_scan = 0, dbg_drains = 0;

    for (;;) {
        const unsigned long long dbg_tj = T->debug ? __builtin_amdgcn_s_memrealtime() : 0ull;
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(T->job_next, 1u);   // (drawing one job ahead was measured: the 2048 claimed jobs lengthen the tail)
        job = (uint32_t)__builtin_amdgcn_readfirstlane((int)job);
        if (job >= n_jobs) break;
        job = T->job_order[job];   // longest first (dfire_bm_order)
        const uint32_t jd = job / (uint32_t)kBmJobRows;
        const int jrow = (int)(job % (uint32_t)kBmJobRows);   // partial-sum row of the entry: (job row of the tile, part of its blocks)
        const int arow = jrow / kBmHalves, b_lo = (jrow % kBmHalves) * (8 / kBmHalves);
        const int a = arow / kBmSplit, la0 = (arow % kBmSplit) * kBmLig;   // ligand subtile a, its atoms la0 .. la0 + kBmLig - 1
        const uint32_t b_mask = ((1u << (8 / kBmHalves)) - 1u) << b_lo;      // the job's blocks (a, b_lo .. b_lo + 8 / kBmHalves - 1)
        const size_t tp = T->jobs[2 * jd];
        const uint32_t lo = T->jobs[2 * jd + 1];
        const uint32_t n = T->tp_count[tp];
        const uint32_t hi = n < lo + part_entries ? n : lo + part_entries;
        const int n_chunks = (int)((hi - lo + 63) / 64);
        const int lt = (int)(tp / (unsigned)n_rt), RT = (int)(tp % (unsigned)n_rt);
        const int ls = lt * 8 + a;
        const BmWaveCtx W{tp, ls, RT, lo};
        const bool lig_tracked = T->m.lig_sub_tracked[ls] != 0;

        // What the job's blocks need that does not depend on the entry, for all 8 receptor subtiles of the tile at once (one
        // latency, together with the masks): lane = (receptor subtile b, atom j).
        const uint32_t roff_all = T->m.rec_rowoff[(size_t)RT * 64 + lane];           // the atom's column in a table row block
        float recf[4];                                                                // the tile's 32 pair records, 256 floats
#pragma unroll
        for (int k = 0; k < 4; k++) recf[k] = reinterpret_cast<const float *>(T->m.rec_pairs + (size_t)RT * 32)[k * 64 + lane];
        const TiledBox my_box = T->m.rec_sub[(size_t)RT * 8 + (lane & 7)];           // lane b (mod 8): subtile b's box
        const uint32_t my_tracked = T->m.rec_sub_tracked[RT * 8 + (lane & 7)];
        uint32_t any_bits = 0;
        {   // the job's block masks: all loads in flight at once
            static_assert(kBmPartEntries == 1024, "16 chunks of 64 entries");
            unsigned long long m[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t e = lo + (uint32_t)k * 64 + lane;
                m[k] = e < hi ? T->ent_mask[tp * T->cap + e] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t bits = (uint32_t)(m[k] >> (8 * a)) & b_mask;
                if (k < n_chunks) WS.row_bits[k * 64 + lane] = (unsigned char)bits;
                any_bits |= bits;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) any_bits |= (uint32_t)__shfl_xor((int)any_bits, off, 64);
        any_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)any_bits);
        if (any_bits == 0) continue;
        dbg_jobs++;

        // the ligand subtile's local coordinates (uniform)
        // (kept in LDS, read back per batch as broadcasts: 24 wave-uniform values in vector registers for the whole job are what
        // pushed the block set-up into scratch)
        if (lane < kBmLig) WS.lig_local[lane] = reinterpret_cast<const float4 *>(T->m.lig_local)[ls * 8 + la0 + lane];
        // table rows of a block -> LDS: kBmCubeRows * 11 pieces of 16 bytes, one LDS-DMA instruction per KiB
        constexpr int kPieces = kBmCubeRows * 11, kDma = (kPieces + 63) / 64;
        uint32_t src_lig[kDma];
#pragma unroll
        for (int t = 0; t < kDma; t++) {
            const int piece = t * 64 + lane, row = piece / 11;
            src_lig[t] = piece < kPieces ? T->m.lig_rowbase[ls * 8 + la0 + (row >> 3)] + (uint32_t)(piece % 11) * 16u : 0u;
        }
        const size_t row_base = (tp * kBmJobRows + (size_t)jrow) * T->cap + lo;
        const size_t ent_base = tp * T->cap + lo;
        uint32_t queued = 0;   // wave-uniform

        if (T->debug) dbg_t_scan += __builtin_amdgcn_s_memrealtime() - dbg_tj;   // job set-up
        for (int b = 0; b < 8; b++) {
            if (!((any_bits >> b) & 1u)) continue;
            const unsigned long long dbg_tblk = T->debug ? __builtin_amdgcn_s_memrealtime() : 0ull;
            {   // stage the block's rows; they land while the entries are scanned
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous block's reads are done
                const unsigned char *rows = reinterpret_cast<const unsigned char *>(T->m.rows);
#pragma unroll
                for (int t = 0; t < kDma; t++) {
                    const int row = (t * 64 + lane) / 11;
                    const uint32_t roff = (uint32_t)__shfl((int)roff_all, b * 8 + (row & 7), 64);
                    if (t * 64 + 63 < kPieces || t * 64 + lane < kPieces)   // (the last KiB may be partial)
                        __builtin_amdgcn_global_load_lds((const global_u32 *)(rows + (src_lig[t] + roff)), (lds_u32 *)(WS.cube + t * 1024), 16, 0, 0);
                }
            }
            // A block with an atom that has an interface-flag slot also queues its pairs closer than 2.5 A (bins 0 and 1,
            // whose slots are the last two of a row) for the exact path, which sets the flags (src/dfire.rs:339-342).
            const uint32_t flag_from = lig_tracked || __builtin_amdgcn_readlane((int)my_tracked, b) != 0 ? kBmNearCode : kBmFlagged;
            constexpr float seed = (float)kBmCellZero + 0.5f;
            // ---- the job's entries that hold block (a, b), in entry order (all 16 chunks' bytes in flight, then the ballots)
            uint32_t n_items = 0;
            {
                uint32_t bits16[16];
#pragma unroll
                for (int k = 0; k < 16; k++) bits16[k] = k < n_chunks ? (uint32_t)WS.row_bits[k * 64 + lane] : 0u;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t bits = bits16[k];
                    const bool act = (bits >> b) & 1u;
                    const unsigned long long m = __ballot(act);
                    if (act) {
                        const uint32_t at = n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        const bool first = (bits & ((1u << b) - 1u)) == 0u;   // the entry's first block of this job (bits hold the job's blocks only): nothing to add to yet
                        WS.items[at] = (unsigned short)((uint32_t)(k * 64 + lane) | (first ? 0x8000u : 0u));
                    }
                    n_items += (uint32_t)__popcll(m);
                }
            }
            // receptor subtile b of the tile: 4 pair records, wave-uniform, out of the registers loaded at the job's start
            // The block's distance arithmetic has its origin at the centre c of the receptor subtile's box:
            //   E = seed - |l - r|^2 = (seed - |r - c|^2) - |l - c|^2 + 2 (r - c) . (l - c),     cell' = (u32)E
            // four packed operations per step instead of six (the differences need not be formed), all operands small
            // enough (below 2^17 for every pair within reach of the cutoff) that the roundings stay inside eps.  The LUT
            // is indexed from the far end: a pair beyond its last cell has E < 0, which v_cvt_u32_f32 turns into cell' 0
            // ("miss") like a NaN -- no clamp.
            auto lane_f32 = [](float v, int from) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), from)); };
            const float cbx = 0.5f * (lane_f32(my_box.lox, b) + lane_f32(my_box.hix, b));
            const float cby = 0.5f * (lane_f32(my_box.loy, b) + lane_f32(my_box.hiy, b));
            const float cbz = 0.5f * (lane_f32(my_box.loz, b) + lane_f32(my_box.hiz, b));
            const float rec_here = (b >> 1) == 0 ? recf[0] : (b >> 1) == 1 ? recf[1] : (b >> 1) == 2 ? recf[2] : recf[3];
            v2f Rx[4], Ry[4], Rz[4], Rs[4];   // 2 (r - c), and seed - |r - c|^2
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int at = (b & 1) * 32 + q * 8;   // record q of the subtile: x0 x1 y0 y1 z0 z1 . .
                const v2f x = v2f{lane_f32(rec_here, at), lane_f32(rec_here, at + 1)} - v2f{cbx, cbx};
                const v2f y = v2f{lane_f32(rec_here, at + 2), lane_f32(rec_here, at + 3)} - v2f{cby, cby};
                const v2f z = v2f{lane_f32(rec_here, at + 4), lane_f32(rec_here, at + 5)} - v2f{cbz, cbz};
                Rs[q] = __builtin_elementwise_fma(-x, x, __builtin_elementwise_fma(-y, y, __builtin_elementwise_fma(-z, z, v2f{seed, seed})));
                Rx[q] = x * v2f{2.f, 2.f};
                Ry[q] = y * v2f{2.f, 2.f};
                Rz[q] = z * v2f{2.f, 2.f};
            }
            // what a lane of a batch needs from memory, loaded one batch ahead
            struct BatchLoads {
                float4 a0, a1, a2;   // the entry's affine map
                double prev;         // the entry's partial of this row so far
                uint32_t prev_cnt;
                uint32_t item;
            };
            auto issue_loads = [&](uint32_t first_item) {
                BatchLoads L;
                const uint32_t at = first_item + (uint32_t)lane;
                L.item = WS.items[at < n_items ? at : first_item];
                const uint32_t el = L.item & 0x7fffu;
                const float4 *ap = reinterpret_cast<const float4 *>(T->ent_rt) + (ent_base + el) * 3;
                L.a0 = ap[0];
                L.a1 = ap[1];
                L.a2 = ap[2];
                L.prev = 0.0;
                L.prev_cnt = 0;
                if (!(L.item & 0x8000u)) {
                    L.prev = T->ent_partial[row_base + el];
                    if (COUNT) L.prev_cnt = T->ent_count[row_base + el];
                }
                return L;
            };
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the item list as every lane wrote it
            BatchLoads next = issue_loads(0);
            if (T->debug) dbg_drains += __builtin_amdgcn_s_memrealtime() - dbg_tblk;   // block set-up
            for (uint32_t done = 0; done < n_items; done += 64) {
                const BatchLoads cur = next;
                dbg_batches++;
                const unsigned long long dbg_tb = T->debug ? __builtin_amdgcn_s_memrealtime() : 0ull;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this batch's loads (and, the first time, the block's rows) are in
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (done + 64 < n_items) next = issue_loads(done + 64);
                const int count = n_items - done >= 64u ? 64 : (int)(n_items - done);
                const bool valid = lane < count;
                const uint32_t el = cur.item & 0x7fffu;
                const Affine A{cur.a0.x, cur.a0.y, cur.a0.z, cur.a0.w - cbx, cur.a1.x, cur.a1.y, cur.a1.z, cur.a1.w - cby, cur.a2.x, cur.a2.y, cur.a2.z, cur.a2.w - cbz};
                v2f lxy[kBmLig], lz2[kBmLig];   // l - c as {x, y} and {z, |l - c|^2}: the packed operations broadcast either half (op_sel)
#pragma unroll
                for (int i = 0; i < kBmLig; i++) {
                    float lx, ly, lz;
                    const float4 L = WS.lig_local[i];
                    bm_apply(A, L.x, L.y, L.z, lx, ly, lz);
                    lxy[i] = v2f{lx, ly};
                    lz2[i] = v2f{lz, __builtin_fmaf(lx, lx, __builtin_fmaf(ly, ly, lz * lz))};
                }
                double acc = 0.0;
                uint32_t cnt = 0;
                // The batch's steps (ligand atom i x the receptor pair record q, 2 atom pairs each) in groups of 8: all cells, all
                // codes, all table values of a group in flight, then the adds in order and one test for flagged cells.
                // Step t of the batch: q = t / kBmLig, i = t % kBmLig.
                constexpr int kGroups = kBmLig / 2;
                auto codes_of_group = [&](int g, uint32_t (&w)[16]) {
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8 += 2) {
                        static_assert(kBmLig == 8, "the two steps of a pair share the receptor record");
                        const int t = g * 8 + s8, q = t / kBmLig, i = t % kBmLig;
                        // Two steps (ligand atoms i, i + 1 against the receptor record q) as ONE block of instructions:
                        //   D = Rs - l2; D = fma(Rz, lz, D); D = fma(Ry, ly, D); D = fma(Rx, lx, D), both halves; cell = (u32)D
                        // Written out because the compiler duplicates the broadcast operands into register pairs (32 moves a
                        // group) instead of using op_sel; as one block because it guards every separate asm statement with an
                        // s_nop (128 per batch); the two dependent chains interleaved.
                        v2f D0, D1;
                        uint32_t c0, c1, c2, c3;
                        asm("v_pk_add_f32 %[d0], %[rs], %[za] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                            "v_pk_add_f32 %[d1], %[rs], %[zb] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                            "v_pk_fma_f32 %[d0], %[rz], %[za], %[d0] op_sel_hi:[1,0,1]\n\t"
                            "v_pk_fma_f32 %[d1], %[rz], %[zb], %[d1] op_sel_hi:[1,0,1]\n\t"
                            "v_pk_fma_f32 %[d0], %[ry], %[xa], %[d0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %[d1], %[ry], %[xb], %[d1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %[d0], %[rx], %[xa], %[d0] op_sel_hi:[1,0,1]\n\t"
                            "v_pk_fma_f32 %[d1], %[rx], %[xb], %[d1] op_sel_hi:[1,0,1]"
                            : [d0] "=&v"(D0), [d1] "=&v"(D1)
                            : [rs] "v"(Rs[q]), [rz] "v"(Rz[q]), [ry] "v"(Ry[q]), [rx] "v"(Rx[q]), [za] "v"(lz2[i]), [xa] "v"(lxy[i]),
                              [zb] "v"(lz2[i + 1]), [xb] "v"(lxy[i + 1]));
                        // (v_cvt_u32_f32 saturates: negative and NaN -> 0)
                        asm("v_cvt_u32_f32 %0, %4\n\tv_cvt_u32_f32 %1, %5\n\tv_cvt_u32_f32 %2, %6\n\tv_cvt_u32_f32 %3, %7"
                            : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(D0.x), "v"(D0.y), "v"(D1.x), "v"(D1.y));
                        w[2 * s8] = S.lut[c0];
                        w[2 * s8 + 1] = S.lut[c1];
                        w[2 * s8 + 2] = S.lut[c2];
                        w[2 * s8 + 3] = S.lut[c3];
                    }
                };
                // (computing the codes of group g + 1 while the table values of group g are on their way was tried: the second
                // wave of the SIMD already fills those waits, and the 16 extra live registers spill the block set-up)
#pragma unroll
                for (int g = 0; g < kGroups; g++) {
                    uint32_t w[16];
                    codes_of_group(g, w);
#pragma unroll
                    for (int k = 0; k < 16; k++) asm("" : "+v"(w[k]));   // 32-bit values from here on (no 16-bit detours on the way to the address)
                    double tv[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const int t = g * 8 + (k >> 1), q = t / kBmLig, i = t % kBmLig;
                        tv[k] = *reinterpret_cast<const double *>(cube + (i * 8 + 2 * q + (k & 1)) * kBmRowBytes + w[k]);
                    }
                    uint32_t wm = 0;
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8++) {
                        acc += tv[2 * s8];
                        acc += tv[2 * s8 + 1];
                        if (COUNT && WS.lig_local[(g * 8 + s8) % kBmLig].w != 0.f)
                            cnt += (w[2 * s8] != 0u && w[2 * s8] < kBmFlagged ? 1u : 0u) + (w[2 * s8 + 1] != 0u && w[2 * s8 + 1] < kBmFlagged ? 1u : 0u);
                        const uint32_t m2 = w[2 * s8] > w[2 * s8 + 1] ? w[2 * s8] : w[2 * s8 + 1];
                        wm = wm > m2 ? wm : m2;
                    }
                    asm volatile("" : "+v"(acc));   // the group's adds end here (the scheduler would park table values in registers)
                    if (__builtin_expect(__ballot(wm >= flag_from) != 0ull, 0)) {
                        // Pairs in flagged cells read 0.0 above; queue them for the exact path.  Most groups of a 1k4c
                        // batch come here for one or two of their 1024 pairs, so the way in is vector-only: every lane
                        // turns its 16 codes into a bit mask (no scalar compare-and-branch per code: 16 of those cost four
                        // times the group's arithmetic), then the few lanes with a bit set push one pair per round.
                        // (two operations per code and no compare: the sign of code - threshold shifted in by v_alignbit_b32)
                        uint32_t fm = 0, fm_only = 0;   // bit k: pair k of the group goes to the exact path / for its flags only
#pragma unroll
                        for (int k = 15; k >= 0; k--) fm = __builtin_amdgcn_alignbit(fm, w[k] - flag_from, 31);   // fm << 1 | (code below the threshold)
                        fm = ~fm & 0xffffu;
                        if (flag_from != kBmFlagged) {   // a block with tracked atoms (wave-uniform, rare)
#pragma unroll
                            for (int k = 15; k >= 0; k--) fm_only = __builtin_amdgcn_alignbit(fm_only, w[k] - kBmFlagged, 31);
                        }
                        if (!valid) fm = 0;
                        unsigned long long live = __ballot(fm != 0u);
                        while (live != 0ull) {
                            if (queued > (uint32_t)kBmQueue - 64u) {   // room for 64 more, always: a pose's sum never depends on its batch
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                bm_drain<COUNT>(T, W, WS.queue, queued, lane);
                                queued = 0;
                            }
                            const uint32_t at = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
                            if (fm != 0u) {
                                const uint32_t k = (uint32_t)__builtin_ctz(fm);
                                const uint32_t t = (uint32_t)g * 8u + (k >> 1), q = t / (uint32_t)kBmLig, i = t % (uint32_t)kBmLig;   // the step, its pair k & 1
                                WS.queue[at] = el | (((uint32_t)la0 + i) * 8u + 2u * q + (k & 1u)) << 10 | (uint32_t)b << 16 |
                                               ((fm_only >> k) & 1u ? kBmFlagsOnly : 0u);
                                fm &= fm - 1u;
                            }
                            queued += (uint32_t)__popcll(live);
                            live = __ballot(fm != 0u);
                        }
                    }
                }
                if (valid) {
                    T->ent_partial[row_base + el] = cur.prev + acc;
                    if (COUNT) T->ent_count[row_base + el] = cur.prev_cnt + cnt;
                }
                if (T->debug) dbg_t_batch += __builtin_amdgcn_s_memrealtime() - dbg_tb;
            }
        }
        if (queued) {
            const unsigned long long td = T->debug ? __builtin_amdgcn_s_memrealtime() : 0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            bm_drain<COUNT>(T, W, WS.queue, queued, lane);
            if (T->debug) dbg_t_drain += __builtin_amdgcn_s_memrealtime() - td;
        }
    }
    if (T->debug != nullptr && lane == 0) {
        unsigned long long *d = T->debug + ((size_t)blockIdx.x * kBmWaves + wave) * 8;
        d[0] = dbg_t0;
        d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = dbg_jobs;
        d[3] = dbg_batches;
        d[4] = dbg_t_batch;
        d[5] = dbg_t_drain;
        d[6] = dbg_drains;
        d[7] = dbg_t_scan;
    }
}

// ---------------------------------------------------------------------------------------------
// dfire_bm_gather: wave = pose; lanes over the ligand tiles, fixed order, then a fixed tree
// ---------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(512) void dfire_bm_gather(const BmLaunch launch_arguments) {
    BmArgs *T = LD_BM_ARGS;
    __shared__ double s_sum[512];
    __shared__ uint32_t s_cnt[512], s_tested[512];
    const int tid = threadIdx.x;
    const int n_lt = T->m.lig.n_tiles, n_rt = T->m.rec_n_tiles;
    // thread = (pose, ligand tile); a workgroup holds 512 / span poses, span = the power of two that covers the ligand's
    // tiles.  The kernel is bound by the latency of three dependent loads per entry: many poses in flight per CU, and
    // per thread the partial sums of four entries x all their rows requested together.
    // A small ligand leaves threads over: `chunks` of them share a (pose, ligand tile), each taking every chunks-th group of
    // four entries (both numbers depend on the molecules only: a pose's sum is the same tree in every launch).
    const int span_lt = bm_gather_span(n_lt), chunks = bm_gather_chunks(n_lt, n_rt), span = span_lt * chunks;
    const int per_wg = 512 / span, sub = tid / span, r0 = tid % span;
    const int lt0 = r0 / chunks, chunk = r0 % chunks;
    const size_t n_rows = bm_rows(T);
    for (size_t first_row = (size_t)blockIdx.x * per_wg; first_row < n_rows; first_row += (size_t)gridDim.x * per_wg) {
    const size_t listed = first_row + sub;
    const long long pp = listed < n_rows ? bm_pose_of(T, listed) : -1;
    const size_t pose = pp < 0 ? 0 : (size_t)pp;
    double s = 0.0;
    uint32_t cnt = 0, tested = 0;
    for (int lt = lt0; pp >= 0 && lt < n_lt; lt += span_lt) {   // the tile's entries in the order the culling listed them, their rows in order
        const size_t slot = pose * (size_t)n_lt + lt;
        if (COUNT && chunk == 0) tested += T->tile_tested[slot];
        const uint32_t n_vis = T->vis_count[slot];
        for (uint32_t v0 = 4u * (uint32_t)chunk; v0 < n_vis; v0 += 4u * (uint32_t)chunks) {
            unsigned long long ent[4];
#pragma unroll
            for (int k = 0; k < 4; k++) ent[k] = v0 + k < n_vis ? T->vis_entry[slot * (size_t)n_rt + v0 + k] : 0ull;
            double part[4][kBmJobRows];
            uint32_t pc[4][kBmJobRows];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const size_t tp = (size_t)lt * n_rt + (size_t)(ent[k] >> 48);
                const size_t at = tp * kBmJobRows * T->cap + (size_t)(ent[k] & 0xffffffffull);
#pragma unroll
                for (int jrow = 0; jrow < kBmJobRows; jrow++) {
                    const int sub_half = jrow / (kBmSplit * kBmHalves) * kBmHalves + jrow % kBmHalves;   // (ligand subtile, part of its blocks)
                    const bool on = (ent[k] >> (32 + sub_half)) & 1ull;
                    part[k][jrow] = on ? T->ent_partial[at + (size_t)jrow * T->cap] : 0.0;
                    pc[k][jrow] = COUNT && on ? T->ent_count[at + (size_t)jrow * T->cap] : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int jrow = 0; jrow < kBmJobRows; jrow++) {
                    s += part[k][jrow];   // (a row without a block of the entry adds 0.0: no bit of the sum changes)
                    cnt += pc[k][jrow];
                }
        }
    }
    s_sum[tid] = s;
    if (COUNT) {
        s_cnt[tid] = cnt;
        s_tested[tid] = tested;
    }
    __syncthreads();
    for (int half = span >> 1; half > 0; half >>= 1) {   // fixed tree, per pose
        if (r0 < half) {
            s_sum[tid] += s_sum[tid + half];
            if (COUNT) {
                s_cnt[tid] += s_cnt[tid + half];
                s_tested[tid] += s_tested[tid + half];
            }
        }
        __syncthreads();
    }
    if (r0 == 0 && pp >= 0) {
        const double total = s_sum[tid] + (double)T->exact_fix[pose] * (1.0 / kBmFixScale);
        T->partial[2 * pose] = total;
        T->partial[2 * pose + 1] = 0.0;
        if (COUNT) {
            T->count_partial[pose] = s_cnt[tid] + T->exact_count[pose];
            if (T->tested_partial) T->tested_partial[pose] = s_tested[tid];
            if (T->exact_partial) T->exact_partial[pose] = T->exact_pairs[pose];
        }
    }
    __syncthreads();   // s_sum is reused by the next rows
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
    if (t.m.rec_n_tiles > 255) return hipErrorInvalidValue;
    const size_t lds = (size_t)t.m.rec_n_tiles * 9 * sizeof(TiledBox) + (size_t)kBmCullWaves * bm_cull_wave_lds(t.m.rec_n_tiles);
    // persistent workgroups (each fills its LDS with the receptor's boxes once): as many as fit the chip at this LDS size
    const size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 512)));
    const size_t cus = t.pairs_groups > 0 ? (size_t)t.pairs_groups : 256;
    const size_t blocks = std::min<size_t>(((t.n_poses + kBmCullPoses - 1) / kBmCullPoses * (size_t)t.m.lig.n_tiles + kBmCullWaves - 1) / kBmCullWaves, cus * per_cu);
    if (lds > 64 * 1024) {   // (a receptor of more than ~100 tiles)
        const hipError_t e = t.ent_count != nullptr
            ? hipFuncSetAttribute(reinterpret_cast<const void *>(&dfire_bm_cull<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
            : hipFuncSetAttribute(reinterpret_cast<const void *>(&dfire_bm_cull<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (t.ent_count != nullptr) hipLaunchKernelGGL((dfire_bm_cull<true>), dim3((unsigned)blocks), dim3(kBmCullWaves * 64), lds, stream, t);
    else hipLaunchKernelGGL((dfire_bm_cull<false>), dim3((unsigned)blocks), dim3(kBmCullWaves * 64), lds, stream, t);
    return hipGetLastError();
}

hipError_t launch_bm_pairs(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(dfire_bm_plan, dim3(1), dim3(1024), 0, stream, t);
    hipLaunchKernelGGL(dfire_bm_census, dim3(128), dim3(kBmOrderWaves * 64), 0, stream, t);
    hipLaunchKernelGGL(dfire_bm_order, dim3(1), dim3(kBmOrderWaves * 64), 0, stream, t);
    const unsigned groups = t.pairs_groups > 0 ? (unsigned)t.pairs_groups : 256u;   // persistent: one workgroup per CU
    if (t.ent_count != nullptr) hipLaunchKernelGGL((dfire_bm_pairs<true>), dim3(groups), dim3(kBmWaves * 64), 0, stream, t);
    else hipLaunchKernelGGL((dfire_bm_pairs<false>), dim3(groups), dim3(kBmWaves * 64), 0, stream, t);
    return hipGetLastError();
}

hipError_t launch_bm_gather(const BmLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    const int span = bm_gather_span(t.m.lig.n_tiles) * bm_gather_chunks(t.m.lig.n_tiles, t.m.rec_n_tiles);
    const size_t per_wg = 512 / span;
    const unsigned blocks = (unsigned)std::min<size_t>((t.n_poses + per_wg - 1) / per_wg, 16384);
    if (t.ent_count != nullptr) hipLaunchKernelGGL((dfire_bm_gather<true>), dim3(blocks), dim3(512), 0, stream, t);
    else hipLaunchKernelGGL((dfire_bm_gather<false>), dim3(blocks), dim3(512), 0, stream, t);
    return hipGetLastError();
}

}  // namespace ld
