// dfire_tiled.hip -- K1 for DFIRE with bounding-box culling (gfx950 / MI355X).
//
// DFIRE only counts pairs closer than 15 A (src/dfire.rs:334): about 1 % of the 11.2 M atom
// pairs of the 1k4c example.  This kernel evaluates the same sum as src/dfire.rs:325-345 but
// throws away whole blocks of pairs by box distance before touching them:
//
//   workgroup = (pose, receptor chunk); up to 16 wave64s.
//   1. receptor chunk: coalesced SoA loads from HBM, ANM deformation (src/dfire.rs:304-320),
//      32-byte f64 records into LDS; then one f32 bounding box (rounded outwards) per 8-atom
//      subtile and per 64-atom tile, also in LDS.
//   2. work item = (ligand tile of 64 atoms, range of receptor tiles), dealt round-robin to
//      the waves.  The wave poses its ligand tile in registers (q v q^-1 + t, then ANM;
//      src/dfire.rs:282-302), parks the posed records in its private LDS slice and builds the
//      8 subtile boxes + the tile box with wave shuffles.
//   3. 64 lanes test the tile box against 64 receptor tile boxes at once (ballot); for every
//      surviving receptor tile, 64 lanes test the 8x8 subtile pairs at once (ballot).
//   4. every surviving subtile pair is one wave iteration: lane (i, j) takes ligand atom i and
//      receptor atom j of the pair -- 64 distinct atom pairs, all operands from LDS -- and runs
//      the reference's pair body: f64 d2 in the reference's operation order, cutoff, distance
//      bin (LUT, exact), potential[type_i][type_j][bin] gather, interface flags.
//   5. wave64 shuffle reduction, cross-wave through LDS, one partial per (pose, chunk).
//
// Box tests are conservative (boxes rounded outwards, cutoff padded), so no in-cutoff pair is
// ever dropped; the pair body itself is bit-identical to the all-pairs kernel.  Only the
// order of the f64 += differs.  Compiled with -ffp-contract=off.  No MFMA (lookup/reduction).
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

struct alignas(16) AtomRec {
    double x, y, z;
    uint32_t tindex;
    int32_t slot;
};
static_assert(sizeof(AtomRec) == 32, "AtomRec must be 32 bytes");

struct alignas(16) Box {
    float lox, loy, loz, pad0;
    float hix, hiy, hiz, pad1;
};
static_assert(sizeof(Box) == 32, "Box must be 32 bytes");

constexpr float kCut2Padded = 225.01f;  // 15 A cutoff + slack for the f32 box arithmetic

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
__device__ __forceinline__ float box_gap2(float lox, float loy, float loz, float hix, float hiy, float hiz, const Box &b) {
    const float gx = axis_gap(lox, hix, b.lox, b.hix);
    const float gy = axis_gap(loy, hiy, b.loy, b.hiy);
    const float gz = axis_gap(loz, hiz, b.loz, b.hiz);
    return gx * gx + gy * gy + gz * gz;
}

__device__ __forceinline__ size_t round16(size_t v) { return (v + 15) & ~size_t(15); }

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

template <bool COUNT>
__global__ __launch_bounds__(1024) void dfire_tiled_pairs(const TiledLaunch T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- LDS carve (every offset a multiple of 16) ---------------------------------------
    AtomRec *rec = reinterpret_cast<AtomRec *>(smem);
    size_t off = (size_t)T.chunk_tiles * 64 * sizeof(AtomRec);
    Box *sbox = reinterpret_cast<Box *>(smem + off);
    off += (size_t)T.chunk_tiles * 8 * sizeof(Box);
    Box *tbox = reinterpret_cast<Box *>(smem + off);
    off += (size_t)T.chunk_tiles * sizeof(Box);
    AtomRec *ligt_all = reinterpret_cast<AtomRec *>(smem + off);
    off += (size_t)T.waves * 64 * sizeof(AtomRec);
    uint8_t *lut = smem + off;
    off += round16(kDfireLutCells);
    double *bin_step = reinterpret_cast<double *>(smem + off);
    off += kDfireSteps * sizeof(double);
    double *red = reinterpret_cast<double *>(smem + off);
    off += kTiledMaxWaves * sizeof(double);
    uint32_t *red_cnt = reinterpret_cast<uint32_t *>(smem + off);  // [kTiledMaxWaves][2]

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t pose = blockIdx.x / (unsigned)T.n_chunks;
    const int chunk = blockIdx.x % (unsigned)T.n_chunks;
    if (T.active != nullptr && T.active[pose] == 0) return;

    const double *row = T.poses + pose * T.stride;
    const double tx = row[0], ty = row[1], tz = row[2];
    const Quat q{row[3], row[4], row[5], row[6]};
    const Quat qinv = qinverse(q);
    const bool anm_rec = T.use_anm && T.rec.num_anm > 0;
    const bool anm_lig = T.use_anm && T.lig.num_anm > 0;
    const double *rec_nm = row + 7;
    const double *lig_nm = row + 7 + (T.use_anm ? T.rec.num_anm : 0);

    // ---- 1. stage the receptor chunk ---------------------------------------------------------
    const int tile0 = chunk * T.chunk_tiles;
    const int rn_tiles = min(T.chunk_tiles, T.rec.n_tiles - tile0);
    const int rn = rn_tiles * 64;
    const int atom0 = tile0 * 64;
    const size_t rec_pad = (size_t)T.rec.n_tiles * 64;
    for (int i = tid; i < rn; i += nthreads) {
        const int a = atom0 + i;
        double x = T.rec.x[a], y = T.rec.y[a], z = T.rec.z[a];
        if (anm_rec) {
            for (int k = 0; k < T.rec.num_anm; k++) {
                const double c = rec_nm[k];
                const double *m = T.rec.modes + (size_t)k * 3 * rec_pad;
                x += m[a] * c;
                y += m[rec_pad + a] * c;
                z += m[2 * rec_pad + a] * c;
            }
        }
        AtomRec r;
        r.x = x;
        r.y = y;
        r.z = z;
        r.tindex = T.rec.tindex[a];
        r.slot = T.rec.slot[a];
        rec[i] = r;
    }
    for (int i = tid; i < kDfireLutCells / 4; i += nthreads)
        reinterpret_cast<uint32_t *>(lut)[i] = reinterpret_cast<const uint32_t *>(T.lut)[i];
    if (tid < kDfireSteps) bin_step[tid] = T.bin_step[tid];
    __syncthreads();

    for (int s = tid; s < rn_tiles * 8; s += nthreads) {  // subtile boxes over the real atoms
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 8; k++) {
            const int i = s * 8 + k;
            if (atom0 + i < T.rec.n_real) {
                const AtomRec r = rec[i];
                lo[0] = fmin(lo[0], r.x); hi[0] = fmax(hi[0], r.x);
                lo[1] = fmin(lo[1], r.y); hi[1] = fmax(hi[1], r.y);
                lo[2] = fmin(lo[2], r.z); hi[2] = fmax(hi[2], r.z);
            }
        }
        Box b;
        b.lox = round_down(lo[0]); b.loy = round_down(lo[1]); b.loz = round_down(lo[2]); b.pad0 = 0.f;
        b.hix = round_up(hi[0]); b.hiy = round_up(hi[1]); b.hiz = round_up(hi[2]); b.pad1 = 0.f;
        sbox[s] = b;
    }
    __syncthreads();
    for (int t = tid; t < rn_tiles; t += nthreads) {  // tile boxes
        Box b = sbox[t * 8];
        for (int k = 1; k < 8; k++) {
            const Box c = sbox[t * 8 + k];
            b.lox = fminf(b.lox, c.lox); b.loy = fminf(b.loy, c.loy); b.loz = fminf(b.loz, c.loz);
            b.hix = fmaxf(b.hix, c.hix); b.hiy = fmaxf(b.hiy, c.hiy); b.hiz = fmaxf(b.hiz, c.hiz);
        }
        tbox[t] = b;
    }
    __syncthreads();

    // ---- 2..4 work items ------------------------------------------------------------------------
    AtomRec *ligt = ligt_all + wave * 64;
    uint32_t *pose_flags = T.flags + pose * (size_t)(T.rec.flag_words + T.lig.flag_words);
    const size_t lig_pad = (size_t)T.lig.n_tiles * 64;
    const int n_lt = T.lig.n_tiles;
    const int items = n_lt * T.segments;
    const int li = lane >> 3, lj = lane & 7;
    double acc = 0.0;
    uint32_t cnt = 0, tested = 0;

    for (int item = wave; item < items; item += T.waves) {
        const int LT = item % n_lt;
        const int seg = item / n_lt;
        const int seg_lo = seg * rn_tiles / T.segments;
        const int seg_hi = (seg + 1) * rn_tiles / T.segments;

        // pose this lane's ligand atom (src/dfire.rs:282-302)
        const int la = LT * 64 + lane;
        const bool valid = la < T.lig.n_real;
        AtomRec me;
        {
            const Quat v{0.0, T.lig.x[la], T.lig.y[la], T.lig.z[la]};
            const Quat r = qmul(qmul(q, v), qinv);
            double px = r.x + tx, py = r.y + ty, pz = r.z + tz;
            if (anm_lig) {
                for (int k = 0; k < T.lig.num_anm; k++) {
                    const double c = lig_nm[k];
                    const double *m = T.lig.modes + (size_t)k * 3 * lig_pad;
                    px += m[la] * c;
                    py += m[lig_pad + la] * c;
                    pz += m[2 * lig_pad + la] * c;
                }
            }
            me.x = valid ? px : 1.0e30;  // padding: far away, opposite side of the receptor's padding
            me.y = valid ? py : 0.0;
            me.z = valid ? pz : 0.0;
            me.tindex = T.lig.tindex[la];
            me.slot = T.lig.slot[la];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // earlier reads of ligt are done
        ligt[lane] = me;

        // subtile boxes (8 lanes each) and the tile box by xor butterflies
        float slox = valid ? round_down(me.x) : INFINITY, shix = valid ? round_up(me.x) : -INFINITY;
        float sloy = valid ? round_down(me.y) : INFINITY, shiy = valid ? round_up(me.y) : -INFINITY;
        float sloz = valid ? round_down(me.z) : INFINITY, shiz = valid ? round_up(me.z) : -INFINITY;
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            slox = fminf(slox, __shfl_xor(slox, m, 64)); shix = fmaxf(shix, __shfl_xor(shix, m, 64));
            sloy = fminf(sloy, __shfl_xor(sloy, m, 64)); shiy = fmaxf(shiy, __shfl_xor(shiy, m, 64));
            sloz = fminf(sloz, __shfl_xor(sloz, m, 64)); shiz = fmaxf(shiz, __shfl_xor(shiz, m, 64));
        }
        float tlox = slox, thix = shix, tloy = sloy, thiy = shiy, tloz = sloz, thiz = shiz;
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            tlox = fminf(tlox, __shfl_xor(tlox, m, 64)); thix = fmaxf(thix, __shfl_xor(thix, m, 64));
            tloy = fminf(tloy, __shfl_xor(tloy, m, 64)); thiy = fmaxf(thiy, __shfl_xor(thiy, m, 64));
            tloz = fminf(tloz, __shfl_xor(tloz, m, 64)); thiz = fmaxf(thiz, __shfl_xor(thiz, m, 64));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // ligt is written before anyone reads it

        // 64 receptor tiles per ballot
        bool tile_near = false;
        if (lane >= seg_lo && lane < seg_hi) tile_near = box_gap2(tlox, tloy, tloz, thix, thiy, thiz, tbox[lane]) <= kCut2Padded;
        unsigned long long rtmask = __ballot(tile_near);
        while (rtmask) {
            const int RT = __ffsll(rtmask) - 1;
            rtmask &= rtmask - 1;
            // lane (li, lj): ligand subtile li (its box is in this lane's registers) x receptor subtile lj
            const bool sub_near = box_gap2(slox, sloy, sloz, shix, shiy, shiz, sbox[RT * 8 + lj]) <= kCut2Padded;
            unsigned long long smask = __ballot(sub_near);
            if (COUNT) tested += (uint32_t)__popcll(smask);
            const AtomRec *rtile = rec + RT * 64;
            while (smask) {
                const int a = (__ffsll(smask) - 1) >> 3;
                uint32_t am = (uint32_t)(smask >> (8 * a)) & 0xffu;
                smask &= ~(0xffull << (8 * a));
                const AtomRec L = ligt[a * 8 + li];
                const double *tab = T.table + L.tindex;
                while (am) {
                    const int b = __ffs(am) - 1;
                    am &= am - 1;
                    const AtomRec R = rtile[b * 8 + lj];
                    // (x1 - la[0])^2 + (y1 - la[1])^2 + (z1 - la[2])^2, src/dfire.rs:331-333
                    const double dx = R.x - L.x, dy = R.y - L.y, dz = R.z - L.z;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 <= 225.0) {
                        // DIST_TO_BINS[(sqrt(d2)*2-1) as usize] - 1, src/dfire.rs:336-337, via the
                        // exact cell LUT (DESIGN.md "bin LUT"); 0x80 marks the few cells whose last
                        // double the correctly rounded sqrt pushes into the next bin.
                        const uint32_t code = lut[(int)(d2 * 4.0)];
                        uint32_t bin = code & 0x7fu;
                        if (code & 0x80u) bin += d2 >= bin_step[bin + 1] ? 1u : 0u;
                        acc += tab[R.tindex + bin];  // src/dfire.rs:338
                        if (COUNT) cnt++;
                        if (d2 <= T.iface_d2) {  // d <= 3.9, src/dfire.rs:339-342
                            if (R.slot >= 0) atomicOr(&pose_flags[R.slot >> 5], 1u << (R.slot & 31));
                            if (L.slot >= 0) atomicOr(&pose_flags[T.rec.flag_words + (L.slot >> 5)], 1u << (L.slot & 31));
                        }
                    }
                }
            }
        }
    }

    // ---- 5. reduction ----------------------------------------------------------------------------
    acc = wave_sum(acc);
    if (COUNT) cnt = wave_sum_u32(cnt);
    if (lane == 0) {
        red[wave] = acc;
        if (COUNT) {
            red_cnt[2 * wave] = cnt;
            red_cnt[2 * wave + 1] = tested;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        uint32_t c = 0, t = 0;
        for (int w = 0; w < T.waves; w++) {
            s += red[w];
            if (COUNT) {
                c += red_cnt[2 * w];
                t += red_cnt[2 * w + 1];
            }
        }
        const size_t slot = pose * (size_t)T.n_chunks + chunk;
        T.partial[2 * slot] = s;
        T.partial[2 * slot + 1] = 0.0;
        if (COUNT) {
            T.count_partial[slot] = c;
            if (T.tested_partial) T.tested_partial[slot] = t;
        }
    }
}

size_t lds_bytes(int chunk_tiles, int waves) {
    size_t b = (size_t)chunk_tiles * 64 * sizeof(AtomRec) + (size_t)chunk_tiles * 8 * sizeof(Box) +
               (size_t)chunk_tiles * sizeof(Box) + (size_t)waves * 64 * sizeof(AtomRec);
    b += ((kDfireLutCells + 15) & ~15) + kDfireSteps * sizeof(double);
    b += kTiledMaxWaves * sizeof(double) + kTiledMaxWaves * 2 * sizeof(uint32_t) + 16;
    return b;
}

}  // namespace

size_t tiled_kernel_lds_bytes(const TiledLaunch &t) { return lds_bytes(t.chunk_tiles, t.waves); }

int tiled_max_chunk_tiles(int waves) {
    const size_t budget = 160 * 1024;
    int tiles = 1;
    while (tiles < 64 && lds_bytes(tiles + 1, waves) <= budget) tiles++;  // 64: one ballot covers the chunk
    return tiles;
}

hipError_t configure_dfire_tiled() {  // > 64 KiB of dynamic LDS has to be requested explicitly, per device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dfire_tiled_pairs<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(dfire_tiled_pairs<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_dfire_tiled(const TiledLaunch &t, hipStream_t stream) {
    if (t.n_poses == 0) return hipSuccess;
    const size_t blocks = t.n_poses * (size_t)t.n_chunks;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = tiled_kernel_lds_bytes(t);
    const dim3 grid((unsigned)blocks), block((unsigned)t.waves * 64);
    if (t.count_partial != nullptr) hipLaunchKernelGGL((dfire_tiled_pairs<true>), grid, block, lds, stream, t);
    else hipLaunchKernelGGL((dfire_tiled_pairs<false>), grid, block, lds, stream, t);
    return hipGetLastError();
}

}  // namespace ld
