// dfire_device.hpp -- device helpers shared by the DFIRE pose-energy kernels (dfire_packed.hip, dfire_bm.hip):
// quaternion posing in the reference's operation order (src/qt.rs, src/dfire.rs:282-302), bounding boxes by DPP
// reductions, wave reductions, and the exact f64 pair (src/dfire.rs:331-345) that every f32 filter falls back on.
// Include inside a .hip translation unit only (device code, -ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "dfire_tiled.hpp"

namespace ld {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(1))) unsigned int global_u32;

constexpr double kCutScaled = 900.0;    // 4 * 15^2, src/dfire.rs:334
constexpr float kCut2Padded = 900.04f;  // the same for the f32 box tests, padded for their rounding

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

// Ligand atom `la` (tile order) of the pose in `row`, in f64, exactly as the reference poses it
// (src/dfire.rs:282-302).  Used when a wave sets up its tile and again by the exact pair path, so
// both see the same bits.
struct Vec3 {
    double x, y, z;
};
__device__ __forceinline__ Vec3 pose_ligand_atom(const TiledLigand &lig, int use_anm, int anm_rec, const double *row, int la) {
    const double tx = row[0], ty = row[1], tz = row[2];
    const Quat q{row[3], row[4], row[5], row[6]};
    const Quat qinv = qinverse(q);
    const Quat v{0.0, lig.x[la], lig.y[la], lig.z[la]};
    const Quat r = qmul(qmul(q, v), qinv);
    Vec3 p{r.x + tx, r.y + ty, r.z + tz};
    if (use_anm && lig.num_anm > 0) {
        const double *lig_nm = row + 7 + anm_rec;
        const size_t pad = (size_t)lig.n_tiles * 64;
        for (int k = 0; k < lig.num_anm; k++) {
            const double c = lig_nm[k];
            const double *m = lig.modes + (size_t)k * 3 * pad;
            p.x += m[la] * c;
            p.y += m[pad + la] * c;
            p.z += m[2 * pad + la] * c;
        }
    }
    return p;
}

// f32 coordinate of the centred frame, scaled by kappa = 2 sqrt(SC) (see dfire_packed.hpp)
__device__ __forceinline__ float frame_coord(double x, double c, double kappa) { return (float)(kappa * (x - c)); }

__device__ __forceinline__ float axis_gap(float lo_a, float hi_a, float lo_b, float hi_b) {
    return fmaxf(0.0f, fmaxf(lo_a - hi_b, lo_b - hi_a));
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
// Bounding boxes by DPP reductions: one v_min/v_max with a lane-permuting operand per level and
// value, no LDS crossbar.  v_min_f32 / v_max_f32 return the other operand for a NaN, so an atom
// with NaN coordinates is in no box -- and in no pair, like in the reference, where NaN <= 225 is
// false.  Invalid lanes enter as the empty box (lo = +inf, hi = -inf).
__device__ __forceinline__ BoxRegs lane_box(bool valid, float fx, float fy, float fz) {
    BoxRegs b;
    b.lox = valid ? fx : INFINITY; b.hix = valid ? fx : -INFINITY;
    b.loy = valid ? fy : INFINITY; b.hiy = valid ? fy : -INFINITY;
    b.loz = valid ? fz : INFINITY; b.hiz = valid ? fz : -INFINITY;
    return b;
}
#define LD_BOX_DPP_LEVEL(ctrl)                                                                                            \
    asm("s_nop 1\n\t"                                                                                                     \
        "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\t"                                       \
        "v_min_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl "\n\t"                                       \
        "v_min_f32_dpp %4, %4, %4 " ctrl "\n\tv_max_f32_dpp %5, %5, %5 " ctrl                                              \
        : "+v"(b.lox), "+v"(b.hix), "+v"(b.loy), "+v"(b.hiy), "+v"(b.loz), "+v"(b.hiz))
// every lane of a group of 8 gets the box of the group
__device__ __forceinline__ void box_reduce8(BoxRegs &b) {
    LD_BOX_DPP_LEVEL("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    LD_BOX_DPP_LEVEL("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    LD_BOX_DPP_LEVEL("row_half_mirror row_mask:0xf bank_mask:0xf");
}
// from group-of-8 boxes to the box of the wave, valid in lane 63
__device__ __forceinline__ void box_reduce64_from8(BoxRegs &b) {
    LD_BOX_DPP_LEVEL("row_mirror row_mask:0xf bank_mask:0xf");
    LD_BOX_DPP_LEVEL("row_bcast:15 row_mask:0xa bank_mask:0xf");
    LD_BOX_DPP_LEVEL("row_bcast:31 row_mask:0xc bank_mask:0xf");
}
#undef LD_BOX_DPP_LEVEL
// The boxes are built from fl32(u): the true u lies within 2^-24 |u| of it, so widening each side by
// 2^-22 of its own magnitude (capped, so that an infinite side stays infinite instead of turning NaN)
// is outwards.  The smallest coordinate bounds the error of every other one on its side of zero.
__device__ __forceinline__ void box_widen(BoxRegs &b) {
    constexpr float w = 2.384185791015625e-07f;  // 2^-22
    constexpr float big = 3.0e38f;
    b.lox = __builtin_fmaf(-w, fminf(fabsf(b.lox), big), b.lox); b.hix = __builtin_fmaf(w, fminf(fabsf(b.hix), big), b.hix);
    b.loy = __builtin_fmaf(-w, fminf(fabsf(b.loy), big), b.loy); b.hiy = __builtin_fmaf(w, fminf(fabsf(b.hiy), big), b.hiy);
    b.loz = __builtin_fmaf(-w, fminf(fabsf(b.loz), big), b.loz); b.hiz = __builtin_fmaf(w, fminf(fabsf(b.hiz), big), b.hiz);
}
__device__ __forceinline__ float lane63_f32(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
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

// One pair entirely in f64, the reference's way (src/dfire.rs:331-345): the distance from the f64
// coordinates, the cutoff, the bin as a count of the exact steps passed, interface flags, the
// table value.  This is what the pairs that the f32 test cannot decide go through (queued by the
// pair loop, done after it), and all pairs of a wave whose queue overflowed.
struct ExactCtx {
    const double *rx, *ry, *rz;  // receptor f64 coordinates (tile order), undeformed
    const double *modes;         // receptor ANM modes [mode][xyz][pad] and this pose's amplitudes, or num_anm = 0
    const double *rec_nm;
    size_t pad;
    int num_anm;
    const uint32_t *rec_tindex;
    const int32_t *rec_slot, *lig_slot;
    const double *step4;         // LDS: 4 * bin_step[]
    const double *table;
    double iface_scaled;
    uint32_t *pose_flags;
    int rec_flag_words;
};
__device__ __forceinline__ double exact_pair(const ExactCtx &c, const Vec3 &p, uint32_t lig_term, int la, int ra, uint32_t &in_cutoff) {
    // (2 x_rec - 2 x_lig)^2 + ... = 4 d2 bit for bit (power-of-two scaling commutes with rounding)
    double rx = c.rx[ra], ry = c.ry[ra], rz = c.rz[ra];
    for (int k = 0; k < c.num_anm; k++) {  // src/dfire.rs:304-320, the same operations as dfire_packed_prepare
        const double a = c.rec_nm[k];
        const double *m = c.modes + (size_t)k * 3 * c.pad;
        rx += m[ra] * a;
        ry += m[c.pad + ra] * a;
        rz += m[2 * c.pad + ra] * a;
    }
    const double dx = 2.0 * rx - 2.0 * p.x, dy = 2.0 * ry - 2.0 * p.y, dz = 2.0 * rz - 2.0 * p.z;
    const double D = dx * dx + dy * dy + dz * dz;
    if (!(D <= kCutScaled)) return 0.0;  // d2 <= 225 (src/dfire.rs:334)
    uint32_t bin = 0;  // src/dfire.rs:336-337 as a count of the steps passed
    for (int b = 1; b <= 20; b++) bin += D >= c.step4[b] ? 1u : 0u;
    if (D <= c.iface_scaled) {  // d <= 3.9 (src/dfire.rs:339-342)
        const int32_t rslot = c.rec_slot[ra], lslot = c.lig_slot[la];
        if (rslot >= 0) atomicOr(&c.pose_flags[rslot >> 5], 1u << (rslot & 31));
        if (lslot >= 0) atomicOr(&c.pose_flags[c.rec_flag_words + (lslot >> 5)], 1u << (lslot & 31));
    }
    in_cutoff++;
    return c.table[(lig_term + c.rec_tindex[ra] + tiled_bin_term(bin)) / 8u];
}

}  // namespace
}  // namespace ld
