// dfire_packed.hpp -- launch interface of the default DFIRE pose-energy kernel (K1, DFIRE).
//
// Same sum as src/dfire.rs:325-345 and the same culling as dfire_tiled.hpp (64x64 tile boxes, 8x8
// subtile boxes), but the pair test runs in packed f32 on 16-byte records and only the pairs whose
// f32 distance cannot decide the reference's f64 result are recomputed in f64:
//
//   record coordinate  u = fl32(kappa (x - c))      c = centre of the receptor's box, kappa = 2 sqrt(SC)
//   D' = fl32(sum (u_rec - u_lig)^2 + 1/2)          = SC * 4 d2 + 1/2 up to eps (dfire_f32_error_bound)
//   cell = min((unsigned)D', 1024 SC)               SC = LUT cells per unit of 4 d2: 1 by default, 2 with LIGHTDOCK_PACKED_CELLS=2
//
// Everything the reference derives from d2 -- the cutoff d2 <= 225 (src/dfire.rs:334), the distance
// bin (:336-337) and the interface test d <= 3.9 (:339) -- is a step function of 4 d2 with steps at
// the squares (k+1)^2, at 4*iface_d2 and at 900.  The half added to D' puts those integer steps in
// the middle of a cell, so a cell holds at most one of them.  A cell whose whole interval
// [(cell - 1/2) / SC - eps, (cell + 1/2) / SC + eps) lies between two steps has ONE answer for every f64 distance
// that can produce it: the LUT word is the table term of that bin (or kPackedMiss beyond the
// cutoff).  The other cells (2.6 % of the in-cutoff pairs with unit cells, half that with half-unit cells) are flagged
// kPackedSlow: there the kernel compares the f32 D with the step and, only if it is within eps of it (5.1e-3 units of 4 d2
// for a 256-unit frame with unit cells, 3.5e-3 with half-unit cells), recomputes the pair
// in f64 from the f64 coordinates (receptor image in HBM/L2, ligand atom re-posed), in the
// reference's operation order.  Bins, cutoff and interface flags are therefore the reference's f64
// results bit for bit, and the f64 `+=` of table values is untouched.
// Atoms further than `ubound` from c in the scaled frame (absurd ANM extents, ligand poses far from
// the receptor) carry kPackedSlow in their type term: every pair of theirs sums to a flagged offset and goes
// through the exact path (a NaN distance converts to cell 1024 SC, beyond the cutoff).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "dfire_tiled.hpp"

namespace ld {

// Wave64s per workgroup, one ligand tile each.  One: the tiles of a pose differ widely in work (four waves
// behind one barrier are busy 64 % of the time the slowest takes) and a wave that is done frees its slot at
// once; the price is a cell LUT per wave, which is why the default LUT has one cell per unit then
// (6.6 KiB of LDS per wave, 6 waves per SIMD).  Measured against 4 (MI355X, same box): 1k4c +2.3..4.5 %,
// GSO 1k4c +4.5 %, GSO 1ppe +1.7 %, 1ppe equal.
constexpr int kPackedWaves = 1;
constexpr int kPackedPartialsPerGroup = 1;
constexpr int kPackedLutCells = 1028;    // per cell of 4 d2: cells 0..1024 (1024 = everything further), padded to 16 bytes
constexpr float kPackedCellMax = 1024.0f;
constexpr int kPackedQueue = 64;         // per wave: pairs waiting for the exact f64 path
// LUT words.  Sum of the ligand term, the receptor term and the word = byte offset into the table.
//   plain cell:    tiled_bin_term(bin)
//   beyond cutoff: kPackedMiss (past the end of the 5.5 MB table: the load returns 0.0, no request)
//   flagged cell:  kPackedSlow | code << 24 | step bin << 16 | upper << 12 ... see kPackedCode*
// kPackedSlow in an atom's type term marks an atom outside the f32 frame: every pair of it goes
// through the exact path.  A sum with bit 30 set is what the pair loop branches on.
constexpr uint32_t kPackedMiss = 0x00800000u;
constexpr uint32_t kPackedSlow = 0x40000000u;
// code (bits 24..27) of a flagged cell
constexpr uint32_t kPackedCodeFlags = 0x10u;  // with Lean: the whole cell lies below the interface distance (src/dfire.rs:339):
                                               // lean unless one of the two atoms has an interface-flag slot
constexpr uint32_t kPackedCodeLean = 0x1u;    // exactly one step, at the cell's middle: bits 0..11 = term below it,
                                               // bits 12..23 = what the term grows by above it (beyond the cutoff:
                                               // up to the unused bin slot 21, which holds 0.0)
constexpr uint32_t kPackedCodeStep = 0x2u;    // full path: a bin step inside, its bin in bits 16..20
constexpr uint32_t kPackedCodeIface = 0x4u;   // full path: the cell reaches down to the interface distance
constexpr uint32_t kPackedCodeCutoff = 0x8u;  // full path: the cutoff inside

// Two receptor atoms as one lane of the pair loop reads them: the atoms (2 q, 2 q + 1) of subtile j
// of a tile (record index 4 j + q).  The operands of v_pk_add_f32 / v_pk_fma_f32 are (x0, x1),
// (y0, y1), (z0, z1) as they lie here.
struct alignas(32) PackedRecPair {
    float x0, x1, y0, y1, z0, z1;
    uint32_t t0, t1;  // tiled_rec_term(type): byte offset of the type's column in a table patch
};
static_assert(sizeof(PackedRecPair) == 32, "PackedRecPair must be 32 bytes");

struct PackedReceptor {
    int n_real = 0;
    int n_tiles = 0;
    const PackedRecPair *pairs = nullptr;       // [n_tiles*32]
    const TiledBox *sub_boxes = nullptr;        // [n_tiles*8], scaled + centred frame
    const TiledBox *tile_boxes = nullptr;       // [n_tiles]; pad0/pad1 = bit per atom of the tile that has a flag slot
    const double *x = nullptr, *y = nullptr, *z = nullptr;  // f64, tile order, padded, undeformed: the exact path reads these ...
    const double *modes = nullptr;              // ... and deforms them itself when the image is per pose: [mode][xyz][n_tiles*64]
    int num_anm = 0;                            // 0 = static image
    const int32_t *slot = nullptr;              // tile order
    const uint32_t *tindex = nullptr;           // tile order: tiled_rec_term(type), what the records carry
    size_t pose_stride_pairs = 0, pose_stride_sub = 0, pose_stride_tile = 0;  // 0 = static image
    int flag_words = 0;
};

struct PackedLaunch {
    PackedReceptor rec;
    TiledLigand lig;
    int use_anm = 0;
    int anm_rec = 0;
    int split = 1;      // waves sharing one ligand tile (each takes every split-th surviving receptor tile)
    int n_groups = 0;   // workgroups per pose = ceil(lig.n_tiles * split / kPackedWaves)
    double cx = 0, cy = 0, cz = 0;  // centre of the f32 frame (unscaled)
    double kappa = 2.0;             // records hold fl32(kappa (x - c)), kappa = 2 sqrt(cells_per_unit)
    int cells_per_unit = 1;         // LUT cells per unit of 4 d2 (1 or 2)
    float ubound = 0.f;             // |u| beyond this: the atom is flagged kPackedSlow
    float eps = 0.f;                // bound on |D_f32 - 4 d2| for records inside ubound, in units of 4 d2
    const double *table = nullptr;  // 2 x 2 x 4 patches, dfire_tiled.hpp
    const uint32_t *lut = nullptr;  // kPackedLutCells words
    const double *bin_step = nullptr;  // kDfireSteps, d2 units
    double iface_scaled = 0.0;      // 4 * iface_d2
    const double *poses = nullptr;
    size_t stride = 0;
    const uint8_t *active = nullptr;
    size_t n_poses = 0;                  // rows in `poses`; upper bound of *pose_count
    // GSO: the rows to evaluate as a list compacted by K2 (src/glowworm.rs:62: only the glowworms that
    // moved).  The count lives on the device only: the launch covers n_poses rows and the workgroups
    // beyond the count leave on one scalar load.  Both null for a plain batch.
    const uint32_t *pose_list = nullptr;
    const uint32_t *pose_count = nullptr;
    double *partial = nullptr;           // [pose][group][2]
    uint32_t *flags = nullptr;
    uint32_t *count_partial = nullptr;   // [pose][group] or nullptr
    uint32_t *tested_partial = nullptr;  // [pose][group]: 8x8 blocks evaluated (diagnostics) or nullptr
    uint32_t *exact_partial = nullptr;   // [pose][group]: pairs recomputed in f64 (diagnostics) or nullptr
};

struct PackedPrepareLaunch {
    int n_real = 0, n_tiles = 0;
    const double *x = nullptr, *y = nullptr, *z = nullptr;  // tile order, padded
    const uint32_t *tindex = nullptr;
    const int32_t *slot = nullptr;
    int num_anm = 0;
    const double *modes = nullptr;  // [mode][xyz][n_tiles*64]
    const double *poses = nullptr;
    size_t stride = 0;
    const uint8_t *active = nullptr;
    size_t n_poses = 0;
    double cx = 0, cy = 0, cz = 0;
    double kappa = 2.0;
    float ubound = 0.f;
    PackedRecPair *pairs_out = nullptr;
    TiledBox *sub_out = nullptr, *tile_out = nullptr;
};

// Bound on |D_f32 - 4 d2| (units of 4 d2) for two records inside `ubound` (record units) whose true
// 4 d2 is below 1100, with the kernel's operation order (three packed subtractions, one fma chain
// seeded with 1/2).
double dfire_f32_error_bound(double ubound, int cells_per_unit);

size_t packed_kernel_lds_bytes(int cells_per_unit);
hipError_t launch_dfire_packed(const PackedLaunch &t, hipStream_t stream);
hipError_t launch_packed_prepare(const PackedPrepareLaunch &p, hipStream_t stream);

}  // namespace ld
