// dfire_tiled.hpp -- launch interface of the tiled DFIRE pose-energy kernel (K1, DFIRE).
//
// Same result as the all-pairs kernel of pose_energy.hip (and as src/dfire.rs:264-363), but
// whole 64x64 and 8x8 blocks of atom pairs whose bounding boxes are further apart than the
// 15 A cutoff are skipped.  Atoms arrive in the spatial tile order of host/spatial_order.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "pose_energy.hpp"

namespace ld {

constexpr int kTiledMaxWaves = 16;
// The tiled kernel reads the potential as table[lig_type][bin 0..20][rec_type] (stride 176):
// the 8 receptor atoms of a subtile are mostly one residue, whose DFIRE types are consecutive
// numbers, so lanes of one ligand atom whose pairs fall in the same distance bin share a 64-byte
// line.  The kernel is bound by outstanding L1 misses of this gather, so fewer
// distinct lines per wave instruction is throughput.  Entry [l][20][r] is what the reference
// reads for r = 15.0 A exactly: potential[r*3380 + l*20 + 20] (src/dfire.rs:338).
constexpr uint32_t kTiledTableStride = 176;
constexpr uint32_t kTiledTableBins = 21;

// 32-byte atom record, the unit both molecules are handled in inside the kernel.
struct alignas(16) TiledAtom {
    double x, y, z;
    uint32_t tindex;  // receptor: type, ligand: type * 21 * 176 (see kTiledTableStride)
    int32_t slot;     // interface-flag bit or -1
};
static_assert(sizeof(TiledAtom) == 32, "TiledAtom must be 32 bytes");

// f32 bounding box rounded outwards; an empty box has lo = +inf, hi = -inf.
struct alignas(16) TiledBox {
    float lox, loy, loz, pad0;
    float hix, hiy, hiz, pad1;
};
static_assert(sizeof(TiledBox) == 32, "TiledBox must be 32 bytes");

// The receptor as the kernel streams it: records in tile order (padding atoms at x = -1e30),
// one box per 8-atom subtile and per 64-atom tile.  Without receptor ANM this is static data
// (pose_stride_* = 0); with it, dfire_prepare_receptor writes one image per pose.
struct TiledReceptor {
    int n_real = 0;
    int n_tiles = 0;
    const TiledAtom *atoms = nullptr;   // [n_tiles*64]
    const TiledBox *sub_boxes = nullptr;   // [n_tiles*8]
    const TiledBox *tile_boxes = nullptr;  // [n_tiles]
    size_t pose_stride_atoms = 0, pose_stride_sub = 0, pose_stride_tile = 0;  // elements per pose image
    int flag_words = 0;
};

// The ligand in tile order, SoA, padded to whole tiles (the kernel re-places padding after posing).
struct TiledLigand {
    int n_real = 0;
    int n_tiles = 0;
    const double *x = nullptr, *y = nullptr, *z = nullptr;
    const uint32_t *tindex = nullptr;
    const int32_t *slot = nullptr;
    int num_anm = 0;
    const double *modes = nullptr;  // [mode][xyz][n_tiles*64]
    int flag_words = 0;
};

struct TiledLaunch {
    TiledReceptor rec;
    TiledLigand lig;
    int use_anm = 0;
    int anm_rec = 0;      // pose-row columns taken by receptor ANM extents
    int waves = 1;        // wave64s per workgroup; each wave owns one ligand tile of one pose
    int split = 1;        // waves sharing one ligand tile (each takes every split-th surviving receptor tile)
    int n_groups = 0;     // workgroups per pose = ceil(lig.n_tiles * split / waves)
    const double *table = nullptr;
    const uint8_t *lut = nullptr;      // cell -> bin | 0x80 if a bin step falls inside the cell
    const double *bin_step = nullptr;  // kDfireSteps
    double iface_d2 = 0.0;
    const double *poses = nullptr;
    size_t stride = 0;
    const uint8_t *active = nullptr;
    size_t n_poses = 0;
    double *partial = nullptr;           // [pose][group][2]
    uint32_t *flags = nullptr;
    uint32_t *count_partial = nullptr;   // [pose][group] or nullptr
    uint32_t *tested_partial = nullptr;  // [pose][group]: 8x8 blocks evaluated (diagnostics) or nullptr
};

// Receptor image per pose for runs with receptor ANM (src/dfire.rs:304-320).
struct PrepareReceptorLaunch {
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
    TiledAtom *atoms_out = nullptr;
    TiledBox *sub_out = nullptr, *tile_out = nullptr;
};

size_t tiled_kernel_lds_bytes(const TiledLaunch &t);
hipError_t launch_dfire_tiled(const TiledLaunch &t, hipStream_t stream);
hipError_t launch_prepare_receptor(const PrepareReceptorLaunch &p, hipStream_t stream);

}  // namespace ld
