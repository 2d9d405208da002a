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
// The tiled kernel reads the potential in 128-byte patches of 2 ligand types x 2 receptor types
// x 4 distance bins:
//   index = ((l/2)*85 + r/2)*96 + (bin/4)*16 + (l%2)*8 + (r%2)*4 + bin%4        (bin 0..20; 169 types -> 85 pairs)
// The atoms of a subtile are mostly one residue, whose DFIRE types are consecutive numbers, and
// bonded atoms sit in the same or the next distance bin of a given partner, so the hits of one
// 8x8 block fall into fewer distinct cache lines than with any row-major order (simulated on the
// 1k4c poses: 0.63 lines per hit; [lig][bin][rec] rows 0.77, the reference's [rec][lig][bin] 0.95).
// The kernel is bound by the L1 fills of this gather, so lines per hit is throughput.
// Bin 20 is what the reference reads for r = 15.0 A exactly: potential[r*3380 + l*20 + 20]
// (src/dfire.rs:338).
constexpr uint32_t kTiledTableBins = 21;
constexpr uint32_t kTiledPatchDoubles = 16;                       // one 128-byte line
constexpr uint32_t kTiledRecStride = 6 * kTiledPatchDoubles;      // 24 bin slots per type pair
constexpr uint32_t kTiledLigStride = 85 * kTiledRecStride;        // 169 types -> 85 pairs
constexpr uint32_t kTiledTableDoubles = 85 * kTiledLigStride;
// The three terms are BYTE offsets; their sum is the buffer-load offset of the table entry.
__host__ __device__ inline uint32_t tiled_lig_term(uint32_t type) { return 8u * ((type >> 1) * kTiledLigStride + (type & 1u) * 8u); }
__host__ __device__ inline uint32_t tiled_rec_term(uint32_t type) { return 8u * ((type >> 1) * kTiledRecStride + (type & 1u) * 4u); }
__host__ __device__ inline uint32_t tiled_bin_term(uint32_t bin) { return 8u * (bin + 12u * (bin >> 2)); }
// Cell LUT of the kernel: one 32-bit word per 0.25 A^2 cell of d2 (cell = (int)(4 d2), 0..903)
//   cells whose every d2 has one bin and no side effect:  tiled_bin_term(bin)
//   cells that need the exact test (a bin step inside, the interface distance, the cutoff
//   cell 900):                                            kTiledLutSlow | 8-bit code (scorer.cpp)
//   cells beyond the cutoff:                              kTiledLutMiss
// kTiledLutMiss pushes the buffer offset past the end of the table: the load returns 0.0 without
// touching memory, so pairs out of range need neither a compare nor a branch.
constexpr uint32_t kTiledLutSlow = 0x40000000u;
constexpr uint32_t kTiledLutMiss = 0x80000000u;

// 32-byte atom record, the unit both molecules are handled in inside the kernel.
struct alignas(16) TiledAtom {
    double x, y, z;
    uint32_t tindex;  // tiled_rec_term(type) / tiled_lig_term(type): byte offsets
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
    const uint32_t *lut = nullptr;     // kDfireLutCells words, see kTiledLutSlow
    const double *bin_step = nullptr;  // kDfireSteps
    double iface_d2 = 0.0;
    double iface_scaled = 0.0;  // 4 * iface_d2 (the kernel works on doubled coordinates)
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
