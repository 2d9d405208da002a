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

// A molecule in tile order, padded to whole 64-atom tiles; padding atoms sit at x = 1e30.
struct TiledMolecule {
    int n_real = 0;   // atoms that are not padding (padding only at the tail)
    int n_tiles = 0;  // 64-atom tiles; arrays hold n_tiles * 64 entries
    const double *x = nullptr, *y = nullptr, *z = nullptr;
    const uint32_t *tindex = nullptr;  // receptor: type*3380, ligand: type*20
    const int32_t *slot = nullptr;     // interface-flag bit or -1
    int num_anm = 0;
    const double *modes = nullptr;     // [mode][xyz][n_tiles*64]
    int flag_words = 0;
};

struct TiledLaunch {
    TiledMolecule rec, lig;
    int use_anm = 0;
    int waves = 8;        // wave64s per workgroup
    int chunk_tiles = 0;  // receptor tiles staged in LDS by one workgroup
    int n_chunks = 0;
    int segments = 1;     // the chunk's tiles are split into this many ranges; work item = (ligand tile, range)
    const double *table = nullptr;
    const uint8_t *lut = nullptr;       // cell -> bin | 0x80 if the cell's last double may belong to the next bin
    const double *bin_step = nullptr;   // kDfireSteps
    double iface_d2 = 0.0;
    const double *poses = nullptr;
    size_t stride = 0;
    const uint8_t *active = nullptr;
    size_t n_poses = 0;
    double *partial = nullptr;          // [pose][chunk][2]
    uint32_t *flags = nullptr;
    uint32_t *count_partial = nullptr;  // [pose][chunk] or nullptr
    uint32_t *tested_partial = nullptr; // [pose][chunk]: 8x8 blocks actually evaluated (diagnostics) or nullptr
};

size_t tiled_kernel_lds_bytes(const TiledLaunch &t);
// Largest chunk (in tiles) whose LDS image fits beside `waves` ligand tiles.
int tiled_max_chunk_tiles(int waves);
hipError_t configure_dfire_tiled();  // once per device, before the first launch
hipError_t launch_dfire_tiled(const TiledLaunch &t, hipStream_t stream);

}  // namespace ld
