// pose_energy.hpp -- launch interface of the batched pose-energy kernels (K1).
//
// K1 evaluates `Score::energy` (reference src/scoring.rs:11-19) for a batch of poses:
// DFIRE (src/dfire.rs:264-363) or DNA (src/dna.rs:410-529).  One workgroup handles one
// (pose, receptor chunk); a second tiny kernel folds the chunk partials and applies the
// restraint / membrane tail (src/dfire.rs:347-361, src/scoring.rs:21-47).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ld {

constexpr int kBlockThreads = 256;  // 4 wave64
constexpr int kWaves = kBlockThreads / 64;
constexpr int kDfireLutCells = 904;  // 4*225 + 1 cells of 0.25 A^2, padded to a multiple of 8
constexpr int kDfireSteps = 24;      // exact first d2 of bins 0..20, +inf padded
constexpr uint32_t kDfireRowStride = 169 * 20;  // src/dfire.rs:338

// Device-resident molecule, SoA, padded to a multiple of 64 atoms.
struct DeviceMolecule {
    int n = 0;      // atoms
    int n_pad = 0;  // allocation length of every per-atom array
    const double *x = nullptr, *y = nullptr, *z = nullptr;
    // DFIRE receptor: type * 3380 (row base into the potential); DFIRE ligand: type * 20
    const uint32_t *tindex = nullptr;
    // tracked-atom slot (bit index into the per-pose interface flag words) or -1
    const int32_t *slot = nullptr;
    // DNA per-atom parameters; well_depth holds sqrt(eps) (src/dna.rs:495 takes sqrt(eps_i*eps_j) per pair)
    const double *charge = nullptr, *well_depth = nullptr, *radius = nullptr;
    // ANM modes re-laid out as [mode][xyz][n_pad] so atom-consecutive lanes load coalesced
    int num_anm = 0;
    const double *modes = nullptr;
    int flag_words = 0;  // uint32 words of interface flags per pose for this side
};

struct PairLaunch {
    DeviceMolecule rec, lig;
    int method = 0;
    int use_anm = 0;
    int chunk_atoms = 0;  // receptor atoms per workgroup
    int n_chunks = 0;
    int split_j = 0;      // 1: every wave walks all ligand groups over a quarter of the chunk
    const double *table = nullptr;  // DFIRE potential (LD_DFIRE_TABLE_LEN)
    const uint8_t *lut = nullptr;   // DFIRE: cell floor(d2*4) -> bin at the cell's lower edge
    const double *bin_step = nullptr;  // DFIRE: exact first d2 of each bin (kDfireSteps)
    double iface_d2 = 0.0;          // pair is "interface" iff d2 <= iface_d2
    // batch
    const double *poses = nullptr;
    size_t stride = 0;
    const uint8_t *active = nullptr;
    size_t n_poses = 0;
    // workspace outputs
    double *partial = nullptr;       // [pose][chunk][2]
    uint32_t *flags = nullptr;       // [pose][rec.flag_words + lig.flag_words], pre-zeroed
    uint32_t *count_partial = nullptr;  // [pose][chunk] or nullptr
};

// Restraint groups / membrane beads expressed over flag slots.
struct TailTables {
    int n_rec_groups = 0, n_lig_groups = 0, n_membrane = 0;
    const uint32_t *rec_group_offsets = nullptr, *rec_group_slots = nullptr;
    const uint32_t *lig_group_offsets = nullptr, *lig_group_slots = nullptr;
    const uint32_t *membrane_slots = nullptr;  // receptor-side slots
};

struct FinishLaunch {
    int method = 0;
    int n_chunks = 0;
    int rec_flag_words = 0, lig_flag_words = 0;
    TailTables tail;
    const double *partial = nullptr;
    const uint32_t *flags = nullptr;
    const uint32_t *count_partial = nullptr;
    const uint8_t *active = nullptr;
    size_t n_poses = 0;
    double *energies = nullptr;
    uint32_t *pair_counts = nullptr;
};

size_t pair_kernel_lds_bytes(const PairLaunch &p);
const char *pair_kernel_name(int method);
hipError_t launch_pair_kernel(const PairLaunch &p, hipStream_t stream);
hipError_t launch_finish_kernel(const FinishLaunch &f, hipStream_t stream);

}  // namespace ld
