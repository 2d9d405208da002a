// scorer.hpp -- the object behind an `ld_scorer*`: device-resident docking models + the
// workspace of the pose-energy kernels.  Plays the role of the reference's
// `Box<dyn Score>` (DFIRE / DNA structs, src/dfire.rs:193-198, src/dna.rs:367-372).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "host/error.hpp"
#include "kernels/dfire_bm.hpp"
#include "kernels/dfire_packed.hpp"
#include "kernels/dfire_tiled.hpp"
#include "kernels/pose_energy.hpp"
#include "lightdock_hip.h"

namespace ld {

inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Error(LD_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// Owns a set of device allocations; freed together.
class DeviceArena {
   public:
    ~DeviceArena();
    template <typename T>
    T *upload(const std::vector<T> &host, size_t min_count = 0) {
        size_t count = host.size() > min_count ? host.size() : min_count;
        if (count == 0) count = 1;
        T *d = static_cast<T *>(alloc_bytes(count * sizeof(T)));
        hip_check(hipMemset(d, 0, count * sizeof(T)), "hipMemset");
        if (!host.empty()) hip_check(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
        return d;
    }
    void *alloc_bytes(size_t bytes);

   private:
    std::vector<void *> blocks_;
};

// A grow-only device buffer.
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    uint64_t generation = 0;  // bumped whenever the block is reallocated
    void reserve(size_t want);
    void release();
};

struct HostMolecule {  // what ld_scorer_model_arrays hands back
    std::vector<double> coordinates;
    std::vector<uint32_t> dfire_types;
    std::vector<double> ele_charges, vdw_charges, vdw_radii;
};

// DFIRE's distance binning as a lookup over cells of 0.25 A^2 (DESIGN.md "bin LUT").
int dfire_bin_reference(double dist2);               // the formula, src/dfire.rs:49-53,336-337
struct DfireBinning {
    std::vector<uint8_t> lut;   // kDfireLutCells: bin at the lower edge of each 0.25 A^2 cell
    std::vector<double> step;   // kDfireSteps: step[b] = first d2 the reference puts in bin >= b
};
DfireBinning build_dfire_binning();                  // throws if the self-check fails
double dfire_interface_d2();                         // largest d2 with sqrt(d2)*2-1 <= 3.9
std::vector<uint32_t> build_packed_lut(int cells_per_unit, double eps, uint32_t zero_bins = 0);  // kPackedLutCells * cells_per_unit words
size_t dfire_bm_reach_count(const double *xyz, size_t n, double reach);   // upper bound on the atoms inside any ball of that radius (n itself below 8192 atoms)
double dfire_bm_fix_scale(double vmax, size_t reach_count, int *extra_bits_out);   // the block-major path's fixed-point units per unit of the potential; 0.0: none fits
std::vector<uint8_t> build_bm_lut(double eps_cells, uint32_t zero_bins = 0);  // kBmLutBytes codes of the block-major kernel (kernels/dfire_bm.hpp)

class Scorer {
   public:
    explicit Scorer(const ld_scorer_desc &desc);
    ~Scorer();
    Scorer(const Scorer &) = delete;
    Scorer &operator=(const Scorer &) = delete;

    int method() const { return method_; }
    bool use_anm() const { return use_anm_; }
    size_t anm_rec() const { return use_anm_ ? (size_t)pair_.rec.num_anm : 0; }
    size_t anm_lig() const { return use_anm_ ? (size_t)pair_.lig.num_anm : 0; }
    size_t pose_len() const { return 7 + anm_rec() + anm_lig(); }
    size_t num_atoms(int side) const { return side ? (size_t)pair_.lig.n : (size_t)pair_.rec.n; }
    const HostMolecule &host_molecule(int side) const { return side ? host_lig_ : host_rec_; }
    int device() const { return device_; }
    hipStream_t stream() const { return stream_; }
    void set_stream(hipStream_t s) { stream_ = s ? s : own_stream_; }  // NULL: back to the handle's own stream
    void set_capturing(bool on) { capturing_ = on; }
    // make sure no allocation happens in the next energy_batch_device call of this size
    void prepare_batch(size_t n_poses) { reserve_workspace(n_poses, false); }
    // Changes whenever a workspace block of this scorer was reallocated (a larger batch came by): a
    // hipGraph captured before that replays kernels into freed memory and must be captured again.
    uint64_t workspace_generation() const;

    // Enqueue K1 for n poses already in HBM.  active / pair_counts may be null.  d_list / d_count
    // (device): the rows to evaluate as a compacted list whose length only the device knows (the GSO
    // loop: the glowworms that moved); they must be the rows `d_active` marks.
    void energy_batch_device(size_t n, const double *d_poses, size_t stride, const uint8_t *d_active,
                             double *d_energies, uint32_t *d_pair_counts, const uint32_t *d_list = nullptr,
                             const uint32_t *d_count = nullptr);
    // Host-pointer convenience: H2D, kernels, D2H, synchronise.
    void energy_batch_host(size_t n, const double *poses, size_t stride, double *energies);

    void kernel_info(ld_kernel_info *out) const;
    // diagnostics of the last counting launch: 8x8 atom-pair blocks evaluated per pose (tiled kernel)
    void last_block_counts(size_t n, uint32_t *out_host);
    uint32_t bm_quiet_subtiles() const { return use_bm_ ? bm_quiet_subtiles_ : 0u; }
    void enable_timing(bool on);
    void pair_kernel_time(double *total_ms, uint64_t *launches);

   private:
    void upload_molecule(const ld_molecule &m, bool is_receptor, DeviceMolecule &dev, HostMolecule &host,
                         std::vector<uint32_t> &group_offsets, std::vector<uint32_t> &group_slots,
                         std::vector<uint32_t> &membrane_slots);
    void reserve_workspace(size_t n_poses, bool counts);
    void build_tiled(const ld_scorer_desc &desc);
    void build_packed(const ld_scorer_desc &desc);  // after build_tiled: shares its table, ligand and tile order
    void build_bm(const ld_scorer_desc &desc);      // after build_packed: the block-major path (rigid molecules; the ANM form for molecules that flex)
    void run_bm(size_t n, const double *d_poses, size_t stride, const uint8_t *d_active, bool counts, const uint32_t *d_list,
                const uint32_t *d_count);
    void frame_of_receptor(const ld_molecule &rec, double centre[3], double *half) const;
    struct TiledSoA {  // a molecule in tile order, SoA, padded to whole tiles
        int n_real = 0, n_tiles = 0;
        const double *x = nullptr, *y = nullptr, *z = nullptr;
        const uint32_t *tindex = nullptr;
        const int32_t *slot = nullptr;
        int num_anm = 0;
        const double *modes = nullptr;
        std::vector<double> hx, hy, hz;   // host copies, tile order (padding included)
        std::vector<uint32_t> htype;      // DFIRE type per slot of the tile order, 0xffffffff = padding
        std::vector<int32_t> hslot;       // interface-flag slot or -1
        std::vector<double> hmodes;       // host copy of `modes`: [mode][xyz][padded atoms]
    };
    void upload_tiled_molecule(const ld_molecule &m, bool is_receptor, TiledSoA &out);
    PrepareReceptorLaunch prepare_launch(const double *poses, size_t stride, const uint8_t *active, size_t n) const;
    PackedPrepareLaunch packed_prepare_launch(const double *poses, size_t stride, const uint8_t *active, size_t n) const;

    int device_ = 0;
    hipStream_t stream_ = nullptr;
    hipStream_t own_stream_ = nullptr;
    bool capturing_ = false;
    int method_ = 0;
    bool use_anm_ = false;
    DeviceArena arena_;
    PairLaunch pair_;     // receptor / ligand / table pointers filled once; batch fields per call
    TailTables tail_;
    bool use_tiled_ = false;  // DFIRE: a bounding-box culled kernel instead of all-pairs
    TiledLaunch tiled_;
    bool use_packed_ = false;  // DFIRE default: culling + packed-f32 pair test with exact f64 path (kernels/dfire_packed.hpp)
    PackedLaunch packed_;
    const uint32_t *packed_lut_full_ = nullptr;  // the LUT without elided zero bins (counting launches)
    uint32_t packed_zero_bins_ = 0;
    uint32_t bm_quiet_subtiles_ = 0;   // receptor subtiles of the block-major path whose atoms' rows of the potential are zero (build_bm)
    DeviceBuffer ws_rec_pairs_, ws_exact_;
    bool use_bm_ = false;      // DFIRE: the block-major path (kernels/dfire_bm.hpp) evaluates every batch
    BmModel bm_;
    TiledSoA tiled_lig_soa_;
    size_t bm_chunk_ = 0;      // poses per block-major pass (bounds the entry workspace)
    std::vector<float> bm_tile_radius_;   // angstrom: the ligand tiles' bounding spheres (the reach of the count-aware fixed-point scale)
    size_t bm_pass_poses(size_t n) const;
    size_t bm_sets(size_t n) const;            // workspace sets a batch of n poses needs (2 while two passes are in flight)
    hipStream_t bm_aux_stream_ = nullptr;   // the second of two passes in flight runs here
    hipEvent_t bm_fork_ = nullptr, bm_join_ = nullptr;
    int n_cus_ = 256;
    DeviceBuffer ws_bm_debug_, ws_bm_jobs_, ws_bm_job_cost_, ws_bm_job_order_, ws_bm_rt_, ws_bm_tp_count_, ws_bm_ent_row_, ws_bm_ent_mask_, ws_bm_queue_, ws_bm_ent_partial_, ws_bm_tile_sum_,
        ws_bm_tile_tested_, ws_bm_exact_fix_, ws_bm_exact_pairs_, ws_bm_amp_;
    TiledSoA tiled_rec_soa_;          // receptor in tile order (input of dfire_prepare_receptor)
    bool rec_anm_per_pose_ = false;   // receptor ANM: one receptor image per pose per launch
    DeviceBuffer ws_rec_atoms_, ws_rec_sub_, ws_rec_tile_;
    std::vector<uint32_t> type_perm_rec_, type_perm_lig_;  // DFIRE type -> number used by the tiled kernel's table layout
    std::vector<int32_t> host_slot_rec_, host_slot_lig_;  // per original atom, as uploaded to the all-pairs path
    HostMolecule host_rec_, host_lig_;
    DeviceBuffer ws_partial_, ws_flags_, ws_counts_, ws_tested_, ws_poses_, ws_energies_;
    bool timing_ = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events_;  // pool, reused
    size_t events_used_ = 0;
    double timed_ms_ = 0.0;
    uint64_t timed_launches_ = 0;
};

}  // namespace ld

struct ld_scorer {
    ld::Scorer impl;
    explicit ld_scorer(const ld_scorer_desc &d) : impl(d) {}
};
