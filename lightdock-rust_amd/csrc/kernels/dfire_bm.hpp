// dfire_bm.hpp -- launch interface of the BLOCK-MAJOR DFIRE pose-energy path (K1, DFIRE: rigid molecules, and -- the ANM
// form -- molecules that flex by normal modes, src/dfire.rs:288-320).
//
// Same sum as src/dfire.rs:325-345, same 64x64 / 8x8 box culling and the same f32-filter-with-exact-f64-path
// numerics as dfire_packed.hpp, but the pair work is ordered by atom-pair BLOCK instead of by pose:
//
//   dfire_bm_pose    one thread per row of the pass: the pose's rotation + translation as an f32 affine map into the record
//                    frame, [row][12], 48 bytes that stay in L2 for the whole pass; the exact path's [row][8] f64 row; clears
//                    the pose's flag words, the pass's sums and the sequence's counters; ANM: the row's amplitudes
//   dfire_bm_rec_boxes  (ANM) lane = row of the pass: the flexed receptor's subtile and tile boxes, f32
//   dfire_bm_cull    persistent waves, one (ligand tile, 8 poses) item at a time: a bounding-sphere test of the tile against the
//                    receptor's tile boxes (in LDS; ANM: the row's own), then the ligand atoms posed in f32 (ANM: and flexed),
//                    boxes, 64x64 and 8x8 box tests; every surviving (ligand tile, receptor tile) pair of a pose becomes one
//                    ENTRY {row of the pass, 64-bit block mask} appended to that tile pair's list (one global atomic per
//                    tile pair per wave)
//   dfire_bm_plan    one workgroup: every tile pair's entries cut into equal parts of at most P entries
//   dfire_bm_census  one wave per (tile pair, part): per ligand-subtile row the block bits of its entries -> the estimated
//                    length of the JOB (tile pair, part, row), and the job's 16-byte record
//   dfire_bm_order   one workgroup: the jobs that have any work, longest first (counting sort into classes)
//   dfire_bm_pairs   persistent workgroups of 4 independent waves (two per CU); a wave takes a job and, for each of its 8
//                    blocks (a, b): the 64 table rows T[type_i][type_j][.] of the block are staged in the wave's slice of LDS
//                    ONCE (dense L2 -> LDS copies by LDS-DMA), the entries whose mask holds the block are compacted into
//                    batches of 64, and a batch runs lane = pose: the lane poses the 8 ligand atoms of subtile a (uniform
//                    local coordinates, its pose's affine map from the L2-resident [row][12] table; ANM: + its pose's
//                    deformation of both subtiles, modes from LDS) and walks the 64 atom pairs of the block, whose receptor
//                    atoms and table rows are wave-uniform: E = cell-zero + 1/2 - 64 d2 in packed f32 (coordinates relative
//                    to the receptor subtile's box centre), cell = (u32)E, code = lut[cell] (u8, LDS), value = row[code]
//                    (64-bit fixed point, LDS; the row's address is an instruction constant), integer add.  A flagged cell's
//                    slot holds a MARKER that names the pair (below).  An (entry, ligand subtile)'s finished sum goes to the
//                    pose's (row, ligand tile) sum by an integer atomic.  No gather ever leaves the CU.
//   dfire_bm_gather  eight lanes per row: the row's ligand-tile sums + the exact path's sum -> f64 -> the [pose][1][2]
//                    partials that pose_energy_finish folds (restraint / membrane tail, src/dfire.rs:347-361)
//
// Numerics: records are u = fl32(8 (x - c)); the ligand is posed by an f32 affine map whose error
// is part of `eps`; a LUT cell (1/16 of a unit of 4 d2) whose interval, widened by eps, holds a bin step or the cutoff is
// FLAGGED: its pairs read the row's MARKER instead of a table value (below) and are recomputed in f64 (the exact path of
// dfire_bm.hip: f64 coordinates, the reference's quaternion posing and operation order); in a block with an atom that has an
// interface-flag slot so are the pairs of bins 0 and 1 (r < 2.5 A, where the interface distance lies).  Bins, cutoff decisions
// and interface flags are therefore the reference's, bit for bit.  The liberty this path takes is the SUM: every table value
// is rounded ONCE, on the host, to 64-bit fixed point -- rint(v * 2^(44 - e - x)), 2^e >= the table's largest |value|, x = 0
// unless one ligand tile can reach more than 8191 receptor atoms (scorer.cpp, dfire_bm_fix_scale) -- and a pose's sum is
// integer adds, exact in any order.  Error model: |error of a pose's sum| <= N_pairs * 2^-(45 - e - x) units of the
// potential, times the 0.0157 of src/dfire.rs:347 on the energy: 2^-40 units for the synthetic table (e = 4), below 8e-10 on
// an energy of 1k4c's 114 k pairs in the worst case, 2-4e-12 observed.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "dfire_packed.hpp"

namespace ld {

constexpr int kBmCells = 16;                 // LUT cells per unit of 4 d2
constexpr double kBmKappa = 8.0;             // records hold fl32(kappa (x - c)): D'' = sum (du)^2 = 16 * 4 d2
constexpr int kBmCutCell = 900 * kBmCells;   // the cutoff 4 d2 = 900
constexpr int kBmLutBytes = 14592;           // cells 0 .. 14591: everything beyond kBmCutCell + eps reads "miss"
constexpr float kBmCellMax = 14591.0f;
constexpr int kBmCellZero = 14583;            // the cell of 64 d2 = 0: E = kBmCellZero + 1/2 - 64 d2 stays inside the LUT for any error below 8 cells
// A table row T[type_i][type_j][.] as the pair kernel reads it: 22 slots of 64-bit FIXED POINT (value * fix_scale, rounded
// once on the host: integer adds are exact in any order, so a pose's sum does not depend on how the launch was cut up):
// slots 0..19 = bins 0..19, slot 20 = 0 ("miss": beyond the cutoff, or a bin that is zero for the whole complex), slot 21 = the
// MARKER a flagged cell reads (the cell holds a bin step or the cutoff -- hence also the reference's read past the row at
// r = 15.0, src/dfire.rs:338).  The LUT code of a cell is the byte offset of its slot.  The marker of the block's row (i, j) is
// (64 + i * 8 + j) << 51, written into the LDS copy of the row: a lane's 64 adds of a block leave marker bits + sum, and the
// marker bits say "no flagged pair", WHICH pair, or "several" (dfire_bm.hip, the exact path) -- detection costs the pair
// loop nothing.  In a block that holds an atom with an interface-flag slot the LDS copy also carries the marker in the slots
// of bins 0 and 1 (r < 2.5 A: the only pairs that can set interface flags, src/dfire.rs:339): the exact path sets them.
constexpr int kBmRowSlots = 22;
constexpr int kBmRowBytes = kBmRowSlots * 8;
constexpr uint32_t kBmMissCode = 8 * 20;     // 160
constexpr uint32_t kBmFlagged = 8 * 21;      // 168
__host__ __device__ inline uint32_t bm_code_of_bin(uint32_t bin) { return 8 * bin; }   // bins 0..19 -> 0..152
constexpr int kBmMarkerShift = 51;           // a lane keeps two sums of 32 pairs each: below 2^50 under it, 32 markers of at most 127 above it: 63 bits
constexpr int kBmJobRows = 8;                // a job = (tile pair, part of its entries, ligand subtile a): the blocks (a, 0..7); one partial sum per (entry, a)
constexpr int kBmPartEntries = 1792;         // entries of a tile pair in one job: what a wave's share of the LDS holds at 3 bytes an entry (the block bits, the item list)
constexpr int kBmEntryMask = 0x7ff;          // an entry's number in its part
constexpr int kBmPassQuantum = 1024;         // poses per pass: a multiple of this
constexpr int kBmOpsFloats = 36;             // BmModel::rec_ops: Rs[4][2], Rz[4][2], Ry[4][2], Rx[4][2], cx, cy, cz, 0
constexpr int kBmMaxModes = 10;              // normal modes per molecule the ANM form of the path takes (the reference's examples: 10 + 10, src/dfire.rs:288-320)
constexpr int kBmModeFloats = 8 * 3 * kBmMaxModes;   // a subtile's modes as a batch reads them: ((atom pair p * 3 + coordinate) * kBmMaxModes + mode) * 2 + atom of the pair
constexpr int kBmAmpFloats = 24;             // a row's amplitudes as the pair kernel loads them: receptor modes 0..9, ligand modes 10..19, [20] != 0: a WILD pose,
                                             // [21]: how far the row's amplitudes can move a ligand atom (record units; the culling kernel's sphere test)
constexpr int kBmAnmPartEntries = 1024;      // entries of a tile pair in one job of the ANM form (its LDS holds two subtiles' modes where the other keeps the entries' rows of the pass)
constexpr float kBmWildUnits = 128.0f;       // a pose whose amplitudes could move a coordinate of an atom further than this (record units: 16 A) is WILD: every
                                             // pair of its blocks goes to the exact path (the f32 arithmetic's error bound covers deformations up to here).
                                             // The test is Cauchy-Schwarz per molecule: |sum_k a_k m_k| <= |a|_2 x max over atoms and coordinates of |m|_2
                                             // (BmModel::rec_mode_norm) -- 14.1 A at most over the starting poses of the reference's ANM examples (1czy),
                                             // 7.8 / 10.7 A for 2uuy.  (Until round 6: 256 units against sum_k |a_k| max |m_k|, 21.5 A for 2uuy; the bound on
                                             // the distance arithmetic's operands, and with it the LUT's eps, is what W buys: 0.87 -> 0.49 cells for 2uuy,
                                             // one flagged cell per bin step instead of three.)
constexpr int kBmCubeRows = 64;              // table rows of a block: 8 ligand x 8 receptor atoms
constexpr int kBmCubeBytes = kBmCubeRows * kBmRowBytes;
constexpr int kBmTypes = 170;                // 169 DFIRE types + one all-zero type for padding atoms
constexpr int kBmWaves = 4;                  // waves per dfire_bm_pairs workgroup: each is compiled with ITS cube's LDS address as a constant
constexpr int kBmGroupsPerCu = 2;            // workgroups of dfire_bm_pairs per CU (what 160 KB of LDS hold)
constexpr int kBmWavesPerCu = kBmWaves * kBmGroupsPerCu;
constexpr int kBmQueuePairs = 8 * kBmPartEntries + 4096 + 512;   // per wave (global memory): 64-bit items, flagged pairs waiting for the exact path: what
                                             // a job can push (one per item), what it may start with, a round of bm_recheck ...
constexpr int kBmQueueCap = kBmQueuePairs + 8 * kBmPartEntries + 64;   // ... and behind them (entry, block) items whose flagged pairs have to be found again
constexpr double kBmFixLimit = 1024.0;       // |table value| the fixed-point sums take; the scale is 2^(44 - e - x), 2^e >= the table's largest |value|:
                                             // 32 pairs of a block stay below 2^49, the 512 of an (entry, ligand subtile) partial below 2^53, a (row, ligand
                                             // tile) sum below 2^63 (x, dfire_bm_fix_scale).  At the limit a value still resolves to 2^-34 (6e-11): a table
                                             // with larger entries (DFIRE's are below 20) runs the pose-major kernels, whose sums are f64
constexpr int kBmCounters = 8;               // words behind tp_count, zeroed per launch: (tile pair, part) pairs listed, jobs drawn, entries per
                                             // part, jobs listed; behind them kBmCullQueueWords item counters of dfire_bm_cull
constexpr int kBmCullQueueWords = 256;
constexpr int kBmCostClasses = 80;           // jobs are drawn in classes of estimated length, longest first
// dfire_bm_cull's LDS: the receptor's boxes (one per tile and 8 per tile of 32 bytes) and per wave of the workgroup a hit list
// -- 14 bytes a hit, room for `hit tiles` poses that reach every receptor tile: a pose adds at most one hit per receptor tile, and
// the list is flushed (one atomic per tile pair) when the next pose might not fit -- plus two words per receptor tile.  The list
// shrinks for a receptor whose boxes leave less room; one of more than ~430 tiles (27 000 atoms) does not fit 160 KB with any
// list: such a complex stays with the pose-major kernels (scorer.cpp, build_bm).
#ifndef LD_BM_CULL_WAVES
#define LD_BM_CULL_WAVES 4
#endif
constexpr int kBmCullWaves = LD_BM_CULL_WAVES;   // independent waves per dfire_bm_cull workgroup
constexpr size_t kBmLdsPerCu = 160 * 1024;
__host__ __device__ inline size_t bm_cull_lds_for(int n_rt, int hit_tiles) {
    const int hit_cap = hit_tiles * n_rt > 192 ? hit_tiles * n_rt : 192;
    return (size_t)n_rt * 9 * 32 + (size_t)kBmCullWaves * (((size_t)hit_cap * 14 + (size_t)n_rt * 8 + 15) / 16 * 16);
}
__host__ __device__ inline int bm_cull_hit_tiles(int n_rt) {   // 4 while they fit (every receptor up to 266 tiles), then 3, 2, 1
    int k = 4;
    while (k > 1 && bm_cull_lds_for(n_rt, k) + 1024 > kBmLdsPerCu) k--;
    return k;
}
__host__ __device__ inline int bm_cull_hit_cap(int n_rt) { const int k = bm_cull_hit_tiles(n_rt); return k * n_rt > 192 ? k * n_rt : 192; }
__host__ __device__ inline size_t bm_cull_wave_lds(int n_rt) { return ((size_t)bm_cull_hit_cap(n_rt) * 14 + (size_t)n_rt * 8 + 15) / 16 * 16; }
__host__ __device__ inline size_t bm_cull_lds_bytes(int n_rt) { return bm_cull_lds_for(n_rt, bm_cull_hit_tiles(n_rt)); }
constexpr float kBmBoxCutUnits2 = 14400.0f * 1.00005f;   // (8 * 15 A)^2 in record units, padded for the rounding of the box test: a receptor box's reach unless BmModel::rec_sub says less
constexpr size_t kBmMaxPassPoses = 262144;   // a job keeps its entries' rows of the pass as 18-bit numbers (LDS)

struct BmModel {
    // receptor (static image in the kappa = 8 frame; no receptor ANM on this path)
    int rec_n_real = 0, rec_n_tiles = 0;
    const PackedRecPair *rec_pairs = nullptr;   // [n_tiles*32]
    const TiledBox *rec_sub = nullptr;          // [n_tiles*8]; pad0 = the subtile's reach for the culling kernel's box test, squared record units (scorer.cpp, build_bm)
    const TiledBox *rec_tile = nullptr;         // [n_tiles]; pad0 = the largest reach of its subtiles
    const uint32_t *rec_rowoff = nullptr;       // [n_tiles*64]: byte offset of the atom's type column in a table row block
    const float *rec_ops = nullptr;             // [n_tiles*8][kBmOpsFloats]: a receptor subtile as the pair kernel's batches take it -- the centre c of its box
                                                // and per pair record the packed operands seed - |r - c|^2, 2 (r - c)_z, _y, _x (f32, formed on the host by
                                                // the operations bm_recheck repeats on the device): a block's set-up is three scalar loads
    const double *rec_x = nullptr, *rec_y = nullptr, *rec_z = nullptr;  // f64, tile order (exact path)
    const uint32_t *rec_tindex = nullptr;       // tile order: tiled_rec_term (exact path reads the patch table)
    const int32_t *rec_slot = nullptr;
    int rec_flag_words = 0;
    // DFIRE with normal modes (src/dfire.rs:288-320: both molecules flex per pose).  The block-major form: the receptor's boxes per
    // pose from dfire_packed_prepare (BmLaunch::anm_sub / anm_tile), the ligand's atoms flexed by the culling kernel; a lane of a
    // batch flexes its pose's 8 + 8 atoms itself -- the two subtiles' modes lie in LDS, the pose's amplitudes come with its map.
    int anm_rec = 0, anm_lig = 0;               // modes of either molecule; 0 + 0: the rigid form
    const float *rec_modes_f32 = nullptr;       // [rec subtiles][kBmModeFloats]: kappa x the modes, f32, in the batch's order (absent modes 0)
    const float *lig_modes_f32 = nullptr;       // [lig subtiles][kBmModeFloats]
    const float *lig_modes_atom = nullptr;      // [lig atoms (tile order)][32]: the same modes per atom, [mode][x y z] (the culling kernel: lane = atom)
    const float *rec_modes_atom = nullptr;      // [rec atoms (tile order)][32]: the same for the receptor (dfire_bm_rec_boxes)
    float rec_box_pad = 0.f;                    // record units: how far an f32-flexed receptor atom of a pose that is not wild can be from the exact one, per coordinate
    const double *rec_modes_exact = nullptr;    // f64 [atom (tile order)][kBmMaxModes][x y z]: what the exact path reads of an atom's modes, 240 contiguous bytes
    const double *lig_modes_exact = nullptr;    //   (src/dfire.rs:288-320; absent modes 0, never read)
    float rec_mode_norm = 0.f, lig_mode_norm = 0.f;   // kappa x max over atoms and coordinates of the 2-norm of the ten mode components (f32 values, rounded up):
                                                      // |amplitudes|_2 x this bounds every coordinate's deformation and every partial sum of it (Cauchy-Schwarz)
    float lig_mode_norm_vec = 0.f;                    // the same over an atom's thirty components: |amplitudes|_2 x this bounds how far a ligand atom moves
    // ligand
    TiledLigand lig;                            // f64, tile order (exact path, overflow tiles)
    const double *lig_exact = nullptr, *rec_exact = nullptr;   // [atom][4]: x, y, z (f64) and {table term, interface-flag slot}: what the exact path reads of an atom, in two loads
    const float *lig_local = nullptr;           // [n_tiles*64][4]: x, y, z (angstrom, f32), 1.0 = real atom
    const uint32_t *lig_rowbase = nullptr;      // [n_tiles*64]: byte offset of the atom's type block in `rows`
    const float *lig_tile_sphere = nullptr;     // [n_tiles][4]: centre (local, angstrom) and radius (record units, rounded up) of a sphere around the tile
    // tables
    const long long *rows = nullptr;            // [kBmTypes lig][kBmTypes rec][kBmRowSlots], fixed point
    const long long *rows_ones = nullptr;       // the same with 1 in every bin's slot: a counting launch sums pairs
    double fix_scale = 0.0;                     // fixed-point units per unit of the potential
    const uint8_t *lut = nullptr;               // kBmLutBytes codes, cell' = floor(kBmCellZero + 1/2 - 64 d2)
    const uint8_t *lut_full = nullptr;          // the same without elided zero bins (counting launches)
    const uint8_t *rec_sub_tracked = nullptr;   // [rec subtiles]: 1 = holds an atom with a flag slot
    const uint8_t *lig_sub_tracked = nullptr;   // [lig subtiles]
    const double *table = nullptr;              // 2 x 2 x 4 patches (dfire_tiled.hpp): the exact path's table
    const double *bin_step = nullptr;
    double iface_scaled = 0.0;                  // 4 * iface_d2
    double cx = 0, cy = 0, cz = 0;
    float ubound = 0.f;                         // |u| beyond this: the atom is further than the cutoff from every receptor atom
                                                // (the frame holds the receptor + 16 A) and joins no box
    float box_pad = 0.f;                        // absolute widening of the ligand boxes (error of the f32 affine map)
};

struct BmLaunch {
    BmModel m;
    const double *poses = nullptr;
    size_t stride = 0;
    const uint8_t *active = nullptr;
    const uint32_t *pose_list = nullptr;   // GSO: compacted rows + device-side count (both null for a plain batch)
    const uint32_t *pose_count = nullptr;
    size_t first = 0;                      // this pass covers listed rows [first, first + n_poses); row r of the pass = listed row first + r
    size_t n_poses = 0;
    size_t cap = 0;                        // entries per tile pair the workspace has room for (>= n_poses)
    int count_mode = 0;                    // 1: a counting launch -- full LUT, rows of ones, the sums are in-cutoff pair counts (-> count_partial)
    // workspace of the pass; "row" = row of the pass
    float *amp = nullptr;                  // ANM: [row][kBmAmpFloats] f32 amplitudes (dfire_bm_pose), or nullptr
    double *amp_exact = nullptr;           // ANM: [row][2 * kBmMaxModes] the pose's own f64 amplitudes, receptor's then ligand's (the exact path: 16-byte loads)
    TiledBox *anm_sub = nullptr;           // ANM: the flexed receptor's subtile boxes per ROW [row][n_rt * 8], as {lo, -hi} pairs (dfire_bm_rec_boxes) ...
    TiledBox *anm_tile = nullptr;          // ... and tile boxes [row][n_rt]
    uint32_t part_cap = 0;                 // entries per part at most (kBmPartEntries, or kBmAnmPartEntries for the ANM form)
    float *rt = nullptr;                   // [row][12]: the pose as an f32 affine map
    double *rt_exact = nullptr;            // [row][8]: the pose as the exact path reads it: t, q (f64, the launch's own numbers), its index in the launch
    uint32_t *tp_count = nullptr;          // [n_tile_pairs], zeroed per launch
    uint32_t *ent_row = nullptr;           // [tile pair][cap]
    unsigned long long *ent_mask = nullptr;  // [tile pair][cap]
    long long *ent_partial = nullptr;      // [wave of dfire_bm_pairs][kBmPartEntries], fixed point: the partial sums of the wave's CURRENT job, one per entry of the
                                           // part (8 KB a wave that never leave the L2; until round 5 a sparse [tile pair][8][cap] array: 33 M scattered HBM read-modify-writes a launch)
    uint32_t *jobs = nullptr;              // [(tile pair, part)][2]: tile pair, first entry; written by dfire_bm_plan
    uint32_t *job_count = nullptr;         // [kBmCounters], zeroed per launch: (tile pair, part) pairs listed, jobs drawn (job_next = job_count + 1),
                                           // entries per part, jobs listed in job_order
    uint32_t *job_next = nullptr;
    uint32_t *job_cost = nullptr;          // [(tile pair, part)][kBmJobRows]: estimated length of the job (0: no block of that row in any entry)
    uint32_t *job_rec = nullptr;           // [(tile pair, part)][kBmJobRows][4]: tile pair, first entry, one past the last, row -- written by dfire_bm_census
    uint32_t *job_order = nullptr;         // the jobs ((tile pair, part) index * kBmJobRows + row) that have work, longest first
    unsigned long long *queue = nullptr;   // [waves of dfire_bm_pairs][kBmQueueCap]: pairs waiting for the exact path
    int pairs_groups = 0;                  // CUs dfire_bm_pairs / dfire_bm_cull may fill (0: the 256 of an MI355X)
    unsigned long long *debug = nullptr;   // diagnostics (LIGHTDOCK_BM_DEBUG): per wave of dfire_bm_pairs {start, end (100 MHz), jobs, batches, ...}
    long long *tile_sum = nullptr;         // [lig tile][cap rows], zeroed by dfire_bm_pose: the pair kernel adds every finished (entry, ligand subtile) sum (fixed point);
                                           // rows contiguous: a batch's lanes are runs of consecutive rows, whose atomics then share 64-byte requests
    uint32_t *tile_tested = nullptr;       // [row][lig tiles]: 8x8 blocks let through (diagnostics) or nullptr
    long long *exact_fix = nullptr;        // [row], zeroed by dfire_bm_pose: exact-path sum, fixed point
    uint32_t *exact_pairs = nullptr;       // [row]: pairs recomputed in f64 (diagnostics) or nullptr
    uint32_t *flags = nullptr;             // [pose][rec words + lig words], zeroed by dfire_bm_pose
    double *partial = nullptr;             // out: [pose][1][2] for pose_energy_finish
    uint32_t *count_partial = nullptr;     // out: [pose][1] or nullptr
    uint32_t *tested_partial = nullptr;    // out: [pose][1] or nullptr
    uint32_t *exact_partial = nullptr;     // out: [pose][1] or nullptr
};

// Bound on |D''_f32 - 64 d2 - 1/2| (LUT cells) for a ligand atom posed by the f32 affine map and a receptor record,
// both inside `ubound` (record units), pairs within 1100 units of 4 d2; `lig_extent` = largest |local coordinate| of
// the ligand (angstrom).  Twice the derived bound.
double dfire_bm_error_bound(double ubound, double lig_extent, bool anm = false);   // anm: molecules that flex (poses that are not wild)
// Largest distance (record units) between an f32-posed ligand atom inside ubound and its exactly posed position.
double dfire_bm_pose_error(double ubound, double lig_extent, bool anm = false);

size_t bm_pairs_lds_bytes();
hipError_t launch_bm_pose(const BmLaunch &t, hipStream_t stream);
hipError_t launch_bm_cull(const BmLaunch &t, hipStream_t stream);
hipError_t launch_bm_pairs(const BmLaunch &t, hipStream_t stream);
hipError_t launch_bm_gather(const BmLaunch &t, hipStream_t stream);

}  // namespace ld
