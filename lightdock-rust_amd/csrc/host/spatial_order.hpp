// spatial_order.hpp -- atom ordering for the tiled pose-energy kernel.
//
// The reference walks atoms in PDB order (src/dfire.rs:325-345); the energy is a plain sum
// over pairs, so any order gives the same value up to f64 rounding.  The tiled kernel wants
// atoms that are close in space to be close in memory: leaves of 8 atoms ("subtiles") and
// runs of 8 leaves ("tiles", 64 atoms) with small bounding boxes, so that whole 8x8 and
// 64x64 blocks of pairs can be discarded with one box-distance test.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace ld {

// Returns slot -> original atom index, length = ceil(n/64)*64; UINT32_MAX marks a padding
// slot.  Built by recursive median splits along the longest axis, cut at multiples of 64
// (or 8 below 64 atoms) so only the trailing leaf is short.
// Inside each full tile, and across consecutive tiles, a swap refinement then tightens the 8-atom subtile boxes; sweeps over ALL
// full subtiles (each against its nearest ones: a smooth cost first, then the box cost) and a regrouping of the tiles by whole
// subtiles follow (round 6: 10 % fewer 8x8 blocks per 1k4c pose; tools/cluster_sim.py).  Deterministic.
std::vector<uint32_t> spatial_tile_order(const double *xyz /* n x 3 */, size_t n);

// Renumbering of the DFIRE atom types (0..168) of one molecule for the tiled kernel's table
// layout, which keeps the potential of two consecutive type numbers in the same 128-byte patch
// (kernels/dfire_tiled.hpp): types that sit close together inside the subtiles of `order` (bonded
// atoms of one residue, mostly) become the pairs (2k, 2k+1).  Greedy matching on a closeness-
// weighted co-occurrence count; deterministic.  Returns old type -> new type, a bijection.
std::vector<uint32_t> pair_types_for_patches(const double *xyz, const uint32_t *types, const std::vector<uint32_t> &order,
                                             uint32_t n_types);

// Second refinement of a tile order, aware of the type pairing: inside every full tile atoms are
// swapped between subtiles while that lowers  sum over subtiles of  box cost x (1 + mu x share of
// atoms whose patch partner type (number ^ 1) is not in the subtile).  With mu = 0.15 this costs no
// 8x8 blocks on the examples and removes another 8 % of the gather lines (bonded atoms that the
// median splits had separated come back together).
void refine_order_for_pairs(const double *xyz, const uint32_t *types, const std::vector<uint32_t> &perm,
                            std::vector<uint32_t> &order, size_t n, double mu);

// What the DFIRE scorer actually uses for one molecule: spatial_tile_order, then the type pairing,
// the pairing-aware refinement (mu = 0.15) and the final pairing.
struct DfireTileLayout {
    std::vector<uint32_t> order;      // slot -> atom, UINT32_MAX = padding
    std::vector<uint32_t> type_perm;  // DFIRE type -> number in the patch layout, 169 entries
};
DfireTileLayout dfire_tile_layout(const double *xyz, const uint32_t *types, size_t n);

}  // namespace ld
