// spatial_order.hpp -- atom ordering for the tiled pose-energy kernel.
//
// The reference walks atoms in PDB order (src/dfire.rs:325-345); the energy is a plain sum
// over pairs, so any order gives the same value up to f64 rounding.  The tiled kernel wants
// atoms that are close in space to be close in memory: leaves of 8 atoms ("subtiles") and
// runs of 8 leaves ("tiles", 64 atoms) with small bounding boxes, so that whole 8x8 and
// 64x64 blocks of pairs can be discarded with one box-distance test.
#pragma once

#include <cstdint>
#include <vector>

namespace ld {

// Returns slot -> original atom index, length = ceil(n/64)*64; UINT32_MAX marks a padding
// slot.  Built by recursive median splits along the longest axis, cut at multiples of 64
// (or 8 below 64 atoms) so only the trailing leaf is short.
// Inside each full tile a swap refinement then tightens the 8-atom subtile boxes.
std::vector<uint32_t> spatial_tile_order(const double *xyz /* n x 3 */, size_t n);

}  // namespace ld
