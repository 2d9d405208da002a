// io.hpp -- the file formats on either side of the path:
// DCparams, rec_nm.npy / lig_nm.npy, initial_positions_N.dat, setup.json.
#pragma once

#include <cstdint>
#include <map>
#include <optional>
#include <string>
#include <vector>

namespace ld {

// DFIRE::load_potentials, src/dfire.rs:236-257
std::vector<double> load_dcparams(const std::string &path);

// flat C-order <f8 array, as npyz reads it (src/bin/lightdock-rust.rs:221-252)
std::vector<double> read_npy_f64(const std::string &path);

// parse_input_coordinates, src/bin/lightdock-rust.rs:60-75; all rows must have equal length
struct Positions {
    size_t rows = 0, cols = 0;
    std::vector<double> values;  // rows x cols
};
Positions parse_positions(const std::string &path);

// the fields of SetupFile (src/bin/lightdock-rust.rs:27-48) the run actually uses
struct SetupFile {
    std::optional<uint64_t> seed;
    bool use_anm = false;
    size_t anm_rec = 0, anm_lig = 0;
    std::string receptor_pdb, ligand_pdb;
    std::optional<std::map<std::string, std::vector<std::string>>> receptor_restraints, ligand_restraints;
};
// Throws Error(LD_ERR_IO) carrying the serde-like message for a missing/mistyped field.
SetupFile read_setup(const std::string &path);

}  // namespace ld
