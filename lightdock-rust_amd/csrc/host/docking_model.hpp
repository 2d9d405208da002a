// docking_model.hpp -- host-side mirror of the reference's per-molecule docking models:
// DFIREDockingModel (src/dfire.rs:104-190) and DNADockingModel (src/dna.rs:235-364).
// Built once per run on the CPU; the arrays are then uploaded to HBM by the scorer.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "lightdock_hip.h"
#include "structure.hpp"

namespace ld {

struct DockingModel {
    // common to both scoring functions
    std::vector<double> coordinates;             // n x 3 (src/dfire.rs:106)
    std::vector<uint32_t> membrane;              // atom indices of MMB.BJ beads (src/dfire.rs:107,146-149)
    std::vector<std::string> restraint_ids;      // active restraint residues found in the PDB
    std::vector<uint32_t> restraint_offsets;     // CSR over restraint_atoms, size = groups + 1
    std::vector<uint32_t> restraint_atoms;
    size_t num_anm = 0;
    std::vector<double> nmodes;                  // num_anm x n x 3
    // DFIRE
    std::vector<uint32_t> dfire_types;           // 0..167 (src/dfire.rs:105)
    // DNA
    std::vector<double> ele_charges, vdw_charges, vdw_radii;  // src/dna.rs:243-245

    size_t num_atoms() const { return coordinates.size() / 3; }
    ld_molecule view() const;  // borrowed pointers for ld_scorer_create
};

// DFIRE atom type of (residue, atom), 0..167; throws Error(LD_ERR_UNSUPPORTED) with the
// reference's panic text for an unknown residue or atom (src/dfire.rs:43,180).
uint32_t dfire_atom_type(const std::string &res_name, const std::string &atom_name);

struct DnaAtomParams { double well_depth, radius, charge; };
// DNA parameters of (residue, atom) incl. the H1/H2/H3 -> "<res>-H" rule (src/dna.rs:314-358).
// generic_fallback = PYDOCK: an atom the tables do not know is typed by the first letter of its
// name ("*-C", "*-F", "*-H", "*-N", "*-O", "*-S"; src/pydock.rs:332-345).
DnaAtomParams dna_atom_params(const std::string &res_name, const std::string &atom_name, bool generic_fallback = false);

DockingModel build_docking_model(int method, const Structure &structure,
                                 const std::vector<std::string> &active_restraints,
                                 const std::vector<std::string> &passive_restraints,
                                 const std::vector<double> &nmodes, size_t num_anm);

}  // namespace ld
