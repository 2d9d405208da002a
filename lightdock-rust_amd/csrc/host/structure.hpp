// structure.hpp -- minimal PDB structure reader for the host-side model builders.
//
// The reference parses PDB files with pdbtbx 0.11.0 (Cargo.toml:15) and walks
// chains -> residues -> atoms (src/dfire.rs:133-144, src/dna.rs:269-289); atom indices,
// restraint atom lists and the f64 summation order all follow that walk.  This reader
// produces the same walk order from fixed-column ATOM/HETATM records.
#pragma once

#include <string>
#include <vector>

namespace ld {

struct AtomRecord {
    std::string name;      // columns 13-16, trimmed
    std::string res_name;  // columns 18-20, trimmed
    std::string chain_id;  // column 22
    long res_seq = 0;      // columns 23-26
    std::string icode;     // column 27, empty when blank
    char alt_loc = ' ';    // column 17
    double x = 0, y = 0, z = 0;

    // "<chain>.<resname>.<serial><icode>", src/dfire.rs:139-142
    std::string residue_id() const;
};

struct Structure {
    std::vector<AtomRecord> atoms;  // in chain -> residue -> conformer -> atom walk order
    size_t atom_count() const { return atoms.size(); }
};

// Throws ld::Error(LD_ERR_IO) when the file cannot be read.
Structure read_pdb(const std::string &path);

}  // namespace ld
