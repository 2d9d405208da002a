#include "docking_model.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>
#include <unordered_map>

#include "error.hpp"

namespace ld {

namespace {

// DFIRE's 167 protein heavy-atom types are numbered residue by residue: every residue owns
// a consecutive block starting at `first_type`, in the atom order listed here; the
// membrane bead MMB.BJ is type 167 and the dummy MMY.DU aliases ALA.N.  This is the content
// of r3_to_numerical / ATOMNUMBER / ATOMRES (src/dfire.rs:18-46,56-77,80-101) in closed form.
struct ResidueBlock {
    const char *res;
    uint32_t first_type;
    const char *atoms;  // space separated, block order
};
const ResidueBlock kDfireBlocks[] = {
    {"CYS", 0, "N CA C O CB SG"},
    {"MET", 6, "N CA C O CB CG SD CE"},
    {"PHE", 14, "N CA C O CB CG CD1 CD2 CE1 CE2 CZ"},
    {"ILE", 25, "N CA C O CB CG1 CG2 CD1"},
    {"LEU", 33, "N CA C O CB CG CD1 CD2"},
    {"VAL", 41, "N CA C O CB CG1 CG2"},
    {"TRP", 48, "N CA C O CB CG CD1 CD2 CE2 NE1 CE3 CZ3 CH2 CZ2"},
    {"TYR", 62, "N CA C O CB CG CD1 CD2 CE1 CE2 CZ OH"},
    {"ALA", 74, "N CA C O CB"},
    {"GLY", 79, "N CA C O"},
    {"THR", 83, "N CA C O CB OG1 CG2"},
    {"SER", 90, "N CA C O CB OG"},
    {"GLN", 96, "N CA C O CB CG CD OE1 NE2"},
    {"ASN", 105, "N CA C O CB CG OD1 ND2"},
    {"GLU", 113, "N CA C O CB CG CD OE1 OE2"},
    {"ASP", 122, "N CA C O CB CG OD1 OD2"},
    {"HIS", 130, "N CA C O CB CG ND1 CD2 CE1 NE2"},
    {"ARG", 140, "N CA C O CB CG CD NE CZ NH1 NH2"},
    {"LYS", 151, "N CA C O CB CG CD CE NZ"},
    {"PRO", 160, "N CA C O CB CG CD"},
    {"MMB", 167, "BJ"},
    {"MMY", 74, "DU"},
};

const std::unordered_map<std::string, uint32_t> &dfire_type_map() {
    static const std::unordered_map<std::string, uint32_t> map = [] {
        std::unordered_map<std::string, uint32_t> m;
        for (const ResidueBlock &b : kDfireBlocks) {
            std::istringstream names(b.atoms);
            std::string atom;
            uint32_t t = b.first_type;
            while (names >> atom) m.emplace(std::string(b.res) + "/" + atom, t++);
        }
        return m;
    }();
    return map;
}

bool dfire_known_residue(const std::string &res) {
    for (const ResidueBlock &b : kDfireBlocks)
        if (res == b.res) return true;
    return false;
}

struct DnaRecord {
    const char *key;
    double well_depth, radius, charge;
};
const DnaRecord kDnaRecords[] = {
#include "dna_params.inc"
};
constexpr size_t kNumDnaRecords = sizeof(kDnaRecords) / sizeof(kDnaRecords[0]);

const DnaRecord *find_dna_record(const std::string &key) {
    const DnaRecord *end = kDnaRecords + kNumDnaRecords;
    const DnaRecord *it = std::lower_bound(kDnaRecords, end, key, [](const DnaRecord &r, const std::string &k) {
        return std::strcmp(r.key, k.c_str()) < 0;
    });
    return (it != end && key == it->key) ? it : nullptr;
}

}  // namespace

uint32_t dfire_atom_type(const std::string &res_name, const std::string &atom_name) {
    if (!dfire_known_residue(res_name))
        throw Error(LD_ERR_UNSUPPORTED, "Residue name not supported in DFIRE scoring function");
    const auto &m = dfire_type_map();
    auto it = m.find(res_name + "/" + atom_name);
    if (it == m.end()) throw Error(LD_ERR_UNSUPPORTED, "Not supported atom type \"" + res_name + atom_name + "\"");
    return it->second;
}

DnaAtomParams dna_atom_params(const std::string &res_name, const std::string &atom_name, bool generic_fallback) {
    const std::string who = generic_fallback ? "PYDOCK" : "DNA";
    std::string key = res_name + "-" + atom_name;
    const DnaRecord *rec = find_dna_record(key);
    if (!rec) {
        if (atom_name == "H1" || atom_name == "H2" || atom_name == "H3") {
            key = res_name + "-H";  // N-terminal hydrogens fall back to the backbone amide H
            rec = find_dna_record(key);
        } else if (generic_fallback) {
            // the six generic records PYDOCK adds to the tables: AMBER type = the element, charge below;
            // well depth / radius are those of that AMBER type (borrowed from a record that has it)
            static const struct { char element; const char *like; double charge; } kGeneric[] = {
                {'C', "ALA-C", 0.5973}, {'F', nullptr, -0.342}, {'H', "ALA-H", 0.2719},
                {'N', "ALA-N", -0.4157}, {'O', "ALA-O", -0.5679}, {'S', "MET-SD", -0.2737}};
            if (atom_name.empty()) throw Error(LD_ERR_UNSUPPORTED, "PYDOCK Error: Atom element could not be guessed from [\"\"]");
            key = std::string("*-") + atom_name[0];
            for (const auto &g : kGeneric) {
                if (g.element != atom_name[0]) continue;
                if (g.like == nullptr) return DnaAtomParams{0.061, 1.75, g.charge};  // AMBER "F" (src/pydock.rs VDW tables)
                const DnaRecord *like = find_dna_record(g.like);
                return DnaAtomParams{like->well_depth, like->radius, g.charge};
            }
        }
    }
    if (!rec) throw Error(LD_ERR_UNSUPPORTED, who + " Error: Atom [\"" + key + "\"] not supported");
    if (std::isnan(rec->charge))
        throw Error(LD_ERR_UNSUPPORTED, who + " Error: Atom [\"" + key + "\"] electrostatics charge not found");
    if (std::isnan(rec->well_depth))
        throw Error(LD_ERR_UNSUPPORTED, who + " Error: Atom [\"" + key + "\"] VDW charge not found");
    if (std::isnan(rec->radius))
        throw Error(LD_ERR_UNSUPPORTED, who + " Error: Atom [\"" + key + "\"] VDW radius not found");
    return DnaAtomParams{rec->well_depth, rec->radius, rec->charge};
}

DockingModel build_docking_model(int method, const Structure &structure,
                                 const std::vector<std::string> &active_restraints,
                                 const std::vector<std::string> &passive_restraints,
                                 const std::vector<double> &nmodes, size_t num_anm) {
    (void)passive_restraints;  // stored but never read by the reference's energy (src/dfire.rs:164-175)
    DockingModel m;
    const size_t n = structure.atom_count();
    m.coordinates.reserve(3 * n);
    m.num_anm = num_anm;
    m.nmodes = nmodes;

    std::unordered_map<std::string, size_t> group_of;  // restraint residue id -> group
    std::vector<std::vector<uint32_t>> groups;

    for (size_t i = 0; i < n; i++) {
        const AtomRecord &a = structure.atoms[i];
        if (a.res_name == "MMB" && a.name == "BJ") m.membrane.push_back(static_cast<uint32_t>(i));

        const std::string rid = a.residue_id();
        if (std::find(active_restraints.begin(), active_restraints.end(), rid) != active_restraints.end()) {
            auto ins = group_of.emplace(rid, groups.size());
            if (ins.second) {
                groups.emplace_back();
                m.restraint_ids.push_back(rid);
            }
            groups[ins.first->second].push_back(static_cast<uint32_t>(i));
        }

        if (method == LD_METHOD_DFIRE) {
            m.dfire_types.push_back(dfire_atom_type(a.res_name, a.name));
        } else if (method == LD_METHOD_DNA || method == LD_METHOD_PYDOCK) {
            DnaAtomParams p = dna_atom_params(a.res_name, a.name, method == LD_METHOD_PYDOCK);
            m.ele_charges.push_back(p.charge);
            m.vdw_charges.push_back(p.well_depth);
            m.vdw_radii.push_back(p.radius);
        } else {
            throw Error(LD_ERR_UNSUPPORTED, "Error: method not supported");
        }
        m.coordinates.push_back(a.x);
        m.coordinates.push_back(a.y);
        m.coordinates.push_back(a.z);
    }

    m.restraint_offsets.push_back(0);
    for (const auto &g : groups) {
        m.restraint_atoms.insert(m.restraint_atoms.end(), g.begin(), g.end());
        m.restraint_offsets.push_back(static_cast<uint32_t>(m.restraint_atoms.size()));
    }
    return m;
}

ld_molecule DockingModel::view() const {
    ld_molecule v;
    std::memset(&v, 0, sizeof v);
    v.n_atoms = num_atoms();
    v.coordinates = coordinates.data();
    v.dfire_types = dfire_types.empty() ? nullptr : dfire_types.data();
    v.ele_charges = ele_charges.empty() ? nullptr : ele_charges.data();
    v.vdw_charges = vdw_charges.empty() ? nullptr : vdw_charges.data();
    v.vdw_radii = vdw_radii.empty() ? nullptr : vdw_radii.data();
    v.n_membrane = membrane.size();
    v.membrane = membrane.data();
    v.n_restraint_groups = restraint_offsets.size() - 1;
    v.restraint_offsets = restraint_offsets.data();
    v.restraint_atoms = restraint_atoms.data();
    v.num_anm = num_anm;
    v.nmodes = nmodes.empty() ? nullptr : nmodes.data();
    return v;
}

}  // namespace ld
