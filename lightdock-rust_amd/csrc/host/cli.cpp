// cli.cpp -- `lightdock-rust <setup.json> <initial_positions_N.dat> <steps> <method>` on the
// HIP engine.  Observable behaviour follows src/bin/lightdock-rust.rs:77-333: same argv,
// same path rules (PDBs next to setup.json with the "lightdock_" prefix; swarm_N/,
// rec_nm.npy, lig_nm.npy and $LIGHTDOCK_DATA|data/DCparams relative to the CWD), same
// stdout lines, usage errors to stderr with exit status 0.  Panics of the reference
// surface as ld::Error -> exit status 101 in ld_cli_main.
#include "cli.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../gso.hpp"
#include "../scorer.hpp"
#include "docking_model.hpp"
#include "error.hpp"
#include "io.hpp"

namespace ld {

namespace {

// `{:?}` of a str for the characters that occur in paths
std::string debug_quote(const std::string &s) {
    std::string out = "\"";
    for (char c : s) {
        if (c == '"' || c == '\\') out.push_back('\\');
        out.push_back(c);
    }
    out.push_back('"');
    return out;
}

bool parse_u32(const std::string &s, uint32_t &out) {  // u32::from_str
    size_t i = (!s.empty() && s[0] == '+') ? 1 : 0;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); i++) {
        if (!std::isdigit((unsigned char)s[i])) return false;
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > 0xffffffffULL) return false;
    }
    out = (uint32_t)v;
    return true;
}

// parse_swarm_id, src/bin/lightdock-rust.rs:150-156
bool parse_swarm_id(const std::string &path, int32_t &id) {
    size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const std::string prefix = "initial_positions_", suffix = ".dat";
    if (name.size() <= prefix.size() + suffix.size()) return false;
    if (name.compare(0, prefix.size(), prefix) != 0) return false;
    if (name.compare(name.size() - suffix.size(), suffix.size(), suffix) != 0) return false;
    std::string num = name.substr(prefix.size(), name.size() - prefix.size() - suffix.size());
    size_t i = (num[0] == '+' || num[0] == '-') ? 1 : 0;
    if (i >= num.size()) return false;
    for (size_t k = i; k < num.size(); k++)
        if (!std::isdigit((unsigned char)num[k])) return false;
    errno = 0;
    long long v = std::strtoll(num.c_str(), nullptr, 10);
    if (errno || v > 2147483647LL || v < -2147483648LL) return false;
    id = (int32_t)v;
    return true;
}

const std::vector<std::string> &restraint_list(const std::optional<std::map<std::string, std::vector<std::string>>> &m,
                                               const char *which) {
    static const std::vector<std::string> empty;
    if (!m) return empty;
    auto it = m->find(which);
    // restraints["active"] on a map without the key panics in the reference (bin:257-272)
    if (it == m->end()) throw Error(LD_ERR_INVALID, std::string("restraints map has no \"") + which + "\" list");
    return it->second;
}

}  // namespace

int cli_main(int argc, char **argv) {
    // one swarm per process: 200 poses per launch, tune the scorer for latency (scorer.cpp, build_tiled)
    setenv("LIGHTDOCK_TILED_LATENCY", "1", /* overwrite */ 0);
    if (argc != 5) {
        std::fprintf(stderr, "Wrong command line. Usage: %s setup_filename swarm_filename steps method\n",
                     argc > 0 ? argv[0] : "lightdock-hip");
        return 0;
    }
    const std::string setup_filename = argv[1], swarm_filename = argv[2];
    uint32_t steps = 0;
    if (!parse_u32(argv[3], steps)) {
        std::fprintf(stderr, "Error: steps argument must be a number\n");
        return 0;
    }
    std::string method_type = argv[4];
    std::transform(method_type.begin(), method_type.end(), method_type.begin(), [](unsigned char c) { return std::tolower(c); });
    int method;
    const char *method_name;
    if (method_type == "dfire") { method = LD_METHOD_DFIRE; method_name = "DFIRE"; }
    else if (method_type == "dna") { method = LD_METHOD_DNA; method_name = "DNA"; }
    else if (method_type == "pydock") { method = LD_METHOD_PYDOCK; method_name = "PYDOCK"; }
    else {
        std::fprintf(stderr, "Error: method not supported\n");
        return 0;
    }

    SetupFile setup;
    try {
        setup = read_setup(setup_filename);
    } catch (const Error &e) {
        std::fprintf(stderr, "Error reading setup file [%s]: %s\n", debug_quote(setup_filename).c_str(),
                     debug_quote(e.what()).c_str());
        return 0;
    }
    std::string simulation_path;
    {
        size_t slash = setup_filename.find_last_of('/');
        if (slash != std::string::npos) simulation_path = slash == 0 ? "/" : setup_filename.substr(0, slash);
    }

    const uint64_t seed = setup.seed ? *setup.seed : 324324ULL;  // DEFAULT_SEED
    std::printf("Reading starting positions from %s\n", debug_quote(swarm_filename).c_str());
    int32_t swarm_id = 0;
    if (!parse_swarm_id(swarm_filename, swarm_id)) throw Error(LD_ERR_INVALID, "Could not parse swarm from swarm filename");
    std::printf("Swarm ID %d\n", swarm_id);
    const std::string swarm_directory = "swarm_" + std::to_string(swarm_id);
    struct stat sb;
    if (stat(swarm_directory.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) {
        std::fprintf(stderr, "Output directory does not exist for swarm %d, creating it\n", swarm_id);
        if (mkdir(swarm_directory.c_str(), 0777) != 0) throw Error(LD_ERR_IO, "Error creating directory");
    }
    std::printf("Writing to swarm dir %s\n", debug_quote(swarm_directory).c_str());
    Positions positions = parse_positions(swarm_filename);

    const std::string prefix = simulation_path.empty() ? std::string("lightdock_") : simulation_path + "/lightdock_";
    const std::string receptor_filename = prefix + setup.receptor_pdb;
    std::printf("Reading receptor input structure: %s\n", receptor_filename.c_str());
    Structure receptor = read_pdb(receptor_filename);
    const std::string ligand_filename = prefix + setup.ligand_pdb;
    std::printf("Reading ligand input structure: %s\n", ligand_filename.c_str());
    Structure ligand = read_pdb(ligand_filename);

    std::vector<double> rec_nm, lig_nm;
    if (setup.use_anm) {
        if (setup.anm_rec > 0) {
            rec_nm = read_npy_f64("rec_nm.npy");
            if (rec_nm.size() != receptor.atom_count() * 3 * setup.anm_rec)
                throw Error(LD_ERR_INVALID, "Number of read ANM in receptor does not correspond to the number of atoms");
        }
        if (setup.anm_lig > 0) {
            lig_nm = read_npy_f64("lig_nm.npy");
            if (lig_nm.size() != ligand.atom_count() * 3 * setup.anm_lig)
                throw Error(LD_ERR_INVALID, "Number of read ANM in ligand does not correspond to the number of atoms");
        }
    }
    const auto &rec_active = restraint_list(setup.receptor_restraints, "active");
    const auto &rec_passive = restraint_list(setup.receptor_restraints, "passive");
    const auto &lig_active = restraint_list(setup.ligand_restraints, "active");
    const auto &lig_passive = restraint_list(setup.ligand_restraints, "passive");

    std::printf("Loading %s scoring function\n", method_name);
    std::vector<double> potential;
    if (method == LD_METHOD_DFIRE) {
        const char *data = std::getenv("LIGHTDOCK_DATA");
        potential = load_dcparams(std::string(data ? data : "data") + "/DCparams");
    }
    DockingModel rm = build_docking_model(method, receptor, rec_active, rec_passive, rec_nm, setup.anm_rec);
    DockingModel lm = build_docking_model(method, ligand, lig_active, lig_passive, lig_nm, setup.anm_lig);
    ld_scorer_desc desc;
    std::memset(&desc, 0, sizeof desc);
    desc.method = method;
    desc.use_anm = setup.use_anm ? 1 : 0;
    desc.receptor = rm.view();
    desc.ligand = lm.view();
    desc.potential = potential.empty() ? nullptr : potential.data();
    if (const char *dev = std::getenv("LIGHTDOCK_DEVICE")) hip_check(hipSetDevice(std::atoi(dev)), "hipSetDevice");
    Scorer scorer(desc);

    // Swarm::add_glowworms, src/swarm.rs:26-64: 7 pose columns (+ ANM extents when use_anm)
    const size_t pose_len = scorer.pose_len();
    if (positions.rows == 0) throw Error(LD_ERR_INVALID, "no starting positions");
    if (positions.cols < 7 || (setup.use_anm && positions.cols != pose_len))
        throw Error(LD_ERR_INVALID, "starting positions have " + std::to_string(positions.cols) + " columns, expected " +
                                        std::to_string(pose_len));
    std::vector<double> rows(positions.rows * pose_len);
    for (size_t r = 0; r < positions.rows; r++)
        std::memcpy(&rows[r * pose_len], &positions.values[r * positions.cols], pose_len * sizeof(double));

    std::printf("Creating GSO with %zu glowworms\n", positions.rows);
    Gso gso(scorer, 1, positions.rows, rows.data(), &seed);
    std::printf("Starting optimization (%u steps)\n", steps);
    std::fflush(stdout);
    // GSO::run, src/lib.rs:46-58: save after step 1 and after every 10th step; the steps in
    // between run back to back on the device (hipGraph replay)
    uint32_t done = 0;
    while (done < steps) {
        uint32_t next = done == 0 ? 1 : std::min(steps, (done / 10 + 1) * 10);
        gso.run(next - done);
        done = next;
        if (done % 10 == 0 || done == 1) gso.save(0, done, swarm_directory);
    }
    hip_check(hipStreamSynchronize(scorer.stream()), "hipStreamSynchronize");
    return 0;
}

}  // namespace ld
