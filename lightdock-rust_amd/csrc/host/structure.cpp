#include "structure.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <map>
#include <tuple>

#include "error.hpp"

namespace ld {

namespace {

std::string field(const std::string &line, size_t begin, size_t width) {
    if (begin >= line.size()) return std::string();
    std::string s = line.substr(begin, width);
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return std::string();
    size_t e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
}

}  // namespace

std::string AtomRecord::residue_id() const {
    return chain_id + "." + res_name + "." + std::to_string(res_seq) + icode;
}

Structure read_pdb(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw Error(LD_ERR_IO, "cannot open PDB file " + path);

    // pdbtbx files each record under (chain id) -> (serial, insertion code) -> (residue
    // name, alt loc); every level keeps first-appearance order.  Keys below reproduce
    // that: a record sorts by the first-appearance rank of its chain, then of its
    // residue inside the chain, then of its conformer inside the residue, then by file
    // position.
    struct Keyed {
        size_t chain, residue, conformer, pos;
        AtomRecord atom;
    };
    std::vector<Keyed> rows;
    std::map<std::string, size_t> chain_rank;
    std::map<std::tuple<size_t, long, std::string>, size_t> residue_rank;
    std::map<std::tuple<size_t, std::string, char>, size_t> conformer_rank;

    std::string line;
    while (std::getline(in, line)) {
        if (line.size() < 54) continue;
        if (line.compare(0, 6, "ATOM  ") != 0 && line.compare(0, 6, "HETATM") != 0) continue;
        AtomRecord a;
        a.name = field(line, 12, 4);
        a.alt_loc = line[16];
        a.res_name = field(line, 17, 3);
        a.chain_id = field(line, 21, 1);
        a.res_seq = std::strtol(field(line, 22, 4).c_str(), nullptr, 10);
        a.icode = field(line, 26, 1);
        a.x = std::strtod(field(line, 30, 8).c_str(), nullptr);
        a.y = std::strtod(field(line, 38, 8).c_str(), nullptr);
        a.z = std::strtod(field(line, 46, 8).c_str(), nullptr);

        size_t pos = rows.size();
        size_t c = chain_rank.emplace(a.chain_id, chain_rank.size()).first->second;
        size_t r = residue_rank.emplace(std::make_tuple(c, a.res_seq, a.icode), pos).first->second;
        size_t f = conformer_rank.emplace(std::make_tuple(r, a.res_name, a.alt_loc), pos).first->second;
        rows.push_back(Keyed{c, r, f, pos, std::move(a)});
    }

    std::vector<size_t> order(rows.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        const Keyed &x = rows[a], &y = rows[b];
        return std::tie(x.chain, x.residue, x.conformer, x.pos) < std::tie(y.chain, y.residue, y.conformer, y.pos);
    });

    Structure s;
    s.atoms.reserve(rows.size());
    for (size_t i : order) s.atoms.push_back(std::move(rows[i].atom));
    return s;
}

}  // namespace ld
