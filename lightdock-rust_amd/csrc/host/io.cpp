#include "io.hpp"

#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "error.hpp"

namespace ld {

std::vector<double> load_dcparams(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw Error(LD_ERR_IO, "Unable to open DFIRE parameters: " + path);
    std::vector<double> table;
    table.reserve(LD_DFIRE_TABLE_LEN);
    std::string line;
    while (table.size() < (size_t)LD_DFIRE_TABLE_LEN && std::getline(in, line)) {
        const char *b = line.c_str();
        char *e = nullptr;
        double v = std::strtod(b, &e);
        if (e == b) throw Error(LD_ERR_IO, "DFIRE parameters: line " + std::to_string(table.size() + 1) + " is not a number");
        table.push_back(v);
    }
    if (table.size() < (size_t)LD_DFIRE_TABLE_LEN)
        throw Error(LD_ERR_IO, "DFIRE parameters: expected " + std::to_string(LD_DFIRE_TABLE_LEN) + " values, got " +
                                   std::to_string(table.size()));
    return table;
}

std::vector<double> read_npy_f64(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw Error(LD_ERR_IO, "Error reading ANM file [\"" + path + "\"]: " + std::strerror(errno));
    unsigned char head[12] = {0};
    in.read(reinterpret_cast<char *>(head), 10);
    if (!in || std::memcmp(head, "\x93NUMPY", 6) != 0) throw Error(LD_ERR_IO, path + ": not an .npy file");
    size_t header_len = head[8] | (size_t(head[9]) << 8);
    if (head[6] >= 2) {
        in.read(reinterpret_cast<char *>(head + 10), 2);
        header_len |= (size_t(head[10]) << 16) | (size_t(head[11]) << 24);
    }
    std::string header(header_len, '\0');
    in.read(&header[0], (std::streamsize)header_len);
    if (!in) throw Error(LD_ERR_IO, path + ": truncated .npy header");
    if (header.find("'<f8'") == std::string::npos) throw Error(LD_ERR_IO, path + ": expected little-endian f64 data");
    if (header.find("'fortran_order': True") != std::string::npos) throw Error(LD_ERR_IO, path + ": expected C order");
    size_t count = 1;
    size_t sh = header.find("'shape'");
    size_t open = sh == std::string::npos ? sh : header.find('(', sh);
    size_t close = open == std::string::npos ? open : header.find(')', open);
    if (close == std::string::npos) throw Error(LD_ERR_IO, path + ": bad .npy shape");
    std::string dims = header.substr(open + 1, close - open - 1);
    for (char &c : dims)
        if (c == ',') c = ' ';
    std::istringstream ds(dims);
    size_t d;
    while (ds >> d) count *= d;
    std::vector<double> data(count);
    in.read(reinterpret_cast<char *>(data.data()), (std::streamsize)(count * sizeof(double)));
    if ((size_t)in.gcount() != count * sizeof(double)) throw Error(LD_ERR_IO, path + ": truncated .npy data");
    return data;
}

Positions parse_positions(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw Error(LD_ERR_IO, "Error reading the input file: " + path);
    Positions p;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        size_t cols = 0;
        size_t start = 0;
        for (;;) {  // split(' '), every piece must be a float (the reference unwraps the parse)
            size_t sp = line.find(' ', start);
            std::string piece = line.substr(start, sp == std::string::npos ? std::string::npos : sp - start);
            const char *b = piece.c_str();
            char *e = nullptr;
            double v = std::strtod(b, &e);
            while (e && (*e == '\t')) e++;
            if (e == b || *e != 0)
                throw Error(LD_ERR_IO, path + ": line " + std::to_string(p.rows + 1) + ": invalid float literal");
            p.values.push_back(v);
            cols++;
            if (sp == std::string::npos) break;
            start = sp + 1;
        }
        if (p.rows == 0) p.cols = cols;
        else if (cols != p.cols)
            throw Error(LD_ERR_IO, path + ": line " + std::to_string(p.rows + 1) + " has " + std::to_string(cols) +
                                       " columns, expected " + std::to_string(p.cols));
        p.rows++;
    }
    return p;
}

// ---------------------------------------------------------------------------------------
// setup.json
// ---------------------------------------------------------------------------------------
namespace {

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string text;  // number literal or string value
    std::vector<Json> items;
    std::vector<std::pair<std::string, Json>> members;

    const Json *get(const std::string &key) const {
        const Json *hit = nullptr;
        for (const auto &m : members)
            if (m.first == key) hit = &m.second;  // serde keeps the last duplicate
        return hit;
    }
};

class JsonParser {
   public:
    explicit JsonParser(const std::string &s) : s_(s) {}
    Json parse_document() {
        Json v = value();
        ws();
        if (pos_ != s_.size()) bad("trailing characters");
        return v;
    }

   private:
    [[noreturn]] void bad(const std::string &why) { throw Error(LD_ERR_IO, why + " at offset " + std::to_string(pos_)); }
    void ws() {
        while (pos_ < s_.size() && std::isspace((unsigned char)s_[pos_])) pos_++;
    }
    bool eat(const char *lit) {
        size_t n = std::strlen(lit);
        if (s_.compare(pos_, n, lit) == 0) {
            pos_ += n;
            return true;
        }
        return false;
    }
    std::string string() {
        if (s_[pos_] != '"') bad("expected string");
        pos_++;
        std::string out;
        while (pos_ < s_.size() && s_[pos_] != '"') {
            char c = s_[pos_++];
            if (c == '\\' && pos_ < s_.size()) {
                char e = s_[pos_++];
                switch (e) {
                    case 'n': c = '\n'; break;
                    case 't': c = '\t'; break;
                    case 'r': c = '\r'; break;
                    case 'b': c = '\b'; break;
                    case 'f': c = '\f'; break;
                    case 'u': {
                        unsigned code = (unsigned)std::strtoul(s_.substr(pos_, 4).c_str(), nullptr, 16);
                        pos_ += 4;
                        c = code < 0x80 ? (char)code : '?';
                        break;
                    }
                    default: c = e;
                }
            }
            out.push_back(c);
        }
        if (pos_ >= s_.size()) bad("unterminated string");
        pos_++;
        return out;
    }
    Json value() {
        ws();
        if (pos_ >= s_.size()) bad("unexpected end of input");
        Json v;
        char c = s_[pos_];
        if (c == '{') {
            v.kind = Json::Object;
            pos_++;
            ws();
            if (pos_ < s_.size() && s_[pos_] == '}') { pos_++; return v; }
            for (;;) {
                ws();
                std::string key = string();
                ws();
                if (pos_ >= s_.size() || s_[pos_] != ':') bad("expected ':'");
                pos_++;
                v.members.emplace_back(std::move(key), value());
                ws();
                if (pos_ < s_.size() && s_[pos_] == ',') { pos_++; continue; }
                if (pos_ < s_.size() && s_[pos_] == '}') { pos_++; return v; }
                bad("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = Json::Array;
            pos_++;
            ws();
            if (pos_ < s_.size() && s_[pos_] == ']') { pos_++; return v; }
            for (;;) {
                v.items.push_back(value());
                ws();
                if (pos_ < s_.size() && s_[pos_] == ',') { pos_++; continue; }
                if (pos_ < s_.size() && s_[pos_] == ']') { pos_++; return v; }
                bad("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = Json::String; v.text = string(); return v; }
        if (eat("true")) { v.kind = Json::Bool; v.b = true; return v; }
        if (eat("false")) { v.kind = Json::Bool; v.b = false; return v; }
        if (eat("null")) { v.kind = Json::Null; return v; }
        const char *b = s_.c_str() + pos_;
        char *e = nullptr;
        (void)std::strtod(b, &e);
        if (e == b) bad("expected value");
        v.kind = Json::Number;
        v.text.assign(b, (size_t)(e - b));
        pos_ += (size_t)(e - b);
        return v;
    }
    const std::string &s_;
    size_t pos_ = 0;
};

const Json &required(const Json &root, const char *key, Json::Kind kind, const char *expected) {
    const Json *v = root.get(key);
    if (!v) throw Error(LD_ERR_IO, std::string("missing field `") + key + "`");
    if (v->kind != kind) throw Error(LD_ERR_IO, std::string("invalid type for `") + key + "`, expected " + expected);
    return *v;
}
uint64_t as_unsigned(const Json &v, const char *key) {
    if (v.text.find_first_of(".eE-") != std::string::npos)
        throw Error(LD_ERR_IO, std::string("invalid type for `") + key + "`, expected unsigned integer");
    return std::strtoull(v.text.c_str(), nullptr, 10);
}
void optional_string(const Json &root, const char *key) {
    const Json *v = root.get(key);
    if (v && v->kind != Json::Null && v->kind != Json::String)
        throw Error(LD_ERR_IO, std::string("invalid type for `") + key + "`, expected a string");
}
std::optional<std::map<std::string, std::vector<std::string>>> optional_restraints(const Json &root, const char *key) {
    const Json *v = root.get(key);
    if (!v || v->kind == Json::Null) return std::nullopt;
    if (v->kind != Json::Object) throw Error(LD_ERR_IO, std::string("invalid type for `") + key + "`, expected a map");
    std::map<std::string, std::vector<std::string>> out;
    for (const auto &m : v->members) {
        if (m.second.kind != Json::Array) throw Error(LD_ERR_IO, std::string("invalid type in `") + key + "`, expected a sequence");
        std::vector<std::string> list;
        for (const Json &item : m.second.items) {
            if (item.kind != Json::String) throw Error(LD_ERR_IO, std::string("invalid type in `") + key + "`, expected a string");
            list.push_back(item.text);
        }
        out[m.first] = std::move(list);
    }
    return out;
}

}  // namespace

SetupFile read_setup(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw Error(LD_ERR_IO, std::string(std::strerror(errno)) + " (os error " + std::to_string(errno) + ")");
    std::stringstream buf;
    buf << in.rdbuf();
    const std::string text = buf.str();
    Json root = JsonParser(text).parse_document();
    if (root.kind != Json::Object) throw Error(LD_ERR_IO, "invalid type: expected struct SetupFile");

    SetupFile s;
    if (const Json *seed = root.get("seed"); seed && seed->kind != Json::Null) {
        if (seed->kind != Json::Number) throw Error(LD_ERR_IO, "invalid type for `seed`, expected unsigned integer");
        s.seed = as_unsigned(*seed, "seed");
    }
    // every non-Option field of SetupFile must be present and well typed, used or not
    (void)as_unsigned(required(root, "anm_seed", Json::Number, "unsigned integer"), "anm_seed");
    optional_string(root, "ftdock_file");
    (void)required(root, "noh", Json::Bool, "a boolean");
    s.anm_rec = (size_t)as_unsigned(required(root, "anm_rec", Json::Number, "unsigned integer"), "anm_rec");
    s.anm_lig = (size_t)as_unsigned(required(root, "anm_lig", Json::Number, "unsigned integer"), "anm_lig");
    (void)as_unsigned(required(root, "swarms", Json::Number, "unsigned integer"), "swarms");
    (void)as_unsigned(required(root, "starting_points_seed", Json::Number, "unsigned integer"), "starting_points_seed");
    (void)required(root, "verbose_parser", Json::Bool, "a boolean");
    (void)required(root, "noxt", Json::Bool, "a boolean");
    (void)required(root, "now", Json::Bool, "a boolean");
    optional_string(root, "restraints");
    s.use_anm = required(root, "use_anm", Json::Bool, "a boolean").b;
    (void)as_unsigned(required(root, "glowworms", Json::Number, "unsigned integer"), "glowworms");
    (void)required(root, "membrane", Json::Bool, "a boolean");
    s.receptor_pdb = required(root, "receptor_pdb", Json::String, "a string").text;
    s.ligand_pdb = required(root, "ligand_pdb", Json::String, "a string").text;
    s.receptor_restraints = optional_restraints(root, "receptor_restraints");
    s.ligand_restraints = optional_restraints(root, "ligand_restraints");
    return s;
}

}  // namespace ld
