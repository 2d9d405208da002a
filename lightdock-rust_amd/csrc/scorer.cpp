#include "scorer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "host/spatial_order.hpp"

namespace ld {

// ---------------------------------------------------------------------------------------
// device memory helpers
// ---------------------------------------------------------------------------------------
DeviceArena::~DeviceArena() {
    for (void *p : blocks_) (void)hipFree(p);
}
void *DeviceArena::alloc_bytes(size_t bytes) {
    void *p = nullptr;
    hip_check(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc");
    blocks_.push_back(p);
    return p;
}
void DeviceBuffer::reserve(size_t want) {
    if (want <= bytes) return;
    generation++;  // the old block is gone: whoever baked its address (a captured hipGraph) must notice
    release();
    size_t grow = want + want / 4 + 256;
    hip_check(hipMalloc(&ptr, grow), "hipMalloc(workspace)");
    bytes = grow;
}
void DeviceBuffer::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

// ---------------------------------------------------------------------------------------
// DFIRE distance binning
// ---------------------------------------------------------------------------------------
namespace {
// DIST_TO_BINS (src/dfire.rs:49-53) in closed form for the reachable indices 0..29 (d <= 29):
// half-angstrom bins up to 8 A, then one-angstrom bins; value = bin + 1.
int dist_to_bins(size_t idx) {
    if (idx < 3) return 1;
    if (idx < 15) return (int)idx - 1;
    if (idx < 49) return 14 + (int)(idx - 15) / 2;
    return idx == 49 ? 31 : 32;
}
}  // namespace

int dfire_bin_reference(double dist2) {
    const double d = std::sqrt(dist2) * 2.0 - 1.0;  // src/dfire.rs:336
    size_t idx = d > 0.0 ? (size_t)d : 0;            // `d as usize` saturates negatives/NaN to 0
    if (idx > 50) idx = 50;
    return dist_to_bins(idx) - 1;
}

// Smallest double x in [0, 225] for which pred(x) holds, pred monotone (false..true).
template <typename Pred>
static double first_true(Pred pred) {
    double lo = 0.0, hi = 225.0;
    if (pred(lo)) return lo;
    if (!pred(hi)) return std::numeric_limits<double>::infinity();
    uint64_t a, b;
    std::memcpy(&a, &lo, 8);
    std::memcpy(&b, &hi, 8);
    while (b - a > 1) {  // positive doubles order like their bit patterns
        const uint64_t mid = a + (b - a) / 2;
        double m;
        std::memcpy(&m, &mid, 8);
        if (pred(m)) b = mid; else a = mid;
    }
    double out;
    std::memcpy(&out, &b, 8);
    return out;
}

DfireBinning build_dfire_binning() {
    // The bin index floor(2r - 1) steps at r = (k+1)/2, i.e. near d2 = (k+1)^2/4 -- a multiple
    // of 0.25 -- so the bin at the lower edge of a 0.25-wide d2 cell is right for the whole
    // cell EXCEPT possibly its very last double: the correctly rounded sqrt may round the
    // predecessor of a step up onto it.  So: lut[cell] = bin at the cell's lower edge, and
    // step[b] = the exact first double that the reference formula puts in bin >= b (found by
    // bisection on the formula itself); bin(d2) = lut[cell] + (d2 >= step[lut[cell] + 1]).
    DfireBinning t;
    t.lut.assign(kDfireLutCells, 0);
    for (int c = 0; c <= 900; c++) t.lut[c] = (uint8_t)dfire_bin_reference(c * 0.25);
    for (int c = 901; c < kDfireLutCells; c++) t.lut[c] = t.lut[900];
    t.step.assign(kDfireSteps, std::numeric_limits<double>::infinity());
    t.step[0] = 0.0;
    for (int b = 1; b <= 20; b++) t.step[b] = first_true([b](double d2) { return dfire_bin_reference(d2) >= b; });
    // self-check on both ends of every cell
    for (int c = 0; c < 900; c++) {
        const double ends[2] = {c * 0.25, std::nextafter((c + 1) * 0.25, 0.0)};
        for (double d2 : ends) {
            const int cell = (int)(d2 * 4.0);
            int b = t.lut[cell];
            if (d2 >= t.step[b + 1]) b++;
            if (cell != c || b != dfire_bin_reference(d2))
                throw Error(LD_ERR_INVALID, "DFIRE binning self-check failed in cell " + std::to_string(c));
        }
    }
    return t;
}

double dfire_interface_d2() {
    // Largest double d2 with fl(fl(sqrt(d2)) * 2 - 1) <= 3.9 (src/dfire.rs:336,339).  The
    // left side is monotone in d2, so bisect on the bit pattern.
    auto inside = [](double d2) { return std::sqrt(d2) * 2.0 - 1.0 <= 3.9; };
    double lo = 0.0, hi = 225.0;  // inside(lo), !inside(hi)
    uint64_t a, b;
    std::memcpy(&a, &lo, 8);
    std::memcpy(&b, &hi, 8);
    while (b - a > 1) {
        uint64_t mid = a + (b - a) / 2;
        double m;
        std::memcpy(&m, &mid, 8);
        if (inside(m)) a = mid; else b = mid;
    }
    double out;
    std::memcpy(&out, &a, 8);
    return out;
}

// ---------------------------------------------------------------------------------------
// Scorer
// ---------------------------------------------------------------------------------------
namespace {

void check_molecule(const ld_molecule &m, int method, const char *who, bool use_anm) {
    const std::string w(who);
    if (m.n_atoms == 0) throw Error(LD_ERR_INVALID, w + ": molecule has no atoms");
    if (m.n_atoms > (size_t)1 << 24) throw Error(LD_ERR_INVALID, w + ": too many atoms");
    if (!m.coordinates) throw Error(LD_ERR_INVALID, w + ": coordinates missing");
    if (method == LD_METHOD_DFIRE) {
        if (!m.dfire_types) throw Error(LD_ERR_INVALID, w + ": dfire_types missing");
        for (size_t i = 0; i < m.n_atoms; i++)
            if (m.dfire_types[i] > 167) throw Error(LD_ERR_INVALID, w + ": DFIRE atom type out of range");
    } else {
        if (!m.ele_charges || !m.vdw_charges || !m.vdw_radii) throw Error(LD_ERR_INVALID, w + ": DNA parameters missing");
    }
    if (m.n_membrane && !m.membrane) throw Error(LD_ERR_INVALID, w + ": membrane indices missing");
    for (size_t k = 0; k < m.n_membrane; k++)
        if (m.membrane[k] >= m.n_atoms) throw Error(LD_ERR_INVALID, w + ": membrane index out of range");
    if (m.n_restraint_groups) {
        if (!m.restraint_offsets || !m.restraint_atoms) throw Error(LD_ERR_INVALID, w + ": restraint CSR missing");
        for (size_t g = 0; g < m.n_restraint_groups; g++)
            if (m.restraint_offsets[g] > m.restraint_offsets[g + 1]) throw Error(LD_ERR_INVALID, w + ": restraint offsets not sorted");
        for (uint32_t k = 0; k < m.restraint_offsets[m.n_restraint_groups]; k++)
            if (m.restraint_atoms[k] >= m.n_atoms) throw Error(LD_ERR_INVALID, w + ": restraint atom out of range");
    }
    if (use_anm && m.num_anm > 0 && !m.nmodes) throw Error(LD_ERR_INVALID, w + ": ANM modes missing");
    if (m.num_anm > 64) throw Error(LD_ERR_INVALID, w + ": more than 64 ANM modes");
}

}  // namespace

void Scorer::upload_molecule(const ld_molecule &m, bool is_receptor, DeviceMolecule &dev, HostMolecule &host,
                             std::vector<uint32_t> &group_offsets, std::vector<uint32_t> &group_slots,
                             std::vector<uint32_t> &membrane_slots) {
    const size_t n = m.n_atoms;
    const size_t n_pad = (n + 63) / 64 * 64;
    dev.n = (int)n;
    dev.n_pad = (int)n_pad;

    host.coordinates.assign(m.coordinates, m.coordinates + 3 * n);
    std::vector<double> x(n_pad, 0.0), y(n_pad, 0.0), z(n_pad, 0.0);
    for (size_t i = 0; i < n; i++) {
        x[i] = m.coordinates[3 * i];
        y[i] = m.coordinates[3 * i + 1];
        z[i] = m.coordinates[3 * i + 2];
    }
    dev.x = arena_.upload(x);
    dev.y = arena_.upload(y);
    dev.z = arena_.upload(z);

    if (method_ == LD_METHOD_DFIRE) {
        host.dfire_types.assign(m.dfire_types, m.dfire_types + n);
        std::vector<uint32_t> t(n_pad, 0);
        // potential[atoma*169*20 + atomb*20 + bin] (src/dfire.rs:338): receptor carries the
        // row base, ligand the column base.
        for (size_t i = 0; i < n; i++) t[i] = m.dfire_types[i] * (is_receptor ? kDfireRowStride : 20u);
        dev.tindex = arena_.upload(t);
    } else {
        host.ele_charges.assign(m.ele_charges, m.ele_charges + n);
        host.vdw_charges.assign(m.vdw_charges, m.vdw_charges + n);
        host.vdw_radii.assign(m.vdw_radii, m.vdw_radii + n);
        dev.charge = arena_.upload(host.ele_charges, n_pad);
        {   // the kernel multiplies sqrt(eps_i) * sqrt(eps_j) instead of sqrt(eps_i * eps_j) per pair
            std::vector<double> root(host.vdw_charges);
            for (double &v : root) v = std::sqrt(v);
            dev.well_depth = arena_.upload(root, n_pad);
        }
        dev.radius = arena_.upload(host.vdw_radii, n_pad);
    }

    // Interface flags are only needed for restraint atoms and membrane beads
    // (src/scoring.rs:21-47): give each such atom one bit ("slot") of the per-pose flag set.
    std::vector<int32_t> slot(n_pad, -1);
    uint32_t next = 0;
    auto slot_of = [&](uint32_t atom) {
        if (slot[atom] < 0) slot[atom] = (int32_t)next++;
        return (uint32_t)slot[atom];
    };
    group_offsets.assign(1, 0);
    group_slots.clear();
    for (size_t g = 0; g < m.n_restraint_groups; g++) {
        for (uint32_t k = m.restraint_offsets[g]; k < m.restraint_offsets[g + 1]; k++)
            group_slots.push_back(slot_of(m.restraint_atoms[k]));
        group_offsets.push_back((uint32_t)group_slots.size());
    }
    membrane_slots.clear();
    for (size_t k = 0; k < m.n_membrane; k++) membrane_slots.push_back(slot_of(m.membrane[k]));
    dev.slot = arena_.upload(slot);
    dev.flag_words = (int)((next + 31) / 32);
    (is_receptor ? host_slot_rec_ : host_slot_lig_).assign(slot.begin(), slot.begin() + (long)n);

    dev.num_anm = 0;
    dev.modes = nullptr;
    if (use_anm_ && m.num_anm > 0) {
        // (mode, atom, xyz) -> [mode][xyz][n_pad]
        std::vector<double> modes(m.num_anm * 3 * n_pad, 0.0);
        for (size_t k = 0; k < m.num_anm; k++)
            for (size_t i = 0; i < n; i++)
                for (int c = 0; c < 3; c++) modes[(k * 3 + c) * n_pad + i] = m.nmodes[k * n * 3 + i * 3 + c];
        dev.modes = arena_.upload(modes);
        dev.num_anm = (int)m.num_anm;
    }
}

Scorer::Scorer(const ld_scorer_desc &desc) {
    if (desc.method != LD_METHOD_DFIRE && desc.method != LD_METHOD_DNA && desc.method != LD_METHOD_PYDOCK)
        throw Error(LD_ERR_UNSUPPORTED, "Error: method not supported");
    // PYDOCK's energy is DNA's (src/pydock.rs:425-545 == src/dna.rs:411-529); only the model builder differs
    method_ = desc.method == LD_METHOD_PYDOCK ? LD_METHOD_DNA : desc.method;
    use_anm_ = desc.use_anm != 0;
    check_molecule(desc.receptor, method_, "receptor", use_anm_);
    check_molecule(desc.ligand, method_, "ligand", use_anm_);
    if (method_ == LD_METHOD_DFIRE && !desc.potential) throw Error(LD_ERR_IO, "Unable to open DFIRE parameters");

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw Error(LD_ERR_DEVICE, "no HIP device available: the pose-energy path has no CPU fallback");
    hip_check(hipGetDevice(&device_), "hipGetDevice");
    hipDeviceProp_t prop;
    hip_check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties");
    n_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    bool arch_ok = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
#ifdef LD_DIAG_BUILD   // (diagnostic builds only, tools/build_variant.sh: the shipped library has no switch that waives the check)
    if (std::getenv("LIGHTDOCK_ALLOW_ANY_ARCH")) arch_ok = true;
#endif
    if (!arch_ok)
        throw Error(LD_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");

    // own non-blocking stream: kernels of this handle never serialise with the legacy default
    // stream, and the GSO loop can be captured into a hipGraph
    hip_check(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking), "hipStreamCreate");
    stream_ = own_stream_;

    std::vector<uint32_t> rgo, rgs, rms, lgo, lgs, lms;
    upload_molecule(desc.receptor, true, pair_.rec, host_rec_, rgo, rgs, rms);
    upload_molecule(desc.ligand, false, pair_.lig, host_lig_, lgo, lgs, lms);
    // src/scoring.rs:38-47 is only ever applied to the receptor (src/dfire.rs:357); ligand
    // beads get a slot but no reader.
    tail_.n_rec_groups = (int)rgo.size() - 1;
    tail_.n_lig_groups = (int)lgo.size() - 1;
    tail_.n_membrane = (int)rms.size();
    tail_.rec_group_offsets = arena_.upload(rgo);
    tail_.rec_group_slots = arena_.upload(rgs);
    tail_.lig_group_offsets = arena_.upload(lgo);
    tail_.lig_group_slots = arena_.upload(lgs);
    tail_.membrane_slots = arena_.upload(rms);

    pair_.method = method_;
    pair_.use_anm = use_anm_ ? 1 : 0;
    if (method_ == LD_METHOD_DFIRE) {
        std::vector<double> table(desc.potential, desc.potential + LD_DFIRE_TABLE_LEN);
        pair_.table = arena_.upload(table);
        const DfireBinning binning = build_dfire_binning();
        pair_.lut = arena_.upload(binning.lut);
        pair_.bin_step = arena_.upload(binning.step);
        pair_.iface_d2 = dfire_interface_d2();
    } else {
        pair_.iface_d2 = 3.9 * 3.9;  // INTERFACE_CUTOFF2, src/constants.rs:15
    }

    // Receptor chunking: <= 512 (DFIRE) / 256 (DNA) atoms per workgroup = 16 KiB of LDS
    // records, balanced over the chunks.  Fixed per scorer so that a pose's energy does not
    // depend on the batch it is evaluated in.
    int max_chunk = method_ == LD_METHOD_DFIRE ? 512 : 256;
    if (const char *e = std::getenv("LIGHTDOCK_CHUNK_ATOMS")) {
        int v = std::atoi(e);
        if (v >= 64 && v <= 2048) max_chunk = v;
    }
    pair_.n_chunks = (pair_.rec.n + max_chunk - 1) / max_chunk;
    pair_.chunk_atoms = (pair_.rec.n + pair_.n_chunks - 1) / pair_.n_chunks;
    const int n_groups = (pair_.lig.n + 63) / 64;
    pair_.split_j = (n_groups % kWaves != 0 && n_groups < 4 * kWaves) ? 1 : 0;

    if (method_ == LD_METHOD_DFIRE) {
        const char *k = std::getenv("LIGHTDOCK_DFIRE_KERNEL");
        // "packed" (default): culling + packed-f32 pair test with exact f64 path; "tiled": the same
        // culling with an all-f64 pair test; "allpairs": no culling
        use_tiled_ = !(k && std::strcmp(k, "allpairs") == 0);
        if (use_tiled_) build_tiled(desc);
        if (use_tiled_ && !(k && std::strcmp(k, "tiled") == 0)) build_packed(desc);
        // "bm" / default: the block-major path (kernels/dfire_bm.hpp; its ANM form for molecules that flex); "packed": the pose-major
        // kernel for everything (what LIGHTDOCK_TILED_LATENCY=1, the single-swarm CLI, and the complexes build_bm declines use anyway)
        const char *latency = std::getenv("LIGHTDOCK_TILED_LATENCY");
        if (use_packed_ && !(k && std::strcmp(k, "packed") == 0) && !(latency && std::atoi(latency) > 0 && !(k && std::strcmp(k, "bm") == 0)))
            build_bm(desc);
    }
}

// Tile-ordered SoA copy of one molecule (host/spatial_order.hpp); padding atoms at -1e30
// (receptor) / +1e30 (ligand) so that no padding/padding pair can ever look close.
void Scorer::upload_tiled_molecule(const ld_molecule &m, bool is_receptor, TiledSoA &out) {
    const size_t n = m.n_atoms;
    const DfireTileLayout layout = dfire_tile_layout(m.coordinates, m.dfire_types, n);
    const std::vector<uint32_t> &order = layout.order;
    const size_t np = order.size();
    const uint32_t kPad = std::numeric_limits<uint32_t>::max();
    std::vector<double> x(np, is_receptor ? -1.0e30 : 1.0e30), y(np, 0.0), z(np, 0.0);
    std::vector<uint32_t> t(np, 0);
    std::vector<int32_t> slot(np, -1);
    const std::vector<int32_t> &hslot = is_receptor ? host_slot_rec_ : host_slot_lig_;
    // type numbers as the patch layout of the potential wants them (bonded atoms paired up)
    std::vector<uint32_t> &perm = is_receptor ? type_perm_rec_ : type_perm_lig_;
    perm = layout.type_perm;
    for (size_t i = 0; i < np; i++) {
        const uint32_t a = order[i];
        if (a == kPad) {
            if (i < n) throw Error(LD_ERR_INVALID, "spatial order: padding before the tail");
            continue;
        }
        x[i] = m.coordinates[3 * (size_t)a];
        y[i] = m.coordinates[3 * (size_t)a + 1];
        z[i] = m.coordinates[3 * (size_t)a + 2];
        t[i] = is_receptor ? tiled_rec_term(perm[m.dfire_types[a]]) : tiled_lig_term(perm[m.dfire_types[a]]);
        slot[i] = hslot[a];
    }
    out.n_real = (int)n;
    out.n_tiles = (int)(np / 64);
    out.hx = x;
    out.hy = y;
    out.hz = z;
    out.htype.assign(np, kPad);
    out.hslot = slot;
    for (size_t i = 0; i < np; i++)
        if (order[i] != kPad) out.htype[i] = m.dfire_types[order[i]];
    out.x = arena_.upload(x);
    out.y = arena_.upload(y);
    out.z = arena_.upload(z);
    out.tindex = arena_.upload(t);
    out.slot = arena_.upload(slot);
    out.num_anm = 0;
    out.modes = nullptr;
    if (use_anm_ && m.num_anm > 0) {
        std::vector<double> modes(m.num_anm * 3 * np, 0.0);
        for (size_t k = 0; k < m.num_anm; k++)
            for (size_t i = 0; i < np; i++) {
                if (order[i] == kPad) continue;
                for (int c = 0; c < 3; c++) modes[(k * 3 + c) * np + i] = m.nmodes[k * n * 3 + (size_t)order[i] * 3 + c];
            }
        out.modes = arena_.upload(modes);
        out.num_anm = (int)m.num_anm;
        out.hmodes = std::move(modes);
    }
}

void Scorer::build_tiled(const ld_scorer_desc &desc) {
    upload_tiled_molecule(desc.receptor, true, tiled_rec_soa_);
    TiledSoA &lig = tiled_lig_soa_;
    upload_tiled_molecule(desc.ligand, false, lig);
    tiled_.lig.n_real = lig.n_real;
    tiled_.lig.n_tiles = lig.n_tiles;
    tiled_.lig.x = lig.x;
    tiled_.lig.y = lig.y;
    tiled_.lig.z = lig.z;
    tiled_.lig.tindex = lig.tindex;
    tiled_.lig.slot = lig.slot;
    tiled_.lig.num_anm = lig.num_anm;
    tiled_.lig.modes = lig.modes;
    tiled_.lig.flag_words = pair_.lig.flag_words;
    tiled_.rec.n_real = tiled_rec_soa_.n_real;
    tiled_.rec.n_tiles = tiled_rec_soa_.n_tiles;
    tiled_.rec.flag_words = pair_.rec.flag_words;
    tiled_.use_anm = use_anm_ ? 1 : 0;
    tiled_.anm_rec = (int)anm_rec();
    {   // potential re-laid out in 2 x 2 x 4 patches, see dfire_tiled.hpp
        std::vector<double> t2(kTiledTableDoubles, 0.0);
        for (uint32_t l = 0; l < 168; l++)
            for (uint32_t b = 0; b < kTiledTableBins; b++)
                for (uint32_t r = 0; r < 168; r++)
                    t2[(tiled_lig_term(type_perm_lig_[l]) + tiled_rec_term(type_perm_rec_[r]) + tiled_bin_term(b)) / 8] =
                        desc.potential[(size_t)r * kDfireRowStride + l * 20 + b];
        tiled_.table = arena_.upload(t2);
    }
    tiled_.bin_step = pair_.bin_step;
    tiled_.iface_d2 = pair_.iface_d2;
    tiled_.iface_scaled = 4.0 * pair_.iface_d2;
    {   // cell code = bin at the cell's lower edge | 0x80 when a bin step falls inside the cell
        // | 0x40 when the cell reaches below the interface distance (src/dfire.rs:339)
        const DfireBinning b = build_dfire_binning();
        std::vector<uint8_t> code(b.lut);
        for (int c = 0; c <= 900; c++) {
            const int bin = b.lut[c];
            if (b.step[bin + 1] < (c + 1) * 0.25) code[c] |= 0x80u;
            if (c * 0.25 <= pair_.iface_d2) code[c] |= 0x40u;
        }
        std::vector<uint32_t> words(kDfireLutCells, kTiledLutMiss);
        for (int c = 0; c <= 900; c++)  // cell 900 holds the cutoff itself: d2 = 225 is in, the rest of the cell out
            words[c] = (code[c] & 0xc0u) || c == 900 ? kTiledLutSlow | code[c] : tiled_bin_term(code[c]);
        tiled_.lut = arena_.upload(words);
    }
    int waves = 4;  // measured on MI355X (1k4c, 1ppe): 4 waves per workgroup beat 1, 2 and 8
    if (const char *e = std::getenv("LIGHTDOCK_TILED_WAVES")) {
        int v = std::atoi(e);
        if (v >= 1 && v <= kTiledMaxWaves) waves = v;
    }
    // `split` items share one ligand tile, each taking every split-th surviving receptor tile.
    // With thousands of poses per launch there are enough waves anyway and split = 1 is best or
    // equal (measured on MI355X at 16 384+ poses: 2uuy, 7 ligand tiles, +12 % over split 3; 1k4c,
    // 52 tiles, +11 %; 1ppe, 4 tiles, +5 % with the packed kernel).  A launch of one swarm (200 poses) of a small ligand
    // does not fill the GPU: split 3 halves its latency on 1ppe.  The split is fixed per scorer (a
    // pose's energy must not depend on the batch it travels in), so the default serves
    // throughput and LIGHTDOCK_TILED_LATENCY=1 -- set by the single-swarm CLI -- serves latency.
    // Splits sharing a factor with the 4 waves of a workgroup (2, 4) are consistently slower.
    const char *latency = std::getenv("LIGHTDOCK_TILED_LATENCY");
    const int small_ligand = (latency && std::atoi(latency) > 0) ? 32 : 0;
    int split = tiled_.lig.n_tiles < small_ligand ? 3 : 1;
    if (const char *e = std::getenv("LIGHTDOCK_TILED_SPLIT")) {
        int v = std::atoi(e);
        if (v >= 1 && v <= 8) split = v;
    }
    tiled_.waves = waves;
    tiled_.split = split;
    tiled_.n_groups = (tiled_.lig.n_tiles * split + waves - 1) / waves;

    rec_anm_per_pose_ = use_anm_ && tiled_rec_soa_.num_anm > 0;
    if (!rec_anm_per_pose_) {
        // static receptor image (records + subtile/tile boxes), built once by the same kernel
        // that builds the per-pose images when the receptor has ANM
        const size_t pad = (size_t)tiled_.rec.n_tiles * 64;
        TiledAtom *atoms = static_cast<TiledAtom *>(arena_.alloc_bytes(pad * sizeof(TiledAtom)));
        TiledBox *sub = static_cast<TiledBox *>(arena_.alloc_bytes(pad / 8 * sizeof(TiledBox)));
        TiledBox *tile = static_cast<TiledBox *>(arena_.alloc_bytes(pad / 64 * sizeof(TiledBox)));
        PrepareReceptorLaunch p = prepare_launch(nullptr, 0, nullptr, 1);
        p.num_anm = 0;
        p.atoms_out = atoms;
        p.sub_out = sub;
        p.tile_out = tile;
        hip_check(launch_prepare_receptor(p, stream_), "launch dfire_prepare_receptor");
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        tiled_.rec.atoms = atoms;
        tiled_.rec.sub_boxes = sub;
        tiled_.rec.tile_boxes = tile;
    }
}

double dfire_f32_error_bound(double ubound, int cells_per_unit) {
    // In record units (D' = cells_per_unit * 4 d2 + 1/2).  Records: |fl32(u) - u| <= 2^-25 U for
    // |u| < U = a power of two.  Differences of pairs in range (4 d2 < 1100): the exact difference
    // of two records is within 2 e_u of the true one and below 128, so its rounding adds at most
    // 2^-25 * 128.  Then the squares and the three fma roundings (results below 4096).
    const double e_u = std::ldexp(ubound, -25);
    const double e_d = 2.0 * e_u + std::ldexp(128.0, -25);
    const double span = std::sqrt(3.0 * 1100.0 * cells_per_unit);
    const double eps = 2.0 * e_d * span + 3.0 * e_d * e_d + 3.0 * std::ldexp(4096.0, -25);
    return 2.0 * eps / cells_per_unit;  // twice the bound, in units of 4 d2
}

// The cell LUT of the packed DFIRE kernel (kernels/dfire_packed.hpp) for `sc` cells per unit of
// 4 d2 and an f32 distance error of at most `eps` (units of 4 d2).
//
// `zero_bins`: bit b set = the potential is 0.0 in bin b for every (receptor type, ligand type) of this
// complex (DFIRE's reference state makes the last shell before the cutoff exactly that).  A cell whose
// pairs can only land in such bins reads "miss": the sum does not change by a bit (x + 0.0 = x) and the
// pair costs no table line.  Cells that may set interface flags are left alone.
std::vector<uint32_t> build_packed_lut(int sc, double eps, uint32_t zero_bins) {
    // cell LUT.  Cell k holds the pairs with D' = sc * 4 d2 + 1/2 (f32) in [k, k+1), i.e. a true
    // 4 d2 within ((k - 1/2) / sc - eps, (k + 1/2) / sc + eps); the last cell everything further.
    const DfireBinning b = build_dfire_binning();
    const double e = eps;
    const double iface_scaled = 4.0 * dfire_interface_d2();
    const int n_cells = 1024 * sc;
    std::vector<uint32_t> words((size_t)kPackedLutCells * sc, kPackedMiss);
    for (int k = 0; k < n_cells; k++) {
        const double ilo = (k - 0.5) / sc - e, ihi = (k + 0.5) / sc + e;
        if (ilo > 900.0) continue;  // beyond the cutoff for sure
        uint32_t code = 0;
        int steps_inside = 0, base_bin = 0, step_bin = 0;
        double step_at = 0.0;
        for (int s = 1; s <= 20; s++) {
            const double at = 4.0 * b.step[s];
            if (at < ilo) base_bin = s;  // already in bin s at the lower end
            else if (at <= ihi) {
                steps_inside++;
                step_bin = s;
                step_at = at;
            }
        }
        if (steps_inside > 1) throw Error(LD_ERR_INVALID, "DFIRE cell LUT: two bin steps in one cell");
        if (steps_inside) code |= kPackedCodeStep;
        const bool below_iface = ihi < iface_scaled;  // every pair of the cell sets interface flags
        if (ilo <= iface_scaled && !below_iface) code |= kPackedCodeIface;
        if (ihi >= 900.0) code |= kPackedCodeCutoff;
        if (!code && !below_iface) {
            if (dfire_bin_reference(std::max(ilo, 0.0) / 4.0) != base_bin || dfire_bin_reference(ihi / 4.0) != base_bin)
                throw Error(LD_ERR_INVALID, "DFIRE cell LUT self-check failed in cell " + std::to_string(k));
            words[k] = (zero_bins >> base_bin) & 1u ? kPackedMiss : tiled_bin_term((uint32_t)base_bin);
            continue;
        }
        // Lean form: the only step of the cell sits at its middle (the squares (n+1)^2 are integers;
        // the exact first double of a bin can lie an ulp below, where the correctly rounded sqrt of
        // the reference rounds up onto the step -- far inside the eps band that goes to the exact
        // path anyway) and everything below / above it has one answer.  The cutoff cell is lean
        // when the last bin step and the cutoff coincide at 900: above it = miss.
        const bool mid = steps_inside == 1 && std::fabs(step_at * sc - (double)k) <= 1e-9 && step_bin == base_bin + 1;
        const uint32_t flags_code = below_iface ? kPackedCodeFlags : 0u;
        if (code == 0) {  // below the interface distance, no step: lean with nothing to grow by
            words[k] = kPackedSlow | ((kPackedCodeLean | flags_code) << 24) | tiled_bin_term((uint32_t)base_bin);
        } else if (code == kPackedCodeStep && mid && !below_iface && ((zero_bins >> base_bin) & 3u) == 3u) {
            words[k] = kPackedMiss;  // zero on both sides of the step
        } else if (code == kPackedCodeStep && mid) {
            const uint32_t below = tiled_bin_term((uint32_t)base_bin);
            words[k] = kPackedSlow | ((kPackedCodeLean | flags_code) << 24) | ((tiled_bin_term((uint32_t)base_bin + 1) - below) << 12) | below;
        } else if (code == (kPackedCodeStep | kPackedCodeCutoff) && mid && std::fabs(step_at - 900.0) <= 1e-9 && base_bin == 19 &&
                   ((zero_bins >> 19) & 3u) == 3u) {
            words[k] = kPackedMiss;  // bins 19 and 20 are zero: nothing to read on either side of the cutoff
        } else if (code == (kPackedCodeStep | kPackedCodeCutoff) && mid && std::fabs(step_at - 900.0) <= 1e-9 && base_bin == 19) {
            // beyond the cutoff = the unused bin slot 21 of the same patch group, which holds 0.0
            const uint32_t below = tiled_bin_term(19u);
            words[k] = kPackedSlow | (kPackedCodeLean << 24) | ((tiled_bin_term(21u) - below) << 12) | below;
        } else {
            words[k] = kPackedSlow | (code << 24) | ((uint32_t)step_bin << 16);
        }
    }
    return words;
}

// The default DFIRE kernel: f32 records in a frame centred on the receptor, the cell LUT of
// kernels/dfire_packed.hpp (every cell that cannot decide the reference's f64 result is flagged),
// and the receptor image as pair records.
void Scorer::build_packed(const ld_scorer_desc &desc) {
    double centre[3], half;
    frame_of_receptor(desc.receptor, centre, &half);
    // LUT cells per unit of 4 d2: 2 halves the share of pairs in flagged cells for 4 KiB more LDS
    int sc = kPackedWaves == 1 ? 1 : 2;  // one-wave workgroups: the LUT is per wave, the smaller one keeps 6 waves per SIMD
    if (const char *e = std::getenv("LIGHTDOCK_PACKED_CELLS")) sc = std::atoi(e) == 1 ? 1 : 2;
    const double kappa = 2.0 * std::sqrt((double)sc);
    // records hold kappa (x - c); room for the cutoff and for ANM deformations (32 A), rounded up to a power of two
    double ubound = 128.0;
    while (ubound < kappa * (half + 32.0)) ubound *= 2.0;
    double eps = dfire_f32_error_bound(ubound, sc);  // units of 4 d2
    if (const char *e = std::getenv("LIGHTDOCK_PACKED_EPS_SCALE")) {  // test hook: results must not depend on it
        const double f = std::atof(e);
        if (f >= 1.0 && f <= 1000.0) eps *= f;
    }
    if (!(eps * sc < 0.2)) return;  // a receptor thousands of angstroms across: keep the all-f64 tiled kernel

    PackedLaunch &P = packed_;
    P.lig = tiled_.lig;
    P.use_anm = tiled_.use_anm;
    P.anm_rec = tiled_.anm_rec;
    P.cx = centre[0];
    P.cy = centre[1];
    P.cz = centre[2];
    P.kappa = kappa;
    P.cells_per_unit = sc;
    P.ubound = (float)ubound;
    P.eps = std::nextafter((float)eps, INFINITY);
    P.table = tiled_.table;
    P.bin_step = pair_.bin_step;
    P.iface_scaled = 4.0 * pair_.iface_d2;

    // bins in which this complex's potential is zero throughout (bin 20 = the read past the row at r = 15.0)
    uint32_t zero_bins = 0;
    {
        const char *e = std::getenv("LIGHTDOCK_PACKED_ELIDE_ZERO_BINS");
        if (!(e && std::strcmp(e, "0") == 0)) {
            std::vector<char> rec_has(169, 0), lig_has(169, 0);
            for (size_t i = 0; i < desc.receptor.n_atoms; i++) rec_has[desc.receptor.dfire_types[i]] = 1;
            for (size_t i = 0; i < desc.ligand.n_atoms; i++) lig_has[desc.ligand.dfire_types[i]] = 1;
            for (uint32_t b = 0; b <= 20; b++) {
                bool all_zero = true;
                for (uint32_t r = 0; r < 169 && all_zero; r++)
                    for (uint32_t l = 0; l < 169 && all_zero; l++)
                        if (rec_has[r] && lig_has[l] && (size_t)r * kDfireRowStride + l * 20 + b < LD_DFIRE_TABLE_LEN &&
                            desc.potential[(size_t)r * kDfireRowStride + l * 20 + b] != 0.0)
                            all_zero = false;
                if (all_zero) zero_bins |= 1u << b;
            }
        }
    }
    packed_zero_bins_ = zero_bins;
    P.lut = arena_.upload(build_packed_lut(sc, (double)P.eps, zero_bins));
    packed_lut_full_ = zero_bins ? arena_.upload(build_packed_lut(sc, (double)P.eps, 0)) : P.lut;  // counting launches count every pair

    int split = tiled_.split;
    P.split = split;
    P.n_groups = (P.lig.n_tiles * split + kPackedWaves - 1) / kPackedWaves;

    P.rec.n_real = tiled_rec_soa_.n_real;
    P.rec.n_tiles = tiled_rec_soa_.n_tiles;
    P.rec.flag_words = pair_.rec.flag_words;
    P.rec.slot = tiled_rec_soa_.slot;
    P.rec.tindex = tiled_rec_soa_.tindex;
    if (!rec_anm_per_pose_) {
        const size_t pad = (size_t)P.rec.n_tiles * 64;
        PackedRecPair *pairs = static_cast<PackedRecPair *>(arena_.alloc_bytes(pad / 2 * sizeof(PackedRecPair)));
        TiledBox *sub = static_cast<TiledBox *>(arena_.alloc_bytes(pad / 8 * sizeof(TiledBox)));
        TiledBox *tile = static_cast<TiledBox *>(arena_.alloc_bytes(pad / 64 * sizeof(TiledBox)));
        PackedPrepareLaunch p = packed_prepare_launch(nullptr, 0, nullptr, 1);
        p.num_anm = 0;
        p.pairs_out = pairs;
        p.sub_out = sub;
        p.tile_out = tile;
        hip_check(launch_packed_prepare(p, stream_), "launch dfire_packed_prepare");
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        P.rec.pairs = pairs;
        P.rec.sub_boxes = sub;
        P.rec.tile_boxes = tile;
    }
    P.rec.x = tiled_rec_soa_.x;  // undeformed; the exact path applies the modes of a per-pose image itself
    P.rec.y = tiled_rec_soa_.y;
    P.rec.z = tiled_rec_soa_.z;
    use_packed_ = true;
}

// ---------------------------------------------------------------------------------------
// Block-major DFIRE path (kernels/dfire_bm.hpp)
// ---------------------------------------------------------------------------------------
// What flexing adds to a posed atom's f32 coordinate (record units), for poses that are not WILD (deformations below
// kBmWildUnits = W per coordinate): the ten terms amplitude x mode with both factors rounded to f32 (2^-23 W together), and the
// ten fma roundings of partial sums below `magnitude`.
static double bm_flex_error(double magnitude) {
    const double W = kBmWildUnits;
    return std::ldexp(W, -23) + 10.0 * std::ldexp(1.0, (int)std::floor(std::log2(magnitude)) - 24);
}

double dfire_bm_pose_error(double ubound, double lig_extent, bool anm) {
    // u = fl32(kappa R) x_f32 + fl32(kappa (t - c)), three fmas.  Per coordinate, in record units:
    //   matrix rounding            3 * 2^-24 * kappa * extent      (|kappa R_ij| <= kappa, |x| <= extent)
    //   local coordinate rounding  3 * kappa * 2^-24 * extent
    //   translation rounding + the three fma roundings (partial sums below 2 U): 4 * 2^-24 U
    double e = std::ldexp(6.0 * kBmKappa * lig_extent, -24) + std::ldexp(ubound, -22);
    if (anm) e += bm_flex_error(2.0 * ubound + kBmWildUnits);   // the culling kernel: mode by mode onto the posed coordinate
    return std::sqrt(3.0) * e;
}

double dfire_bm_error_bound(double ubound, double lig_extent, bool anm) {
    // What dfire_bm_pairs computes: coordinates relative to the centre c of the receptor subtile's box (an f32 constant),
    //   D'' = (|r - c|^2 + seed) + |l - c|^2 - 2 (r - c) . (l - c)
    // in f32: 3 + 3 operations for the two squares, one add, three fmas.  For the f32 coordinates this is |l - r|^2 + seed
    // up to those roundings.  Per coordinate, in record units, for atoms inside the frame (|u| <= U = ubound):
    auto half_ulp = [](double magnitude) { return std::ldexp(1.0, (int)std::floor(std::log2(magnitude)) - 24); };   // of values below `magnitude`
    const double reach = kBmKappa * lig_extent;                 // |kappa R x|
    const double e_lig = std::ldexp(6.0 * reach, -24)          // rounding of the matrix and of the local coordinates
                         + 2.0 * half_ulp(ubound)               // of the stored translation, and of translation - c
                         + 3.0 * half_ulp(ubound / 2 + reach);  // of the three fmas: partial sums below |translation - c| + |kappa R x|
    const double e_rec = half_ulp(ubound) + half_ulp(128.0);    // the record's rounding, and that of r - c
    double e_d = e_lig + e_rec;
    // Molecules that flex (poses that are not WILD): either atom's deformation d = sum_k fl(a_k) fl(kappa m_k), ten fmas from 0
    // (partial sums below W), then ONE rounded add onto the posed coordinate / onto r - c.
    const double W = kBmWildUnits;
    if (anm) e_d += 2.0 * bm_flex_error(W * 0.999) + half_ulp(ubound / 2 + reach + W) + half_ulp(128.0 + W);
    const double span = std::sqrt(3.0 * 1100.0 * kBmCells);     // |du| + |dv| + |dw| <= sqrt(3) |d|, pairs within 1100 units of 4 d2
    // The ten roundings of the distance arithmetic: every operand and partial sum of such a pair is below 2^17
    // (|l - c| <= 16.6 A + a subtile's half extent < 45 A = 362 record units; the seed carries the LUT's offset, < 2^15).
    // (flexing: r - c + d up to 128 + W, l - c up to that + the cutoff's 134 units: their squares bound every operand and partial sum)
    const double top = anm ? std::max(131072.0, (128.0 + W + 134.0) * (128.0 + W + 134.0)) : 131072.0;
    const double eps = 2.0 * e_d * span + 3.0 * e_d * e_d + 10.0 * half_ulp(top * 0.999);
    return 2.0 * eps;  // twice the bound, LUT cells
}

// The kernel computes E = kBmCellZero + 1/2 - 64 d2 (f32) and reads cell' = floor(E), everything further than the LUT
// reaches (E < 0) reading cell' 0; the cells above kBmCellZero are what an error of up to 8 cells can turn a distance near 0
// into.  Cell' k' = kBmCellZero - k holds the pairs with 64 d2 within (k - 1/2 - eps, k + 1/2 + eps),
// i.e. a true 4 d2 within ((k - 1/2 - eps) / 16, (k + 1/2 + eps) / 16).  A cell with ONE answer for that whole interval
// carries the byte offset of the bin's slot in the block's table rows (bm_code_of_bin(bin) = 8 bin for bins 0..19;
// kBmMissCode = "miss" beyond the cutoff or a bin that is zero for the whole complex); a cell with a bin step or the cutoff
// inside is flagged (kBmFlagged, the slot of the marker): the pair is recomputed in f64.  The interface distance
// (src/dfire.rs:339) lies inside bin 1: in a block with tracked atoms the kernel puts the marker into the slots of bins 0 and 1
// too (dfire_bm.hpp), so every pair that can set an interface flag reaches the exact path.
std::vector<uint8_t> build_bm_lut(double eps_cells, uint32_t zero_bins) {
    const DfireBinning b = build_dfire_binning();
    const double iface_scaled = 4.0 * dfire_interface_d2();
    if (!(iface_scaled < 4.0 * b.step[2] && iface_scaled >= 4.0 * b.step[1]))
        throw Error(LD_ERR_INVALID, "DFIRE block-major LUT: the interface distance is not inside bin 1");
    std::vector<uint8_t> codes(kBmLutBytes, (uint8_t)kBmMissCode);
    for (int k = kBmCellZero - (kBmLutBytes - 1); k <= kBmCellZero; k++) {
        uint8_t &code = codes[kBmCellZero - k];
        const double ilo = (k - 0.5 - eps_cells) / kBmCells, ihi = (k + 0.5 + eps_cells) / kBmCells;
        if (ilo > 900.0) continue;  // beyond the cutoff for sure
        bool flagged = ihi >= 900.0;
        int base_bin = 0;
        for (int s = 1; s <= 20; s++) {
            const double at = 4.0 * b.step[s];
            if (at < ilo) base_bin = s;
            else if (at <= ihi) flagged = true;
        }
        if (flagged || base_bin > 19) {
            code = (uint8_t)kBmFlagged;
            continue;
        }
        if (dfire_bin_reference(std::max(ilo, 0.0) / 4.0) != base_bin || dfire_bin_reference(ihi / 4.0) != base_bin)
            throw Error(LD_ERR_INVALID, "DFIRE block-major LUT self-check failed in cell " + std::to_string(k));
        // (bins 0 and 1 are never elided: a block with tracked atoms recognises its flag-setting pairs by their slots)
        code = base_bin >= 2 && ((zero_bins >> base_bin) & 1u) ? (uint8_t)kBmMissCode : (uint8_t)bm_code_of_bin((uint32_t)base_bin);
    }
    if (codes[0] != kBmMissCode) throw Error(LD_ERR_INVALID, "DFIRE block-major LUT: the far cell is not a miss");
    return codes;
}

void Scorer::frame_of_receptor(const ld_molecule &rec, double centre[3], double *half) const {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < rec.n_atoms; i++)
        for (int c = 0; c < 3; c++) {
            lo[c] = std::min(lo[c], rec.coordinates[3 * i + c]);
            hi[c] = std::max(hi[c], rec.coordinates[3 * i + c]);
        }
    *half = 0.0;
    for (int c = 0; c < 3; c++) {
        centre[c] = 0.5 * (lo[c] + hi[c]);
        *half = std::max(*half, 0.5 * (hi[c] - lo[c]));
    }
}

// ---- the fixed-point scale of the block-major path, count-aware.  Table values enter the sums as rint(v * 2^(44 - e - x)),
// 2^e >= the table's largest |value|.  What has to hold (dfire_bm.hpp): 32 pairs of a block stay below 2^50 (the markers sit at
// bit 51) -- true for x >= 0 whatever the complex; and a (row, ligand tile) sum, which collects 64 ligand atoms x every receptor
// atom within the cutoff of any of them, stays below 2^63: 64 K 2^(44 - x) < 2^63 with K = the number of receptor atoms one
// ligand tile can reach.  K is bounded by geometry: dfire_bm_reach_count() = an upper bound on the receptor atoms inside ANY
// ball of radius `reach` = cutoff + the largest ligand tile's radius.  x = the bits K takes beyond 2^13 (0 for every real
// protein: ~3 000 atoms in such a ball).
size_t dfire_bm_reach_count(const double *xyz, size_t n, double reach) {
    if (n < 8192) return n;   // cannot overflow whatever the geometry: no search
    // Any ball of radius `reach` with its centre c in a grid cell of side h lies inside the ball of radius reach + h sqrt(3) / 2
    // around the cell's centre; centres outside the atoms' bounding box see a subset of what their projection onto it sees.
    const double h = 4.0, rho = reach + h * 0.8660254037844387, rho2 = rho * rho;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            if (!std::isfinite(xyz[3 * i + c])) return n;   // (no cell for such an atom: the trivial bound)
            lo[c] = std::min(lo[c], xyz[3 * i + c]);
            hi[c] = std::max(hi[c], xyz[3 * i + c]);
        }
    // the atoms binned in cells of side rho: a probe reads the 27 cells around its own
    // (Guards in double, before any cast: an outlier atom -- a 9999.999 dummy coordinate -- or a NaN makes the extent absurd; the
    // probe grid below has (rho / h)^3 ~ 340 points per cell, so it is the PROBE count that bounds the search: beyond ~10^7 probes
    // the trivial bound n is returned instead -- only coarser, never wrong.)
    double cells = 1.0, n_probes = 1.0;
    for (int c = 0; c < 3; c++) {
        const double ext = hi[c] - lo[c];
        if (!(ext >= 0.0) || !(ext < 1e6)) return n;
        cells *= std::floor(ext / rho) + 1.0;
        n_probes *= std::floor(ext / h) + 1.0;
    }
    if (cells > 1e7 || n_probes > 1e7) return n;
    int dim[3];
    for (int c = 0; c < 3; c++) dim[c] = (int)std::floor((hi[c] - lo[c]) / rho) + 1;
    auto cell_of = [&](const double *p, int *ijk) {
        for (int c = 0; c < 3; c++) ijk[c] = std::min(dim[c] - 1, std::max(0, (int)std::floor((p[c] - lo[c]) / rho)));
    };
    std::vector<uint32_t> start((size_t)dim[0] * dim[1] * dim[2] + 1, 0), order(n);
    std::vector<uint32_t> cell(n);
    for (size_t i = 0; i < n; i++) {
        int ijk[3];
        cell_of(xyz + 3 * i, ijk);
        cell[i] = (uint32_t)((ijk[2] * dim[1] + ijk[1]) * dim[0] + ijk[0]);
        start[cell[i] + 1]++;
    }
    for (size_t k = 1; k < start.size(); k++) start[k] += start[k - 1];
    {
        std::vector<uint32_t> at(start.begin(), start.end() - 1);
        for (size_t i = 0; i < n; i++) order[at[cell[i]]++] = (uint32_t)i;
    }
    size_t best = 0;
    int probes[3];
    for (int c = 0; c < 3; c++) probes[c] = (int)std::floor((hi[c] - lo[c]) / h) + 1;
    for (int pz = 0; pz < probes[2]; pz++)
        for (int py = 0; py < probes[1]; py++)
            for (int px = 0; px < probes[0]; px++) {
                const double p[3] = {lo[0] + (px + 0.5) * h, lo[1] + (py + 0.5) * h, lo[2] + (pz + 0.5) * h};
                int ijk[3];
                cell_of(p, ijk);
                size_t count = 0;
                for (int dz = -1; dz <= 1; dz++)
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            const int x = ijk[0] + dx, y = ijk[1] + dy, z = ijk[2] + dz;
                            if (x < 0 || y < 0 || z < 0 || x >= dim[0] || y >= dim[1] || z >= dim[2]) continue;
                            const size_t k = (size_t)(z * dim[1] + y) * dim[0] + x;
                            for (uint32_t a = start[k]; a < start[k + 1]; a++) {
                                const double *q = xyz + 3 * (size_t)order[a];
                                const double ex = q[0] - p[0], ey = q[1] - p[1], ez = q[2] - p[2];
                                count += ex * ex + ey * ey + ez * ez <= rho2 ? 1 : 0;
                            }
                        }
                best = std::max(best, count);
            }
    return best;
}

// 2^(44 - e - x); returns 0.0 when no scale fits (a table beyond kBmFixLimit, a non-finite value, or an absurd reach count)
double dfire_bm_fix_scale(double vmax, size_t reach_count, int *extra_bits_out) {
    if (!(vmax <= kBmFixLimit)) return 0.0;
    int e = 0;
    while (std::ldexp(1.0, e) < std::max(vmax, 1.0)) e++;
    int x = 0;
    while (((reach_count + ((size_t)1 << x) - 1) >> x) > 8191) x++;   // 64 K 2^(44 - x) < 2^63  <=>  ceil(K / 2^x) <= 8191
    if (extra_bits_out) *extra_bits_out = x;
    if (x > 10) return 0.0;
    return std::ldexp(1.0, 44 - e - x);
}

void Scorer::build_bm(const ld_scorer_desc &desc) {
#ifdef LD_DIAG_BUILD
    // (diagnostic builds only, tools/build_variant.sh -- LIGHTDOCK_BM_DIAG_IGNORE_ANM=1: timing experiments, the block-major kernels
    // on an ANM complex as if it were rigid, wrong sums.  The shipped library does not read the variable.)
    const char *ignore_anm = std::getenv("LIGHTDOCK_BM_DIAG_IGNORE_ANM");
    const bool diag_rigid = ignore_anm && std::atoi(ignore_anm) == 1;
#else
    constexpr bool diag_rigid = false;
#endif
    const TiledSoA &rec = tiled_rec_soa_, &lig = tiled_lig_soa_;
    // Molecules that flex per pose (src/dfire.rs:288-320): the ANM form of the kernels, for up to kBmMaxModes modes a molecule (more:
    // the pose-major kernel -- a stated contract, tests/test_gpu_parity.py::test_which_kernel_a_flexing_complex_gets).  The
    // fixed-point scale's reach count allows for the deformation below (until round 6 a receptor of 8192 atoms or more was declined
    // instead).  LIGHTDOCK_BM_ANM=0: such complexes stay with the pose-major kernels (A/B, tests).
    const bool anm = !diag_rigid && use_anm_ && (rec.num_anm > 0 || lig.num_anm > 0);
    if (anm) {
        const char *e = std::getenv("LIGHTDOCK_BM_ANM");
        if (e && std::atoi(e) == 0) return;
        if (rec.num_anm > kBmMaxModes || lig.num_anm > kBmMaxModes) return;
    }
    if (rec.n_tiles > 1024 || lig.n_tiles > 1024) return;  // an item of the exact path names its atoms in 16 bits each
    if (bm_cull_lds_bytes(rec.n_tiles) + 1024 > kBmLdsPerCu) return;  // the culling kernel keeps every receptor box in LDS: ~430 tiles at most
    // Table values reach a pose's sum as 64-bit fixed point (dfire_bm.hpp): a table that could overflow it or that the scale
    // would resolve too coarsely (beyond kBmFixLimit), or one that holds a value the reference would carry as inf / NaN
    // (src/dfire.rs:338), stays with the pose-major kernels.
    double table_vmax = 0.0;
    for (size_t i = 0; i < LD_DFIRE_TABLE_LEN; i++) {
        if (!(std::fabs(desc.potential[i]) <= kBmFixLimit)) return;
        table_vmax = std::max(table_vmax, std::fabs(desc.potential[i]));
    }
    double centre[3], half;
    frame_of_receptor(desc.receptor, centre, &half);
    // The frame holds the receptor's box + 16 A: a ligand atom outside it is beyond the cutoff of every receptor atom.
    // (flexing: a receptor atom of a pose that is not WILD moves less than kBmWildUnits)
    double ubound = 128.0;
    while (ubound < kBmKappa * (half + 16.5) + (anm ? (double)kBmWildUnits : 0.0)) ubound *= 2.0;
    double extent = 0.0;
    for (size_t i = 0; i < desc.ligand.n_atoms * 3; i++) extent = std::max(extent, std::fabs(desc.ligand.coordinates[i]));
    double eps = dfire_bm_error_bound(ubound, extent, anm);  // LUT cells
    if (const char *e = std::getenv("LIGHTDOCK_PACKED_EPS_SCALE")) {  // test hook: results must not depend on it
        const double f = std::atof(e);
        if (f >= 1.0 && f <= 1000.0) eps *= f;
    }
    if (!(eps < 8.0)) return;  // a complex thousands of angstroms across: the pose-major kernels

    BmModel &M = bm_;
    M.rec_n_real = rec.n_real;
    M.rec_n_tiles = rec.n_tiles;
    M.rec_x = rec.x;
    M.rec_y = rec.y;
    M.rec_z = rec.z;
    M.rec_tindex = rec.tindex;
    M.rec_slot = rec.slot;
    M.rec_flag_words = pair_.rec.flag_words;
    M.lig = tiled_.lig;
    M.cx = centre[0];
    M.cy = centre[1];
    M.cz = centre[2];
    M.ubound = (float)ubound;
    M.box_pad = std::nextafter((float)(2.0 * dfire_bm_pose_error(ubound, extent, anm)), INFINITY);
    M.table = tiled_.table;
    M.iface_scaled = 4.0 * pair_.iface_d2;
    {
        const DfireBinning b = build_dfire_binning();
        std::vector<double> step4(b.step);
        for (double &v : step4) v *= 4.0;
        M.bin_step = arena_.upload(step4);
    }
    M.lut = arena_.upload(build_bm_lut(eps, packed_zero_bins_));
    M.lut_full = packed_zero_bins_ ? arena_.upload(build_bm_lut(eps, 0)) : M.lut;  // counting launches count every pair
    {   // subtiles that hold an atom with an interface-flag slot (restraint atoms, membrane beads)
        auto tracked = [](const TiledSoA &m) {
            std::vector<uint8_t> t(m.hslot.size() / 8, 0);
            for (size_t i = 0; i < m.hslot.size(); i++)
                if (m.hslot[i] >= 0) t[i / 8] = 1;
            return t;
        };
        M.rec_sub_tracked = arena_.upload(tracked(rec));
        M.lig_sub_tracked = arena_.upload(tracked(lig));
    }
    {   // receptor image in this frame, by the kernel that builds the packed kernel's
        const size_t pad = (size_t)rec.n_tiles * 64;
        PackedRecPair *pairs = static_cast<PackedRecPair *>(arena_.alloc_bytes(pad / 2 * sizeof(PackedRecPair)));
        TiledBox *sub = static_cast<TiledBox *>(arena_.alloc_bytes(pad / 8 * sizeof(TiledBox)));
        TiledBox *tile = static_cast<TiledBox *>(arena_.alloc_bytes(pad / 64 * sizeof(TiledBox)));
        PackedPrepareLaunch p = packed_prepare_launch(nullptr, 0, nullptr, 1);
        p.num_anm = 0;
        p.cx = centre[0];
        p.cy = centre[1];
        p.cz = centre[2];
        p.kappa = kBmKappa;
        p.ubound = (float)ubound;
        p.pairs_out = pairs;
        p.sub_out = sub;
        p.tile_out = tile;
        hip_check(launch_packed_prepare(p, stream_), "launch dfire_packed_prepare");
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        M.rec_pairs = pairs;
        M.rec_sub = sub;
        M.rec_tile = tile;
        {   // every receptor subtile's packed operands (BmModel::rec_ops): E = (seed - |r - c|^2) - |l - c|^2 + 2 (r - c) . (l - c), dfire_bm.hip.
            // f32, operation by operation what bm_recheck computes on the device (IEEE: the same bits).
            std::vector<PackedRecPair> hp(pad / 2);
            std::vector<TiledBox> hb(pad / 8);
            hip_check(hipMemcpy(hp.data(), pairs, hp.size() * sizeof(PackedRecPair), hipMemcpyDeviceToHost), "D2H receptor records");
            hip_check(hipMemcpy(hb.data(), sub, hb.size() * sizeof(TiledBox), hipMemcpyDeviceToHost), "D2H receptor boxes");
            const float seed = (float)kBmCellZero + 0.5f;
            std::vector<float> ops((pad / 8) * (size_t)kBmOpsFloats, 0.f);
            for (size_t sbt = 0; sbt < pad / 8; sbt++) {
                const TiledBox &box = hb[sbt];
                const float cbx = 0.5f * (box.lox + box.hix), cby = 0.5f * (box.loy + box.hiy), cbz = 0.5f * (box.loz + box.hiz);
                float *o = &ops[sbt * kBmOpsFloats];
                for (int q = 0; q < 4; q++) {
                    const PackedRecPair &r = hp[sbt * 4 + q];
                    const float xs[2] = {r.x0 - cbx, r.x1 - cbx}, ys[2] = {r.y0 - cby, r.y1 - cby}, zs[2] = {r.z0 - cbz, r.z1 - cbz};
                    for (int h = 0; h < 2; h++) {
                        o[2 * q + h] = std::fmaf(-xs[h], xs[h], std::fmaf(-ys[h], ys[h], std::fmaf(-zs[h], zs[h], seed)));
                        o[8 + 2 * q + h] = zs[h] * 2.f;
                        o[16 + 2 * q + h] = ys[h] * 2.f;
                        o[24 + 2 * q + h] = xs[h] * 2.f;
                    }
                }
                o[32] = cbx; o[33] = cby; o[34] = cbz;
            }
            M.rec_ops = arena_.upload(ops);
            // A receptor subtile's REACH for the culling kernel's box tests, in the spare word of its box (and the largest of a tile's
            // eight in the tile's): (8 x 15 A)^2 for every subtile whose atoms can add to a sum.  A receptor type whose rows of the
            // potential are 0.0 against every ligand type of the complex -- all 20 bins and the read past the row at r = 15.0
            // (src/dfire.rs:336-338: bin 20 = the next type's bin 0) -- adds nothing at any distance: a subtile of such atoms only
            // (lightdock's membrane beads, if the DCparams at hand carries zero rows for them) is listed within the interface
            // distance alone (d <= 3.9, i.e. r <= 2.45 A, src/dfire.rs:339) when it or the ligand holds an atom with an
            // interface-flag slot, and never otherwise.  Exact: a block that is not listed holds no pair that changes the sum or a
            // flag.  Counting launches test against the full cutoff (they count pairs, not values).  VERDICT r05 item 8.
            std::vector<char> lig_has(169, 0), quiet_type(169, 1);
            for (size_t i = 0; i < desc.ligand.n_atoms; i++) lig_has[desc.ligand.dfire_types[i]] = 1;
            for (uint32_t r = 0; r < 169; r++)
                for (uint32_t l = 0; l < 169 && quiet_type[r]; l++)
                    for (uint32_t b = 0; b <= 20 && quiet_type[r] && lig_has[l]; b++) {
                        const size_t at = (size_t)r * kDfireRowStride + (size_t)l * 20 + b;
                        if (at >= LD_DFIRE_TABLE_LEN || desc.potential[at] != 0.0) quiet_type[r] = 0;   // (past the table's end: not ours to reason about)
                    }
            bool lig_tracked_any = false;
            for (int32_t v : lig.hslot) lig_tracked_any = lig_tracked_any || v >= 0;
            const float full_cut = kBmBoxCutUnits2, iface_cut = 400.0f * 1.00005f;   // (8 x 15 A)^2 and (8 x 2.5 A)^2 record units, padded like the full one
            std::vector<TiledBox> ht(pad / 64);
            hip_check(hipMemcpy(ht.data(), tile, ht.size() * sizeof(TiledBox), hipMemcpyDeviceToHost), "D2H receptor tile boxes");
            bm_quiet_subtiles_ = 0;
            for (size_t t = 0; t < pad / 64; t++) {
                float tile_cut = -1.0f;
                for (size_t sb = 0; sb < 8; sb++) {
                    const size_t sbt = t * 8 + sb;
                    bool quiet = true, any = false, tracked = false;
                    for (size_t k = 0; k < 8; k++) {
                        const uint32_t ty = rec.htype[sbt * 8 + k];
                        if (ty == 0xffffffffu) continue;
                        any = true;
                        quiet = quiet && ty < 169 && quiet_type[ty];
                        tracked = tracked || rec.hslot[sbt * 8 + k] >= 0;
                    }
                    float cut = full_cut;
                    if (any && quiet) {
                        cut = (tracked || lig_tracked_any) ? iface_cut : -1.0f;
                        bm_quiet_subtiles_++;
                    }
                    hb[sbt].pad0 = cut;
                    hb[sbt].pad1 = 0.f;
                    tile_cut = std::max(tile_cut, cut);
                }
                ht[t].pad0 = tile_cut;
                ht[t].pad1 = 0.f;
            }
            hip_check(hipMemcpy(sub, hb.data(), hb.size() * sizeof(TiledBox), hipMemcpyHostToDevice), "H2D receptor boxes");
            hip_check(hipMemcpy(tile, ht.data(), ht.size() * sizeof(TiledBox), hipMemcpyHostToDevice), "H2D receptor tile boxes");
        }
    }
    if (anm) {   // the modes as the kernels read them: kappa x, f32 (BmModel)
        M.anm_rec = rec.num_anm;
        M.anm_lig = lig.num_anm;
        auto tables = [&](const TiledSoA &m, std::vector<float> &by_subtile, std::vector<float> *by_atom, float *norm_coord, float *norm_atom, const double **exact) {
            const size_t pad = (size_t)m.n_tiles * 64;
            std::vector<double> per_atom(pad * 3 * (size_t)kBmMaxModes, 0.0);   // [atom][mode][x y z], the reference's numbers
            for (int k = 0; k < m.num_anm; k++)
                for (size_t i = 0; i < pad; i++)
                    for (int c = 0; c < 3; c++) per_atom[(i * kBmMaxModes + (size_t)k) * 3 + c] = m.hmodes[((size_t)k * 3 + c) * pad + i];
            *exact = arena_.upload(per_atom);
            by_subtile.assign(pad / 8 * (size_t)kBmModeFloats, 0.f);
            if (by_atom) by_atom->assign(pad * 32, 0.f);
            // per atom: the 2-norm of the ten components of a coordinate, and of all thirty -- of the f32 values the kernels multiply with
            std::vector<double> n2((size_t)pad * 3, 0.0);
            for (int k = 0; k < m.num_anm; k++)
                for (size_t i = 0; i < pad; i++)
                    for (int c = 0; c < 3; c++) {
                        const double v = m.hmodes[((size_t)k * 3 + c) * pad + i];
                        const float f = (float)(kBmKappa * v);
                        by_subtile[(i / 8) * (size_t)kBmModeFloats + ((((i % 8) / 2) * 3 + c) * (size_t)kBmMaxModes + k) * 2 + (i & 1)] = f;
                        if (by_atom) (*by_atom)[i * 32 + 3 * (size_t)k + c] = f;
                        n2[i * 3 + c] += (double)f * (double)f;
                    }
            double coord = 0.0, atom = 0.0;
            for (size_t i = 0; i < pad; i++) {
                for (int c = 0; c < 3; c++) coord = std::max(coord, n2[i * 3 + c]);
                atom = std::max(atom, n2[i * 3] + n2[i * 3 + 1] + n2[i * 3 + 2]);
            }
            // (not finite: every pose with a non-zero amplitude is wild; NaN stays NaN and fails the kernel's comparison)
            *norm_coord = std::nextafter((float)(std::sqrt(coord) * 1.000001), INFINITY);
            if (norm_atom) *norm_atom = std::nextafter((float)(std::sqrt(atom) * 1.000001), INFINITY);
        };
        std::vector<float> rsub, lsub, latom, ratom;
        tables(rec, rsub, &ratom, &M.rec_mode_norm, nullptr, &M.rec_modes_exact);
        tables(lig, lsub, &latom, &M.lig_mode_norm, &M.lig_mode_norm_vec, &M.lig_modes_exact);
        M.rec_modes_f32 = arena_.upload(rsub);
        M.lig_modes_f32 = arena_.upload(lsub);
        M.lig_modes_atom = arena_.upload(latom);
        M.rec_modes_atom = arena_.upload(ratom);
        // dfire_bm_rec_boxes' atoms: the static record's rounding, the ten terms and their fma roundings (partial sums inside the frame); x 2
        M.rec_box_pad = std::nextafter((float)(2.0 * (std::ldexp(ubound, -24) + bm_flex_error(ubound * 0.999))), INFINITY);
    }
    const uint32_t kPad = std::numeric_limits<uint32_t>::max();
    {   // per atom: where its type's rows / column sit in the row table; padding atoms take the all-zero type
        std::vector<uint32_t> rowoff(rec.htype.size()), rowbase(lig.htype.size());
        for (size_t i = 0; i < rowoff.size(); i++) rowoff[i] = (rec.htype[i] == kPad ? (uint32_t)kBmTypes - 1 : rec.htype[i]) * (uint32_t)kBmRowBytes;
        for (size_t i = 0; i < rowbase.size(); i++)
            rowbase[i] = (lig.htype[i] == kPad ? (uint32_t)kBmTypes - 1 : lig.htype[i]) * (uint32_t)(kBmTypes * kBmRowBytes);
        M.rec_rowoff = arena_.upload(rowoff);
        M.lig_rowbase = arena_.upload(rowbase);
        std::vector<float> local(lig.htype.size() * 4, 0.f);
        for (size_t i = 0; i < lig.htype.size(); i++) {
            if (lig.htype[i] == kPad) continue;
            local[4 * i] = (float)lig.hx[i];
            local[4 * i + 1] = (float)lig.hy[i];
            local[4 * i + 2] = (float)lig.hz[i];
            local[4 * i + 3] = 1.f;
        }
        M.lig_local = arena_.upload(local);
        // what the exact path reads of an atom, together: x, y, z and {table term, interface-flag slot} in the fourth double's bytes
        auto exact_rows = [&](const TiledSoA &m, bool is_receptor) {
            const std::vector<uint32_t> &perm = is_receptor ? type_perm_rec_ : type_perm_lig_;
            std::vector<double> rows(m.htype.size() * 4, 0.0);
            for (size_t i = 0; i < m.htype.size(); i++) {
                rows[4 * i] = m.hx[i];
                rows[4 * i + 1] = m.hy[i];
                rows[4 * i + 2] = m.hz[i];
                const uint32_t term = m.htype[i] == kPad ? 0u : is_receptor ? tiled_rec_term(perm[m.htype[i]]) : tiled_lig_term(perm[m.htype[i]]);
                const uint32_t words[2] = {term, (uint32_t)m.hslot[i]};
                std::memcpy(&rows[4 * i + 3], words, 8);
            }
            return rows;
        };
        M.lig_exact = arena_.upload(exact_rows(lig, false));
        M.rec_exact = arena_.upload(exact_rows(rec, true));
        // a sphere around every ligand tile (rotation invariant): centre of its box, radius to its farthest atom
        std::vector<float> sphere((size_t)lig.n_tiles * 4, 0.f);
        bm_tile_radius_.clear();
        for (int t = 0; t < lig.n_tiles; t++) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int i = t * 64; i < t * 64 + 64; i++) {
                if (lig.htype[i] == kPad) continue;
                const double c[3] = {lig.hx[i], lig.hy[i], lig.hz[i]};
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::min(lo[k], c[k]);
                    hi[k] = std::max(hi[k], c[k]);
                }
            }
            float ctr[3];
            for (int k = 0; k < 3; k++) ctr[k] = lo[k] <= hi[k] ? (float)(0.5 * (lo[k] + hi[k])) : 0.f;
            double r = 0.0;
            for (int i = t * 64; i < t * 64 + 64; i++) {
                if (lig.htype[i] == kPad) continue;
                const double dx = lig.hx[i] - ctr[0], dy = lig.hy[i] - ctr[1], dz = lig.hz[i] - ctr[2];
                r = std::max(r, std::sqrt(dx * dx + dy * dy + dz * dz));
            }
            for (int k = 0; k < 3; k++) sphere[4 * t + k] = ctr[k];
            sphere[4 * t + 3] = std::nextafter((float)(kBmKappa * r * 1.000001 + 1e-3), INFINITY);
            bm_tile_radius_.push_back((float)(r * 1.000001 + 1e-3));
        }
        M.lig_tile_sphere = arena_.upload(sphere);
    }
    {   // rows[l][r][b] = potential[r * 3380 + l * 20 + b] in fixed point, b = 0..19; slot 20 = 0, slot 21: the kernel's marker
        // the scale: 32 pairs of a block stay below 2^49, under the markers, and a (row, ligand tile) sum -- 64 ligand atoms x
        // the receptor atoms one ligand tile can reach -- inside 63 bits (dfire_bm_fix_scale)
        double tile_radius = 0.0;   // angstrom: the largest ligand tile's bounding sphere
        for (int t = 0; t < lig.n_tiles; t++) tile_radius = std::max(tile_radius, (double)bm_tile_radius_[t]);
        // (flexing: in a pose that is not WILD no coordinate of an atom moves further than kBmWildUnits / kappa = 16 A, the atom no
        // further than sqrt(3) x that = 27.7 A from its place -- the atoms of a ligand tile stay within its radius + 27.7 A of its
        // centre, a receptor atom within the cutoff of one of them rests within another 27.7 A: the ball grows by 55.4 A; a wild
        // pose's sums are the exact path's, pair by pair)
        const double flex_reach = anm ? 2.0 * std::sqrt(3.0) * (double)kBmWildUnits / kBmKappa : 0.0;   // (W bounds a coordinate: sqrt(3) W the atom)
        const size_t reach_count = dfire_bm_reach_count(desc.receptor.coordinates, desc.receptor.n_atoms, 15.0 + tile_radius + 0.01 + flex_reach);
        M.fix_scale = dfire_bm_fix_scale(table_vmax, reach_count, nullptr);
        if (!(M.fix_scale > 0.0)) return;
        std::vector<long long> rows((size_t)kBmTypes * kBmTypes * kBmRowSlots, 0), ones(rows.size(), 0);
        for (uint32_t l = 0; l < (uint32_t)kBmTypes; l++)
            for (uint32_t r = 0; r < (uint32_t)kBmTypes; r++) {
                long long *row = &rows[((size_t)l * kBmTypes + r) * kBmRowSlots], *one = &ones[((size_t)l * kBmTypes + r) * kBmRowSlots];
                if (l >= 169 || r >= 169) continue;   // the all-zero type of padding atoms
                for (uint32_t b = 0; b <= 19; b++) {
                    row[b] = std::llrint(desc.potential[(size_t)r * kDfireRowStride + l * 20 + b] * M.fix_scale);
                    one[b] = 1;
                }
            }
        M.rows = arena_.upload(rows);
        M.rows_ones = arena_.upload(ones);
    }
    // poses per pass: the entry workspace is (tile pairs) x (poses of the pass) x 12 bytes (an entry's row and block mask; until
    // round 5 also 64 bytes of partial sums per entry, which now live in a per-wave scratch: the budgets kept their pass sizes)
    const size_t tile_pairs = (size_t)rec.n_tiles * lig.n_tiles;
    constexpr size_t kEntryBytes = 12;
    if (tile_pairs * kBmPassQuantum * kEntryBytes > ((size_t)2560 << 20)) return;   // even the smallest pass (1024 poses) would not fit 2.5 GiB: the pose-major kernels
    size_t chunk = ((size_t)640 << 20) / (kEntryBytes * tile_pairs);   // a second such workspace exists while two passes are in flight
    chunk = std::min<size_t>(kBmMaxPassPoses, std::max<size_t>(kBmPassQuantum, chunk / kBmPassQuantum * kBmPassQuantum));
    if (const char *e = std::getenv("LIGHTDOCK_BM_CHUNK")) {   // tests, A/B: any pass size that the layout can hold --
        // an entry's index (tile pair * cap + entry) travels in 32 bits (bm_block_item), and the override stays inside the 16 GiB
        // the default's guard admits for the smallest pass
        const long v = std::atol(e);
        const size_t by_index = ((size_t)1 << 32) / tile_pairs - 1, by_bytes = ((size_t)2560 << 20) / (kEntryBytes * tile_pairs);
        if (v >= 1) chunk = std::max<size_t>(1, std::min<size_t>({(size_t)v, kBmMaxPassPoses, by_index, by_bytes}));
    }
    if (tile_pairs * chunk >= ((size_t)1 << 32)) return;   // (unreachable with the 640 MiB default: 12 bytes an entry)
    bm_chunk_ = chunk;
    {
        const char *e = std::getenv("LIGHTDOCK_BM_LANES");
        if (!(e && std::atoi(e) == 1)) {
            hip_check(hipStreamCreateWithFlags(&bm_aux_stream_, hipStreamNonBlocking), "hipStreamCreate");
            hip_check(hipEventCreateWithFlags(&bm_fork_, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&bm_join_, hipEventDisableTiming), "hipEventCreate");
        }
    }
    use_bm_ = true;
}

// Poses per block-major pass: at most bm_chunk_ (the entry workspace).  Batches that need several passes alternate them between two
// streams: the culling and gathering kernels of one pass wait on memory while the pair kernel of the other computes.
size_t Scorer::bm_pass_poses(size_t n) const {
    // (Halving a batch that fits one pass so that two passes overlap was measured and lost: 8192 poses of 1k4c as 2 x 4096 on
    // two streams 3.08 M evaluations/s against 3.43 M in one pass -- each pass pays its own tail of long jobs.)
    return std::min(n, bm_chunk_);
}

// Workspace sets of a block-major batch: one per pass in flight (two when the batch needs several passes and the second
// stream exists).  Everything but the per-pose outputs (flags, partial sums) is indexed by the row of the pass.
size_t Scorer::bm_sets(size_t n) const { return n > bm_pass_poses(n) && bm_aux_stream_ != nullptr ? 2 : 1; }

void Scorer::run_bm(size_t n, const double *d_poses, size_t stride, const uint8_t *d_active, bool counts, const uint32_t *d_list,
                    const uint32_t *d_count) {
    const size_t n_lt = (size_t)bm_.lig.n_tiles, tile_pairs = n_lt * bm_.rec_n_tiles;
    const size_t cap = bm_pass_poses(n);
    const size_t parts = tile_pairs * (cap / 64 + 1);   // at most entries / 64 + tile pairs (tile pair, part) pairs
    const size_t waves = (size_t)n_cus_ * kBmWavesPerCu;
    BmLaunch t;
    t.m = bm_;
    t.poses = d_poses;
    t.stride = stride;
    t.active = d_list ? nullptr : d_active;  // the list holds exactly the active rows
    t.pose_list = d_list;
    t.pose_count = d_list ? d_count : nullptr;
    t.cap = cap;
    t.pairs_groups = n_cus_;
    t.flags = static_cast<uint32_t *>(ws_flags_.ptr);
    t.partial = static_cast<double *>(ws_partial_.ptr);
    if (counts) {
        t.count_partial = static_cast<uint32_t *>(ws_counts_.ptr);
        t.tested_partial = static_cast<uint32_t *>(ws_tested_.ptr);
        t.exact_partial = static_cast<uint32_t *>(ws_exact_.ptr);
    }
    const bool anm = bm_.anm_rec + bm_.anm_lig > 0;
    if (anm) t.part_cap = (uint32_t)kBmAnmPartEntries;
    if (const char *e = std::getenv("LIGHTDOCK_BM_PART_CAP")) {   // A/B: jobs of fewer entries than the LDS has room for (a multiple of 64)
        const long v = std::atol(e);
        if (v >= 64 && v % 64 == 0 && (uint32_t)v <= (anm ? (uint32_t)kBmAnmPartEntries : (uint32_t)kBmPartEntries)) t.part_cap = (uint32_t)v;
    }
    const char *dbg = std::getenv("LIGHTDOCK_BM_DEBUG");
    if (dbg) {
        ws_bm_debug_.reserve(waves * 8 * sizeof(unsigned long long));
        t.debug = static_cast<unsigned long long *>(ws_bm_debug_.ptr);
    }
    // Passes of at most `cap` poses (poses are independent), alternating between this handle's stream and a second
    // one: fork behind what the stream holds so far, join before what follows.
    const bool two_lanes = bm_sets(n) == 2;
    if (two_lanes) {
        hip_check(hipEventRecord(bm_fork_, stream_), "hipEventRecord");
        hip_check(hipStreamWaitEvent(bm_aux_stream_, bm_fork_, 0), "hipStreamWaitEvent");
    }
    int lane = 0;
    for (size_t off = 0; off < n; off += cap, lane ^= 1) {
        hipStream_t st = lane && two_lanes ? bm_aux_stream_ : stream_;
        const size_t w = two_lanes ? (size_t)lane : 0;   // workspace set
        t.first = off;
        t.n_poses = std::min(cap, n - off);
        t.rt = static_cast<float *>(ws_bm_rt_.ptr) + w * cap * 12;
        t.rt_exact = reinterpret_cast<double *>(static_cast<float *>(ws_bm_rt_.ptr) + bm_sets(n) * cap * 12) + w * cap * 8;
        t.tp_count = static_cast<uint32_t *>(ws_bm_tp_count_.ptr) + w * (tile_pairs + kBmCounters + kBmCullQueueWords);
        t.job_count = t.tp_count + tile_pairs;
        t.job_next = t.tp_count + tile_pairs + 1;
        t.jobs = static_cast<uint32_t *>(ws_bm_jobs_.ptr) + w * parts * 2;
        t.job_cost = static_cast<uint32_t *>(ws_bm_job_cost_.ptr) + w * parts * kBmJobRows;
        t.job_order = static_cast<uint32_t *>(ws_bm_job_order_.ptr) + w * parts * kBmJobRows;
        t.job_rec = static_cast<uint32_t *>(ws_bm_job_order_.ptr) + bm_sets(n) * parts * kBmJobRows + w * parts * kBmJobRows * 4;
        t.queue = static_cast<unsigned long long *>(ws_bm_queue_.ptr) + w * waves * kBmQueueCap;
        t.ent_row = static_cast<uint32_t *>(ws_bm_ent_row_.ptr) + w * tile_pairs * cap;
        t.ent_mask = static_cast<unsigned long long *>(ws_bm_ent_mask_.ptr) + w * tile_pairs * cap;
        t.ent_partial = static_cast<long long *>(ws_bm_ent_partial_.ptr) + w * waves * kBmPartEntries;
        t.tile_sum = static_cast<long long *>(ws_bm_tile_sum_.ptr) + w * cap * n_lt;
        t.exact_fix = static_cast<long long *>(ws_bm_exact_fix_.ptr) + w * cap;
        t.amp = anm ? static_cast<float *>(ws_bm_amp_.ptr) + w * cap * kBmAmpFloats : nullptr;
        // the flexed receptor's boxes of the pass's rows: subtile boxes of every set, then the tile boxes
        t.anm_sub = anm ? static_cast<TiledBox *>(ws_rec_sub_.ptr) + w * cap * (size_t)bm_.rec_n_tiles * 8 : nullptr;
        t.anm_tile = anm ? static_cast<TiledBox *>(ws_rec_tile_.ptr) + w * cap * (size_t)bm_.rec_n_tiles : nullptr;
        t.amp_exact = anm ? reinterpret_cast<double *>(static_cast<float *>(ws_bm_amp_.ptr) + bm_sets(n) * cap * kBmAmpFloats) + w * cap * 2 * kBmMaxModes : nullptr;
        // With pair counts wanted the sequence runs twice: first as a counting launch (the same kernels over rows of ones and the
        // full LUT: the sums are the in-cutoff pair counts), then for the energies.
        for (int mode = counts ? 1 : 0; mode >= 0; mode--) {
            t.count_mode = mode;
            t.tile_tested = mode ? static_cast<uint32_t *>(ws_bm_tile_tested_.ptr) + w * cap * n_lt : nullptr;
            t.exact_pairs = mode ? static_cast<uint32_t *>(ws_bm_exact_pairs_.ptr) + w * cap : nullptr;
            // (dfire_bm_pose zeroes the sequence's counters: tp_count and the words behind it)
            hip_check(launch_bm_pose(t, st), "launch dfire_bm_pose");
            hip_check(launch_bm_cull(t, st), "launch dfire_bm_cull");
            hip_check(launch_bm_pairs(t, st), "launch dfire_bm_pairs");
            hip_check(launch_bm_gather(t, st), "launch dfire_bm_gather");
        }
    }
    if (two_lanes) {
        hip_check(hipEventRecord(bm_join_, bm_aux_stream_), "hipEventRecord");
        hip_check(hipStreamWaitEvent(stream_, bm_join_, 0), "hipStreamWaitEvent");
    }
    if (dbg) {   // diagnostics: wave lifetimes of the last pass, one text line per wave
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        std::vector<unsigned long long> h(waves * 8);
        hip_check(hipMemcpy(h.data(), t.debug, h.size() * 8, hipMemcpyDeviceToHost), "D2H debug");
        if (FILE *f = std::fopen(dbg, "w")) {
            for (size_t i = 0; i < h.size(); i += 8)
                std::fprintf(f, "%llu %llu %llu %llu %llu %llu %llu %llu\n", h[i], h[i + 1], h[i + 2], h[i + 3], h[i + 4], h[i + 5], h[i + 6], h[i + 7]);
            std::fclose(f);
        }
    }
}

PackedPrepareLaunch Scorer::packed_prepare_launch(const double *poses, size_t stride, const uint8_t *active, size_t n) const {
    PackedPrepareLaunch p;
    p.n_real = tiled_rec_soa_.n_real;
    p.n_tiles = tiled_rec_soa_.n_tiles;
    p.x = tiled_rec_soa_.x;
    p.y = tiled_rec_soa_.y;
    p.z = tiled_rec_soa_.z;
    p.tindex = tiled_rec_soa_.tindex;
    p.slot = tiled_rec_soa_.slot;
    p.num_anm = tiled_rec_soa_.num_anm;
    p.modes = tiled_rec_soa_.modes;
    p.poses = poses;
    p.stride = stride;
    p.active = active;
    p.n_poses = n;
    p.cx = packed_.cx;
    p.cy = packed_.cy;
    p.cz = packed_.cz;
    p.kappa = packed_.kappa;
    p.ubound = packed_.ubound;
    return p;
}

PrepareReceptorLaunch Scorer::prepare_launch(const double *poses, size_t stride, const uint8_t *active, size_t n) const {
    PrepareReceptorLaunch p;
    p.n_real = tiled_rec_soa_.n_real;
    p.n_tiles = tiled_rec_soa_.n_tiles;
    p.x = tiled_rec_soa_.x;
    p.y = tiled_rec_soa_.y;
    p.z = tiled_rec_soa_.z;
    p.tindex = tiled_rec_soa_.tindex;
    p.slot = tiled_rec_soa_.slot;
    p.num_anm = tiled_rec_soa_.num_anm;
    p.modes = tiled_rec_soa_.modes;
    p.poses = poses;
    p.stride = stride;
    p.active = active;
    p.n_poses = n;
    return p;
}

Scorer::~Scorer() {
    for (auto &e : events_) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (bm_aux_stream_) {
        (void)hipStreamSynchronize(bm_aux_stream_);
        (void)hipStreamDestroy(bm_aux_stream_);
        (void)hipEventDestroy(bm_fork_);
        (void)hipEventDestroy(bm_join_);
    }
    if (own_stream_) {
        (void)hipStreamSynchronize(own_stream_);
        (void)hipStreamDestroy(own_stream_);
    }
    ws_partial_.release();
    ws_flags_.release();
    ws_counts_.release();
    ws_tested_.release();
    ws_rec_atoms_.release();
    ws_rec_sub_.release();
    ws_rec_tile_.release();
    ws_rec_pairs_.release();
    ws_exact_.release();
    for (DeviceBuffer *b : {&ws_bm_rt_, &ws_bm_tp_count_, &ws_bm_ent_row_, &ws_bm_jobs_, &ws_bm_job_cost_, &ws_bm_job_order_, &ws_bm_ent_mask_, &ws_bm_queue_, &ws_bm_ent_partial_, &ws_bm_tile_sum_,
                            &ws_bm_tile_tested_, &ws_bm_exact_fix_, &ws_bm_exact_pairs_, &ws_bm_amp_})
        b->release();
    ws_poses_.release();
    ws_energies_.release();
}

uint64_t Scorer::workspace_generation() const {
    return ws_partial_.generation + ws_flags_.generation + ws_counts_.generation + ws_tested_.generation + ws_exact_.generation +
           ws_rec_atoms_.generation + ws_rec_sub_.generation + ws_rec_tile_.generation + ws_rec_pairs_.generation + ws_bm_rt_.generation +
           ws_bm_tp_count_.generation + ws_bm_ent_row_.generation + ws_bm_jobs_.generation + ws_bm_job_cost_.generation + ws_bm_job_order_.generation + ws_bm_ent_mask_.generation + ws_bm_queue_.generation + ws_bm_ent_partial_.generation +
           ws_bm_tile_sum_.generation + ws_bm_tile_tested_.generation +
           ws_bm_exact_fix_.generation + ws_bm_exact_pairs_.generation + ws_bm_amp_.generation;
}

void Scorer::reserve_workspace(size_t n_poses, bool counts) {
    const size_t words = (size_t)(pair_.rec.flag_words + pair_.lig.flag_words);
    const size_t chunks = (size_t)std::max(pair_.n_chunks, use_packed_ ? packed_.n_groups * kPackedPartialsPerGroup : use_tiled_ ? tiled_.n_groups : 0);
    ws_partial_.reserve(n_poses * chunks * 2 * sizeof(double));
    ws_flags_.reserve(std::max<size_t>(n_poses * words * sizeof(uint32_t), 16));
    if (counts) {
        ws_counts_.reserve(n_poses * chunks * sizeof(uint32_t));
        ws_tested_.reserve(n_poses * chunks * sizeof(uint32_t));
        ws_exact_.reserve(n_poses * chunks * sizeof(uint32_t));
    }
    if (use_bm_) {
        const size_t n_lt = (size_t)bm_.lig.n_tiles, n_rt = (size_t)bm_.rec_n_tiles, tile_pairs = n_lt * n_rt;
        const size_t cap = bm_pass_poses(n_poses), sets = bm_sets(n_poses);   // a second set only while two passes are in flight
        const size_t parts = tile_pairs * (cap / 64 + 1), waves = (size_t)n_cus_ * kBmWavesPerCu;
        ws_bm_rt_.reserve(sets * cap * (12 * sizeof(float) + 8 * sizeof(double)));   // the f32 maps of every set, then the exact path's rows
        ws_bm_tp_count_.reserve(sets * (tile_pairs + kBmCounters + kBmCullQueueWords) * sizeof(uint32_t));   // + the launch's counters
        ws_bm_jobs_.reserve(sets * parts * 2 * sizeof(uint32_t));
        ws_bm_job_cost_.reserve(sets * parts * kBmJobRows * sizeof(uint32_t));
        ws_bm_job_order_.reserve(sets * parts * kBmJobRows * (1 + 4) * sizeof(uint32_t) + 16);   // the order, then the jobs' 16-byte records
        ws_bm_queue_.reserve(sets * waves * kBmQueueCap * sizeof(unsigned long long));
        // (+ one part: a job of dfire_bm_pairs loads its part's 1024 entries without looking at the part's end; what lies beyond is never used)
        ws_bm_ent_row_.reserve((sets * tile_pairs * cap + kBmPartEntries) * sizeof(uint32_t));
        ws_bm_ent_mask_.reserve((sets * tile_pairs * cap + kBmPartEntries) * sizeof(unsigned long long));
        ws_bm_ent_partial_.reserve(sets * waves * kBmPartEntries * sizeof(long long));   // per wave of dfire_bm_pairs: the partial sums of its current job (8 KB, L2 resident)
        ws_bm_tile_sum_.reserve(sets * cap * n_lt * sizeof(long long));
        ws_bm_exact_fix_.reserve(sets * cap * sizeof(long long));
        if (bm_.anm_rec + bm_.anm_lig > 0) {   // the poses' amplitudes by row, and the receptor's boxes by pose
            ws_bm_amp_.reserve(sets * cap * (kBmAmpFloats * sizeof(float) + 2 * kBmMaxModes * sizeof(double)));   // every set's f32 rows, then the f64 ones
            ws_rec_sub_.reserve(sets * cap * n_rt * 8 * sizeof(TiledBox));
            ws_rec_tile_.reserve(sets * cap * n_rt * sizeof(TiledBox));
        }
        if (counts) {
            ws_bm_tile_tested_.reserve(sets * cap * n_lt * sizeof(uint32_t));
            ws_bm_exact_pairs_.reserve(sets * cap * sizeof(uint32_t));
        }
    }
}

void Scorer::energy_batch_device(size_t n, const double *d_poses, size_t stride, const uint8_t *d_active,
                                 double *d_energies, uint32_t *d_pair_counts, const uint32_t *d_list, const uint32_t *d_count) {
    if (n == 0) return;
    if (!d_poses || !d_energies) throw Error(LD_ERR_INVALID, "energy_batch: null pose/energy buffer");
    if (stride < pose_len()) throw Error(LD_ERR_INVALID, "energy_batch: stride shorter than a pose row");
    // the list is the compacted form of the mask: the pair kernels walk the list, the tail kernel the mask
    if (d_list && (!d_active || !d_count)) throw Error(LD_ERR_INVALID, "energy_batch: a pose list needs its device-side count and the matching active mask");
    if (use_tiled_ && rec_anm_per_pose_ && !use_bm_) {   // (the block-major path keeps a pose's receptor BOXES only: 36 bytes an atom less)
        // every pose carries its own deformed receptor image: bound that workspace (8 GiB) by
        // slicing very large batches; poses are independent, so the results do not change
        const size_t pad = (size_t)tiled_.rec.n_tiles * 64;
        const size_t per_pose = (use_packed_ ? pad / 2 * sizeof(PackedRecPair) : pad * sizeof(TiledAtom)) +
                                (pad / 8 + pad / 64) * sizeof(TiledBox);
        static const size_t cap = [] {  // LIGHTDOCK_RECEPTOR_IMAGE_MIB: test hook for the slicing
            const char *e = std::getenv("LIGHTDOCK_RECEPTOR_IMAGE_MIB");
            const long v = e ? std::atol(e) : 0;
            return v > 0 ? size_t(v) << 20 : size_t(8) << 30;
        }();
        const size_t max_n = std::max<size_t>(1, cap / per_pose);
        if (n > max_n) {
            for (size_t off = 0; off < n; off += max_n)
                energy_batch_device(std::min(max_n, n - off), d_poses + off * stride, stride,
                                    d_active ? d_active + off : nullptr, d_energies + off,
                                    d_pair_counts ? d_pair_counts + off : nullptr);  // slices go by the mask, not the list
            return;
        }
    }
    reserve_workspace(n, d_pair_counts != nullptr);

    PairLaunch p = pair_;
    p.poses = d_poses;
    p.stride = stride;
    p.active = d_active;
    p.n_poses = n;
    p.partial = static_cast<double *>(ws_partial_.ptr);
    p.flags = static_cast<uint32_t *>(ws_flags_.ptr);
    p.count_partial = d_pair_counts ? static_cast<uint32_t *>(ws_counts_.ptr) : nullptr;

    const size_t words = (size_t)(p.rec.flag_words + p.lig.flag_words);
    // (the block-major path clears a pose's flag words in dfire_bm_pose: one launch less per step)
    if (words > 0 && !use_bm_) hip_check(hipMemsetAsync(p.flags, 0, n * words * sizeof(uint32_t), stream_), "hipMemsetAsync(flags)");
    const bool timing = timing_ && !capturing_;
    if (timing) {
        if (events_used_ == events_.size()) {
            if (events_.size() >= 4096) {  // fold what is pending so the pool stays bounded
                double ms;
                uint64_t n;
                pair_kernel_time(&ms, &n);
                timed_ms_ = ms;
                timed_launches_ = n;
            } else {
                hipEvent_t a, b;
                hip_check(hipEventCreate(&a), "hipEventCreate");
                hip_check(hipEventCreate(&b), "hipEventCreate");
                events_.emplace_back(a, b);
            }
        }
        hip_check(hipEventRecord(events_[events_used_].first, stream_), "hipEventRecord");
    }
    if (use_bm_) {
        p.n_chunks = 1;  // dfire_bm_gather leaves one partial per pose
        run_bm(n, d_poses, stride, d_active, d_pair_counts != nullptr, d_list, d_count);
    } else if (use_packed_) {
        PackedLaunch t = packed_;
        t.poses = d_poses;
        t.stride = stride;
        t.active = d_list ? nullptr : d_active;  // the list holds exactly the active rows
        t.pose_list = d_list;
        t.pose_count = d_list ? d_count : nullptr;
        t.n_poses = n;
        t.partial = p.partial;
        t.flags = p.flags;
        t.count_partial = p.count_partial;
        if (p.count_partial) t.lut = packed_lut_full_;
        t.tested_partial = p.count_partial ? static_cast<uint32_t *>(ws_tested_.ptr) : nullptr;
        t.exact_partial = p.count_partial ? static_cast<uint32_t *>(ws_exact_.ptr) : nullptr;
        p.n_chunks = t.n_groups * kPackedPartialsPerGroup;  // the tail kernel folds this many partials
        if (rec_anm_per_pose_) {  // one deformed receptor image per pose (src/dfire.rs:304-320)
            const size_t pad = (size_t)t.rec.n_tiles * 64;
            ws_rec_pairs_.reserve(n * (pad / 2) * sizeof(PackedRecPair));
            ws_rec_sub_.reserve(n * (pad / 8) * sizeof(TiledBox));
            ws_rec_tile_.reserve(n * (pad / 64) * sizeof(TiledBox));
            PackedPrepareLaunch pr = packed_prepare_launch(d_poses, stride, d_active, n);
            pr.pairs_out = static_cast<PackedRecPair *>(ws_rec_pairs_.ptr);
            pr.sub_out = static_cast<TiledBox *>(ws_rec_sub_.ptr);
            pr.tile_out = static_cast<TiledBox *>(ws_rec_tile_.ptr);
            hip_check(launch_packed_prepare(pr, stream_), "launch dfire_packed_prepare");
            t.rec.pairs = pr.pairs_out;
            t.rec.sub_boxes = pr.sub_out;
            t.rec.tile_boxes = pr.tile_out;
            t.rec.modes = pr.modes;
            t.rec.num_anm = pr.num_anm;
            t.rec.pose_stride_pairs = pad / 2;
            t.rec.pose_stride_sub = pad / 8;
            t.rec.pose_stride_tile = pad / 64;
        }
        hip_check(launch_dfire_packed(t, stream_), "launch dfire_packed_pairs");
    } else     if (use_tiled_) {
        TiledLaunch t = tiled_;
        t.poses = d_poses;
        t.stride = stride;
        t.active = d_active;
        t.n_poses = n;
        t.partial = p.partial;
        t.flags = p.flags;
        t.count_partial = p.count_partial;
        t.tested_partial = p.count_partial ? static_cast<uint32_t *>(ws_tested_.ptr) : nullptr;
        p.n_chunks = t.n_groups;  // the tail kernel folds this many partials
        if (rec_anm_per_pose_) {  // one deformed receptor image per pose (src/dfire.rs:304-320)
            const size_t pad = (size_t)t.rec.n_tiles * 64;
            ws_rec_atoms_.reserve(n * pad * sizeof(TiledAtom));
            ws_rec_sub_.reserve(n * (pad / 8) * sizeof(TiledBox));
            ws_rec_tile_.reserve(n * (pad / 64) * sizeof(TiledBox));
            PrepareReceptorLaunch pr = prepare_launch(d_poses, stride, d_active, n);
            pr.atoms_out = static_cast<TiledAtom *>(ws_rec_atoms_.ptr);
            pr.sub_out = static_cast<TiledBox *>(ws_rec_sub_.ptr);
            pr.tile_out = static_cast<TiledBox *>(ws_rec_tile_.ptr);
            hip_check(launch_prepare_receptor(pr, stream_), "launch dfire_prepare_receptor");
            t.rec.atoms = pr.atoms_out;
            t.rec.sub_boxes = pr.sub_out;
            t.rec.tile_boxes = pr.tile_out;
            t.rec.pose_stride_atoms = pad;
            t.rec.pose_stride_sub = pad / 8;
            t.rec.pose_stride_tile = pad / 64;
        }
        hip_check(launch_dfire_tiled(t, stream_), "launch dfire_tiled_pairs");
    } else {
        hip_check(launch_pair_kernel(p, stream_), "launch pose_energy_pairs");
    }
    if (timing) {
        hip_check(hipEventRecord(events_[events_used_].second, stream_), "hipEventRecord");
        events_used_++;
    }

    FinishLaunch f;
    f.method = method_;
    f.n_chunks = p.n_chunks;
    f.rec_flag_words = p.rec.flag_words;
    f.lig_flag_words = p.lig.flag_words;
    f.tail = tail_;
    f.partial = p.partial;
    f.flags = p.flags;
    f.count_partial = p.count_partial;
    f.active = d_active;
    f.n_poses = n;
    f.energies = d_energies;
    f.pair_counts = d_pair_counts;
    hip_check(launch_finish_kernel(f, stream_), "launch pose_energy_finish");
}

void Scorer::energy_batch_host(size_t n, const double *poses, size_t stride, double *energies) {
    if (n == 0) return;
    if (!poses || !energies) throw Error(LD_ERR_INVALID, "energy_batch: null pose/energy buffer");
    ws_poses_.reserve(n * stride * sizeof(double));
    ws_energies_.reserve(n * sizeof(double));
    hip_check(hipMemcpyAsync(ws_poses_.ptr, poses, n * stride * sizeof(double), hipMemcpyHostToDevice, stream_), "H2D poses");
    energy_batch_device(n, static_cast<const double *>(ws_poses_.ptr), stride, nullptr,
                        static_cast<double *>(ws_energies_.ptr), nullptr);
    hip_check(hipMemcpyAsync(energies, ws_energies_.ptr, n * sizeof(double), hipMemcpyDeviceToHost, stream_), "D2H energies");
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

void Scorer::last_block_counts(size_t n, uint32_t *out_host) {
    if (!use_tiled_) throw Error(LD_ERR_UNSUPPORTED, "block counts exist for the tiled DFIRE kernel only");
    if (!out_host) throw Error(LD_ERR_INVALID, "null output");
    const size_t groups = (size_t)(use_bm_ ? 1 : use_packed_ ? packed_.n_groups * kPackedPartialsPerGroup : tiled_.n_groups);
    if (ws_tested_.bytes < n * groups * sizeof(uint32_t)) throw Error(LD_ERR_INVALID, "no counting launch of that size has run");
    std::vector<uint32_t> part(n * groups);
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    hip_check(hipMemcpy(part.data(), ws_tested_.ptr, part.size() * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H block counts");
    for (size_t p = 0; p < n; p++) {
        uint32_t t = 0;
        for (size_t g = 0; g < groups; g++) t += part[p * groups + g];
        out_host[p] = t;
    }
}

void Scorer::enable_timing(bool on) { timing_ = on; }

void Scorer::pair_kernel_time(double *total_ms, uint64_t *launches) {
    double ms = timed_ms_;
    uint64_t n = timed_launches_;
    for (size_t i = 0; i < events_used_; i++) {
        hip_check(hipEventSynchronize(events_[i].second), "hipEventSynchronize");
        float t = 0.f;
        hip_check(hipEventElapsedTime(&t, events_[i].first, events_[i].second), "hipEventElapsedTime");
        ms += t;
        n++;
    }
    events_used_ = 0;
    timed_ms_ = 0.0;
    timed_launches_ = 0;
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
}

void Scorer::kernel_info(ld_kernel_info *out) const {
    out->pair_kernel_name = use_bm_ ? "dfire_bm_pairs" : use_packed_ ? "dfire_packed_pairs" : use_tiled_ ? "dfire_tiled_pairs" : pair_kernel_name(method_);
    out->block_threads = use_bm_ ? (uint32_t)kBmWaves * 64 : use_packed_ ? (uint32_t)kPackedWaves * 64 : use_tiled_ ? (uint32_t)tiled_.waves * 64 : (uint32_t)kBlockThreads;
    out->receptor_chunks = (uint32_t)(use_bm_ ? 1 : use_packed_ ? packed_.n_groups : use_tiled_ ? tiled_.n_groups : pair_.n_chunks);
    out->lds_bytes = (uint32_t)(use_bm_ ? bm_pairs_lds_bytes() : use_packed_ ? packed_kernel_lds_bytes(packed_.cells_per_unit) : use_tiled_ ? tiled_kernel_lds_bytes(tiled_) : pair_kernel_lds_bytes(pair_));
    out->pair_tests_per_pose = (uint64_t)pair_.rec.n * (uint64_t)pair_.lig.n;
    // DFIRE 26 B/atom (3 f64 + u16 type), DNA 48 B/atom (6 f64), + 240 B/atom
    // per ANM-deformed molecule (10 modes x 24 B), + 56 B pose in + 8 B energy out.
    const uint64_t atoms = (uint64_t)pair_.rec.n + (uint64_t)pair_.lig.n;
    uint64_t bytes = (method_ == LD_METHOD_DFIRE ? 26 : 48) * atoms + 64;
    if (use_anm_) bytes += 24ull * pair_.rec.num_anm * pair_.rec.n + 24ull * pair_.lig.num_anm * pair_.lig.n;
    out->stream_bytes_per_pose = bytes;
}

}  // namespace ld
