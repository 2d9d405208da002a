// capi.cpp -- extern "C" surface declared in include/lightdock_hip.h.
//
// Every call catches ld::Error and turns it into an ld_status plus a thread-local
// message, which is how this library reports what the reference reports by panicking.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>

#include "gso.hpp"
#include "host/cli.hpp"
#include "host/docking_model.hpp"
#include "host/error.hpp"
#include "host/io.hpp"
#include "host/spatial_order.hpp"
#include "lightdock_hip.h"
#include "scorer.hpp"

namespace {

thread_local std::string g_last_error;

int fail(ld_status code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

template <typename F>
int guarded(F &&f) {
    try {
        f();
        return LD_OK;
    } catch (const ld::Error &e) {
        return fail(e.code(), e.what());
    } catch (const std::bad_alloc &) {
        return fail(LD_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(LD_ERR_INVALID, e.what());
    }
}

std::vector<std::string> to_strings(const char *const *list, size_t n) {
    std::vector<std::string> out;
    for (size_t i = 0; i < n; i++)
        if (list && list[i]) out.emplace_back(list[i]);
    return out;
}

}  // namespace

extern "C" {

const char *ld_last_error(void) { return g_last_error.c_str(); }
const char *ld_version(void) { return "lightdock-hip 0.1.0 (gfx950; path of lightdock-rust 0.3.2)"; }

int ld_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ld_init(int device) {
    return guarded([&] {
        int n = ld_device_count();
        if (n <= 0) throw ld::Error(LD_ERR_DEVICE, "no HIP device available: the pose-energy path has no CPU fallback");
        int dev = device;
        if (dev < 0) {
            const char *e = std::getenv("LIGHTDOCK_DEVICE");
            dev = e ? std::atoi(e) : 0;
        }
        if (dev >= n) throw ld::Error(LD_ERR_DEVICE, "device index out of range");
        ld::hip_check(hipSetDevice(dev), "hipSetDevice");
    });
}

ld_scorer *ld_scorer_create(const ld_scorer_desc *desc) {
    ld_scorer *s = nullptr;
    int rc = guarded([&] {
        if (!desc) throw ld::Error(LD_ERR_INVALID, "ld_scorer_create: null descriptor");
        s = new ld_scorer(*desc);
    });
    return rc == LD_OK ? s : nullptr;
}

ld_scorer *ld_scorer_create_from_pdb(int method, const char *receptor_pdb, const char *ligand_pdb,
                                     const char *const *rec_active, size_t n_rec_active,
                                     const char *const *rec_passive, size_t n_rec_passive, const double *rec_nmodes,
                                     size_t rec_nmodes_len, size_t rec_num_anm, const char *const *lig_active,
                                     size_t n_lig_active, const char *const *lig_passive, size_t n_lig_passive,
                                     const double *lig_nmodes, size_t lig_nmodes_len, size_t lig_num_anm, int use_anm,
                                     const double *potential) {
    ld_scorer *s = nullptr;
    int rc = guarded([&] {
        if (!receptor_pdb || !ligand_pdb) throw ld::Error(LD_ERR_INVALID, "PDB path missing");
        if (method != LD_METHOD_DFIRE && method != LD_METHOD_DNA && method != LD_METHOD_PYDOCK)
            throw ld::Error(LD_ERR_UNSUPPORTED, "Error: method not supported");
        ld::Structure rec = ld::read_pdb(receptor_pdb);
        ld::Structure lig = ld::read_pdb(ligand_pdb);
        std::vector<double> rnm, lnm;
        if (rec_nmodes && rec_nmodes_len) rnm.assign(rec_nmodes, rec_nmodes + rec_nmodes_len);
        if (lig_nmodes && lig_nmodes_len) lnm.assign(lig_nmodes, lig_nmodes + lig_nmodes_len);
        if (use_anm) {  // src/bin/lightdock-rust.rs:233,250
            if (rec_num_anm > 0 && rnm.size() != rec.atom_count() * 3 * rec_num_anm)
                throw ld::Error(LD_ERR_INVALID, "Number of read ANM in receptor does not correspond to the number of atoms");
            if (lig_num_anm > 0 && lnm.size() != lig.atom_count() * 3 * lig_num_anm)
                throw ld::Error(LD_ERR_INVALID, "Number of read ANM in ligand does not correspond to the number of atoms");
        }
        ld::DockingModel rm = ld::build_docking_model(method, rec, to_strings(rec_active, n_rec_active),
                                                      to_strings(rec_passive, n_rec_passive), rnm, rec_num_anm);
        ld::DockingModel lm = ld::build_docking_model(method, lig, to_strings(lig_active, n_lig_active),
                                                      to_strings(lig_passive, n_lig_passive), lnm, lig_num_anm);
        ld_scorer_desc d;
        std::memset(&d, 0, sizeof d);
        d.method = method;
        d.use_anm = use_anm;
        d.receptor = rm.view();
        d.ligand = lm.view();
        d.potential = potential;
        s = new ld_scorer(d);
    });
    return rc == LD_OK ? s : nullptr;
}

void ld_scorer_destroy(ld_scorer *s) { delete s; }

/* ---- host-only model builder ------------------------------------------------------------ */
struct ld_model {
    ld::DockingModel model;
};

ld_model *ld_model_from_pdb(int method, const char *pdb_path, const char *const *active, size_t n_active,
                            const char *const *passive, size_t n_passive, const double *nmodes, size_t nmodes_len,
                            size_t num_anm) {
    ld_model *m = nullptr;
    int rc = guarded([&] {
        if (!pdb_path) throw ld::Error(LD_ERR_INVALID, "PDB path missing");
        ld::Structure st = ld::read_pdb(pdb_path);
        std::vector<double> nm;
        if (nmodes && nmodes_len) nm.assign(nmodes, nmodes + nmodes_len);
        if (num_anm > 0 && !nm.empty() && nm.size() != st.atom_count() * 3 * num_anm)
            throw ld::Error(LD_ERR_INVALID, "Number of read ANM does not correspond to the number of atoms");
        m = new ld_model{ld::build_docking_model(method, st, to_strings(active, n_active), to_strings(passive, n_passive),
                                                 nm, num_anm)};
    });
    return rc == LD_OK ? m : nullptr;
}
int ld_model_view(const ld_model *m, ld_molecule *out) {
    return guarded([&] {
        if (!m || !out) throw ld::Error(LD_ERR_INVALID, "null argument");
        *out = m->model.view();
    });
}
void ld_model_destroy(ld_model *m) { delete m; }

int ld_dfire_bin_lut(uint8_t *lut_out, double *steps_out, double *interface_d2_out) {
    return guarded([&] {
        const ld::DfireBinning t = ld::build_dfire_binning();
        if (lut_out) std::memcpy(lut_out, t.lut.data(), 901);
        if (steps_out) std::memcpy(steps_out, t.step.data(), 21 * sizeof(double));
        if (interface_d2_out) *interface_d2_out = ld::dfire_interface_d2();
    });
}
int ld_dfire_packed_lut(int cells_per_unit, double ubound, uint32_t *words_out, double *eps_out) {
    return guarded([&] {
        if (cells_per_unit != 1 && cells_per_unit != 2) throw ld::Error(LD_ERR_INVALID, "cells_per_unit must be 1 or 2");
        if (!(ubound > 0.0)) throw ld::Error(LD_ERR_INVALID, "ubound must be positive");
        const double eps = (double)std::nextafter((float)ld::dfire_f32_error_bound(ubound, cells_per_unit), INFINITY);
        const std::vector<uint32_t> words = ld::build_packed_lut(cells_per_unit, eps);
        if (words_out) std::memcpy(words_out, words.data(), words.size() * sizeof(uint32_t));
        if (eps_out) *eps_out = eps;
    });
}
int ld_dfire_bm_lut(double ubound, double lig_extent, uint8_t *codes_out, double *eps_cells_out) {
    return guarded([&] {
        if (!(ubound > 0.0) || !(lig_extent >= 0.0)) throw ld::Error(LD_ERR_INVALID, "ubound must be positive, lig_extent non-negative");
        const double eps = ld::dfire_bm_error_bound(ubound, lig_extent);
        const std::vector<uint8_t> codes = ld::build_bm_lut(eps);
        if (codes_out) std::memcpy(codes_out, codes.data(), codes.size());
        if (eps_cells_out) *eps_cells_out = eps;
    });
}
int ld_dfire_bm_fix_scale(const double *rec_xyz, size_t n_rec, double reach, double table_vmax, uint64_t *reach_count_out,
                          int *extra_bits_out, double *scale_out) {
    return guarded([&] {
        if ((!rec_xyz && n_rec) || !(reach > 0.0) || !(table_vmax >= 0.0)) throw ld::Error(LD_ERR_INVALID, "coordinates, a positive reach and a non-negative table maximum");
        const size_t count = ld::dfire_bm_reach_count(rec_xyz, n_rec, reach);
        int extra = 0;
        const double scale = ld::dfire_bm_fix_scale(table_vmax, count, &extra);
        if (reach_count_out) *reach_count_out = (uint64_t)count;
        if (extra_bits_out) *extra_bits_out = extra;
        if (scale_out) *scale_out = scale;
    });
}
size_t ld_spatial_tile_order(const double *xyz, size_t n, uint32_t *order_out) {
    size_t len = 0;
    guarded([&] {
        if (!xyz && n) throw ld::Error(LD_ERR_INVALID, "null coordinates");
        const std::vector<uint32_t> order = ld::spatial_tile_order(xyz, n);
        if (order_out) std::memcpy(order_out, order.data(), order.size() * sizeof(uint32_t));
        len = order.size();
    });
    return len;
}
size_t ld_dfire_tile_layout(const double *xyz, const uint32_t *dfire_types, size_t n, uint32_t *order_out,
                            uint32_t *type_perm_out) {
    size_t len = 0;
    guarded([&] {
        if ((!xyz || !dfire_types) && n) throw ld::Error(LD_ERR_INVALID, "null coordinates / types");
        const ld::DfireTileLayout layout = ld::dfire_tile_layout(xyz, dfire_types, n);
        if (order_out) std::memcpy(order_out, layout.order.data(), layout.order.size() * sizeof(uint32_t));
        if (type_perm_out) std::memcpy(type_perm_out, layout.type_perm.data(), layout.type_perm.size() * sizeof(uint32_t));
        len = layout.order.size();
    });
    return len;
}
void ld_stdrng_key(uint64_t seed, uint32_t key_out[8]) { ld::stdrng_key_from_seed(seed, key_out); }

int ld_load_dcparams(const char *path, double *out) {
    return guarded([&] {
        if (!path || !out) throw ld::Error(LD_ERR_INVALID, "ld_load_dcparams: null argument");
        std::vector<double> t = ld::load_dcparams(path);
        std::memcpy(out, t.data(), sizeof(double) * LD_DFIRE_TABLE_LEN);
    });
}

size_t ld_scorer_num_atoms(const ld_scorer *s, int side) { return s ? s->impl.num_atoms(side) : 0; }
size_t ld_scorer_pose_len(const ld_scorer *s) { return s ? s->impl.pose_len() : 0; }
int ld_scorer_method(const ld_scorer *s) { return s ? s->impl.method() : LD_ERR_INVALID; }

int ld_scorer_model_arrays(const ld_scorer *s, int side, double *coordinates, uint32_t *dfire_types, double *ele_charges,
                           double *vdw_charges, double *vdw_radii) {
    return guarded([&] {
        if (!s) throw ld::Error(LD_ERR_INVALID, "null scorer");
        const ld::HostMolecule &m = s->impl.host_molecule(side);
        auto put = [](auto *dst, const auto &src) {
            if (dst && !src.empty()) std::memcpy(dst, src.data(), src.size() * sizeof(src[0]));
        };
        put(coordinates, m.coordinates);
        put(dfire_types, m.dfire_types);
        put(ele_charges, m.ele_charges);
        put(vdw_charges, m.vdw_charges);
        put(vdw_radii, m.vdw_radii);
    });
}

int ld_scorer_set_stream(ld_scorer *s, void *hip_stream) {
    return guarded([&] {
        if (!s) throw ld::Error(LD_ERR_INVALID, "null scorer");
        s->impl.set_stream(static_cast<hipStream_t>(hip_stream));
    });
}

int ld_scorer_energy(ld_scorer *s, const double translation[3], const double rotation_wxyz[4], const double *rec_nmodes,
                     const double *lig_nmodes, double *energy_out) {
    return guarded([&] {
        if (!s || !translation || !rotation_wxyz || !energy_out) throw ld::Error(LD_ERR_INVALID, "ld_scorer_energy: null argument");
        ld::Scorer &sc = s->impl;
        std::vector<double> row(sc.pose_len(), 0.0);
        std::memcpy(row.data(), translation, 3 * sizeof(double));
        std::memcpy(row.data() + 3, rotation_wxyz, 4 * sizeof(double));
        if (sc.anm_rec()) {
            if (!rec_nmodes) throw ld::Error(LD_ERR_INVALID, "ld_scorer_energy: rec_nmodes missing");
            std::memcpy(row.data() + 7, rec_nmodes, sc.anm_rec() * sizeof(double));
        }
        if (sc.anm_lig()) {
            if (!lig_nmodes) throw ld::Error(LD_ERR_INVALID, "ld_scorer_energy: lig_nmodes missing");
            std::memcpy(row.data() + 7 + sc.anm_rec(), lig_nmodes, sc.anm_lig() * sizeof(double));
        }
        sc.energy_batch_host(1, row.data(), row.size(), energy_out);
    });
}

int ld_scorer_energy_batch(ld_scorer *s, size_t n, const double *poses, size_t stride, double *energies_out) {
    return guarded([&] {
        if (!s) throw ld::Error(LD_ERR_INVALID, "null scorer");
        s->impl.energy_batch_host(n, poses, stride, energies_out);
    });
}

int ld_scorer_energy_batch_device(ld_scorer *s, size_t n, const double *d_poses, size_t stride, const uint8_t *d_active,
                                  double *d_energies_out, uint32_t *d_pair_counts) {
    return guarded([&] {
        if (!s) throw ld::Error(LD_ERR_INVALID, "null scorer");
        s->impl.energy_batch_device(n, d_poses, stride, d_active, d_energies_out, d_pair_counts);
    });
}

int ld_scorer_last_block_counts(ld_scorer *s, size_t n, uint32_t *blocks_out_host) {
    return guarded([&] {
        if (!s) throw ld::Error(LD_ERR_INVALID, "null scorer");
        s->impl.last_block_counts(n, blocks_out_host);
    });
}

int ld_scorer_bm_quiet_subtiles(const ld_scorer *s, uint32_t *count_out) {
    return guarded([&] {
        if (!s || !count_out) throw ld::Error(LD_ERR_INVALID, "null argument");
        *count_out = s->impl.bm_quiet_subtiles();
    });
}

int ld_scorer_kernel_info(const ld_scorer *s, ld_kernel_info *out) {
    return guarded([&] {
        if (!s || !out) throw ld::Error(LD_ERR_INVALID, "null argument");
        s->impl.kernel_info(out);
    });
}

int ld_scorer_enable_timing(ld_scorer *s, int enable) {
    return guarded([&] {
        if (!s) throw ld::Error(LD_ERR_INVALID, "null scorer");
        s->impl.enable_timing(enable != 0);
    });
}
int ld_scorer_pair_kernel_time(ld_scorer *s, double *total_ms_out, uint64_t *launches_out) {
    return guarded([&] {
        if (!s) throw ld::Error(LD_ERR_INVALID, "null scorer");
        s->impl.pair_kernel_time(total_ms_out, launches_out);
    });
}

/* ---- GSO ------------------------------------------------------------------------------ */

ld_gso *ld_gso_create(ld_scorer *scorer, size_t n_swarms, size_t n_glowworms, const double *positions,
                      const uint64_t *seeds) {
    ld_gso *g = nullptr;
    int rc = guarded([&] {
        if (!scorer) throw ld::Error(LD_ERR_INVALID, "ld_gso_create: null scorer");
        g = new ld_gso(scorer->impl, n_swarms, n_glowworms, positions, seeds);
    });
    return rc == LD_OK ? g : nullptr;
}
void ld_gso_destroy(ld_gso *g) { delete g; }

int ld_gso_step(ld_gso *g) {
    return guarded([&] {
        if (!g) throw ld::Error(LD_ERR_INVALID, "null gso");
        g->impl.step();
    });
}
int ld_gso_run(ld_gso *g, uint32_t steps) {
    return guarded([&] {
        if (!g) throw ld::Error(LD_ERR_INVALID, "null gso");
        g->impl.run(steps);
    });
}
uint32_t ld_gso_steps_done(const ld_gso *g) { return g ? g->impl.steps_done() : 0; }
uint64_t ld_gso_num_evals(ld_gso *g) {
    uint64_t n = 0;
    if (g) guarded([&] { n = g->impl.num_evals(); });
    return n;
}
int ld_gso_read(ld_gso *g, size_t swarm, double *poses, double *luciferin, double *vision_range, double *scoring,
                int32_t *n_neighbors, int32_t *moved, int32_t *target) {
    return guarded([&] {
        if (!g) throw ld::Error(LD_ERR_INVALID, "null gso");
        g->impl.read(swarm, poses, luciferin, vision_range, scoring, n_neighbors, moved, target);
    });
}
int ld_gso_save(ld_gso *g, size_t swarm, uint32_t step, const char *dir) {
    return guarded([&] {
        if (!g || !dir) throw ld::Error(LD_ERR_INVALID, "null argument");
        g->impl.save(swarm, step, dir);
    });
}

int ld_gso_save_many(ld_gso *g, size_t n, const size_t *swarms, const char *const *dirs, uint32_t step) {
    return guarded([&] {
        if (!g || (n && (!swarms || !dirs))) throw ld::Error(LD_ERR_INVALID, "null argument");
        std::vector<std::string> d(n);
        for (size_t k = 0; k < n; k++) {
            if (!dirs[k]) throw ld::Error(LD_ERR_INVALID, "null directory");
            d[k] = dirs[k];
        }
        g->impl.save_many(std::vector<size_t>(swarms, swarms + n), step, d);
    });
}

int ld_cli_main(int argc, char **argv) {
    int code = 0;
    int rc = guarded([&] { code = ld::cli_main(argc, argv); });
    if (rc != LD_OK) {
        std::fprintf(stderr, "%s\n", g_last_error.c_str());
        return 101;  // the reference's panic exit code
    }
    return code;
}

}  // extern "C"
