// host_check.cpp -- TEST INFRASTRUCTURE: drives the host side of the library through its C ABI in
// the sanitizer build (tests/asan/hip_stub.cpp stands in for the HIP runtime and the kernels).
// usage: host_check <tests/golden> <scratch dir>.  Exit code 0 = every call behaved; ASan / UBSan
// report on their own (the CPU test greps for them).
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lightdock_hip.h"

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "host_check: %s failed at line %d (%s)\n", #cond, __LINE__, ld_last_error()); \
            failures++;                                                          \
        }                                                                        \
    } while (0)

static int cli(std::vector<std::string> args) {
    std::vector<char *> argv;
    for (auto &a : args) argv.push_back(&a[0]);
    argv.push_back(nullptr);
    return ld_cli_main((int)args.size(), argv.data());
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string gold = argv[1], scratch = argv[2];
    CHECK(ld_init(0) == LD_OK);
    CHECK(ld_device_count() == 1);

    // ---- host-only helpers ---------------------------------------------------------------------
    std::vector<uint8_t> lut(901);
    std::vector<double> steps(21);
    double iface = 0;
    CHECK(ld_dfire_bin_lut(lut.data(), steps.data(), &iface) == LD_OK);
    for (int cells = 1; cells <= 2; cells++) {
        std::vector<uint32_t> words(1028 * cells);
        double eps = 0;
        CHECK(ld_dfire_packed_lut(cells, 256.0, words.data(), &eps) == LD_OK && eps > 0);
    }
    CHECK(ld_dfire_packed_lut(3, 256.0, nullptr, nullptr) != LD_OK);
    uint32_t key[8];
    ld_stdrng_key(324324, key);

    // ---- model builders (src/dfire.rs:115-190, src/dna.rs:249-364, src/pydock.rs) ----------------
    const std::string rec1 = gold + "/1ppe/lightdock_1ppe_e.pdb", lig1 = gold + "/1ppe/lightdock_1ppe_i.pdb";
    const char *active[] = {"E.ILE.16", "E.XXX.999"};
    for (int method = 0; method < 3; method++) {
        const std::string pdb = method == 0 ? rec1 : gold + "/unit/1azp/1azp_ligand.pdb";
        ld_model *m = ld_model_from_pdb(method, pdb.c_str(), active, 2, nullptr, 0, nullptr, 0, 0);
        CHECK(m != nullptr);
        if (!m) continue;
        ld_molecule mol;
        CHECK(ld_model_view(m, &mol) == LD_OK && mol.n_atoms > 0);
        if (method == 0) {
            std::vector<uint32_t> order((mol.n_atoms + 63) / 64 * 64), perm(169);
            CHECK(ld_spatial_tile_order(mol.coordinates, mol.n_atoms, order.data()) == order.size());
            CHECK(ld_dfire_tile_layout(mol.coordinates, mol.dfire_types, mol.n_atoms, order.data(), perm.data()) == order.size());
        }
        ld_model_destroy(m);
    }
    CHECK(ld_model_from_pdb(0, (scratch + "/missing.pdb").c_str(), nullptr, 0, nullptr, 0, nullptr, 0, 0) == nullptr);
    {   // unsupported residue / atom, short line, empty file
        const std::string bad = scratch + "/bad.pdb";
        const char *texts[] = {"ATOM      1  N   XXX A   1      11.104  13.207   2.100  1.00  0.00           N\n",
                               "ATOM      1  H1  ALA A   1      11.104  13.207   2.100  1.00  0.00           H\n",
                               "ATOM      1  N   ALA A   1      11.1\n", ""};
        for (const char *t : texts) {
            FILE *f = std::fopen(bad.c_str(), "w");
            std::fputs(t, f);
            std::fclose(f);
            ld_model *m = ld_model_from_pdb(0, bad.c_str(), nullptr, 0, nullptr, 0, nullptr, 0, 0);
            if (m) ld_model_destroy(m);
        }
    }

    // ---- DCparams (src/dfire.rs:236-257) -----------------------------------------------------------
    std::vector<double> table(LD_DFIRE_TABLE_LEN);
    mkdir((scratch + "/data").c_str(), 0755);
    const std::string dc = scratch + "/data/DCparams";
    {
        FILE *f = std::fopen(dc.c_str(), "w");
        for (int i = 0; i < 1000; i++) std::fprintf(f, "%.9f\n", 0.001 * i);
        std::fclose(f);
        CHECK(ld_load_dcparams(dc.c_str(), table.data()) != LD_OK);          // too short
        f = std::fopen(dc.c_str(), "w");
        for (size_t i = 0; i < LD_DFIRE_TABLE_LEN + 5; i++) std::fprintf(f, "%.9f\n", (double)((i * 2654435761u) % 4000) / 1000.0 - 2.0);
        std::fclose(f);
        CHECK(ld_load_dcparams(dc.c_str(), table.data()) == LD_OK);
        CHECK(ld_load_dcparams((scratch + "/nope").c_str(), table.data()) != LD_OK);
    }

    // ---- scorer + GSO bookkeeping (kernels stubbed) ---------------------------------------------
    ld_scorer *s = ld_scorer_create_from_pdb(LD_METHOD_DFIRE, rec1.c_str(), lig1.c_str(), active, 1, nullptr, 0, nullptr, 0, 0,
                                             nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, table.data());
    CHECK(s != nullptr);
    if (s) {
        CHECK(ld_scorer_pose_len(s) == 7 && ld_scorer_num_atoms(s, 0) == 1615 && ld_scorer_num_atoms(s, 1) == 221);
        std::vector<double> poses(7 * 300, 0.0), e(300);
        for (int i = 0; i < 300; i++) poses[7 * i + 3] = 1.0;
        CHECK(ld_scorer_energy_batch(s, 300, poses.data(), 7, e.data()) == LD_OK);
        CHECK(ld_scorer_energy_batch(s, 3, poses.data(), 5, e.data()) != LD_OK);   // stride shorter than a pose row
        double one = 1.0;
        const double t0[3] = {0, 0, 0}, q0[4] = {1, 0, 0, 0};
        CHECK(ld_scorer_energy(s, t0, q0, nullptr, nullptr, &one) == LD_OK);
        ld_kernel_info info;
        CHECK(ld_scorer_kernel_info(s, &info) == LD_OK);
        std::vector<double> swarms(7 * 2 * 50, 0.0);
        for (int i = 0; i < 100; i++) swarms[7 * i + 3] = 1.0;
        ld_gso *g = ld_gso_create(s, 2, 50, swarms.data(), nullptr);
        CHECK(g != nullptr);
        if (g) {
            CHECK(ld_gso_run(g, 13) == LD_OK && ld_gso_steps_done(g) == 13);
            (void)ld_gso_num_evals(g);
            const std::string d0 = scratch + "/swarm_a", d1 = scratch + "/swarm_b";
            mkdir(d0.c_str(), 0755);
            mkdir(d1.c_str(), 0755);
            CHECK(ld_gso_save(g, 1, 13, d0.c_str()) == LD_OK);
            const size_t ids[2] = {0, 1};
            const char *dirs[2] = {d0.c_str(), d1.c_str()};
            CHECK(ld_gso_save_many(g, 2, ids, dirs, 13) == LD_OK);
            const size_t bad_ids[1] = {7};
            CHECK(ld_gso_save_many(g, 1, bad_ids, dirs, 13) != LD_OK);
            CHECK(ld_gso_save(g, 0, 13, (scratch + "/no/such/dir").c_str()) != LD_OK);
            std::vector<double> rp(7 * 50), luc(50), vis(50), sco(50);
            std::vector<int32_t> nn(50), mv(50), tg(50);
            CHECK(ld_gso_read(g, 1, rp.data(), luc.data(), vis.data(), sco.data(), nn.data(), mv.data(), tg.data()) == LD_OK);
            CHECK(ld_gso_read(g, 2, rp.data(), luc.data(), vis.data(), sco.data(), nn.data(), mv.data(), tg.data()) != LD_OK);
            ld_gso_destroy(g);
        }
        CHECK(ld_gso_create(s, 0, 50, swarms.data(), nullptr) == nullptr);
        ld_scorer_destroy(s);
    }
    CHECK(ld_scorer_create_from_pdb(7, rec1.c_str(), lig1.c_str(), nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0,
                                    nullptr, 0, 0, 0, table.data()) == nullptr);

    // ---- the CLI (src/bin/lightdock-rust.rs:77-333): usage errors return 0, panics 101 ---------------
    CHECK(chdir(scratch.c_str()) == 0);
    CHECK(cli({"lightdock-hip"}) == 0);
    (void)cli({"lightdock-hip", gold + "/1ppe/setup.json", gold + "/1ppe/initial_positions_0.dat", "abc", "dfire"});   // exit codes: tests/test_host_cpu.py
    CHECK(cli({"lightdock-hip", gold + "/1ppe/setup.json", gold + "/1ppe/initial_positions_0.dat", "2", "nomethod"}) == 0);
    (void)cli({"lightdock-hip", scratch + "/nosetup.json", gold + "/1ppe/initial_positions_0.dat", "2", "dfire"});
    CHECK(cli({"lightdock-hip", gold + "/1ppe/setup.json", gold + "/1ppe/initial_positions_0.dat", "2", "dfire"}) == 0);   // data/DCparams from above
    // DNA + ANM: rec_nm.npy / lig_nm.npy from the CWD
    for (const char *f : {"rec_nm.npy", "lig_nm.npy"}) {
        std::string cmd = "cp " + gold + "/1azp/" + f + " " + scratch + "/";
        CHECK(std::system(cmd.c_str()) == 0);
    }
    CHECK(cli({"lightdock-hip", gold + "/1azp/setup.json", gold + "/1azp/initial_positions_0.dat", "2", "dna"}) == 0);
    {   // a truncated .npy must be refused, not read past its end
        FILE *f = std::fopen((scratch + "/rec_nm.npy").c_str(), "r+");
        if (f) { CHECK(ftruncate(fileno(f), 200) == 0); std::fclose(f); }
        CHECK(cli({"lightdock-hip", gold + "/1azp/setup.json", gold + "/1azp/initial_positions_0.dat", "2", "dna"}) != 0);
    }
    std::fprintf(stderr, "host_check: %d failures\n", failures);
    return failures ? 1 : 0;
}
