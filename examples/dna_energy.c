/* Minimal C client of include/lightdock_hip.h: the known-answer test of the reference's DNA scorer
 * (src/dna.rs:539-572: tests/1azp at the identity pose = -364.88126358158974) plus a small batch.
 *
 *   gcc -std=c99 -I include examples/dna_energy.c -L lightdock-rust_amd/lib -llightdock_hip \
 *       -Wl,-rpath,$PWD/lightdock-rust_amd/lib -o dna_energy
 *   ./dna_energy tests/golden/unit/1azp/1azp_receptor.pdb tests/golden/unit/1azp/1azp_ligand.pdb
 */
#include <stdio.h>
#include <stdlib.h>

#include "lightdock_hip.h"

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s <receptor.pdb> <ligand.pdb>\n", argv[0]);
        return 2;
    }
    if (ld_init(-1) != LD_OK) { /* $LIGHTDOCK_DEVICE or device 0 */
        fprintf(stderr, "ld_init: %s\n", ld_last_error());
        return 1;
    }
    ld_scorer *s = ld_scorer_create_from_pdb(LD_METHOD_DNA, argv[1], argv[2], NULL, 0, NULL, 0, NULL, 0, 0, NULL, 0, NULL, 0,
                                             NULL, 0, 0, /* use_anm */ 0, /* potential (DFIRE only) */ NULL);
    if (!s) {
        fprintf(stderr, "ld_scorer_create_from_pdb: %s\n", ld_last_error());
        return 1;
    }
    /* Score::energy(translation, rotation (w, x, y, z), rec_nmodes, lig_nmodes), src/scoring.rs:11-19 */
    const double t[3] = {0.0, 0.0, 0.0}, q[4] = {1.0, 0.0, 0.0, 0.0};
    double e = 0.0;
    if (ld_scorer_energy(s, t, q, NULL, NULL, &e) != LD_OK) {
        fprintf(stderr, "ld_scorer_energy: %s\n", ld_last_error());
        return 1;
    }
    printf("identity pose: %.14f\n", e);

    /* a batch: rows of pose_len doubles [tx ty tz qw qx qy qz | rec_nm | lig_nm], one launch */
    const size_t n = 4, len = ld_scorer_pose_len(s);
    double *poses = (double *)calloc(n * len, sizeof(double)), out[4];
    for (size_t k = 0; k < n; k++) {
        poses[k * len + 0] = 2.0 * (double)k; /* slide the ligand along x */
        poses[k * len + 3] = 1.0;
    }
    if (ld_scorer_energy_batch(s, n, poses, len, out) != LD_OK) {
        fprintf(stderr, "ld_scorer_energy_batch: %s\n", ld_last_error());
        return 1;
    }
    for (size_t k = 0; k < n; k++) printf("x = %4.1f: %.14f\n", 2.0 * (double)k, out[k]);
    free(poses);
    ld_scorer_destroy(s);
    return 0;
}
