/*
 * ld_oracle.h -- CPU ORACLE for the LightDock-Rust GSO + DFIRE/DNA pose-energy path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, f64, single-threaded restatement of
 * the reference algorithm (lightdock-rust v0.3.2), written loop-for-loop after the
 * reference so that known-answer tests and golden files can be reproduced.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * shipped product (lightdock-rust_amd/, include/lightdock_hip.h) never links, calls
 * or falls back to anything in this directory.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - DNA energy, quaternion algebra, StdRng stream, whole GSO loop, gso_N.out
 *     format: PINNED against the reference's own known-answer tests and committed
 *     example outputs (tests/golden).
 *   - DFIRE absolute energies: the potential table data/DCparams is absent from the
 *     reference mount (.MISSING_LARGE_BLOBS), so "parity unpinned" for DFIRE table
 *     VALUES; everything else in DFIRE shares code shape with the pinned DNA path
 *     and opt-in tests run the real goldens when LIGHTDOCK_DATA/DCparams exists.
 *
 * Each function cites the reference file:line it follows (paths under /root/reference).
 */
#ifndef LD_ORACLE_H
#define LD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_METHOD_DFIRE 0
#define ORC_METHOD_DNA 1
#define ORC_METHOD_PYDOCK 2 /* src/pydock.rs: DNA's energy + a generic-element fallback for unknown atoms */
#define ORC_DFIRE_TABLE_LEN (169 * 169 * 20) /* src/dfire.rs:216,251 */

const char *orc_last_error(void);

/* ---- quaternion, src/qt.rs:19-198; q = {w,x,y,z} --------------------------------- */
void orc_q_conjugate(const double q[4], double out[4]);
double orc_q_dot(const double a[4], const double b[4]);
double orc_q_norm2(const double q[4]);
double orc_q_norm(const double q[4]);
void orc_q_normalize(double q[4]);
void orc_q_inverse(const double q[4], double out[4]);
double orc_q_distance(const double a[4], const double b[4]);
void orc_q_mul(const double a[4], const double b[4], double out[4]);
void orc_q_rotate(const double q[4], const double v[3], double out[3]);
void orc_q_lerp(const double a[4], const double b[4], double t, double out[4]);
void orc_q_slerp(const double a[4], const double b[4], double t, double out[4]);

/* ---- rand 0.7.3 StdRng (ChaCha20 + PCG32 seed expander), SURVEY Appendix B -------- */
typedef struct orc_rng orc_rng;
orc_rng *orc_rng_new(uint64_t seed);          /* SeedableRng::seed_from_u64, src/lib.rs:38 */
void orc_rng_free(orc_rng *r);
uint64_t orc_rng_next_u64(orc_rng *r);
double orc_rng_f64(orc_rng *r);                /* rng.gen::<f64>(), src/swarm.rs:118 */
void orc_q_random(orc_rng *r, double out[4]);  /* src/qt.rs:93-103 */

/* ---- DFIRE table --------------------------------------------------------------- */
/* src/dfire.rs:236-257: first 169*169*20 lines of a text file, one f64 per line. */
int orc_load_dcparams(const char *path, double *out /* ORC_DFIRE_TABLE_LEN */);
/* src/dfire.rs:49-53,336-337: (d2) -> table bin, via d = sqrt(d2)*2-1, `as usize`. */
int orc_dfire_bin(double dist2);

/* ---- scorer ---------------------------------------------------------------------- */
typedef struct orc_scorer orc_scorer;

/* DFIRE::new / DNA::new (src/dfire.rs:201-234, src/dna.rs:375-408).  PDB files are parsed
 * here; restraint lists are arrays of "chain.resname.serial[icode]" strings.  nmodes are
 * the flat (mode, atom, xyz) f64 arrays of rec_nm.npy / lig_nm.npy or NULL.
 * potential: DFIRE only, ORC_DFIRE_TABLE_LEN doubles (copied).  Returns NULL on error. */
orc_scorer *orc_scorer_new(int method, const char *receptor_pdb, const char *ligand_pdb,
                           const char *const *rec_active, int n_rec_active,
                           const char *const *rec_passive, int n_rec_passive,
                           const double *rec_nmodes, size_t rec_nmodes_len, int rec_num_anm,
                           const char *const *lig_active, int n_lig_active,
                           const char *const *lig_passive, int n_lig_passive,
                           const double *lig_nmodes, size_t lig_nmodes_len, int lig_num_anm,
                           int use_anm, const double *potential);
void orc_scorer_free(orc_scorer *s);

/* Score::energy (src/scoring.rs:11-19; src/dfire.rs:265-362; src/dna.rs:411-529). */
double orc_scorer_energy(const orc_scorer *s, const double t[3], const double q[4],
                         const double *rec_nm, const double *lig_nm);

/* Same evaluation, with the intermediate quantities the tests/bench need:
 *  stats[0] raw pair sum before the final transform (DFIRE: sum of table entries;
 *           DNA: total_elec before *FACTOR/EPSILON), stats[1] DNA total_vdw,
 *  stats[2] satisfied receptor restraints fraction, stats[3] ligand fraction,
 *  stats[4] membrane intersection fraction, stats[5] #pairs inside the outer cutoff
 *  (DFIRE d2<=225; DNA d2<=900), stats[6] #interface receptor atoms, stats[7] #interface
 *  ligand atoms. */
double orc_scorer_energy_ex(const orc_scorer *s, const double t[3], const double q[4],
                            const double *rec_nm, const double *lig_nm, double stats[8]);

/* n pose rows ([t q rec_nm lig_nm], `stride` doubles apart) on `threads` host threads, thread k taking
 * rows k, k+threads, ...: the stand-in for `ant_thony.py --cores N` (example/1czy/execution.sh:24)
 * that bench.py times as the CPU baseline.  Returns 0, or -1 on bad arguments. */
int orc_scorer_energy_rows_mt(const orc_scorer *s, const double *rows, size_t n, size_t stride, int threads, double *out);

/* model introspection, side: 0 receptor, 1 ligand */
size_t orc_scorer_num_atoms(const orc_scorer *s, int side);
const double *orc_scorer_coordinates(const orc_scorer *s, int side); /* n*3 AoS */
const uint32_t *orc_scorer_dfire_types(const orc_scorer *s, int side);
const double *orc_scorer_ele_charges(const orc_scorer *s, int side);
const double *orc_scorer_vdw_charges(const orc_scorer *s, int side);
const double *orc_scorer_vdw_radii(const orc_scorer *s, int side);
size_t orc_scorer_num_membrane(const orc_scorer *s, int side);
const uint32_t *orc_scorer_membrane(const orc_scorer *s, int side);
/* active restraint groups as CSR (group order = first appearance in the PDB) */
size_t orc_scorer_num_restraint_groups(const orc_scorer *s, int side);
const uint32_t *orc_scorer_restraint_offsets(const orc_scorer *s, int side);
const uint32_t *orc_scorer_restraint_atoms(const orc_scorer *s, int side);

/* ---- GSO (src/lib.rs:27-58, src/swarm.rs, src/glowworm.rs) -------------------- */
typedef struct orc_gso orc_gso;
/* positions: n rows x row_len (7 or 7+anm_rec+anm_lig), src/swarm.rs:26-64 */
orc_gso *orc_gso_new(const double *positions, int n, int row_len, uint64_t seed,
                     const orc_scorer *scorer, int use_anm, int rec_num_anm, int lig_num_anm);
void orc_gso_free(orc_gso *g);
void orc_gso_step(orc_gso *g);                          /* one iteration of lib.rs:47-50 */
int orc_gso_save(const orc_gso *g, int step, const char *dir);      /* swarm.rs:128-167 */
int orc_gso_run(orc_gso *g, int steps, const char *dir);            /* lib.rs:46-58 */
int orc_gso_num_glowworms(const orc_gso *g);
int orc_gso_row_len(const orc_gso *g);
uint64_t orc_gso_num_evals(const orc_gso *g);
/* Snapshot of the swarm: pose rows (n x row_len), luciferin, vision range, scoring,
 * neighbour count, moved flag, and the neighbour id chosen at the last move (own id if none). */
void orc_gso_state(const orc_gso *g, double *poses, double *luciferin, double *vision,
                   double *scoring, int32_t *n_neighbors, int32_t *moved, int32_t *target);
/* neighbour ids of glowworm i from the last movement phase; returns count */
int orc_gso_neighbors(const orc_gso *g, int i, int32_t *out, int cap);

/* ---- file helpers used by the oracle CLI and tests ------------------------------ */
/* initial_positions_N.dat, src/bin/lightdock-rust.rs:60-75. Returns malloc'd rows. */
double *orc_parse_positions(const char *path, int *n_rows, int *row_len);
/* flat <f8 .npy, src/bin/lightdock-rust.rs:221-252. Returns malloc'd data. */
double *orc_read_npy_f64(const char *path, size_t *len);
void orc_free(void *p);

/* the reference CLI (src/bin/lightdock-rust.rs:77-333) as a function */
int orc_cli_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
